"""TEST INFRASTRUCTURE — plan descriptions with the constructor signatures and attributes of `ballista_amd.plan`'s operator
classes, but without the HIP library behind them: plain objects that

  * `oracle/plan_eval.py` evaluates on the CPU (it duck-types on these class NAMES), so the distributed query flows of
    `ballista_amd/distributed.py` run under the 2-rank gloo tests with the oracle as the engine;
  * `tests/proto_encode.py` serialises to `PhysicalPlanNode` bytes for the protobuf decoder's tests.

`schema()` follows the output-schema rules of the operators (rust/core/src/serde/physical_plan/from_proto.rs:58-346 build
them; state columns per SURVEY.md Appendix A), as `ballista_amd/csrc/host/ops_*.cpp` implement them.
Use: `tpch.P = plan_nodes` (the plan builders of ballista_amd/tpch.py take their classes from that module global).
"""
from __future__ import annotations

from ballista_amd import expr as E

PARTIAL, FINAL = "Partial", "Final"
INNER, LEFT, RIGHT = "Inner", "Left", "Right"


class Partitioning:
    UNKNOWN, ROUND_ROBIN, HASH = 0, 1, 2

    def __init__(self, scheme, count, exprs=()):
        self.scheme, self.count, self.exprs = scheme, count, list(exprs)

    @staticmethod
    def Hash(exprs, n): return Partitioning(Partitioning.HASH, n, exprs)
    @staticmethod
    def RoundRobinBatch(n): return Partitioning(Partitioning.ROUND_ROBIN, n)
    @staticmethod
    def UnknownPartitioning(n): return Partitioning(Partitioning.UNKNOWN, n)

    def partition_count(self): return self.count


class _Node:
    def children(self):
        return [self.input] if hasattr(self, "input") else []

    def with_new_children(self, children):
        import copy
        new = copy.copy(self)
        if len(children) == 1:
            new.input = children[0]
        elif len(children) == 2:
            new.left, new.right = children
        return new

    def _types(self):
        return {n: t for n, t, _ in self.schema()}


class MemoryExec(_Node):
    """partitions: list of partitions, each a list of oracle batches (dict name -> OCol)"""

    def __init__(self, partitions, ctx=None, schema=None):
        self._oracle_partitions = [list(p) for p in partitions]
        self._schema = schema
        self.name = None              # set by the proto encoder's callers: the path the scan leaf carries

    def children(self):
        return []

    def schema(self):
        if self._schema is not None:
            return list(self._schema)
        for p in self._oracle_partitions:
            for b in p:
                return [(k, c.dtype, c.valid is not None) for k, c in b.items()]
        raise ValueError("MemoryExec without batches needs a schema")


class FilterExec(_Node):
    def __init__(self, predicate, input):
        self.predicate, self.input = predicate, input

    def schema(self):
        return self.input.schema()


class ProjectionExec(_Node):
    def __init__(self, exprs, input):
        self.exprs, self.input = list(exprs), input

    def schema(self):
        t = self.input._types()
        nul = {n: u for n, _, u in self.input.schema()}
        return [(n, E.expr_type(e, t), nul.get(e.name, True) if isinstance(e, E.Column) else True) for e, n in self.exprs]


class CoalesceBatchesExec(_Node):
    def __init__(self, input, target_batch_size):
        self.input, self.target_batch_size = input, target_batch_size

    def schema(self):
        return self.input.schema()


class MergeExec(_Node):
    def __init__(self, input):
        self.input = input

    def schema(self):
        return self.input.schema()


class GlobalLimitExec(_Node):
    def __init__(self, input, limit):
        self.input, self.limit = input, limit

    def schema(self):
        return self.input.schema()


class LocalLimitExec(GlobalLimitExec):
    pass


def _sum_type(t):
    if t == E.FLOAT64:
        return E.FLOAT64
    return E.UINT64 if t in (E.UINT8, E.UINT64) else E.INT64


class HashAggregateExec(_Node):
    def __init__(self, mode, group_expr, aggr_expr, input):
        self.mode, self.group_expr, self.aggr_expr, self.input = mode, list(group_expr), list(aggr_expr), input

    def schema(self):
        t = self.input._types()
        ins = self.input.schema()
        out = [(n, E.expr_type(e, t), True) for e, n in self.group_expr]
        pos = len(self.group_expr)
        for a in self.aggr_expr:
            if self.mode == PARTIAL:
                at = E.expr_type(a.expr, t)
                if a.fun == "SUM":
                    out.append((a.name + "[sum]", _sum_type(at), True))
                elif a.fun == "AVG":
                    out += [(a.name + "[count]", E.UINT64, False), (a.name + "[sum]", E.FLOAT64, True)]
                elif a.fun == "COUNT":
                    out.append((a.name + "[count]", E.UINT64, False))
                else:
                    out.append((a.name + ("[min]" if a.fun == "MIN" else "[max]"), at, True))
            else:
                if a.fun == "AVG":
                    out.append((a.name, E.FLOAT64, True))
                    pos += 2
                    continue
                out.append((a.name, E.UINT64 if a.fun == "COUNT" else ins[pos][1], a.fun != "COUNT"))
                pos += 1
        return out


class HashJoinExec(_Node):
    def __init__(self, left, right, on, join_type=INNER):
        self.left, self.right, self.on, self.join_type = left, right, list(on), join_type

    def children(self):
        return [self.left, self.right]

    def schema(self):
        out = list(self.left.schema())
        drop = {b for a, b in self.on if a == b}
        return out + [f for f in self.right.schema() if f[0] not in drop]


class SortExec(_Node):
    def __init__(self, expr, input):
        self.expr, self.input = list(expr), input

    def schema(self):
        return self.input.schema()


class RepartitionExec(_Node):
    def __init__(self, input, partitioning):
        self.input, self.partitioning = input, partitioning

    def schema(self):
        return self.input.schema()


class AllGatherExec(_Node):
    """ballista_amd.plan.AllGatherExec: `comm.world` output partitions, partition r = rank r's (concatenated) input.
    comm: an object with world / rank / all_gather(batch) -> [batch] (tests/test_distributed_cpu.py: OracleComm over gloo)"""

    def __init__(self, input, comm):
        self.input, self.comm = input, comm

    def schema(self):
        return self.input.schema()


class ShuffleExchangeExec(_Node):
    """ballista_amd.plan.ShuffleExchangeExec: this rank's partition of Hash([key], world) over every rank's input"""

    def __init__(self, input, comm, key, chunk_rows=0):
        self.input, self.comm, self.key, self.chunk_rows = input, comm, key, chunk_rows

    def schema(self):
        return self.input.schema()
