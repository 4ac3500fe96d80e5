"""TEST INFRASTRUCTURE — evaluates a plan tree with the CPU oracle (oracle/engine.py).

Plans are duck-typed on the operator class NAMES of DataFusion / the reference's serde
(rust/core/src/serde/physical_plan/from_proto.rs:58-346) and on the attributes the host mirror
keeps (predicate, exprs, group_expr, ...).  Leaves are MemoryExec objects carrying
`_oracle_partitions` (list of partitions, each a list of `dict name -> OCol`), attached by the
test helpers — the oracle never reads device memory.
"""
from __future__ import annotations

from collections import OrderedDict

from . import engine as og


def execute(plan, partition):
    """-> list of batches (dict name -> OCol) of one output partition"""
    k = type(plan).__name__
    if k == "MemoryExec":
        return [b for b in plan._oracle_partitions[partition]]
    if k == "FilterExec":
        return [og.filter_batch(b, plan.predicate) for b in execute(plan.input, partition)]
    if k == "ProjectionExec":
        return [og.project(b, plan.exprs) for b in execute(plan.input, partition)]
    if k == "CoalesceBatchesExec":
        return execute(plan.input, partition)
    if k == "MergeExec":
        out = []
        for p in range(n_partitions(plan.input)):
            out.extend(execute(plan.input, p))
        return out
    if k in ("GlobalLimitExec", "LocalLimitExec"):
        batches = [b for b in execute(plan.input, partition) if og.batch_len(b)]
        if not batches:
            return []
        return [og.limit(og.concat_batches(batches), plan.limit)]
    if k == "HashAggregateExec":
        batches = execute(plan.input, partition)
        schema_in = batches[0] if batches else None
        nonempty = [b for b in batches if og.batch_len(b)]
        if nonempty:
            whole = og.concat_batches(nonempty)
        elif schema_in is not None:
            whole = schema_in
        else:
            raise ValueError("oracle: aggregate over a partition without batches needs a schema")
        return [og.hash_aggregate(whole, plan.mode, plan.group_expr, plan.aggr_expr)]
    if k == "HashJoinExec":
        left = []
        for p in range(n_partitions(plan.left)):
            left.extend(execute(plan.left, p))
        lb = og.concat_batches(left)
        right = [b for b in execute(plan.right, partition)]
        # unmatched build rows of a Left join are emitted once per task, after every probe batch
        return [og.hash_join(lb, og.concat_batches(right), plan.on, plan.join_type)]
    if k == "SortExec":
        batches = execute(plan.input, 0)
        return [og.sort_batch(og.concat_batches(batches), plan.expr)]
    if k == "RepartitionExec":
        part = plan.partitioning
        if part.scheme == 2:
            out = []
            for p in range(n_partitions(plan.input)):
                for b in execute(plan.input, p):
                    out.append(og.repartition_hash(b, part.exprs, part.count)[partition])
            return out
        # RoundRobinBatch: batch i of the concatenated input stream -> partition i mod n
        allb = []
        for p in range(n_partitions(plan.input)):
            allb.extend(execute(plan.input, p))
        return [b for i, b in enumerate(allb) if i % part.count == partition]
    if k in ("AllGatherExec", "ShuffleExchangeExec"):
        # the stage boundary between ranks (ballista_amd/csrc/host/exchange.cpp): every partition of the input as one batch goes
        # through plan.comm ONCE per node (a collective call), whichever output partition is asked for first
        if "_exchanged" not in plan.__dict__:
            mine = _whole_input(plan.input)
            plan._exchanged = plan.comm.all_gather(mine) if k == "AllGatherExec" else [plan.comm.shuffle(mine, plan.key)]
        return [plan._exchanged[partition]]
    raise NotImplementedError(k)


def _whole_input(plan):
    """every partition of `plan` as ONE batch (an empty one with the plan's schema when there is no row at all)"""
    import numpy as np
    out = []
    for p in range(n_partitions(plan)):
        out.extend(b for b in execute(plan, p))
    live = [b for b in out if og.batch_len(b)]
    if live:
        return og.concat_batches(live)
    if out:
        return out[0]
    empty = OrderedDict()
    for name, dtype, _ in plan.schema():
        np_t = {"Float64": np.float64, "Int64": np.int64, "UInt64": np.uint64, "UInt8": np.uint8, "Boolean": np.bool_}.get(dtype, np.int32)
        empty[name] = og.OCol(dtype, [] if dtype == "Utf8" else np.zeros(0, np_t))
    return empty


def n_partitions(plan):
    k = type(plan).__name__
    if k == "MemoryExec":
        return len(plan._oracle_partitions)
    if k in ("MergeExec", "SortExec", "GlobalLimitExec"):
        return 1
    if k == "RepartitionExec":
        return plan.partitioning.count
    if k == "AllGatherExec":
        return plan.comm.world
    if k == "ShuffleExchangeExec":
        return 1
    if k == "HashJoinExec":
        return n_partitions(plan.right)
    return n_partitions(plan.input)


def collect(plan):
    """all partitions concatenated -> one batch"""
    out = []
    for p in range(n_partitions(plan)):
        out.extend(execute(plan, p))
    out = [b for b in out]
    return og.concat_batches(out) if out else OrderedDict()
