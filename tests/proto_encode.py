"""TEST INFRASTRUCTURE — a proto3 ENCODER for the physical-plan subset of the reference's wire format, written from the field
numbers of rust/core/proto/ballista.proto (PhysicalPlanNode :294-422, LogicalExprNode :14-161, ScalarValue :685-709,
Schema / Field / ArrowType :611-800) in the way rust/core/src/serde/physical_plan/to_proto.rs:60-511 fills them.
It produces the bytes an executor receives in a task (`TaskDefinition.plan`, `ExecutePartition.plan`); the product's decoder
(`bhip_plan_from_proto`, ballista_amd/csrc/host/proto.cpp) is tested against it.  No reference code travels: only numbers.

Plans are `tests/plan_nodes.py` (or `ballista_amd.plan`) trees; a MemoryExec leaf is written as a CsvScanExecNode whose path
is the leaf's `.name` ("mem://lineitem"): the test's leaf resolver maps it back to a table.
"""
from __future__ import annotations

import struct

from ballista_amd import expr as E


# ---- proto3 wire primitives -------------------------------------------------------------------------------------------

def varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def tag(field: int, wt: int) -> bytes:
    return varint((field << 3) | wt)


def f_varint(field, v, always=False):
    return b"" if (not v and not always) else tag(field, 0) + varint(int(v))


def f_bytes(field, b: bytes, always=True):
    return b"" if (not b and not always) else tag(field, 2) + varint(len(b)) + b


def f_str(field, s: str, always=False):
    return f_bytes(field, s.encode(), always)


def f_double(field, v):
    return tag(field, 1) + struct.pack("<d", v)


def f_float(field, v):
    return tag(field, 5) + struct.pack("<f", v)


def f_packed(field, vals):
    return f_bytes(field, b"".join(varint(v) for v in vals), always=False)


# ---- Arrow types / schema ---------------------------------------------------------------------------------------------

# ArrowType oneof field numbers (ballista.proto:755-790); EmptyMessage payloads
ARROW_TYPE_FIELD = {"Boolean": 2, "UInt8": 3, "Int8": 4, "UInt16": 5, "Int16": 6, "UInt32": 7, "Int32": 8, "UInt64": 9, "Int64": 10,
                    "Float32": 12, "Float64": 13, "Utf8": 14, "LargeUtf8": 32, "Binary": 15, "Float16": 11, "Date32": 17, "Date64": 18}


def arrow_type(t: str) -> bytes:
    if t.startswith("Timestamp"):                    # "Timestamp(Nanosecond)" -> Timestamp{time_unit}
        unit = {"Second": 0, "Millisecond": 1, "Microsecond": 2, "Nanosecond": 3}[t[10:-1]]
        return f_bytes(20, f_varint(1, unit))
    return f_bytes(ARROW_TYPE_FIELD[t], b"")


def field(name, dtype, nullable) -> bytes:
    return f_str(1, name) + f_bytes(2, arrow_type(dtype)) + f_varint(3, 1 if nullable else 0)


def schema(fields) -> bytes:
    return b"".join(f_bytes(1, field(n, t, u)) for n, t, u in fields)


# ---- ScalarValue / LogicalExprNode ---------------------------------------------------------------------------------------

PRIMITIVE_SCALAR = {"Boolean": 0, "UInt8": 1, "Int8": 2, "UInt16": 3, "Int16": 4, "UInt32": 5, "Int32": 6, "UInt64": 7, "Int64": 8,
                    "Float32": 9, "Float64": 10, "Utf8": 11, "LargeUtf8": 12, "Date32": 13}


def scalar_value(lit: E.Literal) -> bytes:
    t, v = lit.dtype, lit.value
    if v is None:
        return f_varint(19, PRIMITIVE_SCALAR[t], always=True)
    if t == "Boolean":
        return f_varint(1, 1 if v else 0, always=True)
    if t == "Utf8":
        return f_str(2, v, always=True)
    if t == "LargeUtf8":
        return f_str(3, v, always=True)
    if t == "Float64":
        return f_double(13, float(v))
    if t == "Float32":
        return f_float(12, float(v))
    num = {"Int8": 4, "Int16": 5, "Int32": 6, "Int64": 7, "UInt8": 8, "UInt16": 9, "UInt32": 10, "UInt64": 11, "Date32": 14}[t]
    return f_varint(num, int(v), always=True)


SCALAR_FN = {"sqrt": 0, "sin": 1, "cos": 2, "tan": 3, "asin": 4, "acos": 5, "atan": 6, "exp": 7, "ln": 8, "log2": 9, "log10": 10,
             "floor": 11, "ceil": 12, "round": 13, "trunc": 14, "abs": 15, "signum": 16, "octet_length": 17, "concat": 18,
             "lower": 19, "upper": 20, "trim": 21, "ltrim": 22, "rtrim": 23, "to_timestamp": 24, "array": 25, "nullif": 26,
             "date_trunc": 27, "md5": 28, "sha224": 29, "sha256": 30, "sha384": 31, "sha512": 32}
AGG_FN = {"MIN": 0, "MAX": 1, "SUM": 2, "AVG": 3, "COUNT": 4}


class Between:
    """logical-only BetweenNode (ballista.proto:133-138): the decoder expands it to `x >= low AND x <= high`"""

    def __init__(self, expr, low, high, negated=False):
        self.expr, self.low, self.high, self.negated = expr, low, high, negated


class Alias:
    def __init__(self, expr, alias):
        self.expr, self.alias = expr, alias


def expr(e) -> bytes:
    """LogicalExprNode"""
    if isinstance(e, E.Column):
        return f_str(1, e.name, always=True)
    if isinstance(e, Alias):
        return f_bytes(2, f_bytes(1, expr(e.expr)) + f_str(2, e.alias))
    if isinstance(e, E.Literal):
        return f_bytes(3, scalar_value(e))
    if isinstance(e, E.BinaryExpr):
        return f_bytes(4, f_bytes(1, expr(e.left)) + f_bytes(2, expr(e.right)) + f_str(3, e.op))
    if isinstance(e, E.IsNullExpr):
        return f_bytes(6, f_bytes(1, expr(e.expr)))
    if isinstance(e, E.IsNotNullExpr):
        return f_bytes(7, f_bytes(1, expr(e.expr)))
    if isinstance(e, E.NotExpr):
        return f_bytes(8, f_bytes(1, expr(e.expr)))
    if isinstance(e, Between):
        return f_bytes(9, f_bytes(1, expr(e.expr)) + f_varint(2, 1 if e.negated else 0) + f_bytes(3, expr(e.low)) + f_bytes(4, expr(e.high)))
    if isinstance(e, E.CaseExpr):
        body = f_bytes(1, expr(e.expr)) if e.expr is not None else b""
        for w, t in e.when_then:
            body += f_bytes(2, f_bytes(1, expr(w)) + f_bytes(2, expr(t)))
        if e.else_expr is not None:
            body += f_bytes(3, expr(e.else_expr))
        return f_bytes(10, body)
    if isinstance(e, E.CastExpr):
        return f_bytes(11, f_bytes(1, expr(e.expr)) + f_bytes(2, arrow_type(e.dtype)))
    if isinstance(e, E.NegativeExpr):
        return f_bytes(13, f_bytes(1, expr(e.expr)))
    if isinstance(e, E.InListExpr):
        return f_bytes(14, f_bytes(1, expr(e.expr)) + b"".join(f_bytes(2, expr(v)) for v in e.list) + f_varint(3, 1 if e.negated else 0))
    if isinstance(e, E.ScalarFunctionExpr):
        return f_bytes(16, f_varint(1, SCALAR_FN[e.fun]) + b"".join(f_bytes(2, expr(a)) for a in e.args))
    raise TypeError(f"cannot encode {e!r}")


def aggregate_expr(a: E.AggregateExpr) -> bytes:
    return f_bytes(5, f_varint(1, AGG_FN[a.fun]) + f_bytes(2, expr(a.expr)))


def sort_expr(s: E.PhysicalSortExpr) -> bytes:
    return f_bytes(12, f_bytes(1, expr(s.expr)) + f_varint(2, 0 if s.descending else 1) + f_varint(3, 1 if s.nulls_first else 0))


# ---- PhysicalPlanNode -----------------------------------------------------------------------------------------------------

def plan(p) -> bytes:
    k = type(p).__name__
    if k == "MemoryExec":                            # -> CsvScanExecNode (field 2)
        sch = p.schema()
        body = (f_str(1, p.name) + f_packed(2, list(range(len(sch)))) + f_bytes(3, schema(sch)) + f_str(4, ".tbl") +
                f_varint(6, 32768) + f_str(7, "|"))
        return f_bytes(2, body)
    if k == "EmptyExec":
        return f_bytes(3, f_varint(1, 1 if p.produce_one_row else 0) + f_bytes(2, schema(p.schema())))
    if k == "ProjectionExec":
        return f_bytes(4, f_bytes(1, plan(p.input)) + b"".join(f_bytes(2, expr(e)) for e, _ in p.exprs) +
                       b"".join(f_str(3, n, always=True) for _, n in p.exprs))
    if k == "GlobalLimitExec":
        return f_bytes(6, f_bytes(1, plan(p.input)) + f_varint(2, p.limit))
    if k == "LocalLimitExec":
        return f_bytes(7, f_bytes(1, plan(p.input)) + f_varint(2, p.limit))
    if k == "HashAggregateExec":
        body = b"".join(f_bytes(1, expr(e)) for e, _ in p.group_expr)
        body += b"".join(f_bytes(2, aggregate_expr(a)) for a in p.aggr_expr)
        body += f_varint(3, 0 if p.mode == "Partial" else 1)
        body += f_bytes(4, plan(p.input))
        body += b"".join(f_str(5, n, always=True) for _, n in p.group_expr)
        body += b"".join(f_str(6, a.name, always=True) for a in p.aggr_expr)
        body += f_bytes(7, schema(p.input.schema()))
        return f_bytes(8, body)
    if k == "HashJoinExec":
        body = f_bytes(1, plan(p.left)) + f_bytes(2, plan(p.right))
        body += b"".join(f_bytes(3, f_str(1, a) + f_str(2, b)) for a, b in p.on)
        body += f_varint(4, {"Inner": 0, "Left": 1, "Right": 2}[p.join_type])
        return f_bytes(9, body)
    if k == "SortExec":
        return f_bytes(11, f_bytes(1, plan(p.input)) + b"".join(f_bytes(2, sort_expr(s)) for s in p.expr))
    if k == "CoalesceBatchesExec":
        return f_bytes(12, f_bytes(1, plan(p.input)) + f_varint(2, p.target_batch_size))
    if k == "FilterExec":
        return f_bytes(13, f_bytes(1, plan(p.input)) + f_bytes(2, expr(p.predicate)))
    if k == "MergeExec":
        return f_bytes(14, f_bytes(1, plan(p.input)))
    if k == "RepartitionExec":
        part = p.partitioning
        body = f_bytes(1, plan(p.input))
        if part.scheme == 2:
            body += f_bytes(3, b"".join(f_bytes(1, expr(e)) for e in part.exprs) + f_varint(2, part.count))
        elif part.scheme == 1:
            body += f_varint(2, part.count, always=True)
        else:
            body += f_varint(4, part.count, always=True)
        return f_bytes(16, body)
    raise TypeError(f"cannot encode plan node {k}")


def shuffle_reader(locations, fields) -> bytes:
    """ShuffleReaderExecNode (field 10): locations = [(job_id, stage_id, partition_id, executor_id, host, port)]"""
    body = b""
    for job, stage, part, ex_id, host, port in locations:
        pid = f_str(1, job) + f_varint(2, stage) + f_varint(4, part)
        meta = f_str(1, ex_id) + f_str(2, host) + f_varint(3, port)
        body += f_bytes(1, f_bytes(1, pid) + f_bytes(2, meta))
    return f_bytes(10, body + f_bytes(2, schema(fields)))


def unresolved_shuffle(stage_ids, fields, partition_count) -> bytes:
    return f_bytes(15, f_packed(1, stage_ids) + f_bytes(2, schema(fields)) + f_varint(3, partition_count))


def parquet_scan(filenames, projection, num_partitions=1, batch_size=32768) -> bytes:
    return f_bytes(1, b"".join(f_str(1, f, always=True) for f in filenames) + f_packed(2, projection) + f_varint(3, num_partitions) +
                   f_varint(4, batch_size))
