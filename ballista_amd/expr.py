"""Physical expressions — the host-side mirror of the expression kinds a Ballista executor
can receive.

The set is exactly what the reference's physical-plan serde accepts
(rust/core/src/serde/physical_plan/to_proto.rs:380-511; rebuilt by `compile_expr`,
from_proto.rs:348-364): Column, Literal, BinaryExpr, CastExpr, CaseExpr, NotExpr,
IsNullExpr, IsNotNullExpr, InListExpr, NegativeExpr, ScalarFunctionExpr — and the
aggregate kinds of to_proto.rs:348-378 (Sum, Avg, Count; Min/Max exist in the proto but
are not serialisable there, we accept them anyway).  Binary operator names are the wire
strings of rust/core/src/serde/logical_plan/from_proto.rs:937-957.

These classes are plain descriptions; `ballista_amd.plan` lowers them to the flat VM
program the C-ABI takes (include/ballista_hip.h, `bhip_expr_*`).  As in DataFusion's
*physical* plans both sides of a BinaryExpr must already have equal types — use
`coerce()` to insert the casts DataFusion's planner would.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

# Arrow type names as in rust/benchmarks/tpch/src/main.rs:267-360 / SURVEY §8(b)
INT32, INT64, UINT8, UINT64 = "Int32", "Int64", "UInt8", "UInt64"
FLOAT64, DATE32, BOOLEAN, UTF8 = "Float64", "Date32", "Boolean", "Utf8"
# the other primitive types the serde ships (rust/core/proto/ballista.proto:755-790)
INT8, INT16, UINT16, UINT32, FLOAT32, DATE64 = "Int8", "Int16", "UInt16", "UInt32", "Float32", "Date64"
TIMESTAMP_S, TIMESTAMP_MS = "Timestamp(Second)", "Timestamp(Millisecond)"
TIMESTAMP_US, TIMESTAMP_NS = "Timestamp(Microsecond)", "Timestamp(Nanosecond)"
BINARY = "Binary"              # schemas only: variable-length bytes (the sha* digests) in the buffer layout of Utf8
LARGE_UTF8 = "LargeUtf8"       # schemas only (Arrow / IPC form of a Utf8 device column with 64-bit offsets)
ALL_TYPES = (INT32, INT64, UINT8, UINT64, FLOAT64, DATE32, BOOLEAN, UTF8, INT8, INT16, UINT16, UINT32, FLOAT32, DATE64,
             TIMESTAMP_S, TIMESTAMP_MS, TIMESTAMP_US, TIMESTAMP_NS)

BINARY_OPS = ("And", "Or", "Eq", "NotEq", "LtEq", "Lt", "Gt", "GtEq",
              "Plus", "Minus", "Multiply", "Divide", "Like", "NotLike")
COMPARE_OPS = ("Eq", "NotEq", "LtEq", "Lt", "Gt", "GtEq")
ARITH_OPS = ("Plus", "Minus", "Multiply", "Divide")

MATH_FUNCTIONS = ("sqrt", "abs", "floor", "ceil", "round", "trunc", "signum",
                  "exp", "ln", "log2", "log10", "sin", "cos", "tan", "asin", "acos", "atan")
# Utf8 -> Utf8 (rust/core/src/serde/logical_plan/from_proto.rs:914-918) and Utf8 -> Int32 (:910)
STRING_FUNCTIONS = ("lower", "upper", "trim", "ltrim", "rtrim")
# Utf8 -> Binary digests (from_proto.rs:924-927)
SHA_FUNCTIONS = ("sha224", "sha256", "sha384", "sha512")
SCALAR_FUNCTIONS = MATH_FUNCTIONS + STRING_FUNCTIONS + SHA_FUNCTIONS + ("octet_length",)


class PhysicalExpr:
    """Base class (DataFusion `PhysicalExpr`)."""

    # small builder sugar used by tests / tpch plans
    def _bin(self, op, other):
        return BinaryExpr(self, op, other if isinstance(other, PhysicalExpr) else lit(other))

    def __add__(self, o): return self._bin("Plus", o)
    def __sub__(self, o): return self._bin("Minus", o)
    def __mul__(self, o): return self._bin("Multiply", o)
    def __truediv__(self, o): return self._bin("Divide", o)
    def __lt__(self, o): return self._bin("Lt", o)
    def __le__(self, o): return self._bin("LtEq", o)
    def __gt__(self, o): return self._bin("Gt", o)
    def __ge__(self, o): return self._bin("GtEq", o)
    def eq(self, o): return self._bin("Eq", o)
    def ne(self, o): return self._bin("NotEq", o)
    def and_(self, o): return self._bin("And", o)
    def or_(self, o): return self._bin("Or", o)
    def __neg__(self): return NegativeExpr(self)


@dataclass(eq=False)
class Column(PhysicalExpr):
    name: str


@dataclass(eq=False)
class Literal(PhysicalExpr):
    """ScalarValue (rust/core/proto/ballista.proto:685-709). value None = typed NULL."""
    value: object
    dtype: str


@dataclass(eq=False)
class BinaryExpr(PhysicalExpr):
    left: PhysicalExpr
    op: str
    right: PhysicalExpr

    def __post_init__(self):
        if self.op not in BINARY_OPS:
            raise ValueError(f"Unsupported binary operator '{self.op}'")


@dataclass(eq=False)
class CastExpr(PhysicalExpr):
    expr: PhysicalExpr
    dtype: str


@dataclass(eq=False)
class CaseExpr(PhysicalExpr):
    """CASE [expr] WHEN w THEN t ... [ELSE e] END."""
    expr: Optional[PhysicalExpr]
    when_then: Sequence[Tuple[PhysicalExpr, PhysicalExpr]]
    else_expr: Optional[PhysicalExpr] = None


@dataclass(eq=False)
class NotExpr(PhysicalExpr):
    expr: PhysicalExpr


@dataclass(eq=False)
class IsNullExpr(PhysicalExpr):
    expr: PhysicalExpr


@dataclass(eq=False)
class IsNotNullExpr(PhysicalExpr):
    expr: PhysicalExpr


@dataclass(eq=False)
class InListExpr(PhysicalExpr):
    expr: PhysicalExpr
    list: Sequence[PhysicalExpr]
    negated: bool = False


@dataclass(eq=False)
class NegativeExpr(PhysicalExpr):
    expr: PhysicalExpr


@dataclass(eq=False)
class ScalarFunctionExpr(PhysicalExpr):
    fun: str
    args: Sequence[PhysicalExpr]

    def __post_init__(self):
        if self.fun not in SCALAR_FUNCTIONS:
            raise NotImplementedError(f"scalar function '{self.fun}' is not supported")


# ---- aggregates (DataFusion `AggregateExpr`) -----------------------------------------

@dataclass(eq=False)
class AggregateExpr:
    fun: str            # "SUM" | "AVG" | "COUNT" | "MIN" | "MAX"
    expr: PhysicalExpr
    name: str

    def __post_init__(self):
        if self.fun not in ("SUM", "AVG", "COUNT", "MIN", "MAX"):
            raise NotImplementedError(f"aggregate function '{self.fun}' is not supported")


def Sum(expr, name): return AggregateExpr("SUM", expr, name)
def Avg(expr, name): return AggregateExpr("AVG", expr, name)
def Count(expr, name): return AggregateExpr("COUNT", expr, name)
def Min(expr, name): return AggregateExpr("MIN", expr, name)
def Max(expr, name): return AggregateExpr("MAX", expr, name)


@dataclass(eq=False)
class PhysicalSortExpr:
    """ballista.proto PhysicalSortExprNode; SQL default ASC NULLS FIRST (Appendix A)."""
    expr: PhysicalExpr
    descending: bool = False
    nulls_first: bool = True


# ---- helpers -------------------------------------------------------------------------

def col(name: str) -> Column:
    return Column(name)


def lit(v, dtype: Optional[str] = None) -> Literal:
    if isinstance(v, Literal):
        return v
    if dtype is None:
        if isinstance(v, bool):
            dtype = BOOLEAN
        elif isinstance(v, int):
            dtype = INT64
        elif isinstance(v, float):
            dtype = FLOAT64
        elif isinstance(v, str):
            dtype = UTF8
        else:
            raise TypeError(f"cannot infer literal type of {v!r}")
    return Literal(v, dtype)


def date32(s: str) -> Literal:
    """`date 'YYYY-MM-DD'` literal as days since 1970-01-01 (Appendix A)."""
    import datetime
    d = datetime.date.fromisoformat(s)
    return Literal((d - datetime.date(1970, 1, 1)).days, DATE32)


# DataFusion's numerical_coercion (4.0.0-SNAPSHOT, physical_plan/expressions/coercion.rs): equal types stay, otherwise the FIRST of
# Float64, Float32, Int64, Int32, Int16, Int8, UInt64, UInt32, UInt16, UInt8 that either side has
_NUMERIC_RANK = {FLOAT64: 10, FLOAT32: 9, INT64: 8, INT32: 7, INT16: 6, INT8: 5, UINT64: 4, UINT32: 3, UINT16: 2, UINT8: 1}
_TEMPORAL = (DATE32, DATE64, TIMESTAMP_S, TIMESTAMP_MS, TIMESTAMP_US, TIMESTAMP_NS)
_INT_RANGE = {INT8: (-2**7, 2**7 - 1), INT16: (-2**15, 2**15 - 1), INT32: (-2**31, 2**31 - 1), INT64: (-2**63, 2**63 - 1),
              UINT8: (0, 2**8 - 1), UINT16: (0, 2**16 - 1), UINT32: (0, 2**32 - 1), UINT64: (0, 2**64 - 1), DATE32: (-2**31, 2**31 - 1),
              DATE64: (-2**63, 2**63 - 1), TIMESTAMP_S: (-2**63, 2**63 - 1), TIMESTAMP_MS: (-2**63, 2**63 - 1),
              TIMESTAMP_US: (-2**63, 2**63 - 1), TIMESTAMP_NS: (-2**63, 2**63 - 1)}


def expr_type(e: PhysicalExpr, schema: dict) -> str:
    """Result type of `e` over `schema` (name -> Arrow type name)."""
    if isinstance(e, Column):
        if e.name not in schema:
            raise KeyError(f"No field named '{e.name}'")
        return schema[e.name]
    if isinstance(e, Literal):
        return e.dtype
    if isinstance(e, BinaryExpr):
        if e.op in ARITH_OPS:
            return expr_type(e.left, schema)
        return BOOLEAN
    if isinstance(e, CastExpr):
        return e.dtype
    if isinstance(e, CaseExpr):
        return expr_type(e.when_then[0][1], schema)
    if isinstance(e, (NotExpr, IsNullExpr, IsNotNullExpr, InListExpr)):
        return BOOLEAN
    if isinstance(e, NegativeExpr):
        return expr_type(e.expr, schema)
    if isinstance(e, ScalarFunctionExpr):
        return UTF8 if e.fun in STRING_FUNCTIONS else BINARY if e.fun in SHA_FUNCTIONS else INT32 if e.fun == "octet_length" else FLOAT64
    raise TypeError(f"not a PhysicalExpr: {e!r}")


def coerce(e: PhysicalExpr, schema: dict) -> PhysicalExpr:
    """Insert the casts DataFusion's physical planner inserts so that both sides of every
    BinaryExpr / InList / Case branch have one type (numeric literals and columns are
    widened towards the higher-ranked side; e.g. Int64 literal 1 -> Float64)."""
    def cast_to(x, t):
        if expr_type(x, schema) == t:
            return x
        if isinstance(x, Literal) and x.value is not None and (t in _NUMERIC_RANK or t in _TEMPORAL) and x.dtype in _NUMERIC_RANK:
            # a numeric literal is re-typed when the value survives; otherwise it stays a CAST (NULL at run time)
            if t in (FLOAT64, FLOAT32):
                import struct
                v = float(x.value)
                return Literal(struct.unpack("f", struct.pack("f", v))[0] if t == FLOAT32 else v, t)
            if not isinstance(x.value, float) or float(x.value).is_integer():
                v = int(x.value)
                if _INT_RANGE[t][0] <= v <= _INT_RANGE[t][1]:
                    return Literal(v, t)
        return CastExpr(x, t)

    def common(a, b):
        if a == b:
            return a
        if a in _NUMERIC_RANK and b in _NUMERIC_RANK:
            return a if _NUMERIC_RANK[a] >= _NUMERIC_RANK[b] else b
        if a in _TEMPORAL and b in _NUMERIC_RANK and b not in (FLOAT64, FLOAT32):
            return a
        if b in _TEMPORAL and a in _NUMERIC_RANK and a not in (FLOAT64, FLOAT32):
            return b
        raise TypeError(f"cannot coerce {a} and {b}")

    if isinstance(e, BinaryExpr):
        l, r = coerce(e.left, schema), coerce(e.right, schema)
        if e.op in ARITH_OPS or e.op in COMPARE_OPS:
            t = common(expr_type(l, schema), expr_type(r, schema))
            l, r = cast_to(l, t), cast_to(r, t)
        return BinaryExpr(l, e.op, r)
    if isinstance(e, CastExpr):
        return CastExpr(coerce(e.expr, schema), e.dtype)
    if isinstance(e, NotExpr):
        return NotExpr(coerce(e.expr, schema))
    if isinstance(e, IsNullExpr):
        return IsNullExpr(coerce(e.expr, schema))
    if isinstance(e, IsNotNullExpr):
        return IsNotNullExpr(coerce(e.expr, schema))
    if isinstance(e, NegativeExpr):
        return NegativeExpr(coerce(e.expr, schema))
    if isinstance(e, InListExpr):
        x = coerce(e.expr, schema)
        t = expr_type(x, schema)
        return InListExpr(x, [cast_to(coerce(v, schema), t) for v in e.list], e.negated)
    if isinstance(e, CaseExpr):
        base = coerce(e.expr, schema) if e.expr is not None else None
        wt = [(coerce(w, schema), coerce(t, schema)) for w, t in e.when_then]
        el = coerce(e.else_expr, schema) if e.else_expr is not None else None
        t = expr_type(wt[0][1], schema)
        for _, th in wt[1:]:
            t = common(t, expr_type(th, schema))
        if el is not None:
            t = common(t, expr_type(el, schema))
        wt = [(w if base is None else cast_to(w, expr_type(base, schema)), cast_to(th, t)) for w, th in wt]
        return CaseExpr(base, wt, cast_to(el, t) if el is not None else None)
    if isinstance(e, ScalarFunctionExpr):
        if e.fun not in MATH_FUNCTIONS:
            return ScalarFunctionExpr(e.fun, [coerce(a, schema) for a in e.args])
        return ScalarFunctionExpr(e.fun, [cast_to(coerce(a, schema), FLOAT64) for a in e.args])
    return e
