"""TEST INFRASTRUCTURE — CPU restatement of the reference's operator semantics.
PARITY UNPINNED BY THE REFERENCE (see oracle/__init__.py).

Each function restates one DataFusion operator the Ballista executor can be handed
(operator inventory: rust/core/src/serde/physical_plan/from_proto.rs:58-346; semantics:
SURVEY.md Appendix A).  Evaluation is column-at-a-time with one materialised numpy array
per expression node — numpy's elementwise f64 ops are separately rounded IEEE operations,
exactly like arrow-rs' arithmetic kernels (no FMA).

Batches are `dict name -> OCol`; expressions are duck-typed on the class *names* of
DataFusion's physical expressions (Column, Literal, BinaryExpr, CastExpr, CaseExpr,
NotExpr, IsNullExpr, IsNotNullExpr, InListExpr, NegativeExpr, ScalarFunctionExpr), so the
oracle imports nothing from the product.
"""
from __future__ import annotations

import ctypes
import math
import os
import re
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NP_TYPES = {"Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "UInt64": np.uint64,
            "Float64": np.float64, "Date32": np.int32, "Boolean": np.bool_,
            # the other primitive types of the serde (rust/core/proto/ballista.proto:755-790)
            "Int8": np.int8, "Int16": np.int16, "UInt16": np.uint16, "UInt32": np.uint32, "Float32": np.float32,
            "Date64": np.int64, "Timestamp(Second)": np.int64, "Timestamp(Millisecond)": np.int64,
            "Timestamp(Microsecond)": np.int64, "Timestamp(Nanosecond)": np.int64}
FLOAT_TYPES = ("Float64", "Float32")
UNSIGNED_TYPES = ("UInt8", "UInt16", "UInt32", "UInt64")
# units per day of the temporal types (arrow's temporal casts multiply / divide by the ratio)
TEMPORAL_UNITS = {"Date32": 1, "Timestamp(Second)": 86400, "Date64": 86400000, "Timestamp(Millisecond)": 86400000,
                  "Timestamp(Microsecond)": 86400000000, "Timestamp(Nanosecond)": 86400000000000}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        _LIB = ctypes.CDLL(path)
    return _LIB


class OCol:
    """One column: Arrow type name, values (numpy array; Utf8 = object array of str),
    validity (bool array, None = all valid)."""

    def __init__(self, dtype, values, valid=None):
        self.dtype = dtype
        if dtype == "Utf8":
            self.values = np.asarray(values, dtype=object)
        else:
            self.values = np.asarray(values, dtype=NP_TYPES[dtype])
        if valid is not None:
            valid = np.asarray(valid, dtype=np.bool_)
            if valid.all():
                valid = None
        self.valid = valid

    def __len__(self):
        return len(self.values)

    def take(self, idx):
        return OCol(self.dtype, self.values[idx], None if self.valid is None else self.valid[idx])

    def is_valid(self):
        return np.ones(len(self), np.bool_) if self.valid is None else self.valid

    def to_pylist(self):
        v = self.is_valid()
        return [(x.item() if hasattr(x, "item") else x) if ok else None for x, ok in zip(self.values, v)]


def batch_len(batch):
    for c in batch.values():
        return len(c)
    return 0


def _and_valid(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a & b


# ---- expressions -----------------------------------------------------------------------

def _like_to_regex(p):
    out = []
    for ch in p:
        if ch == "%":
            out.append(".*")
        elif ch == "_":
            out.append(".")
        else:
            out.append(re.escape(ch))
    return re.compile("^" + "".join(out) + "$", re.S)


def evaluate(e, batch) -> OCol:
    n = batch_len(batch)
    k = type(e).__name__
    if k == "Column":
        if e.name not in batch:
            raise KeyError(f"No field named '{e.name}'")
        return batch[e.name]
    if k == "Literal":
        if e.value is None:
            z = "" if e.dtype == "Utf8" else 0
            return OCol(e.dtype, [z] * n, np.zeros(n, np.bool_))
        if e.dtype == "Utf8":
            return OCol("Utf8", [e.value] * n)
        return OCol(e.dtype, np.full(n, e.value, dtype=NP_TYPES[e.dtype]))
    if k == "BinaryExpr":
        l, r = evaluate(e.left, batch), evaluate(e.right, batch)
        return _binary(e.op, l, r)
    if k == "CastExpr":
        return _cast(evaluate(e.expr, batch), e.dtype)
    if k == "NotExpr":
        x = evaluate(e.expr, batch)
        if x.dtype != "Boolean":
            raise TypeError("NOT requires Boolean")
        return OCol("Boolean", ~x.values, x.valid)
    if k == "IsNullExpr":
        return OCol("Boolean", ~evaluate(e.expr, batch).is_valid())
    if k == "IsNotNullExpr":
        return OCol("Boolean", evaluate(e.expr, batch).is_valid().copy())
    if k == "NegativeExpr":
        x = evaluate(e.expr, batch)
        with np.errstate(all="ignore"):
            return OCol(x.dtype, -x.values, x.valid)
    if k == "InListExpr":
        x = evaluate(e.expr, batch)
        acc = None
        for item in e.list:
            c = _binary("Eq", x, evaluate(item, batch))
            acc = c if acc is None else _binary("Or", acc, c)
        if e.negated:
            acc = OCol("Boolean", ~acc.values, acc.valid)
        return acc
    if k == "CaseExpr":
        return _case(e, batch)
    if k == "ScalarFunctionExpr":
        args = [evaluate(a, batch) for a in e.args]
        return _scalar_fn(e.fun, args)
    raise NotImplementedError(f"expression kind {k}")


def _binary(op, l: OCol, r: OCol) -> OCol:
    if op in ("And", "Or"):
        if l.dtype != "Boolean" or r.dtype != "Boolean":
            raise TypeError(f"{op} requires Boolean operands")
        lv, rv = l.is_valid(), r.is_valid()
        a, b = l.values & lv, r.values & rv          # definitely true
        fa, fb = (~l.values) & lv, (~r.values) & rv   # definitely false
        if op == "And":                               # Kleene
            val = a & b
            valid = (a & b) | fa | fb
        else:
            val = a | b
            valid = a | b | (fa & fb)
        return OCol("Boolean", val, valid)
    if l.dtype != r.dtype:
        raise TypeError(f"Cannot evaluate binary expression {op} with types {l.dtype} and {r.dtype}")
    valid = _and_valid(l.valid, r.valid)
    if op in ("Like", "NotLike"):
        if l.dtype != "Utf8":
            raise TypeError("LIKE requires Utf8")
        out = np.zeros(len(l), np.bool_)
        cache = {}
        for i, (s, p) in enumerate(zip(l.values, r.values)):
            rx = cache.get(p) or cache.setdefault(p, _like_to_regex(p))
            out[i] = rx.match(s) is not None
        return OCol("Boolean", ~out if op == "NotLike" else out, valid)
    if op in ("Eq", "NotEq", "Lt", "LtEq", "Gt", "GtEq"):
        a, b = l.values, r.values
        if l.dtype == "Utf8":
            a = np.array([s.encode() for s in a], dtype=object)
            b = np.array([s.encode() for s in b], dtype=object)
        with np.errstate(all="ignore"):
            res = {"Eq": a == b, "NotEq": a != b, "Lt": a < b, "LtEq": a <= b,
                   "Gt": a > b, "GtEq": a >= b}[op]
        return OCol("Boolean", np.asarray(res, dtype=np.bool_), valid)
    if l.dtype in ("Utf8", "Boolean"):
        raise TypeError(f"arithmetic on {l.dtype}")
    a, b = l.values, r.values
    with np.errstate(all="ignore"):
        if op == "Plus":
            res = a + b
        elif op == "Minus":
            res = a - b
        elif op == "Multiply":
            res = a * b
        elif op == "Divide":
            if l.dtype in FLOAT_TYPES:
                res = a / b
            else:
                live = r.is_valid() & l.is_valid()
                if np.any((b == 0) & live):
                    raise ZeroDivisionError("Divide by zero error")
                bb = np.where(b == 0, 1, b)
                # Rust integer division truncates toward zero
                res = (np.sign(a) * np.sign(bb) * (np.abs(a) // np.abs(bb))).astype(a.dtype)
        else:
            raise NotImplementedError(op)
    return OCol(l.dtype, res, valid)


def _cast(x: OCol, to: str) -> OCol:
    if x.dtype == to:
        return x
    if x.dtype == "Utf8":
        if to == "Date32":
            import datetime
            vals = [(datetime.date.fromisoformat(s) - datetime.date(1970, 1, 1)).days for s in x.values]
            return OCol("Date32", vals, x.valid)
        raise NotImplementedError(f"cast Utf8 -> {to}")
    if to == "Utf8":
        raise NotImplementedError("cast to Utf8")
    src = x.values
    valid = x.valid
    if x.dtype in TEMPORAL_UNITS and to in TEMPORAL_UNITS:
        uf, ut = TEMPORAL_UNITS[x.dtype], TEMPORAL_UNITS[to]
        w = src.astype(np.int64)
        if ut >= uf:
            w = w * (ut // uf)
        else:
            d = uf // ut
            w = (np.sign(w) * (np.abs(w) // d)).astype(np.int64)          # Rust integer division truncates toward zero
        return OCol(to, w.astype(NP_TYPES[to]), valid)
    if x.dtype in FLOAT_TYPES and to in FLOAT_TYPES:
        return OCol(to, src.astype(NP_TYPES[to]), valid)
    if x.dtype in FLOAT_TYPES:
        info = np.iinfo(NP_TYPES[to]) if to != "Boolean" else None
        with np.errstate(all="ignore"):
            t = np.trunc(src)
            ok = np.isfinite(src)
            if info is not None:
                wide64 = NP_TYPES[to] in (np.int64, np.uint64)
                signed64 = NP_TYPES[to] == np.int64
                ok &= (t >= float(info.min)) & (t <= float(info.max)) if not wide64 else \
                    (t >= -9.223372036854775808e18 if signed64 else t >= 0) & \
                    (t < (9.223372036854775808e18 if signed64 else 1.8446744073709552e19))
            res = np.where(ok, t, 0).astype(NP_TYPES[to])
        valid = _and_valid(valid, ok if not ok.all() else None)
        return OCol(to, res, valid)
    if to in FLOAT_TYPES:
        return OCol(to, src.astype(NP_TYPES[to]), valid)               # (numpy rounds an int64 to float32 once, as Rust's `as f32`)
    if to == "Boolean":
        return OCol(to, src != 0, valid)
    # int -> int: out-of-range becomes NULL (arrow-rs numeric cast)
    info = np.iinfo(NP_TYPES[to])
    wide = src.astype(object) if x.dtype == "UInt64" else src.astype(np.int64)
    ok = np.array([(info.min <= int(v) <= info.max) for v in wide], dtype=np.bool_) if len(wide) else np.zeros(0, np.bool_)
    res = np.where(ok, src, 0).astype(NP_TYPES[to])
    return OCol(to, res, _and_valid(valid, None if ok.all() else ok))


def _case(e, batch) -> OCol:
    n = batch_len(batch)
    base = evaluate(e.expr, batch) if e.expr is not None else None
    thens = [evaluate(t, batch) for _, t in e.when_then]
    dtype = thens[0].dtype
    if dtype == "Utf8":
        out = np.array([""] * n, dtype=object)
    else:
        out = np.zeros(n, NP_TYPES[dtype])
    out_valid = np.zeros(n, np.bool_)
    undecided = np.ones(n, np.bool_)
    for (w, _), t in zip(e.when_then, thens):
        c = evaluate(w, batch)
        if base is not None:
            c = _binary("Eq", base, c)
        hit = undecided & c.values & c.is_valid()
        out[hit] = t.values[hit]
        out_valid[hit] = t.is_valid()[hit]
        undecided &= ~hit
    if e.else_expr is not None:
        el = evaluate(e.else_expr, batch)
        out[undecided] = el.values[undecided]
        out_valid[undecided] = el.is_valid()[undecided]
    return OCol(dtype, out, out_valid)


# Unicode White_Space: what Rust's str::trim / trim_start / trim_end strip (DataFusion 4.0 string_expressions.rs)
_WS = "\t\n\x0b\x0c\r \x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000"


def _scalar_fn(fun, args) -> OCol:
    x = args[0]
    if fun in ("lower", "upper", "trim", "ltrim", "rtrim", "octet_length"):
        if x.dtype != "Utf8":
            raise TypeError(f"{fun} requires Utf8")
        if fun == "octet_length":
            return OCol("Int32", [len(s.encode()) for s in x.values], x.valid)
        f = {"lower": str.lower, "upper": str.upper, "trim": lambda s: s.strip(_WS), "ltrim": lambda s: s.lstrip(_WS),
             "rtrim": lambda s: s.rstrip(_WS)}[fun]
        ok = x.is_valid()
        return OCol("Utf8", [f(s) if k else "" for s, k in zip(x.values, ok)], x.valid)
    if x.dtype != "Float64":
        raise TypeError(f"{fun} requires Float64")
    v = x.values
    with np.errstate(all="ignore"):
        if fun == "round":
            res = np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5))   # Rust f64::round: half away from zero
            big = np.abs(v) >= 4503599627370496.0
            res = np.where(big | ~np.isfinite(v), v, res)
            # floor(v+0.5) double-rounds for 0.49999999999999994; fix like f64::round
            res = np.where(np.abs(v) < 0.5, np.copysign(0.0, v), res)
        elif fun == "signum":
            res = np.where(np.isnan(v), v, np.copysign(1.0, v))            # Rust f64::signum
        else:
            f = {"sqrt": np.sqrt, "abs": np.abs, "floor": np.floor, "ceil": np.ceil, "trunc": np.trunc,
                 "exp": np.exp, "ln": np.log, "log2": np.log2, "log10": np.log10, "sin": np.sin,
                 "cos": np.cos, "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan}[fun]
            res = f(v)
    return OCol("Float64", res, x.valid)


# ---- operators -------------------------------------------------------------------------

def filter_batch(batch, predicate):
    """FilterExec (from_proto.rs:81-92): keep rows whose predicate is true AND valid; order
    preserved; every input column carried."""
    p = evaluate(predicate, batch)
    if p.dtype != "Boolean":
        raise TypeError("Filter predicate must return boolean values")
    idx = np.nonzero(p.values & p.is_valid())[0]
    return OrderedDict((k, c.take(idx)) for k, c in batch.items())


def project(batch, exprs_names):
    """ProjectionExec (from_proto.rs:69-80)."""
    return OrderedDict((name, evaluate(e, batch)) for e, name in exprs_names)


def _key_tuples(cols):
    lists = [c.to_pylist() for c in cols]
    return list(zip(*lists)) if lists else []


def _factorize(cols, n):
    """dense group ids in first-appearance order (NULL is a group value of its own)."""
    if not cols:
        return np.zeros(n, np.int32), [()]
    ids = np.empty(n, np.int32)
    table = {}
    keys = []
    for i, t in enumerate(_key_tuples(cols)):
        g = table.get(t)
        if g is None:
            g = len(keys)
            table[t] = g
            keys.append(t)
        ids[i] = g
    return ids, keys


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def _group_sum(col: OCol, gid, ngroups, batch_rows):
    L = lib()
    n = len(col)
    valid = None if col.valid is None else col.valid.astype(np.uint8)
    cnt = np.zeros(max(ngroups, 1), np.uint64)
    has = np.zeros(max(ngroups, 1), np.uint8)
    gid = np.ascontiguousarray(gid, np.int32)
    if col.dtype in FLOAT_TYPES:
        v = np.ascontiguousarray(col.values, np.float64)
        out = np.zeros(max(ngroups, 1), np.float64)
        L.oracle_batched_group_sum_f64(_ptr(v, ctypes.c_double), _ptr(valid, ctypes.c_uint8),
                                       _ptr(gid, ctypes.c_int32), ctypes.c_int64(n), ctypes.c_int64(batch_rows),
                                       ctypes.c_int32(ngroups), _ptr(out, ctypes.c_double),
                                       _ptr(cnt, ctypes.c_uint64), _ptr(has, ctypes.c_uint8))
        return out[:ngroups], cnt[:ngroups], has[:ngroups].astype(bool)
    if col.dtype in NP_TYPES:
        v = np.ascontiguousarray(col.values.astype(np.int64))
        out = np.zeros(max(ngroups, 1), np.int64)
        L.oracle_batched_group_sum_i64(_ptr(v, ctypes.c_int64), _ptr(valid, ctypes.c_uint8),
                                       _ptr(gid, ctypes.c_int32), ctypes.c_int64(n),
                                       ctypes.c_int32(ngroups), _ptr(out, ctypes.c_int64),
                                       _ptr(cnt, ctypes.c_uint64), _ptr(has, ctypes.c_uint8))
        return out[:ngroups], cnt[:ngroups], has[:ngroups].astype(bool)
    raise TypeError(f"SUM/AVG over {col.dtype}")


def _sum_type(t):
    """sum_return_type of DataFusion 4.0: signed -> Int64, unsigned -> UInt64, Float32 -> Float32, Float64 -> Float64.
    (SUM(Float32) is added here in double and rounded once; the reference keeps a running float — not reproducible bit for bit,
    compared with a float32-sized tolerance in the tests)"""
    if t in FLOAT_TYPES:
        return t
    if t in UNSIGNED_TYPES:
        return "UInt64"
    return "Int64"


def _minmax(col: OCol, gid, ngroups, is_min):
    out = [None] * ngroups
    ok = col.is_valid()
    for v, g, o in zip(col.values, gid, ok):
        if not o:
            continue
        if col.dtype in FLOAT_TYPES and v != v:
            continue
        cur = out[g]
        if cur is None or (v < cur if is_min else v > cur):
            out[g] = v
    z = "" if col.dtype == "Utf8" else 0
    return OCol(col.dtype, [z if x is None else x for x in out], [x is not None for x in out])


def hash_aggregate(batch, mode, group_exprs_names, aggr_exprs, batch_rows=32768):
    """HashAggregateExec (from_proto.rs:173-252).

    mode "Partial": input rows -> group columns + state columns (SUM -> [sum], AVG ->
    [count: UInt64, sum: Float64], COUNT -> [count: UInt64]; Appendix A).
    mode "Final": input = concatenated partial outputs (group columns first, then state
    columns in aggregate order); merges states and evaluates.
    Group output order = first appearance (the reference's is unspecified).
    `aggr_exprs`: objects with .fun/.expr/.name (in Final mode .expr is ignored).
    """
    n = batch_len(batch)
    if mode == "Partial":
        gcols = [evaluate(e, batch) for e, _ in group_exprs_names]
    else:
        gcols = [batch[name] for _, name in group_exprs_names]
    gid, keys = _factorize(gcols, n)
    ng = len(keys) if gcols else 1
    if gcols and n == 0:
        ng = 0
    out = OrderedDict()
    for j, (_, name) in enumerate(group_exprs_names):
        kc = gcols[j]
        vals = [k[j] for k in keys][:ng]
        z = "" if kc.dtype == "Utf8" else 0
        out[name] = OCol(kc.dtype, [z if v is None else v for v in vals], [v is not None for v in vals])
    if mode == "Partial":
        for a in aggr_exprs:
            x = evaluate(a.expr, batch)
            if a.fun == "COUNT":
                cnt = np.bincount(gid[x.is_valid()], minlength=ng).astype(np.uint64)[:ng] if n else np.zeros(ng, np.uint64)
                out[f"{a.name}[count]"] = OCol("UInt64", cnt)
            elif a.fun in ("SUM", "AVG"):
                s, c, has = _group_sum(x, gid, ng, batch_rows)
                if a.fun == "AVG":
                    if x.dtype != "Float64":
                        s = s.astype(np.float64)                 # (Float32: the sums are doubles already)
                    out[f"{a.name}[count]"] = OCol("UInt64", c)
                    out[f"{a.name}[sum]"] = OCol("Float64", s, has)
                else:
                    st = _sum_type(x.dtype)
                    out[f"{a.name}[sum]"] = OCol(st, s.astype(NP_TYPES[st]), has)
            elif a.fun in ("MIN", "MAX"):
                out[f"{a.name}[{a.fun.lower()}]"] = _minmax(x, gid, ng, a.fun == "MIN")
            else:
                raise NotImplementedError(a.fun)
        return out
    # Final: state columns follow the group columns, in aggregate order
    names = list(batch.keys())
    pos = len(group_exprs_names)
    for a in aggr_exprs:
        if a.fun == "COUNT":
            c = batch[names[pos]]; pos += 1
            s, _, _ = _group_sum(OCol("Int64", c.values.astype(np.int64), c.valid), gid, ng, batch_rows)
            out[a.name] = OCol("UInt64", s.astype(np.uint64))
        elif a.fun == "SUM":
            c = batch[names[pos]]; pos += 1
            s, _, has = _group_sum(c, gid, ng, batch_rows)
            out[a.name] = OCol(c.dtype, s.astype(NP_TYPES[c.dtype]), has)
        elif a.fun == "AVG":
            c = batch[names[pos]]; sc = batch[names[pos + 1]]; pos += 2
            cs, _, _ = _group_sum(OCol("Int64", c.values.astype(np.int64), c.valid), gid, ng, batch_rows)
            ss, _, has = _group_sum(sc, gid, ng, batch_rows)
            with np.errstate(all="ignore"):
                avg = ss / cs.astype(np.float64)
            out[a.name] = OCol("Float64", np.where(has & (cs > 0), avg, 0.0), has & (cs > 0))
        elif a.fun in ("MIN", "MAX"):
            c = batch[names[pos]]; pos += 1
            out[a.name] = _minmax(c, gid, ng, a.fun == "MIN")
        else:
            raise NotImplementedError(a.fun)
    return out


def concat_batches(batches):
    """MergeExec / CoalesceBatchesExec (from_proto.rs:122-132): concatenation."""
    batches = [b for b in batches]
    first = batches[0]
    out = OrderedDict()
    for k in first:
        vals = np.concatenate([b[k].values for b in batches])
        valid = np.concatenate([b[k].is_valid() for b in batches])
        out[k] = OCol(first[k].dtype, vals, valid)
    return out


def limit(batch, n):
    """GlobalLimitExec / LocalLimitExec (from_proto.rs:165-172)."""
    idx = np.arange(min(n, batch_len(batch)))
    return OrderedDict((k, c.take(idx)) for k, c in batch.items())


def hash_join(left, right, on, join_type="Inner"):
    """HashJoinExec (from_proto.rs:253-276): left = build side.  Output = left fields then
    right fields, a right key column dropped when it has the same name as its left partner
    (Appendix A).  Row order unspecified (here: probe order, matches in build order).
    NULL keys never match."""
    ln, rn = batch_len(left), batch_len(right)
    lk = _key_tuples([left[a] for a, _ in on])
    rk = _key_tuples([right[b] for _, b in on])
    table = {}
    for i, t in enumerate(lk):
        if any(v is None for v in t):
            continue
        table.setdefault(t, []).append(i)
    li, ri = [], []
    lmatched = np.zeros(ln, np.bool_)
    for j, t in enumerate(rk):
        rows = table.get(t) if not any(v is None for v in t) else None
        if rows:
            for i in rows:
                li.append(i); ri.append(j)
            lmatched[rows] = True
        elif join_type == "Right":
            li.append(-1); ri.append(j)
    if join_type == "Left":
        for i in np.nonzero(~lmatched)[0]:
            li.append(int(i)); ri.append(-1)
    li = np.asarray(li, np.int64); ri = np.asarray(ri, np.int64)

    def gather(c, idx):
        safe = np.where(idx < 0, 0, idx)
        if len(c) == 0:
            z = "" if c.dtype == "Utf8" else 0
            return OCol(c.dtype, [z] * len(idx), np.zeros(len(idx), np.bool_))
        t = c.take(safe)
        return OCol(c.dtype, t.values, t.is_valid() & (idx >= 0))

    out = OrderedDict()
    for k, c in left.items():
        out[k] = gather(c, li)
    drop = {b for a, b in on if a == b}
    for k, c in right.items():
        if k in drop:
            continue
        out[k] = gather(c, ri)
    return out


def sort_batch(batch, sort_exprs):
    """SortExec (from_proto.rs:291-331): lexicographic, per key descending / nulls_first.
    Stable here; the reference leaves tie order unspecified."""
    n = batch_len(batch)
    order = np.arange(n)
    for se in reversed(list(sort_exprs)):
        c = evaluate(se.expr, batch)
        vals = c.values[order]
        ok = c.is_valid()[order]
        if c.dtype == "Utf8":
            keys = np.array([s.encode() for s in vals], dtype=object)
        else:
            keys = vals
        idx = np.argsort(keys[ok] if True else keys, kind="stable")
        valid_pos = np.nonzero(ok)[0][idx]
        if se.descending:
            # stable descending: reverse groups of equal keys, not the rows inside them
            kv = keys[valid_pos]
            if len(kv):
                change = np.nonzero(np.array([kv[i] != kv[i - 1] for i in range(1, len(kv))], dtype=bool))[0] + 1
                groups = np.split(valid_pos, change)
                valid_pos = np.concatenate(groups[::-1]) if groups else valid_pos
        null_pos = np.nonzero(~ok)[0]
        new = np.concatenate([null_pos, valid_pos]) if se.nulls_first else np.concatenate([valid_pos, null_pos])
        order = order[new.astype(np.int64)]
    return OrderedDict((k, c.take(order)) for k, c in batch.items())


# ---- hash repartition --------------------------------------------------------------------
# RepartitionExec(Hash(exprs, n)) (from_proto.rs:133-147): rows with equal keys land in the
# same output partition; WHICH one is unobservable in the reference (Appendix A).  The
# product documents its own hash (DESIGN.md "Row hash"); it is restated here so partition
# contents can be compared bit for bit.

_M64 = (1 << 64) - 1


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _value_bits(c: OCol, i):
    if not c.is_valid()[i]:
        return 0x6E756C6C6E756C6C  # "nullnull"
    v = c.values[i]
    if c.dtype in FLOAT_TYPES:
        if v == 0.0:
            v = 0.0
        return int(np.float64(v).view(np.uint64))                 # a Float32 value hashes as the double it equals
    if c.dtype == "Utf8":
        h = 0xCBF29CE484222325
        for b in v.encode():
            h = ((h ^ b) * 0x100000001B3) & _M64
        return h
    return int(v) & _M64


def row_hash(cols, n):
    out = np.zeros(n, np.uint64)
    for i in range(n):
        h = 0
        for c in cols:
            h = mix64(h ^ _value_bits(c, i))
        out[i] = h
    return out


def repartition_hash(batch, exprs, nparts):
    cols = [evaluate(e, batch) for e in exprs]
    h = row_hash(cols, batch_len(batch))
    pid = (h % np.uint64(nparts)).astype(np.int64)
    return [OrderedDict((k, c.take(np.nonzero(pid == p)[0])) for k, c in batch.items()) for p in range(nparts)]
