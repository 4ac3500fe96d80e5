"""Shared test plumbing: moving batches between the oracle's representation (dict name -> OCol)
and the product (device RecordBatch), and comparing results the way SURVEY.md §8 prescribes:
bit-exact for integers / strings / row sets, <= 1e-6 relative for SUM/AVG over Float64, row order
and group order unspecified unless a SortExec fixed it."""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import numpy as np

import ballista_amd as ba
from oracle.engine import OCol

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def to_device(ctx, batch) -> ba.RecordBatch:
    cols = []
    for name, c in batch.items():
        vals = list(c.values) if c.dtype == "Utf8" else c.values
        cols.append((name, c.dtype, vals, c.valid))
    return ba.RecordBatch.from_columns(ctx, cols)


def from_device(rb: ba.RecordBatch):
    out = OrderedDict()
    for i in range(rb.num_columns):
        name = rb.column_info(i)[0]
        dtype, values, valid = rb.column(i)
        out[name] = OCol(dtype, values, valid)
    return out


def memory_exec(ctx, partitions) -> ba.MemoryExec:
    """partitions: list of partitions, each a list of oracle batches"""
    dev = [[to_device(ctx, b) for b in part] for part in partitions]
    m = ba.MemoryExec(dev, ctx)
    m._oracle_partitions = [list(p) for p in partitions]
    return m


def collect_product(plan):
    """all output partitions of a product plan as oracle-style batches"""
    return [from_device(b) for b in plan.collect()]


def rows_of(batch):
    cols = [c.to_pylist() for c in batch.values()]
    return list(zip(*cols)) if cols else []


def _sort_key(row):
    return tuple((0, "") if v is None else (1, v) if not isinstance(v, float) else (1, v) for v in row)


def concat(batches):
    from oracle.engine import concat_batches
    batches = [b for b in batches]
    return concat_batches(batches) if batches else OrderedDict()


def assert_same_schema(got, want):
    assert list(got.keys()) == list(want.keys()), (list(got.keys()), list(want.keys()))
    for k in want:
        assert got[k].dtype == want[k].dtype, (k, got[k].dtype, want[k].dtype)


def assert_rows_equal(got, want, ordered=False, float_rtol=0.0, key_cols=None):
    """compare two oracle-style batches.  Float64 columns compare with float_rtol (0 = bit exact),
    everything else exactly.  Unordered: rows are matched after sorting on the non-float columns
    (or key_cols)."""
    if not got or not want:
        # a stream may end without yielding a batch: equivalent to zero rows
        assert len(rows_of(got)) == 0 and len(rows_of(want)) == 0, (len(rows_of(got)), len(rows_of(want)))
        return
    assert_same_schema(got, want)
    g, w = rows_of(got), rows_of(want)
    assert len(g) == len(w), f"row count {len(g)} != {len(w)}"
    names = list(want.keys())
    fcols = [i for i, n in enumerate(names) if want[n].dtype in ("Float64", "Float32")]
    if not ordered:
        if key_cols is None:
            kidx = [i for i in range(len(names)) if i not in fcols] or list(range(len(names)))
        else:
            kidx = [names.index(k) for k in key_cols]
        keyf = lambda r: _sort_key(tuple(r[i] for i in kidx)) + _sort_key(tuple(r[i] for i in range(len(r)) if i not in kidx))
        g, w = sorted(g, key=keyf), sorted(w, key=keyf)
    for ri, (a, b) in enumerate(zip(g, w)):
        for ci, (x, y) in enumerate(zip(a, b)):
            if x is None or y is None:
                assert x is None and y is None, f"row {ri} col {names[ci]}: {x!r} vs {y!r}"
            elif ci in fcols and float_rtol > 0:
                if math.isnan(y):
                    assert math.isnan(x)
                else:
                    assert abs(x - y) <= float_rtol * max(abs(y), 1e-300), f"row {ri} col {names[ci]}: {x!r} vs {y!r}"
            elif ci in fcols:
                assert (x == y and math.copysign(1, x) == math.copysign(1, y)) or (math.isnan(x) and math.isnan(y)), \
                    f"row {ri} col {names[ci]}: {x!r} vs {y!r} (bit exact expected)"
            else:
                assert x == y, f"row {ri} col {names[ci]}: {x!r} vs {y!r}"


# ---- the reference's .tbl fixtures (rust/scheduler/testdata/*, copied as data under tests/golden/tbl)

import datetime


def _days(s):
    return (datetime.date.fromisoformat(s) - datetime.date(1970, 1, 1)).days


def load_tbl(name):
    path = os.path.join(GOLDEN, "tbl", name + ".tbl")
    with open(path) as f:
        return [line.rstrip("\n").split("|")[:-1] for line in f if line.strip()]


def lineitem_fixture(name="lineitem_partition0"):
    r = load_tbl(name)
    return OrderedDict([
        ("l_orderkey", OCol("Int32", [int(x[0]) for x in r])),
        ("l_suppkey", OCol("Int32", [int(x[2]) for x in r])),
        ("l_quantity", OCol("Float64", [float(x[4]) for x in r])),
        ("l_extendedprice", OCol("Float64", [float(x[5]) for x in r])),
        ("l_discount", OCol("Float64", [float(x[6]) for x in r])),
        ("l_tax", OCol("Float64", [float(x[7]) for x in r])),
        ("l_returnflag", OCol("Utf8", [x[8] for x in r])),
        ("l_linestatus", OCol("Utf8", [x[9] for x in r])),
        ("l_shipdate", OCol("Date32", [_days(x[10]) for x in r]))])


def orders_fixture():
    r = load_tbl("orders_orders")
    return OrderedDict([("o_orderkey", OCol("Int32", [int(x[0]) for x in r])),
                        ("o_custkey", OCol("Int32", [int(x[1]) for x in r])),
                        ("o_orderdate", OCol("Date32", [_days(x[4]) for x in r])),
                        ("o_shippriority", OCol("Int32", [int(x[7]) for x in r]))])


def customer_fixture():
    r = load_tbl("customer_customer")
    return OrderedDict([("c_custkey", OCol("Int32", [int(x[0]) for x in r])),
                        ("c_nationkey", OCol("Int32", [int(x[3]) for x in r])),
                        ("c_mktsegment", OCol("Utf8", [x[6] for x in r]))])


def supplier_fixture():
    r = load_tbl("supplier_supplier")
    return OrderedDict([("s_suppkey", OCol("Int32", [int(x[0]) for x in r])),
                        ("s_nationkey", OCol("Int32", [int(x[3]) for x in r]))])


def nation_fixture():
    r = load_tbl("nation_nation")
    return OrderedDict([("n_nationkey", OCol("Int32", [int(x[0]) for x in r])),
                        ("n_name", OCol("Utf8", [x[1] for x in r])),
                        ("n_regionkey", OCol("Int32", [int(x[2]) for x in r]))])


def region_fixture():
    r = load_tbl("region_region")
    return OrderedDict([("r_regionkey", OCol("Int32", [int(x[0]) for x in r])),
                        ("r_name", OCol("Utf8", [x[1] for x in r]))])


def slice_batch(batch, lo, hi):
    idx = np.arange(lo, min(hi, len(next(iter(batch.values())))))
    return OrderedDict((k, c.take(idx)) for k, c in batch.items())
