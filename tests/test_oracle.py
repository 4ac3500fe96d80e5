"""CPU tests of the oracle (the checker): it is pinned against the committed golden vectors —
the reference's own lineitem fixture + exact-rational / Acero-cross-checked synthetic vectors
(tests/golden/make_golden.py) — and its pieces are checked against each other."""
import json
import math
import os

import numpy as np
import pytest

from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit
from oracle import engine as og, gen
from oracle.engine import OCol

import helpers

RTOL = 1e-9


def oracle_q1(partitions):
    q = tpch.q1_parts(tpch.LINEITEM_SCHEMA)
    partials = [og.hash_aggregate(og.filter_batch(b, q["predicate"]), "Partial", q["group"], q["aggs"]) for b in partitions]
    fin = og.hash_aggregate(og.concat_batches(partials), "Final", q["group"], tpch.q1_final_aggs())
    return og.sort_batch(fin, [E.PhysicalSortExpr(col("l_returnflag")), E.PhysicalSortExpr(col("l_linestatus"))])


def check_rows(got, rows, rtol=RTOL):
    assert list(zip(got["l_returnflag"].values, got["l_linestatus"].values)) == [(r["l_returnflag"], r["l_linestatus"]) for r in rows]
    for i, r in enumerate(rows):
        assert int(got["count_order"].values[i]) == r["count_order"]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert abs(got[k].values[i] - r[k]) <= rtol * abs(r[k]), (k, got[k].values[i], r[k])


def test_q1_on_reference_fixture_matches_golden():
    parts = [helpers.lineitem_fixture("lineitem_partition0"), helpers.lineitem_fixture("lineitem_partition1")]
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_fixture.json")))
    check_rows(oracle_q1(parts), g["rows"])
    # group row counts of this fixture
    assert [r["count_order"] for r in g["rows"]] == [2, 14, 4]


@pytest.mark.parametrize("n_part", [1, 3, 7])
def test_q1_on_synthetic_matches_golden(n_part):
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
    li = gen.lineitem(g["sf"])
    per = (g["n_rows"] + n_part - 1) // n_part
    check_rows(oracle_q1([helpers.slice_batch(li, p * per, (p + 1) * per) for p in range(n_part)]), g["rows"])


def test_q6_on_synthetic_matches_golden():
    g = json.load(open(os.path.join(helpers.GOLDEN, "q6_synth.json")))
    li = gen.lineitem(g["sf"])
    q = tpch.q6_parts(tpch.LINEITEM_SCHEMA)
    f = og.filter_batch(li, q["predicate"])
    assert og.batch_len(f) == g["selected"]
    part = og.hash_aggregate(f, "Partial", [], q["aggs"])
    assert abs(part["revenue[sum]"].values[0] - g["revenue"]) <= RTOL * g["revenue"]
    # the BETWEEN bounds are f64 results: 0.07 is excluded, 0.05 included
    assert 0.06 + 0.01 < 0.07 and 0.06 - 0.01 <= 0.05


def test_threaded_port_matches_engine():
    """the C port timed as cpu_baseline computes the same thing as the numpy/C engine"""
    a = gen.lineitem_arrays(0.01)
    keys, state, count = gen.q1_partial_port(a, 5, 2)
    port = gen.q1_final_from_port(keys, state, count)
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
    assert sorted(port) == sorted((r["l_returnflag"], r["l_linestatus"]) for r in g["rows"])
    for r in g["rows"]:
        p = port[(r["l_returnflag"], r["l_linestatus"])]
        assert p["count_order"] == r["count_order"]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert abs(p[k] - r[k]) <= RTOL * abs(r[k])
    s6, c6 = gen.q6_partial_port(a, 4, 2)
    g6 = json.load(open(os.path.join(helpers.GOLDEN, "q6_synth.json")))
    assert int(c6.sum()) == g6["selected"] and abs(s6.sum() - g6["revenue"]) <= RTOL * g6["revenue"]


def test_generator_pin():
    """the CPU generator reproduces the committed first rows and column checksums"""
    g = json.load(open(os.path.join(helpers.GOLDEN, "gen_pin.json")))
    for name, arrs in (("lineitem", gen.lineitem_arrays(g["sf"], dates=True)), ("orders", gen.orders_arrays(g["sf"])),
                       ("customer", gen.customer_arrays(g["sf"])), ("supplier", gen.supplier_arrays(g["sf"]))):
        for k, v in arrs.items():
            want = g[name][k]
            assert len(v) == want["n"]
            assert [x.item() for x in v[:8]] == want["first"], (name, k)
            v64 = v.view(np.uint64) if v.dtype == np.float64 else v.astype(np.uint64)
            assert int(np.bitwise_xor.reduce(v64)) == want["xor"], (name, k)


def test_generator_is_block_reproducible_and_tpch_shaped():
    whole = gen.lineitem_arrays(0.01)
    part = gen.lineitem_arrays(0.01, row0=12345, n=1000)
    for k in ("l_quantity", "l_extendedprice", "l_discount", "l_shipdate", "l_returnflag.data"):
        assert np.array_equal(whole[k][12345:13345], part[k])
    q = whole["l_quantity"]
    assert q.min() == 1 and q.max() == 50
    d = whole["l_discount"]
    assert set(np.round(d * 100).astype(int)) == set(range(11))
    assert set(np.round(whole["l_tax"] * 100).astype(int)) == set(range(9))
    # money has two decimals exactly as a .tbl parse would give
    cents = np.round(whole["l_extendedprice"] * 100)
    assert np.array_equal(cents / 100.0, whole["l_extendedprice"])
    flags = set(zip(whole["l_returnflag.data"].tolist(), whole["l_linestatus.data"].tolist()))
    assert flags == {(ord("A"), ord("F")), (ord("N"), ord("F")), (ord("N"), ord("O")), (ord("R"), ord("F"))}
    o = gen.orders_arrays(0.01)
    assert not np.any(o["o_custkey"] % 3 == 0) and o["o_orderdate"].min() >= 8035 and o["o_orderdate"].max() <= 10440


def test_expression_null_semantics():
    b = {"a": OCol("Int32", [1, 2, 3, 4], [True, False, True, True]),
         "t": OCol("Boolean", [True, False, True, False], [True, True, False, False])}
    # Kleene: false AND null = false ; true AND null = null ; true OR null = true
    r = og.evaluate(col("t").and_(E.Literal(None, E.BOOLEAN)), b)
    assert r.to_pylist() == [None, False, None, None]
    r = og.evaluate(col("t").or_(E.Literal(None, E.BOOLEAN)), b)
    assert r.to_pylist() == [True, None, None, None]
    assert og.evaluate(col("a") + lit(1, E.INT32), b).to_pylist() == [2, None, 4, 5]
    assert og.evaluate(E.IsNullExpr(col("a")), b).to_pylist() == [False, True, False, False]
    # a NULL predicate drops the row
    assert og.filter_batch(b, col("a") > lit(1, E.INT32))["a"].to_pylist() == [3, 4]
    with pytest.raises(TypeError):
        og.evaluate(col("a") + lit(1.5), b)
    with pytest.raises(ZeroDivisionError):
        og.evaluate(col("a") / lit(0, E.INT32), b)


def test_casts():
    b = {"f": OCol("Float64", [1.9, -1.9, 3e10, float("nan"), 2.0]), "i": OCol("Int64", [1, -1, 2**40, 255, 256])}
    assert og.evaluate(E.CastExpr(col("f"), E.INT32), b).to_pylist() == [1, -1, None, None, 2]
    assert og.evaluate(E.CastExpr(col("i"), E.UINT8), b).to_pylist() == [1, None, None, 255, None]
    assert og.evaluate(E.CastExpr(col("i"), E.FLOAT64), b).to_pylist() == [1.0, -1.0, float(2**40), 255.0, 256.0]
    assert og.evaluate(E.CastExpr(lit("1995-03-15"), E.DATE32), {"x": OCol("Int32", [0])}).to_pylist() == [9204]
    assert E.date32("1998-09-02").value == 10471 and E.date32("1994-01-01").value == 8766


def test_aggregate_states_and_empty_input():
    b = {"k": OCol("Int32", [1, 1, 2]), "v": OCol("Float64", [1.0, 2.0, 5.0], [True, False, True])}
    aggs = [E.Sum(col("v"), "s"), E.Avg(col("v"), "a"), E.Count(col("v"), "c"), E.Count(lit(1, E.UINT8), "n")]
    p = og.hash_aggregate(b, "Partial", [(col("k"), "k")], aggs)
    assert list(p.keys()) == ["k", "s[sum]", "a[count]", "a[sum]", "c[count]", "n[count]"]
    assert p["a[count]"].to_pylist() == [1, 1] and p["n[count]"].to_pylist() == [2, 1]
    f = og.hash_aggregate(og.concat_batches([p, p]), "Final", [(col("k"), "k")], aggs)
    assert f["s"].to_pylist() == [2.0, 10.0] and f["a"].to_pylist() == [1.0, 5.0] and f["n"].to_pylist() == [4, 2]
    empty = helpers.slice_batch(b, 0, 0)
    e = og.hash_aggregate(empty, "Partial", [], aggs)
    assert e["s[sum]"].to_pylist() == [None] and e["c[count]"].to_pylist() == [0]     # one row, SUM NULL, COUNT 0
    assert og.batch_len(og.hash_aggregate(empty, "Partial", [(col("k"), "k")], aggs)) == 0


def test_batched_sum_order_is_the_references():
    """per (group, batch) sequential fold, then across batches (Appendix A 'Float summation order')"""
    vals = np.array([1e16, 1.0, -1e16, 1.0] * 4, np.float64)
    c = OCol("Float64", vals)
    gid = np.zeros(len(vals), np.int32)
    s_all, _, _ = og._group_sum(c, gid, 1, 32768)
    seq = 0.0
    for v in vals:
        seq = seq + v
    assert s_all[0] == seq
    s_b, _, _ = og._group_sum(c, gid, 1, 4)              # batches of 4 rows: different association
    per_batch = [((vals[i] + vals[i + 1]) + vals[i + 2]) + vals[i + 3] for i in range(0, 16, 4)]
    t = per_batch[0]
    for x in per_batch[1:]:
        t = t + x
    assert s_b[0] == t


def test_join_sort_repartition_properties():
    rng = np.random.default_rng(0)
    l = {"k": OCol("Int32", rng.integers(0, 20, 50)), "x": OCol("Int32", np.arange(50))}
    r = {"k2": OCol("Int32", rng.integers(0, 30, 200)), "y": OCol("Int32", np.arange(200))}
    j = og.hash_join(l, r, [("k", "k2")], "Inner")
    brute = sorted((int(a), int(b)) for a, ka in zip(l["x"].values, l["k"].values) for b, kb in zip(r["y"].values, r["k2"].values) if ka == kb)
    assert sorted(zip(j["x"].to_pylist(), j["y"].to_pylist())) == brute
    lj = og.hash_join(l, r, [("k", "k2")], "Left")
    assert og.batch_len(lj) == len(brute) + sum(1 for ka in l["k"].values if ka not in set(r["k2"].values))
    s = og.sort_batch(r, [E.PhysicalSortExpr(col("k2"), descending=True), E.PhysicalSortExpr(col("y"))])
    keys = list(zip((-s["k2"].values).tolist(), s["y"].values.tolist()))
    assert keys == sorted(keys)
    parts = og.repartition_hash(r, [col("k2")], 4)
    assert sum(og.batch_len(p) for p in parts) == 200
    seen = {}
    for pi, p in enumerate(parts):
        for k in p["k2"].values:
            assert seen.setdefault(int(k), pi) == pi          # equal keys co-locate
