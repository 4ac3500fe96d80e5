"""GPU parity: TPC-H Q1 / Q6 (scan -> filter -> partial aggregate -> merge -> final aggregate) through
the C ABI vs the CPU oracle and the committed golden vectors.

Integer / Utf8 columns, group sets and counts must match bit-exactly; SUM / AVG over Float64 within
1e-6 relative (BASELINE.json north_star) — the GPU sums in a fixed tree order, the reference
sequentially per batch, so only the summation order differs (per-row values are bit-identical:
kernels are compiled with -ffp-contract=off)."""
import json
import os

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit
from oracle import engine as og, gen, plan_eval

import helpers

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def q1_final_no_sort(partial):
    merged = ba.MergeExec(partial)
    group = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]
    return ba.HashAggregateExec(ba.plan.FINAL, group, tpch.q1_final_aggs(), merged)


def golden_rows(name):
    return json.load(open(os.path.join(helpers.GOLDEN, name)))["rows"]


def check_against_golden(got, rows):
    g = {(a, b): i for i, (a, b) in enumerate(zip(got["l_returnflag"].values, got["l_linestatus"].values))}
    assert sorted(g) == sorted((r["l_returnflag"], r["l_linestatus"]) for r in rows)
    for r in rows:
        i = g[(r["l_returnflag"], r["l_linestatus"])]
        assert int(got["count_order"].values[i]) == r["count_order"]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert abs(got[k].values[i] - r[k]) <= RTOL * abs(r[k]), (k, got[k].values[i], r[k])


def test_q1_reference_fixture(ctx):
    """the reference's own 2 x 10-row lineitem fixture (rust/scheduler/testdata/lineitem)"""
    parts = [[helpers.lineitem_fixture("lineitem_partition0")], [helpers.lineitem_fixture("lineitem_partition1")]]
    scan = helpers.memory_exec(ctx, parts)
    plan = q1_final_no_sort(tpch.q1_stage1(scan))
    got = helpers.concat(helpers.collect_product(plan))
    check_against_golden(got, golden_rows("q1_fixture.json"))
    want = plan_eval.collect(plan)
    helpers.assert_rows_equal(got, want, float_rtol=RTOL)


def test_q1_stage1_state_columns(ctx):
    """Partial mode emits group columns + state columns (SUM -> [sum], AVG -> [count, sum], COUNT -> [count])"""
    parts = [[helpers.lineitem_fixture("lineitem_partition0")]]
    plan = tpch.q1_stage1(helpers.memory_exec(ctx, parts))
    names = [n for n, _, _ in plan.schema()]
    assert names == ["l_returnflag", "l_linestatus", "sum_qty[sum]", "sum_base_price[sum]", "sum_disc_price[sum]",
                     "sum_charge[sum]", "avg_qty[count]", "avg_qty[sum]", "avg_price[count]", "avg_price[sum]",
                     "avg_disc[count]", "avg_disc[sum]", "count_order[count]"]
    got = helpers.concat(helpers.collect_product(plan))
    want = plan_eval.collect(plan)
    helpers.assert_rows_equal(got, want, float_rtol=RTOL)


@pytest.mark.parametrize("n_part", [1, 3])
def test_q1_synthetic_golden(ctx, n_part):
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
    li = gen.lineitem(g["sf"])
    n = g["n_rows"]
    per = (n + n_part - 1) // n_part
    parts = [[helpers.slice_batch(li, p * per, (p + 1) * per)] for p in range(n_part)]
    plan = q1_final_no_sort(tpch.q1_stage1(helpers.memory_exec(ctx, parts)))
    got = helpers.concat(helpers.collect_product(plan))
    check_against_golden(got, g["rows"])
    helpers.assert_rows_equal(got, plan_eval.collect(plan), float_rtol=RTOL)


def test_q6_synthetic_golden(ctx):
    g = json.load(open(os.path.join(helpers.GOLDEN, "q6_synth.json")))
    li = gen.lineitem(g["sf"])
    parts = [[helpers.slice_batch(li, 0, 25000), helpers.slice_batch(li, 25000, 25001)], [helpers.slice_batch(li, 25001, 60000)]]
    plan = tpch.q6_plan(helpers.memory_exec(ctx, parts))
    got = helpers.concat(helpers.collect_product(plan))
    assert list(got.keys()) == ["revenue"]
    assert abs(got["revenue"].values[0] - g["revenue"]) <= RTOL * g["revenue"]
    helpers.assert_rows_equal(got, plan_eval.collect(plan), float_rtol=RTOL)


@pytest.mark.parametrize("n", [1, 63, 64, 255, 256, 1023, 1024, 1025, 4097])
def test_q1_ragged_sizes(ctx, n):
    li = helpers.slice_batch(gen.lineitem(0.001), 0, n)
    plan = q1_final_no_sort(tpch.q1_stage1(helpers.memory_exec(ctx, [[li]])))
    got = helpers.concat(helpers.collect_product(plan))
    helpers.assert_rows_equal(got, plan_eval.collect(plan), float_rtol=RTOL)


def test_q6_no_rows_selected_gives_one_null_row(ctx):
    """no GROUP BY: exactly one output row even when nothing passes the filter (SUM = NULL)"""
    li = helpers.slice_batch(gen.lineitem(0.001), 0, 500)
    s = tpch.LINEITEM_SCHEMA
    flt = ba.FilterExec(E.coerce(col("l_quantity") < lit(0), s), helpers.memory_exec(ctx, [[li]]))
    part = ba.HashAggregateExec(ba.plan.PARTIAL, [], [E.Sum(col("l_extendedprice"), "s"), E.Count(col("l_quantity"), "c")], flt)
    got = helpers.concat(helpers.collect_product(part))
    assert got["s[sum]"].to_pylist() == [None]
    assert got["c[count]"].to_pylist() == [0]


def test_q1_device_generated_input_matches_cpu_generated(ctx):
    """HIP generator -> Q1 == CPU generator -> oracle Q1 (generator + pipeline end to end)"""
    sf, n = 0.01, 60000
    dev = [[ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, 0, 40000)], [ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, 40000, 20000)]]
    plan = q1_final_no_sort(tpch.q1_stage1(ba.MemoryExec(dev, ctx)))
    got = helpers.concat(helpers.collect_product(plan))
    check_against_golden(got, golden_rows("q1_synth.json"))


def test_q1_with_nulls(ctx):
    """validity bitmaps are honoured: NULL shipdate fails the filter, NULL discount is skipped by SUM/AVG"""
    rng = np.random.default_rng(5)
    li = helpers.slice_batch(gen.lineitem(0.001), 0, 3000)
    n = 3000
    li["l_shipdate"] = og.OCol("Date32", li["l_shipdate"].values, rng.random(n) > 0.1)
    li["l_discount"] = og.OCol("Float64", li["l_discount"].values, rng.random(n) > 0.2)
    li["l_returnflag"] = og.OCol("Utf8", li["l_returnflag"].values, rng.random(n) > 0.05)
    plan = q1_final_no_sort(tpch.q1_stage1(helpers.memory_exec(ctx, [[li]])))
    got = helpers.concat(helpers.collect_product(plan))
    helpers.assert_rows_equal(got, plan_eval.collect(plan), float_rtol=RTOL)
