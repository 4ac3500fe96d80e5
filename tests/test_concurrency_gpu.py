"""The executor calls `execute` from several tasks at once in one process (concurrent_tasks, default 4:
rust/executor/executor_config_spec.toml:57-62; one spawned task per partition id,
rust/executor/src/flight_service.rs:100-103).  The library must be re-entrant: one HIP stream per
`execute`, a shared caching allocator, plans shared immutably between threads."""
import threading

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit
from oracle import gen, plan_eval

import helpers

pytestmark = pytest.mark.gpu


def test_four_tasks_share_one_plan(ctx):
    """4 threads each drain a different partition of the SAME plan object, repeatedly"""
    li = gen.lineitem(0.01)
    n = len(li["l_quantity"].values)
    per = n // 4
    parts = [[helpers.slice_batch(li, p * per, n if p == 3 else (p + 1) * per)] for p in range(4)]
    plan = tpch.q1_stage1(helpers.memory_exec(ctx, parts))
    want = [plan_eval.execute(plan, p) for p in range(4)]
    errors, results = [], [[None] * 6 for _ in range(4)]

    def task(p):
        try:
            for it in range(6):
                results[p][it] = helpers.concat([helpers.from_device(b) for b in plan.execute(p)])
        except BaseException as e:      # noqa: BLE001
            errors.append((p, e))

    threads = [threading.Thread(target=task, args=(p,)) for p in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for p in range(4):
        w = helpers.concat(want[p])
        for it in range(6):
            helpers.assert_rows_equal(results[p][it], w, ordered=False, float_rtol=1e-9, key_cols=["l_returnflag", "l_linestatus"])


def test_different_operators_concurrently(ctx):
    """a filter, a join, a sort and an aggregate run at the same time on one context"""
    li = helpers.slice_batch(gen.lineitem(0.01), 0, 30000)
    od = gen.orders(0.01)
    s = tpch.LINEITEM_SCHEMA
    plans = {
        "filter": ba.FilterExec(E.coerce(col("l_quantity") < lit(10), s), helpers.memory_exec(ctx, [[li]])),
        "join": ba.HashJoinExec(helpers.memory_exec(ctx, [[od]]), helpers.memory_exec(ctx, [[li]]), [("o_orderkey", "l_orderkey")], ba.plan.INNER),
        "sort": ba.SortExec([E.PhysicalSortExpr(col("l_extendedprice"), descending=True), E.PhysicalSortExpr(col("l_orderkey"))],
                            helpers.memory_exec(ctx, [[li]])),
        "agg": tpch.q6_plan(helpers.memory_exec(ctx, [[li]])),
    }
    want = {k: plan_eval.collect(p) for k, p in plans.items()}
    got, errors = {}, []

    def task(name):
        try:
            for _ in range(3):
                got[name] = helpers.concat(helpers.collect_product(plans[name]))
        except BaseException as e:      # noqa: BLE001
            errors.append((name, e))

    threads = [threading.Thread(target=task, args=(k,)) for k in plans]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    helpers.assert_rows_equal(got["filter"], want["filter"], ordered=True)
    helpers.assert_rows_equal(got["join"], want["join"], ordered=False)
    assert list(got["sort"]["l_extendedprice"].values) == list(want["sort"]["l_extendedprice"].values)
    helpers.assert_rows_equal(got["agg"], want["agg"], ordered=False, float_rtol=1e-9)
