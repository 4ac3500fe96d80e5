"""TPC-H Q12 through the general paths (expression VM: IN-list on Utf8, column-vs-column comparisons, CASE WHEN with
OR / AND of string comparisons; join with a non-unique build side; Utf8 group key of 4 bytes) against the CPU oracle.
Query: rust/benchmarks/tpch/queries/q12.sql.  Integer sums and the group set must match exactly."""
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import tpch
from oracle import gen, plan_eval
from oracle.engine import OCol

import helpers

pytestmark = pytest.mark.gpu
MODES = ["REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"]
PRIOS = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]


def tables(sf, seed):
    rng = np.random.default_rng(seed)
    a = gen.lineitem_arrays(sf, dates=True)
    n = len(a["l_quantity"])
    li = OrderedDict([("l_orderkey", OCol("Int32", a["l_orderkey"])), ("l_shipdate", OCol("Date32", a["l_shipdate"])),
                      ("l_commitdate", OCol("Date32", a["l_commitdate"])), ("l_receiptdate", OCol("Date32", a["l_receiptdate"])),
                      ("l_shipmode", OCol("Utf8", [MODES[k] for k in rng.integers(0, 7, n)]))])
    o = gen.orders_arrays(sf)
    od = OrderedDict([("o_orderkey", OCol("Int32", o["o_orderkey"])),
                      ("o_orderpriority", OCol("Utf8", [PRIOS[k] for k in rng.integers(0, 5, len(o["o_orderkey"]))]))])
    return li, od


@pytest.mark.parametrize("sf,n_part", [(0.002, 1), (0.01, 3)])
def test_q12_matches_oracle(ctx, sf, n_part):
    li, od = tables(sf, 12)
    n = len(li["l_orderkey"].values)
    per = (n + n_part - 1) // n_part
    parts = [[helpers.slice_batch(li, p * per, min(n, (p + 1) * per))] for p in range(n_part)]
    plan = tpch.q12_plan(helpers.memory_exec(ctx, [[od]]), helpers.memory_exec(ctx, parts))
    got = helpers.concat(helpers.collect_product(plan))
    want = plan_eval.collect(plan)
    assert list(got["l_shipmode"].values) == ["MAIL", "SHIP"]
    helpers.assert_rows_equal(got, want, ordered=True)
    assert int(sum(got["high_line_count"].values) + sum(got["low_line_count"].values)) > 0
