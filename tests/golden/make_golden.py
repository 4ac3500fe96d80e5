#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

Nothing of the reference can be executed (its hot path is the un-vendored Rust dependency
datafusion/arrow rev 46161d2, no Rust toolchain), so the vectors come from the
reference's own DATA fixtures and three independent CPU computations that must agree:

  1. q1_fixture.json — TPC-H Q1 over the reference's 10-row lineitem fixture
     (rust/scheduler/testdata/lineitem/partition{0,1}.tbl, two identical partitions = 20 rows;
     copied as data to tests/golden/tbl/).  Expected values: exact rational arithmetic over the
     separately-rounded f64 per-row values (fractions.Fraction), cross-checked against pyarrow /
     Acero `group_by().aggregate()` and the figures recorded in SURVEY.md §8(c).
  2. q1_synth.json / q6_synth.json — Q1 / Q6 over seeded synthetic lineitem (oracle/tpch_gen.c,
     sf = 0.01, 60 000 rows, 3 partitions): exact-rational expected values + Acero cross-check.
  3. gen_pin.json — first rows and column checksums of the synthetic generator, so the CPU and
     HIP generators are pinned to one spec.
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from oracle import gen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
Q1_DATE = 10471            # 1998-09-02
SF = 0.01
N_PART = 3


def exact_q1(batch):
    """Q1 with exact summation of the separately rounded per-row f64 values"""
    qty = batch["l_quantity"].values
    price = batch["l_extendedprice"].values
    disc = batch["l_discount"].values
    tax = batch["l_tax"].values
    ship = batch["l_shipdate"].values
    disc_price = price * (1.0 - disc)               # numpy: each op rounds once, like arrow kernels
    charge = disc_price * (1.0 + tax)
    groups = {}
    for i in range(len(qty)):
        if ship[i] > Q1_DATE:
            continue
        k = (batch["l_returnflag"].values[i], batch["l_linestatus"].values[i])
        g = groups.setdefault(k, [Fraction(0)] * 5 + [0])
        for j, v in enumerate((qty[i], price[i], disc_price[i], charge[i], disc[i])):
            g[j] += Fraction(float(v))
        g[5] += 1
    rows = []
    for k in sorted(groups):
        s = groups[k]
        n = s[5]
        rows.append(dict(l_returnflag=k[0], l_linestatus=k[1], sum_qty=float(s[0]), sum_base_price=float(s[1]),
                         sum_disc_price=float(s[2]), sum_charge=float(s[3]), avg_qty=float(s[0] / n),
                         avg_price=float(s[1] / n), avg_disc=float(s[4] / n), count_order=n))
    return rows


def acero_q1(batch):
    t = pa.table({k: (pa.array(list(c.values)) if c.dtype == "Utf8" else pa.array(c.values)) for k, c in batch.items()})
    t = t.filter(pc.less_equal(t["l_shipdate"], Q1_DATE))
    dp = pc.multiply(t["l_extendedprice"], pc.subtract(1.0, t["l_discount"]))
    ch = pc.multiply(dp, pc.add(1.0, t["l_tax"]))
    t = t.append_column("dp", dp).append_column("ch", ch)
    r = t.group_by(["l_returnflag", "l_linestatus"]).aggregate(
        [("l_quantity", "sum"), ("l_extendedprice", "sum"), ("dp", "sum"), ("ch", "sum"), ("l_quantity", "mean"),
         ("l_extendedprice", "mean"), ("l_discount", "mean"), ("l_quantity", "count")]).sort_by(
        [("l_returnflag", "ascending"), ("l_linestatus", "ascending")])
    return r.to_pylist()


def close(a, b, rtol=1e-12):
    return abs(a - b) <= rtol * max(abs(a), abs(b), 1e-300)


def check_q1(rows, acero):
    assert len(rows) == len(acero)
    for r, a in zip(rows, acero):
        assert (r["l_returnflag"], r["l_linestatus"]) == (a["l_returnflag"], a["l_linestatus"])
        assert r["count_order"] == a["l_quantity_count"]
        for ours, theirs in (("sum_qty", "l_quantity_sum"), ("sum_base_price", "l_extendedprice_sum"),
                             ("sum_disc_price", "dp_sum"), ("sum_charge", "ch_sum"), ("avg_qty", "l_quantity_mean"),
                             ("avg_price", "l_extendedprice_mean"), ("avg_disc", "l_discount_mean")):
            assert close(r[ours], a[theirs]), (ours, r[ours], a[theirs])


def exact_q6(batch):
    qty, price, disc, ship = (batch[k].values for k in ("l_quantity", "l_extendedprice", "l_discount", "l_shipdate"))
    lo, hi = 0.06 - 0.01, 0.06 + 0.01
    sel = (ship >= 8766) & (ship < 9131) & (disc >= lo) & (disc <= hi) & (qty < 24.0)
    prod = price * disc
    s = sum((Fraction(float(v)) for v in prod[sel]), Fraction(0))
    return dict(revenue=float(s), selected=int(sel.sum()))


def main():
    # ---- 1. the reference's own lineitem fixture --------------------------------------------------
    part = helpers.lineitem_fixture("lineitem_partition0")
    both = helpers.concat([part, helpers.lineitem_fixture("lineitem_partition1")])
    rows = exact_q1(both)
    check_q1(rows, acero_q1(both))
    # figures recorded in SURVEY.md §8(c) for this fixture
    by = {(r["l_returnflag"], r["l_linestatus"]): r for r in rows}
    assert [by[k]["count_order"] for k in (("A", "F"), ("N", "O"), ("R", "F"))] == [2, 14, 4]
    assert by[("A", "F")]["sum_qty"] == 54 and close(by[("A", "F")]["sum_base_price"], 79781.76)
    assert close(by[("A", "F")]["sum_disc_price"], 74994.8544) and close(by[("A", "F")]["sum_charge"], 80244.494208)
    assert by[("N", "O")]["sum_qty"] == 366 and close(by[("N", "O")]["sum_base_price"], 453111.46)
    assert by[("R", "F")]["sum_qty"] == 188 and close(by[("R", "F")]["sum_base_price"], 201709.04)
    json.dump(dict(source="rust/scheduler/testdata/lineitem/partition{0,1}.tbl", rows=rows),
              open(os.path.join(OUT, "q1_fixture.json"), "w"), indent=1)

    # ---- 2. seeded synthetic lineitem ------------------------------------------------------------------
    li = gen.lineitem(SF)
    n = len(li["l_quantity"])
    rows = exact_q1(li)
    check_q1(rows, acero_q1(li))
    json.dump(dict(sf=SF, seed=gen.SEED, n_rows=n, n_partitions=N_PART, rows=rows),
              open(os.path.join(OUT, "q1_synth.json"), "w"), indent=1)
    q6 = exact_q6(li)
    t = pa.table({k: pa.array(li[k].values) for k in ("l_quantity", "l_extendedprice", "l_discount", "l_shipdate")})
    m = pc.and_(pc.and_(pc.greater_equal(t["l_shipdate"], 8766), pc.less(t["l_shipdate"], 9131)),
                pc.and_(pc.and_(pc.greater_equal(t["l_discount"], 0.06 - 0.01), pc.less_equal(t["l_discount"], 0.06 + 0.01)),
                        pc.less(t["l_quantity"], 24.0)))
    tf = t.filter(m)
    assert tf.num_rows == q6["selected"]
    assert close(pc.sum(pc.multiply(tf["l_extendedprice"], tf["l_discount"])).as_py(), q6["revenue"])
    json.dump(dict(sf=SF, seed=gen.SEED, n_rows=n, **q6), open(os.path.join(OUT, "q6_synth.json"), "w"), indent=1)

    # ---- 3. generator pin --------------------------------------------------------------------------------
    a = gen.lineitem_arrays(SF, dates=True)
    o = gen.orders_arrays(SF)
    c = gen.customer_arrays(SF)
    s = gen.supplier_arrays(SF)

    def summary(arrs):
        out = {}
        for k, v in arrs.items():
            v64 = v.view(np.uint64) if v.dtype == np.float64 else v.astype(np.uint64)
            out[k] = dict(n=int(len(v)), first=[x.item() for x in v[:8]],
                          xor=int(np.bitwise_xor.reduce(v64)) if len(v) else 0,
                          sum_mod=int(int(v64.astype(object).sum()) % (1 << 64)))
        return out
    json.dump(dict(sf=SF, seed=gen.SEED, lineitem=summary(a), orders=summary(o), customer=summary(c), supplier=summary(s)),
              open(os.path.join(OUT, "gen_pin.json"), "w"), indent=1)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
