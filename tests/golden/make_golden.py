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
  3. q3_synth.json / q5_synth.json / q12_synth.json — the join queries over the seeded synthetic tables (sf = 0.01), evaluated
     by pyarrow / Acero (`Table.join`, `group_by`, `sort_by`) — an engine that shares no code with oracle/ — with the
     per-group revenue re-summed exactly (fractions.Fraction over the separately rounded per-row products).
     join_cases.json / sort_cases.json — Inner / Left / Right joins with duplicate and NULL keys and multi-key sorts
     (descending, NULL placement) over small seeded tables: inputs AND pyarrow's outputs, so the oracle and the HIP path are
     each checked against a third party rather than against each other.
  4. gen_pin.json — first rows and column checksums of the synthetic generator, so the CPU and
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


def _tbl(batch):
    cols = {}
    for k, c in batch.items():
        vals = c.to_pylist()
        if c.dtype == "Utf8":
            cols[k] = pa.array(vals, pa.string())
        elif c.dtype == "Date32":
            cols[k] = pa.array(vals, pa.int32())
        else:
            cols[k] = pa.array(vals, {"Int32": pa.int32(), "Int64": pa.int64(), "Float64": pa.float64(), "UInt64": pa.uint64(),
                                      "Boolean": pa.bool_(), "UInt8": pa.uint8()}[c.dtype])
    return pa.table(cols)


def _exact_group_sums(keys, values):
    """{key tuple: correctly rounded sum of `values`} with exact rational accumulation"""
    acc = {}
    for k, v in zip(keys, values):
        acc[k] = acc.get(k, Fraction(0)) + Fraction(float(v))
    return {k: float(v) for k, v in acc.items()}


def acero_q3(customer, orders, lineitem):
    c = customer.filter(pc.equal(customer["c_mktsegment"], "BUILDING")).select(["c_custkey"])
    o = orders.filter(pc.less(orders["o_orderdate"], 9204))
    li = lineitem.filter(pc.greater(lineitem["l_shipdate"], 9204)).select(["l_orderkey", "l_extendedprice", "l_discount"])
    j1 = c.join(o, keys="c_custkey", right_keys="o_custkey", join_type="inner").select(["o_orderkey", "o_orderdate", "o_shippriority"])
    j2 = j1.join(li, keys="o_orderkey", right_keys="l_orderkey", join_type="inner")
    rev = pc.multiply(j2["l_extendedprice"], pc.subtract(1.0, j2["l_discount"]))
    j2 = j2.append_column("rev", rev)
    g = j2.group_by(["o_orderkey", "o_orderdate", "o_shippriority"]).aggregate([("rev", "sum")])
    exact = _exact_group_sums(zip(j2["o_orderkey"].to_pylist(), j2["o_orderdate"].to_pylist(), j2["o_shippriority"].to_pylist()),
                              j2["rev"].to_pylist())
    rows = []
    for k, d, p, r in zip(g["o_orderkey"].to_pylist(), g["o_orderdate"].to_pylist(), g["o_shippriority"].to_pylist(), g["rev_sum"].to_pylist()):
        e = exact[(k, d, p)]
        assert close(e, r, 1e-12), (k, e, r)
        rows.append(dict(l_orderkey=k, revenue=e, o_orderdate=d, o_shippriority=p))
    rows.sort(key=lambda r: (-r["revenue"], r["o_orderdate"], r["l_orderkey"]))
    return rows


def acero_q5(customer, orders, lineitem, supplier, nation, region):
    r = region.filter(pc.equal(region["r_name"], "ASIA")).select(["r_regionkey"])
    n = r.join(nation, keys="r_regionkey", right_keys="n_regionkey", join_type="inner").select(["n_nationkey", "n_name"])
    c = n.join(customer, keys="n_nationkey", right_keys="c_nationkey", join_type="inner").select(["c_custkey", "n_nationkey", "n_name"])
    o = orders.filter(pc.and_(pc.greater_equal(orders["o_orderdate"], 8766), pc.less(orders["o_orderdate"], 9131))).select(["o_orderkey", "o_custkey"])
    co = c.join(o, keys="c_custkey", right_keys="o_custkey", join_type="inner").select(["o_orderkey", "n_nationkey", "n_name"])
    col_ = co.join(lineitem.select(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]), keys="o_orderkey", right_keys="l_orderkey",
                   join_type="inner")
    j = supplier.join(col_, keys=["s_suppkey", "s_nationkey"], right_keys=["l_suppkey", "n_nationkey"], join_type="inner")
    rev = pc.multiply(j["l_extendedprice"], pc.subtract(1.0, j["l_discount"]))
    exact = _exact_group_sums(j["n_name"].to_pylist(), rev.to_pylist())
    g = j.append_column("rev", rev).group_by(["n_name"]).aggregate([("rev", "sum")])
    for k, v in zip(g["n_name"].to_pylist(), g["rev_sum"].to_pylist()):
        assert close(exact[k], v, 1e-12)
    return [dict(n_name=k, revenue=v) for k, v in sorted(exact.items(), key=lambda kv: -kv[1])]


def join_goldens():
    from collections import OrderedDict
    from oracle.engine import OCol
    li, od, cu, su = gen.lineitem(SF), gen.orders(SF), gen.customer(SF), gen.supplier(SF)
    q3 = acero_q3(_tbl(cu), _tbl(od), _tbl(li))
    assert len(q3) > 50
    json.dump(dict(sf=SF, seed=gen.SEED, engine=f"pyarrow {pa.__version__} + exact rational sums", rows=q3),
              open(os.path.join(OUT, "q3_synth.json"), "w"))
    q5 = acero_q5(_tbl(cu), _tbl(od), _tbl(li), _tbl(su), _tbl(gen.nation()), _tbl(gen.region()))
    assert 1 <= len(q5) <= 5
    json.dump(dict(sf=SF, seed=gen.SEED, engine=f"pyarrow {pa.__version__} + exact rational sums", rows=q5),
              open(os.path.join(OUT, "q5_synth.json"), "w"), indent=1)

    # joins with duplicate and NULL keys on both sides (SQL semantics: NULL never matches), inputs included
    rng = np.random.default_rng(2026)
    cases = []
    for name, nl, nr, kmax, null_p in (("dups_and_nulls", 40, 70, 12, 0.15), ("unique_build", 30, 90, 30, 0.0), ("no_match", 10, 10, 5, 0.0),
                                       ("all_null_left", 8, 20, 6, 1.0)):
        lk = rng.integers(0, kmax, nl)
        if name == "unique_build":
            lk = rng.permutation(kmax)[:nl]
        rk = rng.integers(0, kmax, nr) + (100 if name == "no_match" else 0)
        lkey = [None if rng.random() < null_p else int(v) for v in lk]
        rkey = [None if rng.random() < null_p * 0.5 else int(v) for v in rk]
        left = dict(k=lkey, lv=[float(i) + 0.5 for i in range(nl)], ls=[None if i % 7 == 3 else f"L{i % 5}" for i in range(nl)])
        right = dict(rk=rkey, rv=[int(i * 3) for i in range(nr)], rs=[f"R{i}" for i in range(nr)])
        lt = pa.table({"k": pa.array(left["k"], pa.int32()), "lv": pa.array(left["lv"], pa.float64()), "ls": pa.array(left["ls"], pa.string())})
        rt = pa.table({"rk": pa.array(right["rk"], pa.int32()), "rv": pa.array(right["rv"], pa.int64()), "rs": pa.array(right["rs"], pa.string())})
        want = {}
        for jt, pj in (("Inner", "inner"), ("Left", "left outer"), ("Right", "right outer")):
            # coalesce_keys=False keeps both key columns, as HashJoinExec does for differently named keys
            out = lt.join(rt, keys="k", right_keys="rk", join_type=pj, coalesce_keys=False).select(["k", "lv", "ls", "rk", "rv", "rs"])
            rows = [tuple(r[c] for c in ("k", "lv", "ls", "rk", "rv", "rs")) for r in out.to_pylist()]
            rows.sort(key=lambda r: tuple((0, 0) if v is None else (1, v) for v in r))
            want[jt] = rows
        cases.append(dict(name=name, left=left, right=right, expected=want))
    json.dump(dict(engine=f"pyarrow {pa.__version__} Table.join", columns=["k", "lv", "ls", "rk", "rv", "rs"], cases=cases),
              open(os.path.join(OUT, "join_cases.json"), "w"))

    # sorts: (Int32 with NULLs, Float64 with ties, Utf8) x (asc / desc) x (nulls first / last, one placement per sort: pyarrow's limit)
    n = 300
    a = [None if rng.random() < 0.1 else int(v) for v in rng.integers(-5, 5, n)]
    f = [None if rng.random() < 0.1 else float(v) / 4 for v in rng.integers(-8, 8, n)]
    sv = [None if rng.random() < 0.1 else "".join(chr(97 + int(c)) for c in rng.integers(0, 3, rng.integers(0, 4))) for _ in range(n)]
    rid = list(range(n))
    t = pa.table({"a": pa.array(a, pa.int32()), "f": pa.array(f, pa.float64()), "s": pa.array(sv, pa.string()), "rid": pa.array(rid, pa.int32())})
    sorts = []
    for keys in ([("a", False)], [("a", True), ("f", False)], [("s", False), ("a", True)], [("f", True), ("s", True), ("a", False)], [("s", True)]):
        for nulls_first in (True, False):
            # rid as the last key makes the expected order total (tie order is unspecified in the operator)
            order = [(k, "descending" if d else "ascending") for k, d in keys] + [("rid", "ascending")]
            idx = pc.sort_indices(t, sort_keys=order, null_placement="at_start" if nulls_first else "at_end")
            sorts.append(dict(keys=[dict(column=k, descending=d, nulls_first=nulls_first) for k, d in keys], order=idx.to_pylist()))
    json.dump(dict(engine=f"pyarrow {pa.__version__} sort_indices", table=dict(a=a, f=f, s=sv, rid=rid), sorts=sorts),
              open(os.path.join(OUT, "sort_cases.json"), "w"))


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

    # ---- 3. join queries, joins, sorts: pyarrow / Acero as the independent engine ---------------------------------
    join_goldens()

    # ---- 4. generator pin --------------------------------------------------------------------------------
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
