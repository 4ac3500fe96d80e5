#!/usr/bin/env python3
"""BASELINE.json configs #3 and #4 on one GPU (run on the GPU box): Q6 at SF100, the standalone Filter
operator on Q6's predicate, and Q3 (two hash joins + 3-key aggregate + sort) at SF<Q3_SF>.

Inputs are generated on the device (lineitem, orders) or with numpy (customer: 5 market segments, uniform);
Q3's answer is checked against a numpy evaluation of the same query on host copies of the same tables
(integer columns and the row set exactly, revenue within 1e-9 relative).  No file under oracle/ is used."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")
import numpy as np
import ballista_amd as ba
from ballista_amd import tpch, expr as E, plan as P
from ballista_amd.expr import col

ctx = ba.Context(0)
out = {}


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        r = fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n, r


# ---- config #3: Q6 SF100 + standalone filter
rows = int(os.environ.get("ROWS", 600_037_902))
if os.environ.get("SKIP_Q6") is None:
    li = ba.plan.tpch_lineitem(ctx, 100.0, tpch.SEED, 0, rows)
    scan = ba.MemoryExec([[li]], ctx)
    q6 = tpch.q6_plan(scan)
    dt, res = timeit(lambda: q6.collect(), n=10)
    ctx.kernel_time(reset=True)
    q6.collect()
    kms, kn = ctx.kernel_time(reset=True)
    out["q6_sf100"] = dict(rows=rows, ms_per_query=dt * 1e3, rows_per_s=rows / dt, kernel=ctx.kernel_name(), kernel_ms=kms / max(kn, 1),
                           kernel_gbs=rows * 28 / (kms / max(kn, 1)) / 1e6, revenue=res[0].to_pydict()["revenue"][0])
    print(json.dumps(out["q6_sf100"]), flush=True)
    # the Filter operator alone, on the four columns Q6 reads (arrow `filter`: every column of surviving rows is copied)
    proj = P.ProjectionExec([(col(n), n) for n in ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"]], scan)
    narrow = proj.collect()
    flt = P.FilterExec(tpch.q6_predicate(tpch.LINEITEM_SCHEMA), ba.MemoryExec([narrow], ctx))
    dt, res = timeit(lambda: flt.collect(), n=5)
    sel = sum(b.num_rows for b in res)
    out["filter_q6_sf100"] = dict(rows=rows, selected=sel, ms=dt * 1e3, rows_per_s=rows / dt,
                                  algorithmic_gbs=(rows * 28 + sel * 4) / dt / 1e9)
    print(json.dumps(out["filter_q6_sf100"]), flush=True)
    del li, scan, q6, proj, narrow, flt, res

# ---- config #4: Q3
sf = float(os.environ.get("Q3_SF", "10"))
n_li = int(round(6_000_379.02 * sf)) if sf != 100 else 600_037_902
n_ord, n_cust = int(1_500_000 * sf), int(150_000 * sf)
t0 = time.perf_counter()
key64 = os.environ.get("KEY64", "0") == "1"          # Int64 order keys, as TPC-H needs at SF1000
li = ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, 0, n_li, key64=key64)
od = ba.plan.tpch_orders(ctx, sf, tpch.SEED, 0, n_ord, key64=key64)
rng = np.random.default_rng(7)
segs = np.array(["AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"])
seg_id = rng.integers(0, 5, n_cust)
import pyarrow as pa
cust_pa = pa.RecordBatch.from_arrays([pa.array(np.arange(1, n_cust + 1, dtype=np.int32)), pa.array(np.zeros(n_cust, np.int32)),
                                      pa.DictionaryArray.from_arrays(pa.array(seg_id.astype(np.int32)), pa.array(segs)).cast(pa.string())],
                                     names=["c_custkey", "c_nationkey", "c_mktsegment"])
cu = ba.RecordBatch.from_pyarrow(ctx, cust_pa)
ctx.synchronize()
print(f"tables ready in {time.perf_counter() - t0:.1f} s: lineitem {n_li}, orders {n_ord}, customer {n_cust}", flush=True)
q3 = tpch.q3_plan(ba.MemoryExec([[cu]], ctx), ba.MemoryExec([[od]], ctx), ba.MemoryExec([[li]], ctx))
t0 = time.perf_counter()
res = q3.collect()
ctx.synchronize()
first = time.perf_counter() - t0
print(f"first run {first * 1e3:.1f} ms, {sum(b.num_rows for b in res)} result rows", flush=True)
dt, res = timeit(lambda: q3.collect(), n=3, warm=1)
n_out = sum(b.num_rows for b in res)
algo = n_li * 24 + n_ord * 16 + n_cust * 17
out["q3"] = dict(sf=sf, lineitem_rows=n_li, ms_per_query=dt * 1e3, lineitem_rows_per_s=n_li / dt, result_rows=n_out,
                 algorithmic_gbs=algo / dt / 1e9)
print(json.dumps(out["q3"]), flush=True)

if os.environ.get("CHECK", "1") == "1" and sf <= 10:
    # numpy evaluation of Q3 on host copies of the same tables
    def host(rb, name):
        names = [rb.column_info(i)[0] for i in range(rb.num_columns)]
        return np.asarray(rb.column(names.index(name))[1])
    building = np.arange(1, n_cust + 1, dtype=np.int32)[seg_id == 1]
    o_key, o_cust, o_date, o_prio = (host(od, n) for n in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"))
    cutoff = 9204                                               # 1995-03-15
    om = (o_date < cutoff) & np.isin(o_cust, building)
    ok, odt, opr = o_key[om], o_date[om], o_prio[om]
    order = np.argsort(ok, kind="stable")
    ok, odt, opr = ok[order], odt[order], opr[order]
    l_key, l_price, l_disc, l_ship = (host(li, n) for n in ("l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"))
    lm = l_ship > cutoff
    lk, rev = l_key[lm], (l_price * (1.0 - l_disc))[lm]
    pos = np.searchsorted(ok, lk)
    pos[pos >= len(ok)] = 0
    hit = ok[pos] == lk
    gsum = np.bincount(pos[hit], weights=rev[hit], minlength=len(ok))
    gcnt = np.bincount(pos[hit], minlength=len(ok))
    want = {int(ok[i]): (gsum[i], int(odt[i]), int(opr[i])) for i in np.nonzero(gcnt)[0]}
    got_n = 0
    for b in res:
        d = b.to_pydict()
        for k, r, dd, pp in zip(d["l_orderkey"], d["revenue"], d["o_orderdate"], d["o_shippriority"]):
            w = want[int(k)]
            dd = (dd - __import__("datetime").date(1970, 1, 1)).days if not isinstance(dd, (int, np.integer)) else dd
            assert abs(r - w[0]) <= 1e-9 * abs(w[0]) and dd == w[1] and pp == w[2], (k, r, w)
            got_n += 1
    assert got_n == len(want), (got_n, len(want))
    revs = np.concatenate([np.asarray(b.to_pydict()["revenue"]) for b in res])
    assert np.all(revs[:-1] >= revs[1:]), "not sorted by revenue desc"
    print(f"Q3 check OK: {got_n} groups equal the numpy evaluation, sorted by revenue desc", flush=True)
print(json.dumps(out))
