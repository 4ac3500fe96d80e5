#!/usr/bin/env python3
"""SortExec at size (run on the GPU box): ORDER BY revenue DESC, o_orderdate over N rows of Q3's result shape
(l_orderkey Int32, revenue Float64, o_orderdate Date32, o_shippriority Int32), checked against numpy's lexsort."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import ballista_amd as ba
from ballista_amd import expr as E, plan as P
from ballista_amd.expr import col

ctx = ba.Context(0)
n = int(os.environ.get("ROWS", 50_000_000))
rng = np.random.default_rng(3)
rev = np.round(rng.uniform(1000.0, 500000.0, n), 2)
date = rng.integers(8035, 9204, n).astype(np.int32)
key = np.arange(n, dtype=np.int32)
prio = np.zeros(n, np.int32)
t = ba.RecordBatch.from_pyarrow(ctx, pa.RecordBatch.from_arrays(
    [pa.array(key), pa.array(rev), pa.array(date).cast(pa.date32()), pa.array(prio)],
    names=["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]))
plan = P.SortExec([E.PhysicalSortExpr(col("revenue"), descending=True), E.PhysicalSortExpr(col("o_orderdate"))], ba.MemoryExec([[t]], ctx))
for it in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    out = plan.collect()
    ctx.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps(dict(rows=n, ms=dt * 1e3, rows_per_s=n / dt, bytes_per_row=20)), flush=True)
got_key = np.asarray(out[0].column(0)[1])
order = np.lexsort((date, -rev))                      # stable: ties keep input order, like the radix passes
assert np.array_equal(got_key, key[order]), "order differs from numpy lexsort"
print("sort check OK: equal to numpy lexsort (revenue desc, o_orderdate asc, stable)")
