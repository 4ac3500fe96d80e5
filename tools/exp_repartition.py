#!/usr/bin/env python3
"""RepartitionExec(Hash(l_orderkey), N) on the probe-side columns of the join queries (run on the GPU box):
the local step of the multi-GPU exchange of BASELINE.json config #5 — row hash, stable split, one gather per
partition.  Checks: every row lands in exactly one partition, rows of a partition keep their input order, the
partition id is the documented row hash % N, evaluated here with numpy."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ballista_amd as ba
from ballista_amd import tpch, plan as P
from ballista_amd.expr import col

ctx = ba.Context(0)
rows = int(os.environ.get("ROWS", 600_037_902))
nparts = int(os.environ.get("PARTS", 8))
li = ba.plan.tpch_lineitem(ctx, 100.0, tpch.SEED, 0, rows)
proj = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]], ba.MemoryExec([[li]], ctx)).collect()[0]
ctx.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    parts = ba.plan.hash_partition(proj, [col("l_orderkey")], nparts)
    ctx.synchronize()
    dt = time.perf_counter() - t0
sizes = [p.num_rows for p in parts]
bytes_row = 4 + 4 + 8 + 8
print(json.dumps(dict(rows=rows, partitions=nparts, ms=dt * 1e3, rows_per_s=rows / dt, gbs_read_plus_write=2 * rows * bytes_row / dt / 1e9,
                      sizes=sizes)), flush=True)
assert sum(sizes) == rows
if rows <= 100_000_000:
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix64(z):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        return z ^ (z >> np.uint64(31))

    key = np.asarray(proj.column(0)[1]).astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        pid = (mix64(key) % np.uint64(nparts)).astype(np.int64)
    price = np.asarray(proj.column(2)[1])
    for p, part in enumerate(parts):
        want = np.nonzero(pid == p)[0]
        assert part.num_rows == len(want), (p, part.num_rows, len(want))
        assert np.array_equal(np.asarray(part.column(0)[1]).astype(np.int64).astype(np.uint64), key[want])     # input order kept
        assert np.array_equal(np.asarray(part.column(2)[1]), price[want])
    print("repartition check OK: partition = row_hash % N, input order kept inside each partition")
