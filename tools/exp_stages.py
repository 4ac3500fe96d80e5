#!/usr/bin/env python3
"""where does a Q1 step go?  wall time of each operator of stages 2+3 on its own (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")
import ballista_amd as ba
from ballista_amd import tpch, expr as E, plan as P
from ballista_amd.expr import col

rows = int(os.environ.get("ROWS", 600_037_902))
ctx = ba.Context(0)
t = ba.plan.tpch_lineitem(ctx, 100.0, tpch.SEED, 0, rows)
scan = ba.MemoryExec([[t]], ctx)


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


stage1 = tpch.q1_stage1(scan)
part = stage1.collect()
timeit("stage1 (scan+filter+partial agg)", lambda: stage1.collect())
ctx.kernel_time(reset=True)
stage1.collect()
ms, n = ctx.kernel_time(reset=True)
print(f"  of which the fused kernel        {ms / max(n, 1):8.3f} ms")
mem = ba.MemoryExec([part], ctx)
group = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]
fin = P.HashAggregateExec(P.FINAL, group, tpch.q1_final_aggs(), P.MergeExec(mem))
timeit("final aggregate (4 rows)", lambda: fin.collect())
fout = fin.collect()
proj = P.ProjectionExec([(col(n), n) for n in ["l_returnflag", "l_linestatus"] + tpch.Q1_AGG_NAMES], ba.MemoryExec([fout], ctx))
timeit("projection (4 rows)", lambda: proj.collect())
pout = proj.collect()
srt = P.SortExec([E.PhysicalSortExpr(col("l_returnflag")), E.PhysicalSortExpr(col("l_linestatus"))], ba.MemoryExec([pout], ctx))
timeit("sort (4 rows)", lambda: srt.collect())
full = tpch.q1_final(stage1)
timeit("whole plan", lambda: full.collect())
