#!/usr/bin/env python3
"""ablation of the fused scan+aggregate kernel by plan shape (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")
import ballista_amd as ba
from ballista_amd import tpch, expr as E
from ballista_amd.expr import col, lit, coerce, Sum, Count

rows = int(os.environ.get("ROWS", 120_000_000))
ctx = ba.Context(0)
t = ba.plan.tpch_lineitem(ctx, 100.0, tpch.SEED, 0, rows)
scan = ba.MemoryExec([[t]], ctx)
S = tpch.LINEITEM_SCHEMA
q1 = tpch.q1_parts(S)
keys = q1["group"]
flt = lambda: ba.FilterExec(q1["predicate"], scan)
dp = coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), S)
shapes = {
    "q6": (tpch.q6_stage1(scan), 28),
    "nogroup_5sums": (ba.HashAggregateExec(ba.plan.PARTIAL, [], q1["aggs"], flt()), 36),
    "nogroup_1sum": (ba.HashAggregateExec(ba.plan.PARTIAL, [], [Sum(col("l_quantity"), "s")], flt()), 12),
    "keys_count_only": (ba.HashAggregateExec(ba.plan.PARTIAL, keys, [Count(lit(1, E.UINT8), "n")], flt()), 14),
    "keys_1sum": (ba.HashAggregateExec(ba.plan.PARTIAL, keys, [Sum(col("l_quantity"), "s")], flt()), 22),
    "key1_5sums": (ba.HashAggregateExec(ba.plan.PARTIAL, keys[:1], q1["aggs"], flt()), 41),
    "q1": (tpch.q1_stage1(scan), 46),
}
for name, (plan, bpr) in shapes.items():
    for _ in range(2):
        plan.collect()
    ctx.kernel_time(reset=True)
    for _ in range(5):
        plan.collect()
    ms, n = ctx.kernel_time(reset=True)
    k = ms / max(n, 1)
    print(f"{name:18s} kernel {k:7.3f} ms  {rows * bpr / k / 1e6:7.0f} GB/s  ({bpr} B/row, {rows / k / 1e6:6.1f} Grows/s)", flush=True)
