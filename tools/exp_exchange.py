#!/usr/bin/env python3
"""host-side cost of the N-rank Q1 step, piece by piece (one process, gloo world of 1; run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch.distributed as dist
import pyarrow as pa
import ballista_amd as ba
from ballista_amd import tpch
from ballista_amd.exchange import all_gather_batches

dist.init_process_group("gloo", rank=0, world_size=1)
ctx = ba.Context(0)
rows = int(os.environ.get("ROWS", 1_000_000))
t = ba.plan.tpch_lineitem(ctx, 100.0, tpch.SEED, 0, rows)
stage1 = tpch.q1_stage1(ba.MemoryExec([[t]], ctx))
acc = {}


def lap(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


N = 50
for it in range(N + 5):
    if it == 5:
        acc.clear()
    t0 = time.perf_counter()
    dev = stage1.collect()[0]
    t0 = lap("stage1.collect", t0)
    part = dev.to_pyarrow()
    t0 = lap("to_pyarrow", t0)
    parts = all_gather_batches(dist, part, device="cpu")
    t0 = lap("all_gather_batches", t0)
    state = pa.Table.from_batches(parts * 8).combine_chunks().to_batches()[0]
    t0 = lap("concat (8 ranks' worth)", t0)
    rb = ba.RecordBatch.from_pyarrow(ctx, state)
    t0 = lap("from_pyarrow", t0)
    merged = ba.MemoryExec([[rb]], ctx)
    final = tpch.q1_final(merged)
    t0 = lap("build final plan", t0)
    res = final.collect()
    t0 = lap("final.collect", t0)
for k, v in acc.items():
    print(f"{k:28s} {v / N * 1e3:7.3f} ms")
dist.destroy_process_group()
