#!/usr/bin/env python3
"""bench.py — TPC-H Q1 at SF100 on MI355X (BASELINE.json configs[1]).

A step = one full pass of Q1 (scan -> filter -> partial aggregate -> merge -> final aggregate ->
projection -> sort) through the C ABI over a synthetic TPC-H-shaped lineitem table that is already
resident in HBM (600,037,902 rows per GPU, generated on the device from a seed).

  python bench.py --gpus N --steps K --warmup W
      N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...` (one rank per
      GPU).  Each rank owns one SF100-sized shard (rows [rank*R, (rank+1)*R) of the seeded table)
      and runs stage 1 on it; the per-rank partial states (<= 16 groups x 13 columns) are exchanged
      with ONE small all_gather (RCCL) and every rank runs the Final aggregate — the MergeExec of
      the reference's stage 2 (rust/scheduler/src/planner.rs:136-148).  Weak scaling.

Prints ONE JSON line (rank 0).  `value` = input rows of all ranks / max-over-ranks wall time.
`roofline`: algorithmic bytes (46 B/row) of one launch of the fused scan kernel /
its average duration, timed with HIP events on the stream it ran on (BHIP_KERNEL_TIMING=1).
`cpu_baseline`: the oracle's threaded port of the same stage in DataFusion's structure
(oracle/oracle_ops.c) on a bounded sample, on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")

SF = 100.0
ROWS_SF100 = 600_037_902
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS_SF100, help="rows per GPU (default: SF100 lineitem)")
    ap.add_argument("--query", default="q1", choices=["q1", "q6"])
    ap.add_argument("--cpu-rows", type=int, default=96_000_000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N-rank flow on a box with fewer GPUs)")
    return ap.parse_args()


def cpu_baseline(query, sample_rows):
    """oracle port (DataFusion structure: 32768-row batches, materialised intermediates, one partition
    per thread) on rows [0, sample_rows) of the same seeded table"""
    import numpy as np
    from oracle import gen
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores (an 8-GPU host has 256): use that many threads
    cores = max(1, min(cores, int(os.environ.get("BHIP_CPU_THREADS", "16"))))
    a = gen.lineitem_arrays(SF, 0, sample_rows)
    parts = max(cores, 1) * 4
    best = None
    t_total = 0.0
    reps = 0
    while reps < 3 or (t_total < 10.0 and reps < 1000):      # about 10 s of CPU work
        t0 = time.perf_counter()
        if query == "q1":
            gen.q1_partial_port(a, parts, cores)
        else:
            gen.q6_partial_port(a, parts, cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        t_total += dt
        reps += 1
    return dict(value=sample_rows / best, unit="rows/s", cores=cores, kind="port",
                sample=f"rows [0,{sample_rows}) of the seeded SF100 lineitem, {parts} partitions, best of {reps} passes "
                       f"({t_total:.1f} s of CPU work), oracle/oracle_ops.c::oracle_{query}_partial")


def pmc_traffic(kernel, rows, query):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written
    by tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs of this command).
    FETCH_SIZE is doubled (gfx950 counts wide streaming reads at half, MI355X_MICROARCH.md §HBM).
    None when no pass exists for this kernel and launch size — counters cannot be read from inside."""
    for name in ("pmc_traffic.json", f"pmc_traffic_{query}.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if kernel in rec.get("kernel", "") and rec.get("rows_per_launch") == rows and rec.get("query", "q1") == query:
            return (2.0 * rec["fetch_size_kb_per_launch"] + rec["write_size_kb_per_launch"]) * 1024.0
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        if args.backend == "gloo":
            local_rank = local_rank % max(n_dev, 1)        # rehearsal: ranks may share a GPU
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend)
    coll_device = "cpu" if args.backend == "gloo" else f"cuda:{local_rank}"

    import ballista_amd as ba
    from ballista_amd import tpch

    ctx = ba.Context(local_rank)
    rows = args.rows
    # each rank holds its own SF100-sized shard of the seeded table
    table = ba.plan.tpch_lineitem(ctx, SF, tpch.SEED, rank * rows, rows)
    scan = ba.MemoryExec([[table]], ctx)
    if args.query == "q1":
        stage1 = tpch.q1_stage1(scan)
        bytes_per_row = tpch.Q1_BYTES_PER_ROW
    else:
        stage1 = tpch.q6_stage1(scan)
        bytes_per_row = tpch.Q6_BYTES_PER_ROW

    if world == 1:
        plan = tpch.q1_final(stage1) if args.query == "q1" else tpch.q6_plan(scan)

        def step():
            return plan.collect()
    else:
        from ballista_amd.exchange import all_gather_batches

        def step():
            part = stage1.collect()[0].to_pyarrow()
            # ONE 16-KiB all_gather (RCCL) of the partial-state batches; tests/test_distributed_cpu.py
            parts = all_gather_batches(dist, part, device=coll_device)
            # one import for all ranks' state rows (MergeExec semantics: the partitions' rows, concatenated)
            import pyarrow as pa
            state = pa.Table.from_batches(parts).combine_chunks().to_batches()[0]
            merged = ba.MemoryExec([[ba.RecordBatch.from_pyarrow(ctx, state)]], ctx)
            if args.query == "q1":
                final = tpch.q1_final(merged)
            else:
                final = ba.HashAggregateExec(ba.plan.FINAL, [], [ba.expr.AggregateExpr("SUM", ba.expr.col("revenue[sum]"), "revenue")],
                                             ba.MergeExec(merged))
            return final.collect()

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
            ctx.synchronize()

    result = None
    for _ in range(args.warmup):
        result = step()
    ctx.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = ctx.kernel_time(reset=True)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_rows = rows * world * args.steps
        value = total_rows / elapsed
        kernel_ms = k_ms / max(k_launches, 1)
        algo_bytes = rows * bytes_per_row
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": f"tpch_{args.query}_sf100_rows_per_sec", "value": value, "unit": "rows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"TPC-H {args.query.upper()} SF100 (scan+filter+group-by aggregate), "
                                   f"{rows} lineitem rows per GPU resident in HBM, Arrow layout ({bytes_per_row} B/row)",
                       "rows_per_gpu": rows, "partitioning": f"{world} x SF100 shard, one partial-state all_gather"},
            "hbm_gbs_whole_step": rows * world * bytes_per_row * args.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(ctx.kernel_name(), rows, args.query),
                         "kernel": ctx.kernel_name(), "kernel_ms": kernel_ms, "launches": int(k_launches),
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        groups = result[0].to_pydict() if result else {}
        out["result_check"] = {"groups": len(next(iter(groups.values()))) if groups else 0,
                               "rows_counted": int(sum(groups.get("count_order", [0])))}
        if world == 1 and not args.no_cpu_baseline and args.cpu_rows > 0:      # the CPU leg runs at N=1 only
            # free the GPU table first? no: host memory only; the sample lives in host RAM
            out["cpu_baseline"] = cpu_baseline(args.query, min(args.cpu_rows, rows))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
