#!/usr/bin/env python3
"""bench.py — TPC-H Q1 at SF100 on MI355X (BASELINE.json configs[1]); --query q6 | q3 | q5 are configs[2..4].

A step = one full pass of the query through the C ABI over synthetic TPC-H-shaped tables that are already resident
in HBM (generated on the device from a seed).  Every step runs a plan whose operators are NEW objects that have never
executed (`tpch.fresh`: a with_new_children clone of the whole tree, made before the timed region as a task's plan is
decoded before `plan.execute`): join build sides, hash tables and path choices are rebuilt inside the timed region,
as in a task that has just decoded its plan (rust/executor/src/flight_service.rs:87-121).

  python bench.py --gpus N --steps K --warmup W [--query q1]
      (two more untimed passes run before the W warmup steps, as part of the setup: they bring the library's caching allocator
      to its steady state, so that no timed step calls hipMalloc whatever W is; `setup_passes` in the output line)
      N > 1: one process per GPU.  When WORLD_SIZE is not set this process only spawns
      `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (before anything touches the
      GPU) and relays its output and exit code.
      STRONG scaling is the reported `value`: the fixed SF100 tables are split N ways by row blocks
      (the metric reads "Q1 SF100 at 1/2/4/8 GPUs"); the weak-scaling figure (every rank its own SF100 shard) is
      the extra key `weak_scaling`.  Q1 / Q6: stage 1 per shard, ONE small all_gather of the partial-state
      batches, Final aggregate on every rank (rust/scheduler/src/planner.rs:136-171).  Q3 / Q5: hash repartition of
      both join sides on the order key + all-to-all (ballista_amd/distributed.py).

Prints ONE JSON line (rank 0).  `value` = driving-table rows of the whole job / max-over-ranks wall time.
`roofline`: algorithmic bytes of one launch of the dominant kernel / its average duration, timed
with HIP events on the stream it ran on (BHIP_KERNEL_TIMING=1).  `cpu_baseline`: the oracle's threaded port of the
same query in DataFusion's structure (oracle/oracle_ops.c) on a bounded sample, on the host cores of the same box;
`cpu_baseline_acero`: Arrow C++ (pyarrow) on the same sample, a second, independent CPU engine.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # (the library's own default, ballista_amd/csrc/host/core.cpp: set before torch initialises HIP)

SF = 100.0
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--query", default="q1", choices=["q1", "q6", "q3", "q5"])
    ap.add_argument("--sf", type=float, default=SF, help="scale factor of the tables (default 100)")
    ap.add_argument("--rows", type=int, default=0, help="lineitem rows of the whole job (default: the scale factor's)")
    ap.add_argument("--key64", action="store_true", help="Int64 order keys (what TPC-H needs at SF1000)")
    ap.add_argument("--cpu-rows", type=int, default=96_000_000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the extra weak-scaling measurement")
    ap.add_argument("--join-exchange", default="shuffle", choices=["shuffle", "broadcast"],
                    help="N > 1, q3 / q5: hash-partition both sides of the order-key join (config #5) or broadcast the build side")
    ap.add_argument("--chunk-rows", type=int, default=0, help="N > 1, --join-exchange shuffle: rows per chunk of the streaming shuffle (0 = the library's default, 64 Mi)")
    ap.add_argument("--configs", default="q6,q3,q5", help="N = 1 and --query q1 (the default run): further queries measured after Q1 and "
                    "reported under \"configs\" (\"\" = none)")
    ap.add_argument("--config-steps", type=int, default=10, help="timed steps of each query under --configs")
    ap.add_argument("--config-cpu-rows", type=int, default=48_000_000, help="CPU-baseline sample rows of each query under --configs")
    ap.add_argument("--allow-host-exchange", action="store_true",
                    help="N > 1 with --backend nccl: if the RCCL communicator cannot be created, move batches through host memory over gloo "
                         "instead of exiting non-zero (the line then says so in exchange_backend)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="transport for N > 1 (nccl = RCCL; gloo only to rehearse the N-rank flow on a box with fewer GPUs)")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks as a CHILD process tree before this process
    touches the GPU (never exec from a process that has), relay rank 0's line and the exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    else:
        sys.stdout.write(proc.stdout)
    return proc.returncode if (proc.returncode != 0 or line is not None) else 1


# ---- CPU baselines (N = 1 only) -----------------------------------------------------------------------------

def _cpu_threads():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores (an 8-GPU host has 256): use that many threads
    return max(1, min(cores, int(os.environ.get("BHIP_CPU_THREADS", "16"))))


def _best_of(fn, budget_s=10.0, min_reps=3):
    best, total, reps = None, 0.0, 0
    while reps < min_reps or (total < budget_s and reps < 1000):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        total += dt
        reps += 1
    return best, total, reps


def cpu_baseline(query, sf, sample_rows, budget_s=10.0):
    """oracle port (DataFusion structure: 32768-row batches, materialised intermediates, one partition per thread) on
    rows [0, sample_rows) of the same seeded lineitem (Q3 / Q5: with the matching prefix of orders and the whole small
    tables)"""
    from oracle import gen
    cores = _cpu_threads()
    parts = cores * 4
    if query in ("q1", "q6"):
        a = gen.lineitem_arrays(sf, 0, sample_rows)
        fn = (lambda: gen.q1_partial_port(a, parts, cores)) if query == "q1" else (lambda: gen.q6_partial_port(a, parts, cores))
        what = f"oracle/oracle_ops.c::oracle_{query}_partial"
    else:
        from ballista_amd import tpch
        port = gen.JoinQueryPort(query, sf, sample_rows, tpch.dimension_arrays(sf))
        fn = lambda: port.run(parts, cores)
        what = f"oracle/oracle_ops.c::oracle_{query}_join_port (both hash-join builds inside the timed call)"
    best, total, reps = _best_of(fn, budget_s)
    return dict(value=sample_rows / best, unit="rows/s", cores=cores, kind="port",
                sample=f"lineitem rows [0,{sample_rows}) of the seeded SF{sf:g} table, {parts} partitions, best of {reps} passes "
                       f"({total:.1f} s of CPU work), {what}")


def cpu_baseline_acero(query, sf, sample_rows, budget_s=5.0):
    """Arrow C++ (pyarrow compute / Acero) on the same sample: an independent CPU engine, NOT the reference"""
    try:
        import numpy as np
        import pyarrow as pa
        import pyarrow.compute as pc
        from oracle import gen
    except ImportError:
        return None
    if query not in ("q1", "q6"):
        return None
    sample_rows = min(sample_rows, 24_000_000)
    a = gen.lineitem_arrays(sf, 0, sample_rows)
    cols = {k: pa.array(a[k]) for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_shipdate")}
    for k in ("l_returnflag", "l_linestatus"):
        cols[k] = pa.Array.from_buffers(pa.string(), sample_rows, [None, pa.py_buffer(a[k + ".off"]), pa.py_buffer(a[k + ".data"])])
    t = pa.table(cols)
    pa.set_cpu_count(_cpu_threads())

    def q1():
        f = t.filter(pc.less_equal(t["l_shipdate"], 10471))
        dp = pc.multiply(f["l_extendedprice"], pc.subtract(1.0, f["l_discount"]))
        ch = pc.multiply(dp, pc.add(1.0, f["l_tax"]))
        f = f.append_column("disc_price", dp).append_column("charge", ch)
        return f.group_by(["l_returnflag", "l_linestatus"]).aggregate(
            [("l_quantity", "sum"), ("l_extendedprice", "sum"), ("disc_price", "sum"), ("charge", "sum"), ("l_quantity", "mean"),
             ("l_extendedprice", "mean"), ("l_discount", "mean"), ("l_quantity", "count")])

    def q6():
        m = pc.and_(pc.and_(pc.greater_equal(t["l_shipdate"], 8766), pc.less(t["l_shipdate"], 9131)),
                    pc.and_(pc.and_(pc.greater_equal(t["l_discount"], 0.06 - 0.01), pc.less_equal(t["l_discount"], 0.06 + 0.01)),
                            pc.less(t["l_quantity"], 24.0)))
        f = t.filter(m)
        return pc.sum(pc.multiply(f["l_extendedprice"], f["l_discount"]))

    best, total, reps = _best_of(q1 if query == "q1" else q6, budget_s=budget_s)
    return dict(value=sample_rows / best, unit="rows/s", cores=_cpu_threads(), kind="acero",
                sample=f"pyarrow {pa.__version__} compute + group_by on rows [0,{sample_rows}), best of {reps} passes ({total:.1f} s)")


def pmc_traffic(kernel, rows, query):
    """(HBM bytes per launch of `kernel`, where that number comes from).  Counters cannot be read from inside a run: the
    bytes are those of the COMMITTED PMC passes (profiles/pmc_traffic*.json, written by tools/profile_bench.sh: FETCH_SIZE
    and WRITE_SIZE in separate rocprofv3 runs of this command), NOT a measurement of this run.  FETCH_SIZE is doubled
    (gfx950 counts wide streaming reads at half, MI355X_MICROARCH.md §HBM).  (None, None) when no pass exists for this
    kernel, query and table size."""
    for name in (f"pmc_traffic_{query}.json", "pmc_traffic.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if kernel and kernel in rec.get("kernel", "") and rec.get("rows_per_launch") == rows and rec.get("query", "q1") == query:
            return ((2.0 * rec["fetch_size_kb_per_launch"] + rec["write_size_kb_per_launch"]) * 1024.0,
                    f"profiles/{name} (committed rocprofv3 --pmc pass of this command, not measured in this run)")
    return None, None


def measure(query, args, ctx, group, world, rank, n, cpu_rows, cpu_budget_s, steps, warmup):
    """load `query`'s tables, time `steps` steps of it, return (line dict on rank 0 | None).  Frees the tables on return."""
    import ballista_amd as ba                                       # noqa: F401
    from ballista_amd import distributed as D

    key_bytes = 8 if args.key64 else 4
    W = D.Workload(query, ctx, group, sf=args.sf, rows=n, key64=args.key64, join_exchange=args.join_exchange, chunk_rows=args.chunk_rows)

    def barrier():
        ctx.synchronize()
        group.barrier()
        ctx.synchronize()

    step_times = os.environ.get("BHIP_BENCH_STEP_TIMES", "0") not in ("", "0")

    def timed(step, steps, warmup):
        result = None
        # two untimed passes before the warmup proper: the caching allocator reaches its steady state (the first pass of a
        # process allocates ~70 device buffers with hipMalloc, the second a few more; from the third on a step allocates none)
        W.prepare(steps + warmup + 2)
        for _ in range(warmup + 2):
            result = step()
        ctx.synchronize()
        ctx.kernel_stats(reset=True)
        ctx.kernel_time(reset=True)
        barrier()
        t0 = time.perf_counter()
        marks = []
        for _ in range(steps):
            result = step()
            if step_times:                                       # diagnosis only (BHIP_BENCH_STEP_TIMES=1): wall time of every step
                marks.append(time.perf_counter())
        barrier()
        if step_times and rank == 0:
            print("step ms:", [round((b - a) * 1e3, 3) for a, b in zip([t0] + marks[:-1], marks)], file=sys.stderr)
        elapsed = group.max_over_ranks(time.perf_counter() - t0)
        return elapsed, result, ctx.kernel_stats(reset=True)

    # ---- strong scaling on the fixed tables (N = 1: the whole tables) -------------------------------------------
    W.load(mode="strong")
    elapsed, result, kstats = timed(W.step, steps, warmup)
    exch = W.exchange_stats(reset=True)
    host = W.host_overhead(reset=True)

    weak = None
    # weak scaling (every rank its own full-size block) is defined for the scans only: a join's row blocks of different ranks must
    # come from ONE pair of tables for their keys to meet
    if world > 1 and not args.no_weak and query in ("q1", "q6"):
        W.load(mode="weak")
        w_steps = max(3, steps // 4)
        w_elapsed, _, _ = timed(W.step, w_steps, 1)
        weak = dict(value=n["lineitem"] * world * w_steps / w_elapsed, unit="rows/s", ms_per_step=w_elapsed / w_steps * 1e3,
                    steps=w_steps, rows_per_gpu=n["lineitem"], note="every rank holds its own full-size shard of the seeded tables")

    out = None
    if rank == 0:
        rows_job = n["lineitem"]
        rows_launch = W.rows_local("lineitem")                      # rows one launch of the dominant kernel covers
        dom = max(kstats.items(), key=lambda kv: kv[1][0]) if kstats else ("", (0.0, 0, 0))
        dom_name, (dom_ms, dom_n, dom_bytes) = dom
        kernel_ms = dom_ms / max(dom_n, 1)
        # algorithmic bytes of ONE launch: stated by the call site (join kernels) or rows x bytes/row of the fused scan
        algo_launch = dom_bytes / max(dom_n, 1) if dom_bytes else W.algorithmic_bytes_of_kernel(dom_name, key_bytes)
        achieved = algo_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        algo_job = W.algorithmic_bytes(key_bytes)
        traffic, traffic_source = pmc_traffic(dom_name, rows_launch, query)
        out = {
            "metric": f"tpch_{query}_sf{args.sf:g}_rows_per_sec", "value": rows_job * steps / elapsed, "unit": "rows/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "setup_passes": 2, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": W.describe(), "lineitem_rows": rows_job, "rows_per_gpu": rows_launch,
                       "partitioning": W.partitioning(), "plan_per_step": "an operator tree that has never run, built before the timed region; join builds and path choices happen inside it"},
            # algorithmic = SURVEY.md §8(d)'s compulsory column bytes of the whole query / step time: NOT a bandwidth-utilisation
            # figure (a fused probe never reads the payload of rows it rejects); the kernel-level number is `roofline`
            "algorithmic_gbs_whole_step": algo_job * steps / elapsed / 1e9,
            "algorithmic_frac_whole_step": algo_job * steps / elapsed / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": dom_name, "kernel_ms": kernel_ms,
                         "launches": int(dom_n), "algorithmic_bytes_per_launch": algo_launch},
            "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1][0])[:12]},
            "result_check": W.result_check(result),
        }
        if world > 1:
            # strong: the fixed SF tables are split N ways (the metric reads "Q1 SF100 at 1/2/4/8 GPUs"); at N = 1 there is nothing to scale
            out["scaling"] = "strong"
            out["exchange_backend"] = group.backend
            out["rccl"] = group.describe()
            if host:
                out["host_overhead"] = host
        if weak is not None:
            out["weak_scaling"] = weak
        if exch:
            out["exchange"] = exch
    del result
    W.unload()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and cpu_rows > 0:      # the CPU legs run at N=1 only
            sample = min(cpu_rows, n["lineitem"])
            out["cpu_baseline"] = cpu_baseline(query, args.sf, sample, cpu_budget_s)
            acero = cpu_baseline_acero(query, args.sf, sample, min(5.0, cpu_budget_s))
            if acero is not None:
                out["cpu_baseline_acero"] = acero
        else:
            out["cpu_baseline"] = None
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import ballista_amd as ba
    from ballista_amd import tpch, distributed as D

    group = D.ProcessGroup.from_env(args.backend, allow_host_exchange=args.allow_host_exchange) if world > 1 else D.ProcessGroup.single()
    ctx = ba.Context(group.device_index(local_rank))
    group.attach(ctx)                      # raises (exit code != 0) when RCCL cannot carry the batches and --allow-host-exchange is not set

    n = tpch.table_rows(args.sf)
    if args.rows:
        n["lineitem"] = args.rows

    out = measure(args.query, args, ctx, group, world, rank, n, args.cpu_rows, 10.0, args.steps, args.warmup)
    # the other single-GPU configurations of BASELINE.json (#3 Q6, #4 Q3, and the one-GPU leg of #5: Q5 at SF100), each with its own
    # step time, dominant-kernel roofline and CPU baseline; the headline metric / value / roofline / cpu_baseline stay Q1's
    extra = [q for q in args.configs.split(",") if q and q != args.query] if (world == 1 and args.query == "q1") else []
    configs = {}
    for q in extra:
        if q not in ("q1", "q6", "q3", "q5"):
            raise SystemExit(f"--configs: unknown query {q}")
        line = measure(q, args, ctx, group, world, rank, n, min(args.cpu_rows, args.config_cpu_rows), 4.0, args.config_steps, 2)
        if line is not None:
            configs[q if q != "q5" else f"q5_sf{args.sf:g}_1gpu"] = line
    if rank == 0:
        if configs:
            out["configs"] = configs
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


if __name__ == "__main__":
    main()
