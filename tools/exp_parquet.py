#!/usr/bin/env python3
"""ParquetExec scan figure (the reference's `--format parquet`, rust/benchmarks/tpch/src/main.rs:147-150,
scan node rust/core/src/serde/physical_plan/from_proto.rs:111-121).

A Snappy-compressed lineitem file of >= 1 GiB is written by pyarrow (the benchmark's `convert` writes Snappy, main.rs:84-86) from
the seeded synthetic table, then read three ways:

  * ParquetExec on the device path: footer + page walk on host threads (parquet_host.cpp: headers, Snappy, levels, run tables,
    string walks), values expanded on the device (dictionary runs, NULL re-insertion, gathers) — end to end, file already in the
    page cache, result resident in HBM;
  * the same with ONE host thread (BHIP_PARQUET_THREADS=1, in a child process: the knob is read once): the host-walk share;
  * pyarrow.parquet.read_table with the box's threads: the stated CPU baseline (an independent engine, not the reference).

    python tools/exp_parquet.py [--rows 40000000] [--dir /tmp]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BHIP_KERNEL_TIMING", os.environ.get("EXP_PARQUET_TIMING", "2"))

Q1_COLS = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]


def write_file(path, rows, row_group):
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    from oracle import gen
    schema = None
    writer = None
    step = 8_000_000
    for lo in range(0, rows, step):
        n = min(step, rows - lo)
        a = gen.lineitem_arrays(100.0, lo, n)
        cols = {"l_orderkey": pa.array(a["l_orderkey"]), "l_suppkey": pa.array(a["l_suppkey"])}
        for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"):
            cols[k] = pa.array(a[k])
        for k in ("l_returnflag", "l_linestatus"):
            cols[k] = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(a[k + ".off"]), pa.py_buffer(a[k + ".data"])])
        cols["l_shipdate"] = pa.array(a["l_shipdate"]).cast(pa.date32())
        # a comment-like column so that the file has TPC-H's share of string bytes (l_comment averages 27 bytes)
        rng = np.random.default_rng(lo)
        words = np.array(["carefully ", "final ", "deposits ", "sleep ", "quickly ", "express ", "ironic ", "packages "])
        pick = rng.integers(0, len(words), (n, 3))
        cols["l_comment"] = pa.array(np.char.add(np.char.add(words[pick[:, 0]], words[pick[:, 1]]), words[pick[:, 2]]))
        t = pa.table(cols)
        if writer is None:
            writer = pq.ParquetWriter(path, t.schema, compression="snappy", use_dictionary=True)
        writer.write_table(t, row_group_size=row_group)
    writer.close()


def device_read(path, columns, reps):
    import ballista_amd as ba
    import pyarrow.parquet as pq
    names = pq.ParquetFile(path).schema_arrow.names
    proj = [names.index(c) for c in columns] if columns else None
    ctx = ba.Context(0)
    best = None
    rows = 0
    for _ in range(reps):
        plan = ba.plan.ParquetExec([path], ctx, proj)
        ctx.synchronize()
        ctx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        out = plan.collect()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        rows = sum(b.num_rows for b in out)
        ks = ctx.kernel_stats(reset=True)
        if best is None or dt < best[0]:
            best = (dt, sum(v[0] for v in ks.values()), {k: round(v[0], 2) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][0])[:6]})
        del out, plan
    return dict(seconds=best[0], rows=rows, device_kernel_ms=round(best[1], 2), top_kernels_ms=best[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=60_000_000)
    ap.add_argument("--row-group", type=int, default=4_000_000)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    path = os.path.join(args.dir, f"lineitem_{args.rows}.snappy.parquet")
    if args.child:
        print(json.dumps(device_read(path, json.loads(args.child) or None, args.reps)))
        return
    if not os.path.exists(path):
        t0 = time.perf_counter()
        write_file(path, args.rows, args.row_group)
        print(f"wrote {path}: {os.path.getsize(path) / 2 ** 30:.2f} GiB in {time.perf_counter() - t0:.0f} s", flush=True)
    size = os.path.getsize(path)
    import pyarrow as pa
    import pyarrow.parquet as pq
    threads = min(16, len(os.sched_getaffinity(0)))
    pa.set_cpu_count(threads)
    out = dict(file=os.path.basename(path), file_bytes=size, rows=args.rows, host_threads=threads)
    for label, cols in (("all_columns", None), ("q1_columns", Q1_COLS)):
        best = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            t = pq.read_table(path, columns=cols, use_threads=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            nbytes = t.nbytes
            del t
        res = dict(pyarrow_read_table=dict(seconds=best, rows_per_s=args.rows / best, decoded_bytes=nbytes, threads=threads))
        for th in (threads, 1):
            env = dict(os.environ, BHIP_PARQUET_THREADS=str(th))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rows", str(args.rows), "--dir", args.dir, "--reps", str(args.reps),
                                "--child", json.dumps(cols or [])], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stdout + r.stderr)
            for ln in r.stderr.splitlines():
                if "bhip-parquet" in ln:
                    print(f"  [{th} threads] {ln}", flush=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            d["rows_per_s"] = d["rows"] / d["seconds"]
            d["file_gbs"] = size / d["seconds"] / 1e9 if cols is None else None
            res[f"device_path_{th}_host_threads"] = d
        a, b = res[f"device_path_{threads}_host_threads"], res["device_path_1_host_threads"]
        res["summary"] = (f"{label}: ParquetExec {a['rows_per_s'] / 1e6:.1f} M rows/s with {threads} host threads ({b['rows_per_s'] / 1e6:.1f} with one; device kernels "
                          f"{a['device_kernel_ms']:.0f} ms of {a['seconds'] * 1e3:.0f} ms), pyarrow.read_table {res['pyarrow_read_table']['rows_per_s'] / 1e6:.1f} M rows/s")
        out[label] = res
        print(res["summary"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
