#!/usr/bin/env python3
"""BASELINE.json config #5 (TPC-H Q5 at SF1000, hash-partition shuffle over 8 GPUs) rehearsed for ONE rank on the one GPU there is.

What rank r of N holds and does, at full per-rank size, with Int64 order keys in dbgen's sparse layout (keys up to 6 x 10^9):

  1. build side: EVERY source rank's orders block (1.5 G / N rows, generated and dropped one after the other) through Q5's
     build-side plan (region(ASIA) |x| nation |x| customer |x| orders(1994)); its rows are hash-partitioned N ways by o_orderkey
     (bhip_batch_hash_partition: the scatter the streaming shuffle runs per chunk) and bucket r is kept — what the exchange
     delivers to rank r;
  2. probe side: EVERY source rank's lineitem block (6 G / N rows x Q5's four columns = 28 B/row) partitioned the same way, bucket
     r kept; the bytes per destination bucket of rank r's own block are the xGMI numerator of the exchange;
  3. the local leg on the received sides: order-key join, supplier join on (suppkey, nationkey), partial aggregate — fresh plan
     per step; ms per step, the join kernels' share, peak device memory;
  4. the streaming shuffle itself (count pass + chunked scatter + placement) on rank r's own lineitem block over a loopback world
     of ONE other rank is not possible at this size on one device; its kernels are the partition_count / partition_scatter
     launches timed in 2.

    python tools/exp_q5_sf1000_rank_leg.py [--world 8] [--rank 0] [--sf 1000] [--steps 5] [--dense-keys]
    BHIP_RANK_WINDOW_LOG2=30 python tools/exp_q5_sf1000_rank_leg.py ...      # the CAS-table A/B partner (round-2 window limit)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BHIP_KERNEL_TIMING", "1")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--sf", type=float, default=1000.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dense-keys", action="store_true", help="order keys 1 .. 1.5e9 instead of dbgen's sparse layout (8 of every 32 values)")
    ap.add_argument("--sources", type=int, default=0, help="source ranks to walk (default: all of --world)")
    args = ap.parse_args()

    import ballista_amd as ba
    from ballista_amd import tpch, distributed as D
    from ballista_amd.expr import col
    P = ba.plan

    ctx = ba.Context(0)
    world, me, sf = args.world, args.rank, args.sf
    n = tpch.table_rows(sf)
    keys = dict(key64=True, sparse_keys=not args.dense_keys)
    li_cols = ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]
    od_cols = ["o_orderkey", "o_custkey", "o_orderdate"]
    out = dict(config=f"TPC-H Q5 SF{sf:g}, rank {me} of {world}, Int64 order keys ({'dense' if args.dense_keys else 'dbgen sparse layout'})",
               rank_window_log2=os.environ.get("BHIP_RANK_WINDOW_LOG2", "36 (default)"),
               lineitem_rows_total=n["lineitem"], orders_rows_total=n["orders"])
    print(json.dumps(dict(memory_plan={f"N={w}": D.q5_memory_plan(sf, w) for w in (2, 4, 8)})), flush=True)

    t0 = time.perf_counter()
    dims = tpch.dimension_tables(ctx, sf, "q5")
    leaf = lambda b: P.MemoryExec([[b]], ctx)
    small = {k: leaf(b) for k, b in dims.items()}
    out["dimension_tables_s"] = round(time.perf_counter() - t0, 2)
    print("small tables resident:", {k: b.num_rows for k, b in dims.items()}, f"{out['dimension_tables_s']} s", flush=True)

    sources = list(range(args.sources or world))
    empty_li = P.tpch_lineitem(ctx, sf, tpch.SEED, 0, 0, columns=li_cols, **keys)
    co_mine, li_mine = [], []
    rows_to = None
    part_ms = []
    for s in sources:
        # ---- build side of source rank s --------------------------------------------------------------------------------------
        lo, cnt = D.row_block(n["orders"], s, world)
        od = leaf(P.tpch_orders(ctx, sf, tpch.SEED, lo, cnt, columns=od_cols, **keys))
        co = tpch.q5_build_side(small["customer"], od, small["nation"], small["region"]).collect()
        co = [b for b in co if b.num_rows]
        co = co[0] if len(co) == 1 else P.concat(ctx, co)
        parts = P.hash_partition(co, [col("o_orderkey")], world)
        co_mine.append(parts[me])
        co_rows = co.num_rows
        del od, co, parts
        # ---- probe side of source rank s ----------------------------------------------------------------------------------------
        lo, cnt = D.row_block(n["lineitem"], s, world)
        li = P.tpch_lineitem(ctx, sf, tpch.SEED, lo, cnt, columns=li_cols, **keys)
        ctx.synchronize()
        ctx.kernel_stats(reset=True)
        t1 = time.perf_counter()
        parts = P.hash_partition(li, [col("l_orderkey")], world)
        ctx.synchronize()
        dt = (time.perf_counter() - t1) * 1e3
        part_ms.append(dt)
        if s == me:
            rows_to = [p.num_rows for p in parts]
        # a bucket is a slice of the scattered block: copy it out so that the block can go
        li_mine.append(P.concat(ctx, [parts[me], empty_li]))
        print(f"source {s}: orders block {D.row_block(n['orders'], s, world)[1]} rows -> {co_rows} build rows; lineitem block {li.num_rows} rows "
              f"partitioned {world} ways in {dt:.1f} ms ({li.num_rows * 28 * 2 / dt / 1e6:.0f} GB/s read+write), bucket {me}: {parts[me].num_rows} rows", flush=True)
        del li, parts
    ctx.synchronize()
    co_all = P.concat(ctx, co_mine) if len(co_mine) > 1 else co_mine[0]
    li_all = P.concat(ctx, li_mine) if len(li_mine) > 1 else li_mine[0]
    del co_mine, li_mine
    ctx.synchronize()
    out["received"] = dict(build_rows=co_all.num_rows, probe_rows=li_all.num_rows, probe_bytes=li_all.memory_size(), build_bytes=co_all.memory_size())
    if rows_to:
        row_bytes = 28
        out["own_block_rows_to_each_rank"] = rows_to
        out["own_block_bytes_to_each_rank"] = [r * row_bytes for r in rows_to]
        remote = sum(r for i, r in enumerate(rows_to) if i != me) * row_bytes
        out["exchange_bytes_out"] = remote
        out["xgmi_floor_ms_at_153GBs_per_link"] = round(max(r for i, r in enumerate(rows_to) if i != me) * row_bytes / 153e9 * 1e3, 2)
        out["bucket_skew_max_over_mean"] = round(max(rows_to) / (sum(rows_to) / len(rows_to)), 4)
    out["partition_ms_per_source_block"] = [round(x, 1) for x in part_ms]

    # ---- the local leg: join + supplier join + partial aggregate on what rank `me` received ----------------------------------------------
    plan = tpch.q5_partial(leaf(co_all), leaf(li_all), small["supplier"])
    cold = [tpch.fresh(plan) for _ in range(args.steps + 3)]
    res = None
    for _ in range(3):
        p = cold.pop()
        res = p.collect()
        del p
    ctx.synchronize()
    ctx.kernel_stats(reset=True)
    times = []
    for _ in range(args.steps):
        p = cold.pop()
        t1 = time.perf_counter()
        res = p.collect()
        ctx.synchronize()
        times.append((time.perf_counter() - t1) * 1e3)
        del p
    ks = ctx.kernel_stats(reset=True)
    in_use, peak = ctx.memory()
    out["local_leg"] = dict(ms_per_step_median=round(sorted(times)[len(times) // 2], 3), ms_per_step_all=[round(x, 3) for x in times],
                            probe_rows_per_s=li_all.num_rows / (sorted(times)[len(times) // 2] * 1e-3),
                            kernels_ms_per_step={k: round(v[0] / args.steps, 4) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][0])[:10]},
                            kernel_launches={k: v[1] for k, v in ks.items()},
                            partial_groups=sum(b.num_rows for b in res))
    out["device_memory"] = dict(in_use_bytes=in_use, peak_bytes=peak, note="peak includes this tool's 8 sequential source blocks and one extra concat copy of "
                                "the received side (the streaming shuffle places rows directly)")
    d = res[0].to_pydict()
    out["partial_result"] = {k: v for k, v in d.items()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
