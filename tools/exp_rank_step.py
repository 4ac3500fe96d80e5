#!/usr/bin/env python3
"""What the exchange nodes add to a rank's step (the N-rank step as lean as the 1-rank step).

A rank's distributed query is ONE operator tree (ballista_amd/distributed.py::rank_plan: AllGatherExec / ShuffleExchangeExec at the
stage boundaries) and a step is ONE bhip_plan_collect on a pre-built clone — no Python plan construction inside the timed loop,
for any world size.  This tool times, on the one GPU there is,

    local     the plan of the same query WITHOUT exchange nodes over a rank's row block            (world = 1)
    world N   N ranks as N threads of this process, each with its own context and a LOOPBACK communicator, every rank on
              its own row block of the same size — the exchange runs the communicator's real code (header gather, pack, the
              grouped point-to-point regions, unpack), only the bytes move by device-to-device copies instead of RCCL

over SMALL blocks (default 1 Mi lineitem rows per rank), where the step is all fixed cost: the difference between the two is
what the stage boundary costs a step — host work, launches and rendezvous of the exchange.  (At bench size the ranks of this
tool would share one GPU's bandwidth, which a real N-GPU run does not.)

    python tools/exp_rank_step.py [--rows 1048576] [--steps 200] [--world 2]
"""
import argparse
import json
import os
import statistics
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--sf", type=float, default=1.0)
    args = ap.parse_args()

    import ballista_amd as ba
    from ballista_amd import tpch, distributed as D
    P = ba.plan

    def tables(ctx, rank, query):
        t = {"lineitem": P.MemoryExec([[P.tpch_lineitem(ctx, args.sf, tpch.SEED, rank * args.rows, args.rows)]], ctx)}
        if query in ("q3", "q5"):
            n_ord = args.rows // 4
            t["orders"] = P.MemoryExec([[P.tpch_orders(ctx, args.sf, tpch.SEED, rank * n_ord, n_ord)]], ctx)
            for k, b in tpch.dimension_tables(ctx, args.sf, query).items():
                t[k] = P.MemoryExec([[b]], ctx)
        return t

    def time_steps(ctx, plan, barrier=None):
        cold = [tpch.fresh(plan) for _ in range(args.steps + 20)]
        for _ in range(20):
            p = cold.pop()
            p.collect()
            del p
        ctx.synchronize()
        if barrier:
            barrier.wait()
        ms = []
        for _ in range(args.steps):
            p = cold.pop()
            t0 = time.perf_counter()
            p.collect()
            ms.append((time.perf_counter() - t0) * 1e3)
            del p
        return ms

    out = dict(rows_per_rank=args.rows, steps=args.steps, world=args.world, note=__doc__.split("\n\n")[1].strip()[:0])
    for query, jx in (("q1", "-"), ("q6", "-"), ("q3", "shuffle"), ("q5", "shuffle"), ("q5", "broadcast")):
        ctx = ba.Context(0)
        local = time_steps(ctx, D.rank_plan(query, None, tables(ctx, 0, query), jx))
        del ctx
        hub = uuid.uuid4().bytes * 8
        res, errs = [None] * args.world, []
        bar = threading.Barrier(args.world)

        def body(r):
            try:
                c = ba.Context(0)
                comm = P.Communicator.loopback(c, hub, args.world, r)
                plan = D.rank_plan(query, comm, tables(c, r, query), jx)
                ms = time_steps(c, plan, bar)
                st = comm.stats(reset=True)
                res[r] = (ms, st)
                comm.close()
            except BaseException as e:          # noqa: BLE001
                errs.append(repr(e))
                bar.abort()

        ts = [threading.Thread(target=body, args=(r,)) for r in range(args.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise SystemExit("\n".join(errs))
        med_local = statistics.median(local)
        med_world = max(statistics.median(ms) for ms, _ in res)
        st = res[0][1]
        calls_per_step = st["calls"] / (args.steps + 20)
        key = f"{query}" + ("" if jx == "-" else f"_{jx}")
        out[key] = dict(local_ms=round(med_local, 4), world_ms=round(med_world, 4), added_by_exchange_ms=round(med_world - med_local, 4),
                        collectives_per_step=round(calls_per_step, 2), seconds_inside_collectives_per_step_ms=round(st["seconds"] / (args.steps + 20) * 1e3, 4),
                        bytes_out_per_step=int(st["bytes_out"] / (args.steps + 20)))
        print(key, out[key], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
