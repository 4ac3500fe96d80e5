"""The library's own exchange (ballista_amd/csrc/host/exchange.cpp) on a one-GPU box:

  * bhip_batch_pack / bhip_batch_unpack — the block form every transport moves — round-trips every column type, NULLs, empty
    batches; an unpacked batch is ordinary operator input;
  * bhip_comm_* over RCCL with a world of ONE rank (all a one-GPU box can form; RCCL refuses two ranks on one device): the
    unique id, communicator creation through dlopen'ed librccl, all_gather / all_to_all return the caller's batches;
  * the N-rank code of the communicator — header matrices, `head_of(src, dst)` indexing, grouped point-to-point regions, both
    rounds of all_gather, the streaming shuffle's count matrix, chunk loop and final placement — with world = 2, 3, 8 over the
    LOOPBACK transport (N communicators in this process, one thread each, device-to-device copies): everything above the two
    byte-moving calls is the code RCCL runs under; expectations come from the oracle's restated row hash;
  * the exchange plan nodes: a rank's distributed Q1 / Q3 / Q5 as ONE tree, world = 2 and 3, against the one-rank answer;
  * `bench.py --gpus 2 --backend gloo`: the N-rank flow of the bench end to end — self-launch, row-block sharding, stage 1 on
    the device, exchange (pack -> gloo -> unpack), Final — with both ranks sharing the GPU, against the one-rank answer.
The N-rank routing itself is covered on CPU by tests/test_distributed_cpu.py with the same flow functions."""
import json
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E
from ballista_amd.expr import col
from oracle import plan_eval
from oracle.engine import OCol

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mixed_batch(n, seed=3):
    rng = np.random.default_rng(seed)
    valid = rng.random(n) > 0.2
    return OrderedDict([
        ("i", OCol("Int32", rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int32))),
        ("l", OCol("Int64", rng.integers(-10 ** 12, 10 ** 12, n), valid)),
        ("f", OCol("Float64", rng.random(n))),
        ("s", OCol("Utf8", ["" if i % 5 == 0 else f"str-{i % 17}-{'x' * (i % 9)}" for i in range(n)], rng.random(n) > 0.1)),
        ("d", OCol("Date32", rng.integers(8000, 11000, n).astype(np.int32))),
        ("b", OCol("Boolean", rng.random(n) > 0.5, rng.random(n) > 0.3)),
        ("u", OCol("UInt64", rng.integers(0, 2 ** 62, n).astype(np.uint64)))])


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1000, 70_001])
def test_pack_unpack_roundtrip(ctx, n):
    t = mixed_batch(n)
    dev = helpers.to_device(ctx, t)
    header, block = ba.plan.pack_batch(dev)
    assert header[0] == n and header[1] == block.size and block.size % 64 == 0
    back = ba.plan.unpack_batch(ctx, dev.schema3(), header, block)
    helpers.assert_rows_equal(helpers.from_device(back), t, ordered=True)
    if n:
        src = ba.MemoryExec([[back]], ctx)              # slices of one block are ordinary operator input
        src._oracle_partitions = [[t]]
        plan = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("d"), "d")], [E.Sum(col("f"), "sf"), E.Count(col("l"), "cl")],
                                    ba.FilterExec(E.IsNotNullExpr(col("s")), src))
        helpers.assert_rows_equal(helpers.concat(helpers.collect_product(plan)), plan_eval.collect(plan), ordered=False, float_rtol=1e-9,
                                  key_cols=["d"])


def test_rccl_communicator_world_of_one(ctx):
    uid = ba.plan.Communicator.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = ba.plan.Communicator(ctx, uid, 1, 0)
    t = mixed_batch(5000)
    dev = helpers.to_device(ctx, t)
    got = comm.all_gather(dev)
    assert len(got) == 1
    helpers.assert_rows_equal(helpers.from_device(got[0]), t, ordered=True)
    parts = ba.plan.hash_partition(dev, [col("i")], 1)
    back = comm.all_to_all(parts)
    assert len(back) == 1 and back[0].num_rows == 5000
    helpers.assert_rows_equal(helpers.from_device(back[0]), t, ordered=True)
    with pytest.raises(ValueError):
        comm.all_to_all(parts + parts)
    comm.close()
    with pytest.raises(ba.BallistaError):
        ba.plan.Communicator(ctx, uid, 2, 5)            # rank outside the world


# ---- the N-rank code paths over the loopback transport --------------------------------------------------------------------------

def run_world(world, fn, hub=None):
    """fn(rank, ctx, comm) on `world` threads, each with its own context and its own loopback communicator -> [result per rank]"""
    import threading
    import uuid
    hub = hub or uuid.uuid4().bytes * 8
    out, errs = [None] * world, []

    def body(r):
        try:
            c = ba.Context(0)
            comm = ba.plan.Communicator.loopback(c, hub, world, r)
            try:
                out[r] = fn(r, c, comm)
            finally:
                comm.close()
        except BaseException as e:          # noqa: BLE001 - re-raised in the test thread
            import traceback
            errs.append((r, "".join(traceback.format_exception(type(e), e, e.__traceback__))))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, "\n".join(f"rank {r}:\n{m}" for r, m in sorted(errs))
    return out


def rank_batch(rank, n, seed=0, fixed=False):
    """rank `rank`'s input: n rows, Int64 keys beyond 2^32 (config #5's key type); fixed: only fixed-width NULL-free columns"""
    rng = np.random.default_rng(1000 * seed + rank)
    cols = [("k", OCol("Int64", rng.integers(1, 6_000_000_000, n))), ("src", OCol("Int32", np.full(n, rank, np.int32))),
            ("row", OCol("Int32", np.arange(n, dtype=np.int32))), ("x", OCol("Float64", rng.random(n)))]
    if not fixed:
        cols += [("s", OCol("Utf8", [f"r{rank}-{i % 13}-{'y' * (i % 5)}" for i in range(n)], None if n == 0 else rng.random(n) > 0.15)),
                 ("l", OCol("Int64", rng.integers(-10 ** 9, 10 ** 9, n), None if n == 0 else rng.random(n) > 0.2))]
    return OrderedDict(cols)


SIZES = {2: [70_001, 0], 3: [5, 20_000, 3_001], 8: [0, 1, 63, 4_097, 50_000, 7, 12_345, 0]}          # uneven, empty, > 16 KiB


@pytest.mark.parametrize("world", [2, 3, 8])
def test_loopback_all_to_all_routes_every_part(world):
    """parts[d] of rank s arrive as out[s] on rank d: the (src, dst) header matrix and the grouped regions, parts of 0 rows,
    of a few rows and of megabytes, Utf8 and NULLs inside"""
    from oracle import engine as og

    def fn(rank, c, comm):
        dev = helpers.to_device(c, rank_batch(rank, SIZES[world][rank]))
        parts = ba.plan.hash_partition(dev, [col("k")], world)
        got = comm.all_to_all(parts)
        assert comm.info() == dict(world=world, rank=rank, transport="loopback")
        return [helpers.from_device(b) for b in got]

    got = run_world(world, fn)
    want = [og.repartition_hash(rank_batch(s, SIZES[world][s]), [col("k")], world) for s in range(world)]
    for d in range(world):
        for s in range(world):
            helpers.assert_rows_equal(got[d][s], want[s][d], ordered=True)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("big", [False, True])
def test_loopback_all_gather_both_rounds(world, big):
    """small batches ride in the header slot (one collective); one batch beyond the 16 KiB slot sends every rank through the
    point-to-point round"""
    sizes = [(3 + r) if not (big and r == world - 1) else 9_000 for r in range(world)]
    if big:
        sizes[0] = 0

    def fn(rank, c, comm):
        got = comm.all_gather(helpers.to_device(c, rank_batch(rank, sizes[rank], seed=5)))
        assert len(got) == world
        return [helpers.from_device(b) for b in got]

    got = run_world(world, fn)
    for me in range(world):
        for r in range(world):
            helpers.assert_rows_equal(got[me][r], rank_batch(r, sizes[r], seed=5), ordered=True)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("chunk_rows", [4096, 16384, 0])
@pytest.mark.parametrize("key", ["Int64", "Int32"])
def test_loopback_streaming_shuffle(world, chunk_rows, key):
    """bhip_comm_shuffle, chunked path: fixed-width NULL-free columns; ranks with different numbers of chunks (one with none);
    the result is the concatenation, in source-rank order, of every rank's partition for this rank — exactly what
    hash_partition + all_to_all + concat gives — and the statistics add up"""
    from oracle import engine as og
    sizes = {2: [70_001, 0], 3: [5, 20_000, 33_001], 8: [0, 1, 4_096, 4_097, 50_000, 7, 12_345, 30_000]}[world]

    def make(rank):
        b = rank_batch(rank, sizes[rank], seed=9, fixed=True)
        if key == "Int32":
            b["k"] = OCol("Int32", (b["k"].values % 2_000_000_000).astype(np.int32))
        return b

    def fn(rank, c, comm):
        out, st = comm.shuffle(helpers.to_device(c, make(rank)), "k", chunk_rows, with_stats=True)
        return helpers.from_device(out), st

    got = run_world(world, fn)
    parts = [og.repartition_hash(make(s), [col("k")], world) for s in range(world)]
    for d in range(world):
        batch, st = got[d]
        live = [parts[s][d] for s in range(world) if og.batch_len(parts[s][d])]
        want = og.concat_batches(live) if live else parts[0][d]
        helpers.assert_rows_equal(batch, want, ordered=True)
        assert st["streamed"] == 1 and st["rows_in"] == sizes[d] and st["rows_out"] == og.batch_len(want)
        assert st["rows_to"] == [og.batch_len(parts[d][t]) for t in range(world)]
        row_bytes = sum({"Int64": 8, "Int32": 4, "Float64": 8}[c.dtype] for c in make(d).values())
        assert st["bytes_sent_remote"] + st["bytes_kept_local"] == sizes[d] * row_bytes
        if chunk_rows:
            assert st["chunks"] == max(1, -(-max(sizes) // chunk_rows))


@pytest.mark.parametrize("world", [2, 3])
def test_loopback_shuffle_general_path(world):
    """Utf8 / NULL-able columns: the same call goes through hash_partition + all_to_all + concat inside the library"""
    from oracle import engine as og
    sizes = [9_000, 0, 777][:world]

    def fn(rank, c, comm):
        out, st = comm.shuffle(helpers.to_device(c, rank_batch(rank, sizes[rank], seed=4)), "k", 4096, with_stats=True)
        assert st["streamed"] == 0
        return helpers.from_device(out)

    got = run_world(world, fn)
    parts = [og.repartition_hash(rank_batch(s, sizes[s], seed=4), [col("k")], world) for s in range(world)]
    for d in range(world):
        live = [parts[s][d] for s in range(world) if og.batch_len(parts[s][d])]
        helpers.assert_rows_equal(got[d], og.concat_batches(live), ordered=True)


def test_loopback_collectives_refuse_mixed_schemas():
    def fn(rank, c, comm):
        a = helpers.to_device(c, OrderedDict([("k", OCol("Int64", np.arange(5, dtype=np.int64)))]))
        b = helpers.to_device(c, OrderedDict([("k", OCol("Int32", np.arange(5, dtype=np.int32)))]))
        with pytest.raises(ba.BallistaError):
            comm.all_to_all([a, b])                      # (refused before anything is exchanged: every rank fails alike)
        return True

    assert run_world(2, fn) == [True, True]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("query,join_exchange", [("q1", "-"), ("q6", "-"), ("q3", "shuffle"), ("q3", "broadcast"), ("q5", "shuffle"), ("q5", "broadcast")])
def test_loopback_rank_plans_equal_the_one_rank_answer(ctx, world, query, join_exchange):
    """a rank's distributed query as ONE operator tree (AllGatherExec / ShuffleExchangeExec at the stage boundaries,
    ballista_amd/distributed.py::rank_plan) executed by bhip_plan_collect on `world` ranks == the query on one rank; Int64 order
    keys in dbgen's sparse layout (config #5's key shape); small chunks so the streaming shuffle walks several"""
    from ballista_amd import tpch, distributed as D
    from oracle import gen
    sf = 0.01
    card = gen.cardinalities(sf)
    keys = dict(key64=True, sparse_keys=True)

    def tables(c, rank, n):
        t = {}
        lo, cnt = D.row_block(card["lineitem"], rank, n)
        t["lineitem"] = ba.MemoryExec([[ba.plan.tpch_lineitem(c, sf, tpch.SEED, lo, cnt, **keys)]], c)
        lo, cnt = D.row_block(card["orders"], rank, n)
        t["orders"] = ba.MemoryExec([[ba.plan.tpch_orders(c, sf, tpch.SEED, lo, cnt, **keys)]], c)
        for k, b in dict(customer=gen.customer(sf), supplier=gen.supplier(sf), nation=gen.nation(), region=gen.region()).items():
            t[k] = helpers.memory_exec(c, [[b]])
        return t

    key = {"q1": ["l_returnflag", "l_linestatus"], "q6": None, "q3": ["l_orderkey"], "q5": ["n_name"]}[query]

    def fn(rank, c, comm):
        plan = D.rank_plan(query, comm, tables(c, rank, world), join_exchange, chunk_rows=8192)
        text = plan.display()
        assert ("AllGatherExec" in text) and (("ShuffleExchangeExec" in text) == (join_exchange == "shuffle"))
        first = helpers.concat([helpers.from_device(b) for b in tpch.fresh(plan).collect()])
        again = helpers.concat([helpers.from_device(b) for b in tpch.fresh(plan).collect()])      # a clone shares the communicator
        # (SUM(Float64) is gated at 1e-6 relative; the merge order of a scan's per-workgroup partial sums is not pinned run to run)
        helpers.assert_rows_equal(first, again, ordered=False, float_rtol=1e-12, key_cols=key)
        return first

    got = run_world(world, fn)
    want = helpers.concat([helpers.from_device(b) for b in D.rank_plan(query, None, tables(ctx, 0, 1), join_exchange).collect()])
    key = {"q1": ["l_returnflag", "l_linestatus"], "q6": None, "q3": ["l_orderkey"], "q5": ["n_name"]}[query]
    for r in range(world):
        helpers.assert_rows_equal(got[r], want, ordered=False, float_rtol=1e-9, key_cols=key)


def test_host_transport_moves_the_same_blocks(ctx):
    """bhip_comm_create_host with a world of one rank's worth of callbacks that are never needed, and with two ranks whose
    callbacks hand the bytes over in this process: the `host` transport under the same communicator code"""
    import threading
    world = 2
    box = {"ag": [None] * world, "ex": {}}
    bar = threading.Barrier(world)

    def make(rank):
        def ag(send, recv):
            box["ag"][rank] = bytes(send)
            bar.wait()
            recv[:] = b"".join(box["ag"])
            bar.wait()

        def ex(sends, recvs):
            for i, (v, peer) in enumerate(sends):
                box["ex"].setdefault((rank, peer), []).append(bytes(v))
            bar.wait()
            taken = {}
            for v, peer in recvs:
                k = taken.get(peer, 0)
                v[:] = box["ex"][(peer, rank)][k]
                taken[peer] = k + 1
            bar.wait()
            if rank == 0:
                box["ex"].clear()
            bar.wait()
        return ag, ex

    out, errs = [None] * world, []

    def body(r):
        try:
            c = ba.Context(0)
            ag, ex = make(r)
            comm = ba.plan.Communicator.host(c, world, r, ag, ex)
            g = comm.all_gather(helpers.to_device(c, rank_batch(r, 3 + r, seed=2)))
            big = comm.all_gather(helpers.to_device(c, rank_batch(r, 5000 + r, seed=3)))
            sh = comm.shuffle(helpers.to_device(c, rank_batch(r, 20_000 + 17 * r, seed=6, fixed=True)), "k", 4096)
            out[r] = ([helpers.from_device(b) for b in g], [helpers.from_device(b) for b in big], helpers.from_device(sh), comm.info()["transport"])
            comm.close()
        except BaseException as e:          # noqa: BLE001
            import traceback
            errs.append("".join(traceback.format_exception(type(e), e, e.__traceback__)))
            bar.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, "\n".join(errs)
    from oracle import engine as og
    parts = [og.repartition_hash(rank_batch(s, 20_000 + 17 * s, seed=6, fixed=True), [col("k")], world) for s in range(world)]
    for me in range(world):
        g, big, sh, name = out[me]
        assert name == "host"
        for r in range(world):
            helpers.assert_rows_equal(g[r], rank_batch(r, 3 + r, seed=2), ordered=True)
            helpers.assert_rows_equal(big[r], rank_batch(r, 5000 + r, seed=3), ordered=True)
        helpers.assert_rows_equal(sh, og.concat_batches([parts[s][me] for s in range(world)]), ordered=True)


@pytest.mark.parametrize("query,extra", [("q1", []), ("q3", ["--join-exchange", "shuffle"]), ("q5", ["--join-exchange", "broadcast"])])
def test_bench_two_ranks_share_the_gpu_gloo(query, extra):
    """answer(2 ranks) == answer(1 rank) through bench.py itself, small tables"""
    def run(gpus):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--query", query, "--sf", "0.05",
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-weak"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and lines, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
        return json.loads(lines[-1])
    one, two = run(1), run(2)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and one["n_gpus"] == 1
    a, b = one["result_check"], two["result_check"]
    assert a["result_rows"] == b["result_rows"] > 0
    if query == "q1":
        assert a["rows_counted"] == b["rows_counted"] and a["groups"] == b["groups"] == 4
    else:
        assert np.allclose(a.get("revenue", []), b.get("revenue", []), rtol=1e-9)
    if query != "q1":
        assert two["exchange"]["calls"] > 0
