"""The library's own exchange (ballista_amd/csrc/host/exchange.cpp) on a one-GPU box:

  * bhip_batch_pack / bhip_batch_unpack — the block form every transport moves — round-trips every column type, NULLs, empty
    batches; an unpacked batch is ordinary operator input;
  * bhip_comm_* over RCCL with a world of ONE rank (all a one-GPU box can form; RCCL refuses two ranks on one device): the
    unique id, communicator creation through dlopen'ed librccl, all_gather / all_to_all return the caller's batches;
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
