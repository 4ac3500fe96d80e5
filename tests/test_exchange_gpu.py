"""Device-to-device exchange of partition columns through RCCL (ballista_amd/exchange.py::all_to_all_device).
A one-GPU box can only form a world of ONE rank: this checks the plumbing the N-rank run relies on — device
pointers of partition slices seen as torch tensors, uneven-split all_to_all into one receive buffer per column,
the received buffers wrapped as a batch without a copy and usable as operator input.  The N-rank routing logic
(who sends what to whom, source-rank order) is covered by the two-rank gloo test of the host-staged variant
(tests/test_distributed_cpu.py), which shares the count exchange and the ordering rules.

Runs in a child process that imports torch BEFORE the library, as bench.py does: PyTorch-ROCm ships its own HIP
runtime and must be the first to initialise the GPU in a process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from collections import OrderedDict
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import ballista_amd as ba
from ballista_amd import expr as E
from ballista_amd.exchange import all_to_all_device
from ballista_amd.expr import col
from oracle import plan_eval
from oracle.engine import OCol
import helpers
ctx = ba.Context(0)
rng = np.random.default_rng(4)
n = 50_001
t = OrderedDict([("k", OCol("Int32", rng.integers(0, 10 ** 6, n).astype(np.int32))), ("a", OCol("Float64", rng.random(n))),
                 ("c", OCol("Int64", rng.integers(0, 10 ** 12, n)))])
parts = ba.plan.hash_partition(helpers.to_device(ctx, t), [col("k")], 1)
got = all_to_all_device(dist, parts, ctx, "cuda:0")
assert got.num_rows == n
helpers.assert_rows_equal(helpers.from_device(got), t, ordered=True)
src = ba.MemoryExec([[got]], ctx)                      # the received batch is ordinary operator input
src._oracle_partitions = [[t]]
plan = ba.HashAggregateExec(ba.plan.PARTIAL, [], [E.Sum(col("a"), "s"), E.Count(E.lit(1, E.UINT8), "n")], src)
helpers.assert_rows_equal(helpers.concat(helpers.collect_product(plan)), plan_eval.collect(plan), ordered=False, float_rtol=1e-9)
empty = ba.plan.hash_partition(helpers.to_device(ctx, helpers.slice_batch(t, 0, 0)), [col("k")], 1)
assert all_to_all_device(dist, empty, ctx, "cuda:0").num_rows == 0      # an empty partition travels too
# the bench's exchange (bench.py --gpus N): one RCCL all_gather of the packed partial-state batches, device buffers
import pyarrow as pa
from ballista_amd.exchange import all_gather_batches
state = pa.RecordBatch.from_arrays([pa.array(["A", "N", "R"]), pa.array([1.5, 2.5, 3.5]), pa.array([7, 8, 9], pa.uint64())],
                                   names=["l_returnflag", "sum_qty[sum]", "count_order[count]"])
back = all_gather_batches(dist, state, device="cuda:0")
assert len(back) == 1 and back[0].equals(state)
dist.destroy_process_group()
print("EXCHANGE OK")
'''


def test_all_to_all_device_world_of_one():
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EXCHANGE OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
