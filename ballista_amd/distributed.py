"""One process per GPU: how the stages of a query are spread over the ranks of one node and what moves between them.

The reference splits a query into stages at every exchange (rust/scheduler/src/planner.rs:136-171): stage 1 runs per
input partition, its output is pulled by the next stage through `ShuffleReaderExec`
(rust/core/src/execution_plans/shuffle_reader.rs:77-99), and a stage that repartitions by key carries
`RepartitionExec(Hash(exprs, n))` (rust/core/src/serde/physical_plan/from_proto.rs:133-147).  Here a rank is a
partition, the exchange is a collective over xGMI instead of Flight over TCP:

  * Q1 / Q6 — scan -> filter -> partial aggregate per row block; ONE small all_gather of the partial-state batches;
    MergeExec -> Final aggregate -> Sort on every rank.  No data-path collective.
  * Q3 / Q5 — the order-key join's sides are not co-partitioned.  `shuffle` (BASELINE.json config #5): both sides are
    split by `row_hash(orderkey) % N` on the device (`bhip_batch_hash_partition`) and exchanged all-to-all; every rank
    joins and aggregates its own keys; the partial states are all_gathered.  `broadcast`: the (filtered, small) build
    side is all_gathered instead and the probe side never moves — the reference's collect-left join, where every task
    sees the whole build side (from_proto.rs:253-276).

The flows are written against two small interfaces so that the SAME code runs under the CPU tests
(tests/test_distributed_cpu.py: 2 gloo ranks, the oracle as the engine) and on the GPUs:

  Engine  — how one rank computes: ProductEngine = the plans of ballista_amd.plan on the HIP library.
  Group   — how batches travel: RcclGroup (bhip_comm_*: RCCL on device buffers, inside libballista_hip.so),
            GlooGroup (the same blocks through host memory over torch.distributed gloo: rehearsal and CPU tests), SingleGroup.
"""
from __future__ import annotations

import os
import time
from typing import List, Sequence

from . import tpch
from .expr import col


# ---- engines ---------------------------------------------------------------------------------------------------

class ProductEngine:
    """plans on libballista_hip.so; batches are device-resident ballista_amd.plan.RecordBatch"""

    def __init__(self, ctx):
        from . import plan as P
        self.P, self.ctx = P, ctx

    def leaf(self, partitions):
        """MemoryExec over `partitions` (a batch, or a list of batches = one partition each)"""
        if not isinstance(partitions, (list, tuple)):
            partitions = [partitions]
        return self.P.MemoryExec([[b] for b in partitions], self.ctx)

    def run(self, plan):
        """execute every partition of a NEW copy of the plan (tpch.fresh) -> one batch"""
        plan = tpch.fresh(plan)
        out = [b for b in plan.collect() if b.num_rows]
        if not out:
            return self.empty(plan.schema())
        return out[0] if len(out) == 1 else self.P.concat(self.ctx, out)

    def empty(self, schema):
        import numpy as np
        from . import expr as E
        cols = []
        for name, dtype, _ in schema:
            vals = [] if dtype in (E.UTF8,) else np.zeros(0, self.P.NP_DTYPE.get(dtype, np.bool_))
            cols.append((name, dtype, vals, None))
        return self.P.RecordBatch.from_columns(self.ctx, cols)

    def hash_partition(self, batch, key, n):
        return self.P.hash_partition(batch, [col(key)], n)

    def concat(self, batches):
        live = [b for b in batches if b.num_rows]
        if not live:
            return batches[0]
        return live[0] if len(live) == 1 else self.P.concat(self.ctx, live)

    def num_rows(self, batch):
        return batch.num_rows

    def nbytes(self, batch):
        return batch.memory_size()

    def to_wire(self, batch):
        """the block form the library's exchange moves (bhip_batch_pack), as host bytes: header words, then the block"""
        import numpy as np
        header, block = self.P.pack_batch(batch)
        return np.concatenate([np.array([header.size], np.int64).view(np.uint8), header.view(np.uint8), block])

    def from_wire(self, raw, like):
        import numpy as np
        raw = np.ascontiguousarray(raw, np.uint8)
        n = int(raw[:8].view(np.int64)[0])
        header = raw[8:8 + 8 * n].view(np.int64).copy()
        return self.P.unpack_batch(self.ctx, like.schema3(), header, raw[8 + 8 * n:])


# ---- groups ----------------------------------------------------------------------------------------------------

class SingleGroup:
    rank, world, backend = 0, 1, "single"

    def device_index(self, local_rank):
        return local_rank

    def attach(self, ctx):
        pass

    def barrier(self):
        pass

    def max_over_ranks(self, x):
        return x

    def all_gather(self, eng, batch):
        return [batch]

    def all_to_all(self, eng, parts):
        return list(parts)

    def describe(self):
        return dict(world=1, ranks_seen=[0], transport="none")

    def close(self):
        pass


class GlooGroup:
    """torch.distributed (gloo) for control AND payload: batches travel through host memory in the engine's wire form
    (ProductEngine: the packed block of bhip_batch_pack; the CPU tests' oracle engine: Arrow IPC stream bytes).
    The CPU tests' transport; on a GPU box it rehearses the N-rank flow with ranks sharing the GPU."""
    backend = "gloo"

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_dev = None

    def device_index(self, local_rank):
        import ctypes
        try:                                      # ranks may share a GPU in a rehearsal
            n = ctypes.c_int(0)
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGetDeviceCount(ctypes.byref(n))
            return local_rank % max(1, n.value)
        except OSError:
            return 0

    def attach(self, ctx):
        pass

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, eng, batch):
        """every rank's batch, in rank order (the order MergeExec concatenates partitions in)"""
        import torch
        raw = eng.to_wire(batch)
        size = torch.tensor([raw.size], dtype=torch.int64)
        sizes = [torch.empty_like(size) for _ in range(self.world)]
        self.dist.all_gather(sizes, size)
        sizes = [int(s.item()) for s in sizes]
        cap = max(sizes)
        buf = torch.zeros(cap, dtype=torch.uint8)
        buf[:raw.size] = torch.from_numpy(raw.copy())
        out = torch.empty(self.world * cap, dtype=torch.uint8)
        self.dist.all_gather_into_tensor(out, buf)
        host = out.numpy().reshape(self.world, cap)
        return [batch if r == self.rank else eng.from_wire(host[r, :sizes[r]], batch) for r in range(self.world)]

    def all_to_all(self, eng, parts):
        """parts[d] goes to rank d; returns what every rank holds for me, in source-rank order"""
        import torch
        if len(parts) != self.world:
            raise ValueError(f"need one outgoing batch per rank ({self.world}), got {len(parts)}")
        payload = [eng.to_wire(p) for p in parts]
        sizes = torch.tensor([p.size for p in payload], dtype=torch.int64)
        all_sizes = [torch.empty_like(sizes) for _ in range(self.world)]
        self.dist.all_gather(all_sizes, sizes)
        incoming = [int(all_sizes[src][self.rank].item()) for src in range(self.world)]
        send = [torch.from_numpy(payload[d].copy()) for d in range(self.world)]
        recv = [torch.empty(incoming[s], dtype=torch.uint8) for s in range(self.world)]
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            ops.append(self.dist.P2POp(self.dist.isend, send[peer], peer))
            ops.append(self.dist.P2POp(self.dist.irecv, recv[peer], peer))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        return [parts[s] if s == self.rank else eng.from_wire(recv[s].numpy(), parts[s]) for s in range(self.world)]

    def describe(self):
        return dict(world=self.world, ranks_seen=list(range(self.world)), transport="gloo (host memory)")

    def close(self):
        self.dist.destroy_process_group()


class RcclGroup(GlooGroup):
    """Control (barrier, timing reduce, the RCCL unique id) over gloo on the CPU; every batch moves device to device
    through the library's own RCCL communicator (bhip_comm_*, csrc/host/exchange.cpp): the process never initialises a
    second HIP runtime through PyTorch."""
    backend = "nccl"

    def __init__(self, dist, allow_host_exchange=False):
        super().__init__(dist)
        self.comm = None
        self.allow_host_exchange = allow_host_exchange
        self.ranks_seen = []

    def device_index(self, local_rank):
        return local_rank

    def describe(self):
        if self.comm is None:
            return dict(world=self.world, ranks_seen=[], transport="NONE: RCCL unavailable, batches through host memory over gloo")
        return dict(world=self.world, ranks_seen=self.ranks_seen, transport="rccl")

    def attach(self, ctx):
        import torch
        from . import plan as P
        uid = torch.zeros(P.Communicator.UNIQUE_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            uid = torch.frombuffer(bytearray(P.Communicator.unique_id()), dtype=torch.uint8).clone()
        self.dist.broadcast(uid, src=0)
        # The communicator is created and exercised once (a one-row all_gather checked against what every rank must see).  If any
        # rank cannot do that — no RCCL on the box, a transport the fabric refuses — ALL ranks agree (over gloo) to move batches
        # through host memory instead, and the bench line says so: a scaling run still completes and reports what it measured.
        err = None
        try:
            self.comm = P.Communicator(ctx, bytes(uid.numpy().tobytes()), self.world, self.rank)
            mine = P.RecordBatch.from_columns(ctx, [("r", "Int32", [self.rank], None)])
            got = self.comm.all_gather(mine)
            seen = [int(b.column(0)[1][0]) for b in got]
            self.ranks_seen = seen
            if seen != list(range(self.world)):
                err = f"all_gather self-test returned {seen}"
        except Exception as e:                                   # noqa: BLE001 — any failure means the same thing here
            err = f"{type(e).__name__}: {e}"
        flag = torch.tensor([1 if err else 0], dtype=torch.int32)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if int(flag[0]):
            if self.comm is not None:
                try:
                    self.comm.close()
                except Exception:                                # noqa: BLE001
                    pass
            self.comm = None
            why = "RCCL communicator unavailable: " + (err or "on another rank")
            if not self.allow_host_exchange:
                # a scaling record must not show N ranks and a plausible value with RCCL never having moved a byte
                raise RuntimeError(why + " — refusing to fall back to host-staged gloo (pass --allow-host-exchange to bench.py to rehearse that way)")
            self.backend = "gloo, batches through host memory (" + why + ")"

    def all_gather(self, eng, batch):
        if self.comm is None:
            return super().all_gather(eng, batch)
        return self.comm.all_gather(batch)

    def all_to_all(self, eng, parts):
        if self.comm is None:
            return super().all_to_all(eng, parts)
        return self.comm.all_to_all(parts)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        super().close()


class ProcessGroup:
    @staticmethod
    def single():
        return SingleGroup()

    @staticmethod
    def from_env(backend="nccl", allow_host_exchange=False):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them"""
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo")
        return RcclGroup(dist, allow_host_exchange) if backend == "nccl" else GlooGroup(dist)


# ---- distributed query flows (engine- and transport-agnostic) -------------------------------------------------------

class ExchangeStats:
    def __init__(self):
        self.reset()

    def reset(self):
        self.seconds, self.bytes_out, self.calls = 0.0, 0, 0

    def add(self, dt, nbytes):
        self.seconds += dt
        self.bytes_out += nbytes
        self.calls += 1


def _shuffle(eng, group, batch, key, stats=None, sync=None):
    """RepartitionExec(Hash([key], world)) + the shuffle read: my rows of every rank's batch"""
    t0 = time.perf_counter()
    parts = eng.hash_partition(batch, key, group.world)
    out_bytes = sum(eng.nbytes(p) for r, p in enumerate(parts) if r != group.rank)
    got = group.all_to_all(eng, parts)
    mine = eng.concat(got)
    if sync is not None:
        sync()
    if stats is not None:
        stats.add(time.perf_counter() - t0, out_bytes)
    return mine


def q1_distributed(eng, group, lineitem, query="q1"):
    """stage 1 on this rank's rows, all_gather of the partial states, Merge -> Final (-> Sort) on every rank"""
    if query == "q1":
        part = eng.run(tpch.q1_stage1(lineitem))
        states = group.all_gather(eng, part)
        return eng.run(tpch.q1_final(eng.leaf(states)))
    part = eng.run(tpch.q6_stage1(lineitem))
    states = group.all_gather(eng, part)
    return eng.run(tpch.q6_final(eng.leaf(states)))


def q3_distributed(eng, group, customer, orders, lineitem, join_exchange="shuffle", stats=None, sync=None):
    """customer is replicated, orders / lineitem are this rank's row blocks"""
    j1 = eng.run(tpch.q3_build_side(customer, orders))
    if join_exchange == "broadcast":
        t0 = time.perf_counter()
        j1_all = eng.concat(group.all_gather(eng, j1))
        if stats is not None:
            if sync is not None:
                sync()
            stats.add(time.perf_counter() - t0, eng.nbytes(j1) * (group.world - 1))
        partial = eng.run(tpch.q3_partial(eng.leaf(j1_all), tpch.q3_probe_side(lineitem)))
    else:
        li = eng.run(tpch.q3_probe_side(lineitem))
        j1_mine = _shuffle(eng, group, j1, "o_orderkey", stats, sync)
        li_mine = _shuffle(eng, group, li, "l_orderkey", stats, sync)
        partial = eng.run(tpch.q3_partial(eng.leaf(j1_mine), eng.leaf(li_mine)))
    states = group.all_gather(eng, partial)
    return eng.run(tpch.q3_final(eng.leaf(states)))


def q5_distributed(eng, group, customer, orders, lineitem, supplier, nation, region, join_exchange="shuffle", stats=None, sync=None):
    """customer / supplier / nation / region are replicated, orders / lineitem are this rank's row blocks"""
    co = eng.run(tpch.q5_build_side(customer, orders, nation, region))
    if join_exchange == "broadcast":
        t0 = time.perf_counter()
        co_all = eng.concat(group.all_gather(eng, co))
        if stats is not None:
            if sync is not None:
                sync()
            stats.add(time.perf_counter() - t0, eng.nbytes(co) * (group.world - 1))
        partial = eng.run(tpch.q5_partial(eng.leaf(co_all), tpch.q5_probe_side(lineitem), supplier))
    else:
        li = eng.run(tpch.q5_probe_side(lineitem))
        co_mine = _shuffle(eng, group, co, "o_orderkey", stats, sync)
        li_mine = _shuffle(eng, group, li, "l_orderkey", stats, sync)
        partial = eng.run(tpch.q5_partial(eng.leaf(co_mine), eng.leaf(li_mine), supplier))
    states = group.all_gather(eng, partial)
    return eng.run(tpch.q5_final(eng.leaf(states)))


# ---- bench.py's workloads ------------------------------------------------------------------------------------------

class Workload:
    """tables in HBM + one step of a query, for bench.py.  mode "strong": the fixed tables split N ways by row
    block; "weak": every rank its own full-size block."""

    def __init__(self, query, ctx, group, sf, rows, key64=False, join_exchange="shuffle"):
        self.query, self.ctx, self.group, self.sf, self.rows, self.key64 = query, ctx, group, sf, dict(rows), key64
        self.join_exchange = join_exchange
        self.eng = ProductEngine(ctx)
        self.stats = ExchangeStats()
        self.local = {}
        self.t = {}
        self.plan = None
        self.cold, self.spent = [], []

    def _block(self, table, mode):
        n = self.rows[table]
        if mode == "weak" or self.group.world == 1:
            return self.group.rank * n, n
        per = (n + self.group.world - 1) // self.group.world
        lo = min(n, self.group.rank * per)
        return lo, min(n, lo + per) - lo

    def load(self, mode):
        P, ctx = self.eng.P, self.ctx
        self.t.clear()                         # release the previous tables first
        self.plan = None
        self.cold, self.spent = [], []
        lo, n = self._block("lineitem", mode)
        self.local["lineitem"] = n
        li = P.tpch_lineitem(ctx, self.sf, tpch.SEED, lo, n, key64=self.key64)
        self.t["lineitem"] = self.eng.leaf(li)
        if self.query in ("q3", "q5"):
            lo, n = self._block("orders", mode)
            self.local["orders"] = n
            self.t["orders"] = self.eng.leaf(P.tpch_orders(ctx, self.sf, tpch.SEED, lo, n, key64=self.key64))
            for k, b in tpch.dimension_tables(ctx, self.sf).items():
                self.t[k] = self.eng.leaf(b)
        t = self.t
        if self.group.world == 1:
            self.plan = {"q1": lambda: tpch.q1_plan(t["lineitem"]), "q6": lambda: tpch.q6_plan(t["lineitem"]),
                         "q3": lambda: tpch.q3_plan(t["customer"], t["orders"], t["lineitem"]),
                         "q5": lambda: tpch.q5_plan(t["customer"], t["orders"], t["lineitem"], t["supplier"], t["nation"], t["region"])}[self.query]()
        ctx.synchronize()

    def unload(self):
        """release the tables and every plan over them (the next workload's tables need the HBM)"""
        self.t.clear()
        self.plan = None
        self.cold, self.spent = [], []
        import gc
        gc.collect()
        self.ctx.synchronize()

    def host_overhead(self, reset=False):
        return None

    def prepare(self, n):
        """n operator trees that have never run (tpch.fresh), built ahead of the timed region: a task's plan is decoded from the
        wire before `plan.execute(partition)` is called (rust/executor/src/flight_service.rs:87-121); what a step times is the
        execution of a cold tree — join builds, path choices and all"""
        self.spent = []
        self.cold = [tpch.fresh(self.plan) for _ in range(n)] if self.group.world == 1 else []

    def step(self):
        t, g = self.t, self.group
        if g.world == 1:
            # the previous step's operator tree goes first, as a task's plan does when the task is done: its join build sides
            # return to the allocator's cache and this step's builds take them from there (kept until the end of the timed
            # region, every step went to hipMalloc for its build sides: 7 calls per Q3 step, 0.2-5 ms depending on the driver's mood)
            self.spent.clear()
            plan = self.cold.pop() if getattr(self, "cold", None) else tpch.fresh(self.plan)
            self.spent.append(plan)
            return plan.collect()
        if self.query in ("q1", "q6"):
            return [q1_distributed(self.eng, g, t["lineitem"], self.query)]
        if self.query == "q3":
            return [q3_distributed(self.eng, g, t["customer"], t["orders"], t["lineitem"], self.join_exchange, self.stats, self.ctx.synchronize)]
        return [q5_distributed(self.eng, g, t["customer"], t["orders"], t["lineitem"], t["supplier"], t["nation"], t["region"],
                               self.join_exchange, self.stats, self.ctx.synchronize)]

    # -- reporting
    def rows_local(self, table):
        return self.local.get(table, 0)

    def describe(self):
        shape = {"q1": "scan+filter+group-by aggregate", "q6": "predicate-heavy scan, selection", "q3": "3-way hash join, group-by, sort",
                 "q5": "6-table hash join, group-by, sort"}[self.query]
        return (f"TPC-H {self.query.upper()} SF{self.sf:g} ({shape}), {self.rows['lineitem']} lineitem rows resident in HBM, Arrow layout, "
                f"{'Int64' if self.key64 else 'Int32'} order keys")

    def partitioning(self):
        w = self.group.world
        if w == 1:
            return "1 GPU, whole tables"
        if self.query in ("q1", "q6"):
            return f"{w} row blocks of the fixed tables, one partial-state all_gather ({self.group.backend})"
        return f"{w} row blocks, order-key join by {self.join_exchange} ({self.group.backend}), partial-state all_gather"

    def algorithmic_bytes(self, key_bytes=4):
        r = self.rows
        if self.query == "q1":
            return r["lineitem"] * tpch.Q1_BYTES_PER_ROW
        if self.query == "q6":
            return r["lineitem"] * tpch.Q6_BYTES_PER_ROW
        if self.query == "q3":
            return tpch.q3_algorithmic_bytes(r["lineitem"], r["orders"], r["customer"], key_bytes)
        return tpch.q5_algorithmic_bytes(r["lineitem"], r["orders"], r["customer"], r["supplier"], key_bytes)

    def algorithmic_bytes_of_kernel(self, kernel, key_bytes=4):
        """bytes ONE launch of `kernel` must read and write at least (DESIGN.md §3 lists them per kernel)"""
        n = self.rows_local("lineitem")
        if kernel.startswith("scan_agg_"):
            return n * (tpch.Q1_BYTES_PER_ROW if self.query == "q1" else tpch.Q6_BYTES_PER_ROW)
        return tpch.KERNEL_BYTES.get(kernel, lambda n, kb: 0)(n, key_bytes)

    def exchange_stats(self, reset=False):
        s = self.stats
        if not s.calls:
            return None
        out = dict(calls=s.calls, seconds=s.seconds, bytes_out_per_rank=s.bytes_out,
                   gbs_per_rank=s.bytes_out / s.seconds / 1e9 if s.seconds else 0.0,
                   gbs_per_link=s.bytes_out / s.seconds / 1e9 / max(1, self.group.world - 1) if s.seconds else 0.0,
                   note="hash partition + all-to-all + concat, wall time on rank 0; one xGMI link per peer")
        if reset:
            s.reset()
        return out

    def result_check(self, result):
        if not result:
            return {}
        n_rows = sum(b.num_rows for b in result)
        out = {"result_rows": n_rows}
        head = result[0]
        if head.num_rows > 64:
            head = self.eng.run(self.eng.P.GlobalLimitExec(self.eng.leaf(head), 8))
        d = head.to_pydict()
        if "count_order" in d:
            out["groups"] = len(d["count_order"])
            out["rows_counted"] = int(sum(d["count_order"]))
        elif "revenue" in d:
            out["revenue"] = d["revenue"][:8]
        return out
