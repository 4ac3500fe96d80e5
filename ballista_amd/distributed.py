"""One process per GPU: how the stages of a query are spread over the ranks of one node and what moves between them.

The reference splits a query into stages at every exchange (rust/scheduler/src/planner.rs:136-171): stage 1 runs per
input partition, its output is pulled by the next stage through `ShuffleReaderExec`
(rust/core/src/execution_plans/shuffle_reader.rs:77-99), and a stage that repartitions by key carries
`RepartitionExec(Hash(exprs, n))` (rust/core/src/serde/physical_plan/from_proto.rs:133-147).  Here a rank is a
partition and the stage boundary is a NODE of the rank's plan — `AllGatherExec` / `ShuffleExchangeExec`
(csrc/host/exchange.cpp) — that moves batches between the ranks' GPUs over xGMI when it is executed.  A rank's whole
distributed query is ONE operator tree, a step is ONE `bhip_plan_collect`:

  * Q1 / Q6 — scan -> filter -> partial aggregate per row block -> AllGatherExec (the partial-state batches, a few hundred
    bytes) -> MergeExec -> Final aggregate -> Sort, on every rank.  No data-path collective.
  * Q3 / Q5 — the order-key join's sides are not co-partitioned.  `shuffle` (BASELINE.json config #5): both sides go
    through ShuffleExchangeExec(Hash([orderkey], N)): split by `row_hash(orderkey) % N` on the device and exchanged (a
    streaming, chunked all-to-all; rows land at their final position); every rank joins and aggregates its own keys; the
    partial states are all_gathered.  `broadcast`: the (filtered, small) build side goes through AllGatherExec instead and
    the probe side never moves — the reference's collect-left join, where every task sees the whole build side
    (from_proto.rs:253-276).

The plan builders below take their operator classes from `tpch.P`, so the SAME definitions run on the HIP library
(ballista_amd.plan) and, under the CPU tests, on the oracle (tests/plan_nodes.py evaluated by oracle/plan_eval.py over two
gloo ranks: tests/test_distributed_cpu.py).

Groups — the control plane of a run (barrier, timing reduce, how the communicator is formed):
  RcclGroup    control over gloo on the CPU, batches over the library's RCCL communicator (bhip_comm_create);
  GlooGroup    batches through host memory over gloo, as the `host` transport of the SAME communicator code
               (bhip_comm_create_host): rehearsal of the N-rank flow on a box with fewer GPUs;
  SingleGroup  one rank.
"""
from __future__ import annotations

import time

from . import tpch


# ---- groups ----------------------------------------------------------------------------------------------------

class SingleGroup:
    rank, world, backend = 0, 1, "single"
    comm = None

    def device_index(self, local_rank):
        return local_rank

    def attach(self, ctx):
        pass

    def barrier(self):
        pass

    def max_over_ranks(self, x):
        return x

    def describe(self):
        return dict(world=1, ranks_seen=[0], transport="none")

    def close(self):
        pass


class GlooGroup:
    """torch.distributed (gloo) for control AND payload.  The payload side is the `host` transport of the library's
    communicator: header matrices, block layout and the streaming shuffle are the same C++ as over RCCL, only the bytes
    travel through host memory.  On a GPU box it rehearses the N-rank flow with ranks sharing the GPU."""
    backend = "gloo"

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.comm = None
        self.ranks_seen = []

    def device_index(self, local_rank):
        import ctypes
        try:                                      # ranks may share a GPU in a rehearsal
            n = ctypes.c_int(0)
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGetDeviceCount(ctypes.byref(n))
            return local_rank % max(1, n.value)
        except OSError:
            return 0

    # -- the two callbacks of bhip_comm_host_transport, over gloo
    def _host_all_gather(self, send, recv):
        import torch
        n = len(send)
        if n == 0:
            return
        out = torch.empty(n * self.world, dtype=torch.uint8)
        self.dist.all_gather_into_tensor(out, torch.frombuffer(bytearray(send), dtype=torch.uint8))
        recv[:] = memoryview(out.numpy()).cast("B")

    def _host_exchange(self, sends, recvs):
        """regions of one peer pair travel as ONE message in list order (both sides know every size)"""
        import torch
        out, inc = {}, {}
        for view, peer in sends:
            out.setdefault(peer, []).append(view)
        for view, peer in recvs:
            inc.setdefault(peer, []).append(view)
        sbuf = {p: torch.frombuffer(bytearray(b"".join(bytes(v) for v in vs)), dtype=torch.uint8) for p, vs in out.items()}
        rbuf = {p: torch.empty(sum(len(v) for v in vs), dtype=torch.uint8) for p, vs in inc.items()}
        ops = []
        for p in sorted(set(sbuf) | set(rbuf)):
            if p in sbuf:
                ops.append(self.dist.P2POp(self.dist.isend, sbuf[p], p))
            if p in rbuf:
                ops.append(self.dist.P2POp(self.dist.irecv, rbuf[p], p))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for p, vs in inc.items():
            raw, at = memoryview(rbuf[p].numpy()).cast("B"), 0
            for v in vs:
                v[:] = raw[at:at + len(v)]
                at += len(v)

    def attach(self, ctx):
        from . import plan as P
        self.comm = P.Communicator.host(ctx, self.world, self.rank, self._host_all_gather, self._host_exchange)
        self._self_test(ctx)

    def _self_test(self, ctx):
        """a one-row all_gather checked against what every rank must see"""
        from . import plan as P
        mine = P.RecordBatch.from_columns(ctx, [("r", "Int32", [self.rank], None)])
        self.ranks_seen = [int(b.column(0)[1][0]) for b in self.comm.all_gather(mine)]
        if self.ranks_seen != list(range(self.world)):
            raise RuntimeError(f"all_gather self-test returned {self.ranks_seen}")

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def describe(self):
        return dict(world=self.world, ranks_seen=self.ranks_seen, transport=self.comm.info()["transport"] if self.comm else "none")

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.dist.destroy_process_group()


class RcclGroup(GlooGroup):
    """Control (barrier, timing reduce, the RCCL unique id) over gloo on the CPU; every batch moves device to device
    through the library's own RCCL communicator (bhip_comm_*, csrc/host/exchange.cpp): the process never initialises a
    second HIP runtime through PyTorch."""
    backend = "nccl"

    def __init__(self, dist, allow_host_exchange=False):
        super().__init__(dist)
        self.allow_host_exchange = allow_host_exchange

    def device_index(self, local_rank):
        return local_rank

    def attach(self, ctx):
        import torch
        from . import plan as P
        uid = torch.zeros(P.Communicator.UNIQUE_ID_BYTES, dtype=torch.uint8)
        err = None
        if self.rank == 0:
            try:
                uid = torch.frombuffer(bytearray(P.Communicator.unique_id()), dtype=torch.uint8).clone()
            except Exception as e:                               # noqa: BLE001 — reported through the agreement below
                err = f"{type(e).__name__}: {e}"
        self.dist.broadcast(uid, src=0)
        # The communicator is created and exercised once (a one-row all_gather checked against what every rank must see).  If any
        # rank cannot do that — no RCCL on the box, a transport the fabric refuses — ALL ranks learn it (over gloo) and the run
        # FAILS: a scaling record must not show N ranks and a plausible value with RCCL never having moved a byte.  Only with
        # allow_host_exchange (bench.py --allow-host-exchange) do the ranks agree to move batches through host memory instead.
        if err is None:
            try:
                self.comm = P.Communicator(ctx, bytes(uid.numpy().tobytes()), self.world, self.rank)
                self._self_test(ctx)
            except Exception as e:                               # noqa: BLE001 — any failure means the same thing here
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
                raise RuntimeError(why + " — refusing to fall back to host-staged gloo (bench.py --allow-host-exchange rehearses that way)")
            self.backend = "gloo, batches through host memory (" + why + ")"
            GlooGroup.attach(self, ctx)


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


# ---- a rank's plan of a distributed query ---------------------------------------------------------------------------------
# `comm`: the exchange handle the nodes are built over (ballista_amd.plan.Communicator; the CPU tests pass their own with the
# same `world` attribute); None or world == 1: no exchange nodes at all.  orders / lineitem are this rank's row blocks, the
# small tables are replicated.  `chunk_rows`: rows per chunk of the streaming shuffle (0 = the library's default).

def _many(comm):
    return comm is not None and comm.world > 1


def q1_rank_plan(comm, lineitem, query="q1"):
    """stage 1 on this rank's rows, all_gather of the partial states, Merge -> Final (-> Sort) on every rank"""
    stage1 = tpch.q1_stage1(lineitem) if query == "q1" else tpch.q6_stage1(lineitem)
    states = tpch.P.AllGatherExec(stage1, comm) if _many(comm) else stage1
    return tpch.q1_final(states) if query == "q1" else tpch.q6_final(states)


def q3_rank_plan(comm, customer, orders, lineitem, join_exchange="shuffle", chunk_rows=0):
    P = tpch.P
    j1, li = tpch.q3_build_side(customer, orders), tpch.q3_probe_side(lineitem)
    if _many(comm):
        if join_exchange == "broadcast":
            j1 = P.AllGatherExec(j1, comm)                   # world partitions: the join drains them all (collect-left)
        else:
            j1 = P.ShuffleExchangeExec(j1, comm, "o_orderkey", chunk_rows)
            li = P.ShuffleExchangeExec(li, comm, "l_orderkey", chunk_rows)
    partial = tpch.q3_partial(j1, li)
    return tpch.q3_final(P.AllGatherExec(partial, comm) if _many(comm) else partial)


def q5_rank_plan(comm, customer, orders, lineitem, supplier, nation, region, join_exchange="shuffle", chunk_rows=0):
    P = tpch.P
    co, li = tpch.q5_build_side(customer, orders, nation, region), tpch.q5_probe_side(lineitem)
    if _many(comm):
        if join_exchange == "broadcast":
            co = P.AllGatherExec(co, comm)
        else:
            co = P.ShuffleExchangeExec(co, comm, "o_orderkey", chunk_rows)
            li = P.ShuffleExchangeExec(li, comm, "l_orderkey", chunk_rows)
    partial = tpch.q5_partial(co, li, supplier)
    return tpch.q5_final(P.AllGatherExec(partial, comm) if _many(comm) else partial)


def rank_plan(query, comm, t, join_exchange="shuffle", chunk_rows=0):
    """t: {table name: leaf plan}"""
    if query in ("q1", "q6"):
        return q1_rank_plan(comm, t["lineitem"], query)
    if query == "q3":
        return q3_rank_plan(comm, t["customer"], t["orders"], t["lineitem"], join_exchange, chunk_rows)
    return q5_rank_plan(comm, t["customer"], t["orders"], t["lineitem"], t["supplier"], t["nation"], t["region"], join_exchange, chunk_rows)


def row_block(n, rank, world):
    """(first row, rows) of rank's block of an n-row table split `world` ways"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per) - lo


# ---- per-rank memory plan of the order-key shuffle (BASELINE.json config #5) ------------------------------------------------

def q5_memory_plan(sf, world, key_bytes=8, chunk_rows=64 << 20, hbm_bytes=288e9):
    """Device bytes ONE rank holds at the peak of Q5's order-key exchange at scale factor `sf` over `world` GPUs, from the row
    counts alone (lineitem l_orderkey + l_suppkey + 2 x f64, orders o_orderkey + o_custkey + o_orderdate).
    `resident` stays for the whole query; `peak_staging` is what the streaming shuffle adds (the received side at its exact
    size + two chunks); `general_path` is what hash_partition + all_to_all + concat would hold instead (N partitions, N packed
    blocks, N received blocks, the concatenation: ~4 more copies)."""
    n = tpch.table_rows(sf)
    li_row, od_row = key_bytes + 4 + 16, key_bytes + 8
    li_rows, od_rows = n["lineitem"] / world, n["orders"] / world
    od_kept = od_rows * 0.152                                    # orders of 1994 that survive customer(ASIA): ~15.2 % x 1/5 ... upper bound: the date filter alone
    resident = dict(lineitem=li_rows * li_row, orders=od_rows * (key_bytes + 12), customer=n["customer"] * 8, supplier=n["supplier"] * 8)
    received = li_rows * li_row + od_kept * (key_bytes + 4 + 16)  # my share of every rank's rows ~= my own row count
    staging = 2 * min(chunk_rows, li_rows) * li_row
    join = od_kept * 2 * (key_bytes + 8) + li_rows * 0.152 * 24   # rank map / table + the matches handed on (upper bounds)
    peak = sum(resident.values()) + received + staging + join
    general = sum(resident.values()) + 5 * li_rows * li_row + join
    return dict(sf=sf, world=world, lineitem_rows_per_rank=int(li_rows), orders_rows_per_rank=int(od_rows),
                resident_bytes=int(sum(resident.values())), received_bytes=int(received), staging_bytes=int(staging),
                join_bytes=int(join), peak_bytes=int(peak), general_path_peak_bytes=int(general), hbm_bytes=int(hbm_bytes),
                fits=bool(peak < 0.92 * hbm_bytes), general_path_fits=bool(general < 0.92 * hbm_bytes),
                exchange_bytes_out_per_rank=int((li_rows * li_row + od_kept * (key_bytes + 4 + 16)) * (world - 1) / world),
                xgmi_floor_ms=(li_rows * li_row + od_kept * (key_bytes + 4 + 16)) / world / 153e9 * 1e3)


# ---- bench.py's workloads ------------------------------------------------------------------------------------------

class Workload:
    """tables in HBM + one step of a query, for bench.py.  mode "strong": the fixed tables split N ways by row
    block; "weak" (Q1 / Q6): every rank its own full-size block."""

    def __init__(self, query, ctx, group, sf, rows, key64=False, join_exchange="shuffle", chunk_rows=0):
        from . import plan as P
        self.P = P
        self.query, self.ctx, self.group, self.sf, self.rows, self.key64 = query, ctx, group, sf, dict(rows), key64
        self.join_exchange, self.chunk_rows = join_exchange, chunk_rows
        self.local = {}
        self.t = {}
        self.plan = None
        self.cold, self.spent = [], []
        self.step_seconds = []

    def _block(self, table, mode):
        n = self.rows[table]
        if mode == "weak" or self.group.world == 1:
            return self.group.rank * n, n
        return row_block(n, self.group.rank, self.group.world)

    def _leaf(self, batch):
        return self.P.MemoryExec([[batch]], self.ctx)

    def load(self, mode):
        P, ctx = self.P, self.ctx
        self.unload()                          # release the previous tables first
        # only the columns the query reads: an SF1000 rank block of all nine lineitem columns would be twice Q5's four
        li_cols = {"q1": ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"],
                   "q6": ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"],
                   "q3": ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"],
                   "q5": ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]}[self.query]
        lo, n = self._block("lineitem", mode)
        self.local["lineitem"] = n
        self.t["lineitem"] = self._leaf(P.tpch_lineitem(ctx, self.sf, tpch.SEED, lo, n, key64=self.key64, columns=li_cols))
        if self.query in ("q3", "q5"):
            lo, n = self._block("orders", mode)
            self.local["orders"] = n
            od_cols = ["o_orderkey", "o_custkey", "o_orderdate"] + (["o_shippriority"] if self.query == "q3" else [])
            self.t["orders"] = self._leaf(P.tpch_orders(ctx, self.sf, tpch.SEED, lo, n, key64=self.key64, columns=od_cols))
            for k, b in tpch.dimension_tables(ctx, self.sf, self.query).items():
                self.t[k] = self._leaf(b)
        self.plan = rank_plan(self.query, self.group.comm, self.t, self.join_exchange, self.chunk_rows)
        ctx.synchronize()

    def unload(self):
        """release the tables and every plan over them (the next workload's tables need the HBM)"""
        self.t.clear()
        self.plan = None
        self.cold, self.spent = [], []
        import gc
        gc.collect()
        self.ctx.synchronize()

    def prepare(self, n):
        """n operator trees that have never run (tpch.fresh), built ahead of the timed region: a task's plan is decoded from the
        wire before `plan.execute(partition)` is called (rust/executor/src/flight_service.rs:87-121); what a step times is the
        execution of a cold tree — join builds, path choices, exchanges and all.  The same for every world size: a rank's
        distributed query is one tree (exchange nodes included), so no plan is built inside the timed loop."""
        self.spent = []
        self.cold = [tpch.fresh(self.plan) for _ in range(n)]
        self.step_seconds = []

    def step(self):
        # the previous step's operator tree goes first, as a task's plan does when the task is done: its join build sides
        # return to the allocator's cache and this step's builds take them from there (kept until the end of the timed
        # region, every step went to hipMalloc for its build sides: 7 calls per Q3 step, 0.2-5 ms depending on the driver's mood)
        self.spent.clear()
        plan = self.cold.pop() if self.cold else tpch.fresh(self.plan)
        self.spent.append(plan)
        t0 = time.perf_counter()
        out = plan.collect()                                      # ONE C call: bhip_plan_collect
        self.step_seconds.append(time.perf_counter() - t0)
        return out

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
        comm = self.group.comm
        if comm is None:
            return None
        s = comm.stats(reset)
        if not s["calls"]:
            return None
        sec, out = s["seconds"], s["bytes_out"]
        return dict(calls=s["calls"], seconds=sec, bytes_out_per_rank=out, gbs_per_rank=out / sec / 1e9 if sec else 0.0,
                    gbs_per_link=out / sec / 1e9 / max(1, self.group.world - 1) if sec else 0.0, transport=comm.info()["transport"],
                    note="collective calls of this rank since the last reset (all_gather, shuffle: count pass + scatter + grouped "
                         "send / receive), host wall time inside the library; one xGMI link per peer")

    def host_overhead(self, reset=False):
        """per-step wall time of the ONE C call a step is, on this rank (bench.py subtracts nothing from it: kernels and
        transport are inside); the Python side of a step is what the step loop adds on top"""
        if not self.step_seconds:
            return None
        s = sorted(self.step_seconds)
        out = dict(collect_ms_median=s[len(s) // 2] * 1e3, collect_ms_min=s[0] * 1e3, steps=len(s))
        if reset:
            self.step_seconds = []
        return out

    def result_check(self, result):
        if not result:
            return {}
        n_rows = sum(b.num_rows for b in result)
        out = {"result_rows": n_rows}
        head = result[0]
        if head.num_rows > 64:
            head = self.P.GlobalLimitExec(self._leaf(head), 8).collect()[0]
        d = head.to_pydict()
        if "count_order" in d:
            out["groups"] = len(d["count_order"])
            out["rows_counted"] = int(sum(d["count_order"]))
        elif "revenue" in d:
            out["revenue"] = d["revenue"][:8]
        return out
