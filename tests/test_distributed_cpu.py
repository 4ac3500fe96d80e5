"""The N > 1 plans of ballista_amd/distributed.py on CPU: two gloo ranks, each owning one row block of the seeded tables, build
EXACTLY the per-rank plan `bench.py --gpus N` builds (distributed.rank_plan: the query with AllGatherExec / ShuffleExchangeExec at
its stage boundaries) and evaluate it with the ORACLE (oracle/plan_eval.py; the exchange nodes call an OracleComm over gloo) —
there is no GPU in this tier.  Under test: row-block sharding, hash routing, source-rank order, the
shuffle-vs-broadcast variants, framing of the batches on the wire, and the property the strong-scaling bench relies on:
    answer(2 ranks) == answer(1 rank)      (reference: stage 1 Partial -> exchange -> Final, rust/scheduler/src/planner.rs:136-171;
                                            RepartitionExec(Hash), rust/core/src/serde/physical_plan/from_proto.rs:133-147)."""
import os
import socket
import sys
from collections import OrderedDict

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SF = 0.002


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_PA = {"Int32": "int32", "Int64": "int64", "UInt64": "uint64", "UInt8": "uint8", "Float64": "float64", "Utf8": "string", "Date32": "date32",
       "Boolean": "bool_"}


def to_arrow(batch):
    import pyarrow as pa
    arrays, names = [], []
    for name, c in batch.items():
        vals = c.to_pylist()
        if c.dtype == "Date32":
            arrays.append(pa.array([None if v is None else int(v) for v in vals], type=pa.int32()).cast(pa.date32()))
        else:
            arrays.append(pa.array(vals, type=getattr(pa, _PA[c.dtype])()))
        names.append(name)
    return pa.RecordBatch.from_arrays(arrays, names=names)


def from_arrow(rb):
    import pyarrow as pa
    from oracle.engine import OCol
    inv = {"int32": "Int32", "int64": "Int64", "uint64": "UInt64", "uint8": "UInt8", "double": "Float64", "string": "Utf8", "bool": "Boolean"}
    out = OrderedDict()
    for name, col in zip(rb.schema.names, rb.columns):
        if pa.types.is_date32(col.type):
            vals = col.cast(pa.int32()).to_pylist()
            dt = "Date32"
        else:
            vals = col.to_pylist()
            dt = inv[str(col.type)]
        valid = None if col.null_count == 0 else np.array([v is not None for v in vals])
        if dt == "Utf8":
            out[name] = OCol(dt, ["" if v is None else v for v in vals], valid)
        else:
            np_t = {"Int32": np.int32, "Date32": np.int32, "Int64": np.int64, "UInt64": np.uint64, "UInt8": np.uint8, "Float64": np.float64,
                    "Boolean": np.bool_}[dt]
            out[name] = OCol(dt, np.array([0 if v is None else v for v in vals], np_t), valid)
    return out


class OracleComm:
    """the exchange handle of the plan nodes (tests/plan_nodes.py AllGatherExec / ShuffleExchangeExec, evaluated by
    oracle/plan_eval.py) over torch.distributed gloo: oracle batches travel as Arrow IPC stream bytes.  `world` / `rank` /
    all_gather(batch) / shuffle(batch, key) — the surface of ballista_amd.plan.Communicator the plan builders rely on."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.calls = 0

    @staticmethod
    def to_wire(batch):
        import pyarrow as pa
        rb = to_arrow(batch)
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, rb.schema) as w:
            w.write_batch(rb)
        return np.frombuffer(sink.getvalue(), dtype=np.uint8)

    @staticmethod
    def from_wire(raw, like):
        import pyarrow as pa
        t = pa.ipc.open_stream(pa.py_buffer(np.ascontiguousarray(raw).tobytes())).read_all()
        b = t.combine_chunks().to_batches()
        return from_arrow(b[0]) if b else OrderedDict((k, c.take(np.zeros(0, np.int64))) for k, c in like.items())

    def all_gather(self, batch):
        """every rank's batch, in rank order (the order MergeExec concatenates partitions in)"""
        import torch
        self.calls += 1
        raw = self.to_wire(batch)
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
        return [batch if r == self.rank else self.from_wire(host[r, :sizes[r]], batch) for r in range(self.world)]

    def all_to_all(self, parts):
        """parts[d] goes to rank d; returns what every rank holds for me, in source-rank order"""
        import torch
        assert len(parts) == self.world
        payload = [self.to_wire(p) for p in parts]
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
        return [parts[s] if s == self.rank else self.from_wire(recv[s].numpy(), parts[s]) for s in range(self.world)]

    def shuffle(self, batch, key):
        """RepartitionExec(Hash([key], world)) + the shuffle read: my rows of every rank's batch, source-rank order"""
        from oracle import engine as og
        from ballista_amd.expr import col
        self.calls += 1
        got = self.all_to_all(og.repartition_hash(batch, [col(key)], self.world))
        live = [b for b in got if og.batch_len(b)]
        return og.concat_batches(live) if live else got[0]


def leaf(batch):
    import plan_nodes as N
    return N.MemoryExec([[batch]])


def _tables(rank, world):
    """replicated small tables, row blocks of orders / lineitem (the sharding of distributed.Workload._block)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from oracle import gen, engine as og

    def block(b):
        n = og.batch_len(b)
        per = (n + world - 1) // world
        lo = min(n, rank * per)
        return helpers.slice_batch(b, lo, min(n, lo + per))

    full = dict(lineitem=gen.lineitem(SF), orders=gen.orders(SF), customer=gen.customer(SF), supplier=gen.supplier(SF), nation=gen.nation(),
                region=gen.region())
    mine = dict(full)
    mine["lineitem"], mine["orders"] = block(full["lineitem"]), block(full["orders"])
    return full, mine


def _rank_main(rank, world, port, query, join_exchange, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import plan_nodes as N
        import helpers
        import torch.distributed as dist
        from oracle import plan_eval, engine as og
        from ballista_amd import tpch, distributed as D
        tpch.P = N                                       # the plan builders over GPU-free plan descriptions
        dist.init_process_group("gloo")
        comm = OracleComm(dist)
        assert (comm.rank, comm.world) == (rank, world)
        full, mine = _tables(rank, world)

        def run(c, t):
            # EXACTLY the plan bench.py's Workload builds for this rank (distributed.rank_plan), evaluated by the oracle
            return plan_eval.collect(D.rank_plan(query, c, {k: leaf(b) for k, b in t.items()}, join_exchange))

        got = run(comm, mine)
        want = run(None, full)                           # the same builders on one rank over the whole tables: no exchange nodes
        key = {"q1": ["l_returnflag", "l_linestatus"], "q6": None, "q3": ["l_orderkey"], "q5": ["n_name"]}[query]
        helpers.assert_rows_equal(got, want, ordered=False, float_rtol=1e-9, key_cols=key)
        assert og.batch_len(got) > 0
        if query in ("q3", "q5"):
            # exchange nodes executed: the partial-state all_gather + (build-side all_gather | two shuffles)
            assert comm.calls == (2 if join_exchange == "broadcast" else 3), comm.calls
            # ORDER BY survives the distribution (every rank sorts the gathered states itself)
            rev = got["revenue"].to_pylist()
            assert all(a >= b for a, b in zip(rev, rev[1:]))
        else:
            assert comm.calls == 1
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException as e:          # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def _spawn(world, *args):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_main, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in sorted(results):
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("query,join_exchange", [("q1", "-"), ("q6", "-"), ("q3", "shuffle"), ("q3", "broadcast"), ("q5", "shuffle"),
                                                 ("q5", "broadcast")])
def test_two_rank_flows_equal_one_rank_gloo(query, join_exchange):
    pytest.importorskip("pyarrow")
    _spawn(2, query, join_exchange)


def _routing_main(rank, world, port, _a, _b, q):
    """OracleComm.all_to_all / all_gather / shuffle: who gets what, in which order, empty parts included"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        from oracle import engine as og
        from oracle.engine import OCol
        from ballista_amd.expr import col
        dist.init_process_group("gloo")
        comm = OracleComm(dist)

        def batch(src, dst, n):
            return OrderedDict([("src", OCol("Int32", np.full(n, src, np.int32))), ("dst", OCol("Int32", np.full(n, dst, np.int32))),
                                ("s", OCol("Utf8", [f"{src}->{dst}#{i}" for i in range(n)], None if n == 0 else np.arange(n) % 3 != 0)),
                                ("x", OCol("Float64", np.arange(n) * 0.5 + src))])

        sizes = lambda s, d: 0 if (s + d) % 3 == 0 else 5 + 7 * s + 3 * d      # some parts are empty
        parts = [batch(rank, d, sizes(rank, d)) for d in range(world)]
        got = comm.all_to_all(parts)
        assert len(got) == world
        for s, b in enumerate(got):                                               # source-rank order, payload intact
            want = batch(s, rank, sizes(s, rank))
            assert og.batch_len(b) == sizes(s, rank)
            assert b["s"].to_pylist() == want["s"].to_pylist() and list(b["x"].values) == list(want["x"].values)
            assert all(v == s for v in b["src"].values) and all(v == rank for v in b["dst"].values)
        gathered = comm.all_gather(batch(rank, -1, 3 + rank))
        assert [og.batch_len(b) for b in gathered] == [3 + r for r in range(world)]
        assert all(all(v == r for v in b["src"].values) for r, b in enumerate(gathered))
        # shuffle = every rank's rows whose key hashes to me, source-rank order, input order inside a source
        mk = lambda r: OrderedDict([("k", OCol("Int64", np.arange(100 * r, 100 * r + 40 + r, dtype=np.int64) * 7919)), ("src", OCol("Int32", np.full(40 + r, r, np.int32)))])
        mine = comm.shuffle(mk(rank), "k")
        want = og.concat_batches([og.repartition_hash(mk(r), [col("k")], world)[rank] for r in range(world)])
        assert list(mine["k"].values) == list(want["k"].values) and list(mine["src"].values) == list(want["src"].values)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException as e:          # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def test_three_rank_routing_gloo():
    pytest.importorskip("pyarrow")
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_routing_main, args=(r, 3, port, None, None, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in sorted(results):
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_bench_self_launches_its_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher spawns torch.distributed.run as a child and relays rank 0's line"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        returncode = 0
        stdout = 'noise\n{"metric": "tpch_q1_sf100_rows_per_sec", "value": 1.0}\n'

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_launch(bench.parse(["--gpus", "4", "--steps", "2"])) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
