"""N > 1 path of bench.py on CPU: two gloo ranks, each owning one shard of the seeded lineitem table,
exchange their partial-state batches with ballista_amd.exchange.all_gather_batches (the same call the
RCCL run makes) and merge them.  The scan / aggregate arithmetic here is the ORACLE's (no GPU in this
tier) — what is under test is the product's exchange: framing, rank order, schema fidelity, and that
merge(partials of shards) == aggregate(whole table), the property the weak-scaling bench relies on
(reference: stage 1 Partial -> MergeExec -> Final, rust/scheduler/src/planner.rs:136-171)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SF = 0.002


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_PA = {"Int32": "int32", "Int64": "int64", "UInt64": "uint64", "Float64": "float64", "Utf8": "string", "Date32": "date32",
       "Boolean": "bool_"}


def _to_arrow(batch):
    import pyarrow as pa
    arrays, names = [], []
    for name, c in batch.items():
        vals = c.to_pylist()
        if c.dtype == "Date32":
            arrays.append(pa.array(np.asarray([0 if v is None else v for v in vals], dtype=np.int32), type=pa.int32()).cast(pa.date32()))
        else:
            arrays.append(pa.array(vals, type=getattr(pa, _PA[c.dtype])()))
        names.append(name)
    return pa.RecordBatch.from_arrays(arrays, names=names)


def _from_arrow(rb):
    from collections import OrderedDict
    from oracle.engine import OCol
    inv = {"int32": "Int32", "int64": "Int64", "uint64": "UInt64", "double": "Float64", "string": "Utf8", "bool": "Boolean"}
    out = OrderedDict()
    import pyarrow as pa
    for name, col in zip(rb.schema.names, rb.columns):
        if pa.types.is_date32(col.type):
            out[name] = OCol("Date32", np.asarray(col.cast(pa.int32()).to_pylist(), dtype=np.int32))
            continue
        vals = col.to_pylist()
        dt = inv[str(col.type)]
        valid = None if col.null_count == 0 else np.array([v is not None for v in vals])
        if dt == "Utf8":
            out[name] = OCol(dt, ["" if v is None else v for v in vals], valid)
        else:
            out[name] = OCol(dt, np.array([0 if v is None else v for v in vals]), valid)
    return out


def _rank_main(rank, world, port, query, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from ballista_amd import expr as E, tpch
    from ballista_amd.exchange import all_gather_batches
    from oracle import engine as og, gen
    import helpers
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        li = gen.lineitem(SF)
        n = og.batch_len(li)
        per = (n + world - 1) // world
        shard = helpers.slice_batch(li, rank * per, min(n, (rank + 1) * per))
        parts = tpch.q1_parts(tpch.LINEITEM_SCHEMA) if query == "q1" else tpch.q6_parts(tpch.LINEITEM_SCHEMA)
        partial = og.hash_aggregate(og.filter_batch(shard, parts["predicate"]), "Partial", parts["group"], parts["aggs"])
        gathered = all_gather_batches(dist, _to_arrow(partial), device="cpu")
        assert len(gathered) == world
        # rank order and payload fidelity: my own slot comes back bit-identical
        assert gathered[rank].equals(_to_arrow(partial))
        merged = og.concat_batches([_from_arrow(b) for b in gathered])
        if query == "q1":
            fin = og.hash_aggregate(merged, "Final", parts["group"], tpch.q1_final_aggs())
            fin = og.sort_batch(fin, [E.PhysicalSortExpr(E.col("l_returnflag")), E.PhysicalSortExpr(E.col("l_linestatus"))])
        else:
            fin = og.hash_aggregate(merged, "Final", [], [E.AggregateExpr("SUM", E.col("revenue[sum]"), "revenue")])
        # the same answer as one rank scanning the whole table
        whole = og.hash_aggregate(og.filter_batch(li, parts["predicate"]), "Partial", parts["group"], parts["aggs"])
        if query == "q1":
            ref = og.hash_aggregate(whole, "Final", parts["group"], tpch.q1_final_aggs())
            ref = og.sort_batch(ref, [E.PhysicalSortExpr(E.col("l_returnflag")), E.PhysicalSortExpr(E.col("l_linestatus"))])
        else:
            ref = og.hash_aggregate(whole, "Final", [], [E.AggregateExpr("SUM", E.col("revenue[sum]"), "revenue")])
        helpers.assert_rows_equal(fin, ref, ordered=(query == "q1"), float_rtol=1e-9)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException as e:          # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


@pytest.mark.parametrize("query", ["q1", "q6"])
def test_two_rank_partial_state_exchange_gloo(query):
    pytest.importorskip("pyarrow")
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_main, args=(r, 2, port, query, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in sorted(results):
        assert msg == "ok", f"rank {rank}:\n{msg}"


def _join_rank_main(rank, world, port, q):
    """repartitioned join: both sides sharded by row block, exchanged by hash(join key) % world, joined locally"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from ballista_amd import expr as E
    from ballista_amd.exchange import all_to_all_batches
    from oracle import engine as og, gen
    import helpers
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        od, li = gen.orders(SF), helpers.slice_batch(gen.lineitem(SF), 0, 9000)

        def shard(b):
            n = og.batch_len(b)
            per = (n + world - 1) // world
            return helpers.slice_batch(b, rank * per, min(n, (rank + 1) * per))

        def exchange(b, key):
            parts = og.repartition_hash(b, [E.col(key)], world)          # the row-hash spec of DESIGN.md §6
            got = all_to_all_batches(dist, [_to_arrow(p) for p in parts], device="cpu")
            assert len(got) == world
            return og.concat_batches([_from_arrow(g) for g in got if g.num_rows] or [_from_arrow(got[0])])

        mine_o, mine_l = exchange(shard(od), "o_orderkey"), exchange(shard(li), "l_orderkey")
        # co-location: every key I hold hashes to me
        for b, key in ((mine_o, "o_orderkey"), (mine_l, "l_orderkey")):
            h = og.row_hash([b[key]], og.batch_len(b))
            assert all(int(x) % world == rank for x in h)
        local = og.hash_join(mine_o, mine_l, [("o_orderkey", "l_orderkey")], "Inner")
        whole = og.hash_join(od, li, [("o_orderkey", "l_orderkey")], "Inner")
        hw = og.row_hash([whole["o_orderkey"]], og.batch_len(whole))
        keep = [i for i, x in enumerate(hw) if int(x) % world == rank]
        want = helpers.slice_batch(whole, 0, 0) if not keep else type(whole)((k, c.take(np.array(keep))) for k, c in whole.items())
        helpers.assert_rows_equal(local, want, ordered=False)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException as e:          # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def test_two_rank_repartitioned_join_gloo():
    pytest.importorskip("pyarrow")
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_join_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in sorted(results):
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_exchange_slot_roundtrip_and_overflow():
    pa = pytest.importorskip("pyarrow")
    from ballista_amd.exchange import pack_batch, unpack_batch, SLOT_BYTES
    rb = pa.RecordBatch.from_arrays([pa.array(["A", "N", None]), pa.array([1.5, -0.0, 3.0]), pa.array([1, 2, 3], type=pa.uint64())],
                                    names=["k", "s[sum]", "c[count]"])
    slot = pack_batch(rb)
    assert slot.dtype == np.uint8 and slot.size == SLOT_BYTES
    assert unpack_batch(slot).equals(rb)
    big = pa.RecordBatch.from_arrays([pa.array(np.arange(10000, dtype=np.int64))], names=["x"])
    with pytest.raises(ValueError):
        pack_batch(big)
    with pytest.raises(ValueError):
        unpack_batch(np.zeros(SLOT_BYTES, dtype=np.uint8))
