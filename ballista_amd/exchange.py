"""Stage boundary between ranks: the MergeExec / shuffle-read side of the reference's stage 2.

The reference moves the partial-state batches of stage 1 to the task that runs the Final aggregate as
Arrow IPC over Flight (rust/core/src/execution_plans/shuffle_reader.rs:78-103,
rust/core/src/client.rs:139-183).  With one process per GPU the same bytes travel in ONE fixed-size
all_gather (RCCL over xGMI on the GPU box, gloo in the CPU tests); the payload is at most a few KiB
(<= 16 groups x 13 columns for Q1), so the collective is latency-bound and there is nothing to bucket.

Nothing here computes: batches are serialised, exchanged and parsed.
"""
import numpy as np

SLOT_BYTES = 16384          # fixed exchange slot per rank (length prefix + Arrow IPC stream)


def pack_batch(batch, slot_bytes=SLOT_BYTES):
    """pyarrow.RecordBatch -> uint8[slot_bytes]: int64 length, then the Arrow IPC stream bytes."""
    import pyarrow as pa
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    raw = np.frombuffer(sink.getvalue(), dtype=np.uint8)
    if raw.size + 8 > slot_bytes:
        raise ValueError(f"partial-state batch of {raw.size} IPC bytes does not fit the {slot_bytes}-byte exchange slot")
    slot = np.zeros(slot_bytes, dtype=np.uint8)
    slot[:8] = np.frombuffer(np.int64(raw.size).tobytes(), dtype=np.uint8)
    slot[8:8 + raw.size] = raw
    return slot


def unpack_batch(slot):
    """inverse of pack_batch"""
    import pyarrow as pa
    slot = np.ascontiguousarray(slot, dtype=np.uint8)
    n = int(np.frombuffer(slot[:8].tobytes(), dtype=np.int64)[0])
    if n <= 0 or n + 8 > slot.size:
        raise ValueError(f"corrupt exchange slot (length {n})")
    batches = pa.ipc.open_stream(pa.py_buffer(slot[8:8 + n].tobytes())).read_all().to_batches()
    if len(batches) != 1:
        raise ValueError(f"exchange slot holds {len(batches)} batches, expected 1")
    return batches[0]


def _ipc_bytes(batch):
    import pyarrow as pa
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return np.frombuffer(sink.getvalue(), dtype=np.uint8)


def all_to_all_batches(dist, outgoing, device="cpu"):
    """The exchange step of a repartitioned join (RepartitionExec(Hash(keys), N),
    rust/core/src/serde/physical_plan/from_proto.rs:133-147, read back by ShuffleReaderExec): rank r holds
    `outgoing[d]` = the rows of its shard whose key hashes to rank d (ballista_amd.plan.hash_partition) and
    receives, in source-rank order, the batches every rank holds for r.

    Two phases: byte counts by all_gather, then one point-to-point send/recv per peer pair (xGMI is
    point-to-point: 7 peers = 7 links busy at once).  Payload = Arrow IPC bytes of each batch, staged through
    host memory in this round; exchanging the column buffers device-to-device is the next step."""
    import pyarrow as pa
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if len(outgoing) != world:
        raise ValueError(f"need one outgoing batch per rank ({world}), got {len(outgoing)}")
    payload = [_ipc_bytes(b) for b in outgoing]
    sizes = torch.tensor([p.size for p in payload], dtype=torch.int64, device=device)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    incoming = [int(all_sizes[src][rank].item()) for src in range(world)]
    send = [torch.from_numpy(payload[d].copy()).to(device) for d in range(world)]
    recv = [torch.empty(incoming[s], dtype=torch.uint8, device=device) for s in range(world)]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        ops.append(dist.P2POp(dist.isend, send[peer], peer))
        ops.append(dist.P2POp(dist.irecv, recv[peer], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    recv[rank] = send[rank]
    out = []
    for s in range(world):
        raw = recv[s].cpu().numpy().tobytes()
        batches = pa.ipc.open_stream(pa.py_buffer(raw)).read_all().to_batches()
        if len(batches) > 1:
            batches = [pa.Table.from_batches(batches).combine_chunks().to_batches()[0]]
        out.append(batches[0] if batches else pa.RecordBatch.from_pylist([], schema=outgoing[0].schema))
    return out


class _DeviceBytes:
    """a device buffer as torch sees it (the CUDA array interface; PyTorch-ROCm implements it for HIP memory)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def all_to_all_device(dist, outgoing, ctx, device):
    """The same exchange with the column buffers moving device to device (RCCL over xGMI): `outgoing[d]` is the
    device-resident batch for rank d, as `ballista_amd.plan.hash_partition` returns them — fixed-width NULL-free
    columns (the probe / build columns of the TPC-H joins).  One int64 all_gather of the row counts, then ONE
    all_to_all per column with uneven splits straight out of the partition slices into one receive buffer per
    column; the received batch wraps those buffers without a copy.  Source-rank order, input order inside."""
    import torch
    from . import plan as P
    world, rank = dist.get_world_size(), dist.get_rank()
    if len(outgoing) != world:
        raise ValueError(f"need one outgoing batch per rank ({world}), got {len(outgoing)}")
    schema = outgoing[0].schema()
    widths = []
    for name, dtype in schema:
        if dtype not in P.NP_DTYPE:
            raise NotImplementedError(f"all_to_all_device: column {name} is {dtype}; use all_to_all_batches")
        widths.append(np.dtype(P.NP_DTYPE[dtype]).itemsize)
    for b in outgoing:
        if any(b.column_info(i)[4] for i in range(b.num_columns)):
            raise NotImplementedError("all_to_all_device: columns with NULLs; use all_to_all_batches")
    ctx.synchronize()                                   # the partitions are complete before another stream reads them
    counts = torch.tensor([b.num_rows for b in outgoing], dtype=torch.int64, device=device)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    incoming = [int(all_counts[src][rank].item()) for src in range(world)]
    total = sum(incoming)
    recv_bufs = []
    for ci, w in enumerate(widths):
        send = []
        for d in range(world):
            n = outgoing[d].num_rows
            ptr = outgoing[d].column_device(ci)[0]
            send.append(torch.as_tensor(_DeviceBytes(ptr, n * w), device=device) if n else torch.empty(0, dtype=torch.uint8, device=device))
        buf = torch.empty(max(total * w, 1), dtype=torch.uint8, device=device)
        recv = list(buf[:total * w].split([n * w for n in incoming])) if total else [buf[:0] for _ in range(world)]
        dist.all_to_all(recv, send)
        recv_bufs.append(buf)
    torch.cuda.synchronize(device)
    cols = [(name, dtype, buf.data_ptr()) for (name, dtype), buf in zip(schema, recv_bufs)]
    return P.RecordBatch.from_device_pointers(ctx, cols, total, keep=(recv_bufs, outgoing))


def all_gather_batches(dist, batch, device="cpu", slot_bytes=SLOT_BYTES):
    """Every rank contributes one small pyarrow.RecordBatch and receives the batches of all ranks, in
    rank order (the order MergeExec concatenates partitions in, rust/scheduler/src/planner.rs:136-148).

    dist   : an initialised torch.distributed module ("nccl" = RCCL on the GPU box, "gloo" on CPU)
    device : where the exchange buffers live ("cuda:N" for RCCL)."""
    import torch
    world = dist.get_world_size()
    buf = torch.from_numpy(pack_batch(batch, slot_bytes)).to(device)
    # one output tensor, one copy back to the host (a list of per-rank tensors costs a device-to-host copy per rank)
    out = torch.empty(world * buf.numel(), dtype=buf.dtype, device=device)
    dist.all_gather_into_tensor(out, buf)
    host = out.cpu().numpy().reshape(world, buf.numel())
    return [unpack_batch(host[r]) for r in range(world)]
