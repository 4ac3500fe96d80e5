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


def all_gather_batches(dist, batch, device="cpu", slot_bytes=SLOT_BYTES):
    """Every rank contributes one small pyarrow.RecordBatch and receives the batches of all ranks, in
    rank order (the order MergeExec concatenates partitions in, rust/scheduler/src/planner.rs:136-148).

    dist   : an initialised torch.distributed module ("nccl" = RCCL on the GPU box, "gloo" on CPU)
    device : where the exchange buffers live ("cuda:N" for RCCL)."""
    import torch
    buf = torch.from_numpy(pack_batch(batch, slot_bytes)).to(device)
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(out, buf)
    return [unpack_batch(t.cpu().numpy()) for t in out]
