"""Stage outputs on disk: the step right after `plan.execute(partition)` in an executor task and the step right
before it in the next stage (§8(f) rank 1).

Host-side mirror of
  * `utils::write_stream_to_disk`                       rust/core/src/utils.rs:49-84
  * the ExecutePartition arm of `do_get`                rust/executor/src/flight_service.rs:95-150
    (work_dir/<job>/<stage>/<partition>/data.arrow, reply = 1-row batch {path, partition_stats})
  * `PartitionStats` and its Arrow struct form          rust/core/src/serde/scheduler/mod.rs:94-190
  * the FetchPartition arm / ShuffleReaderExec          rust/executor/src/flight_service.rs:193-228,
                                                        rust/core/src/execution_plans/shuffle_reader.rs:77-103
so that a GPU stage writes exactly the file an unmodified CPU executor (or the next GPU stage) reads: an
Arrow IPC *file* with the stream's schema and one message per batch.  The files are written and read by the library
itself (csrc/host/ipc.cpp); this module only names the paths and frames the `{path, partition_stats}` reply (pyarrow)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

from . import plan as P


@dataclass(frozen=True)
class PartitionStats:
    """rust/core/src/serde/scheduler/mod.rs:94-121"""
    num_rows: Optional[int] = None
    num_batches: Optional[int] = None
    num_bytes: Optional[int] = None

    @staticmethod
    def arrow_struct_fields():
        import pyarrow as pa
        return [pa.field("num_rows", pa.uint64(), False), pa.field("num_batches", pa.uint64(), False),
                pa.field("num_bytes", pa.uint64(), False)]

    def to_arrow(self):
        """StructArray of length 1 (to_arrow_arrayref, mod.rs:136-165)"""
        import pyarrow as pa
        fields = self.arrow_struct_fields()
        cols = [pa.array([v], type=pa.uint64()) for v in (self.num_rows, self.num_batches, self.num_bytes)]
        return pa.StructArray.from_arrays(cols, fields=fields)

    @staticmethod
    def from_arrow(struct_array) -> "PartitionStats":
        """from_arrow_struct_array, mod.rs:167-190"""
        d = struct_array[0].as_py()
        return PartitionStats(d["num_rows"], d["num_batches"], d["num_bytes"])


def write_stream_to_disk(stream, path: str) -> PartitionStats:
    """Drain a RecordBatchStream into an Arrow IPC file; count rows, batches and array bytes (utils.rs:49-84).
    The file is written by the LIBRARY's own IPC writer (bhip_stream_write_ipc / bhip_ipc_write_file, csrc/host/ipc.cpp).

    `stream`: a ballista_amd.RecordBatchStream (what ExecutionPlan.execute returns) or a pyarrow.RecordBatchReader."""
    if hasattr(stream, "write_ipc"):
        return PartitionStats(**stream.write_ipc(path))
    return PartitionStats(**P.ipc_write_file(stream, path))


def execute_partition(plan: P.ExecutionPlan, job_id: str, stage_id: int, partition: int, work_dir: str):
    """One executor task (flight_service.rs:95-150): run `plan.execute(partition)`, write
    work_dir/job_id/stage_id/partition/data.arrow, return the 1-row reply batch {path, partition_stats}."""
    import pyarrow as pa
    d = os.path.join(work_dir, job_id, str(stage_id), str(partition))
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "data.arrow")
    stats = write_stream_to_disk(plan.execute(partition), path)
    schema = pa.schema([pa.field("path", pa.string(), False),
                        pa.field("partition_stats", pa.struct(PartitionStats.arrow_struct_fields()), False)])
    return pa.RecordBatch.from_arrays([pa.array([path], type=pa.string()), stats.to_arrow()], schema=schema)


def fetch_partition(path: str) -> List:
    """FetchPartition (flight_service.rs:193-228): the batches of a materialised partition file"""
    import pyarrow as pa
    with pa.memory_map(path, "r") as src:
        return pa.ipc.open_file(src).read_all().to_batches()


def shuffle_reader(ctx: P.Context, paths: List[str], schema=None) -> P.ExecutionPlan:
    """ShuffleReaderExec (shuffle_reader.rs:55-103): one output partition per input partition file, read by the library's own
    IPC reader (bhip_plan_ipc_files) and imported to the device through the Arrow C Data Interface."""
    return P.IpcFileExec(paths, ctx)
