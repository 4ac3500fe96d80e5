"""Stage outputs as Arrow IPC files (ballista_amd/shuffle.py): the executor-task wrapper around
`plan.execute(partition)` — rust/executor/src/flight_service.rs:95-150, rust/core/src/utils.rs:49-84,
rust/core/src/serde/scheduler/mod.rs:94-190."""
import json
import os

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from ballista_amd import shuffle


def test_partition_stats_arrow_round_trip():
    s = shuffle.PartitionStats(10, 2, 4096)
    arr = s.to_arrow()
    assert str(arr.type) == "struct<num_rows: uint64 not null, num_batches: uint64 not null, num_bytes: uint64 not null>"
    assert len(arr) == 1
    assert shuffle.PartitionStats.from_arrow(arr) == s


def test_write_stream_to_disk_counts_and_file(tmp_path):
    schema = pa.schema([("k", pa.string()), ("v", pa.float64())])
    batches = [pa.RecordBatch.from_arrays([pa.array(["A", "N"]), pa.array([1.0, 2.5])], schema=schema),
               pa.RecordBatch.from_arrays([pa.array(["R"]), pa.array([-0.0])], schema=schema)]
    reader = pa.RecordBatchReader.from_batches(schema, batches)
    path = str(tmp_path / "data.arrow")
    st = shuffle.write_stream_to_disk(reader, path)
    assert (st.num_rows, st.num_batches) == (3, 2)
    # num_bytes = the buffer bytes of the arrays (values + n+1 offsets + validity when there are NULLs); the reference's
    # get_array_memory_size adds arrow-rs' per-array bookkeeping, pyarrow's nbytes leaves out one offset: bounded, not pinned
    assert sum(c.nbytes for b in batches for c in b.columns) <= st.num_bytes <= 64
    back = shuffle.fetch_partition(path)
    assert [b.num_rows for b in back] == [2, 1]
    assert pa.Table.from_batches(back).equals(pa.Table.from_batches(batches))
    # an empty stream still yields a valid file carrying the schema
    st = shuffle.write_stream_to_disk(pa.RecordBatchReader.from_batches(schema, []), path)
    assert (st.num_rows, st.num_batches, st.num_bytes) == (0, 0, 0)
    assert pa.ipc.open_file(path).schema.equals(schema)


def test_write_stream_to_disk_reports_unwritable_path(tmp_path):
    from ballista_amd import _lib as L
    schema = pa.schema([("k", pa.int32())])
    with pytest.raises(L.ExecutionError, match="Failed to create partition file"):
        shuffle.write_stream_to_disk(pa.RecordBatchReader.from_batches(schema, []), str(tmp_path / "no" / "such" / "dir" / "data.arrow"))


@pytest.mark.gpu
def test_q1_through_stage_files(ctx, tmp_path):
    """stage 1 (2 tasks) -> data.arrow files -> ShuffleReader -> stage 2, equal to the golden vector"""
    import ballista_amd as ba
    from ballista_amd import tpch
    from oracle import gen
    import helpers
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
    li = gen.lineitem(g["sf"])
    n = g["n_rows"]
    parts = [[helpers.slice_batch(li, 0, n // 3)], [helpers.slice_batch(li, n // 3, n)]]
    stage1 = tpch.q1_stage1(helpers.memory_exec(ctx, parts))
    replies = [shuffle.execute_partition(stage1, "job1", 1, p, str(tmp_path)) for p in range(2)]
    paths = []
    for p, r in enumerate(replies):
        assert r.schema.names == ["path", "partition_stats"] and r.num_rows == 1
        path = r.column(0)[0].as_py()
        assert path == os.path.join(str(tmp_path), "job1", "1", str(p), "data.arrow") and os.path.exists(path)
        st = shuffle.PartitionStats.from_arrow(r.column(1))
        assert st.num_rows == 4 and st.num_batches == 1 and st.num_bytes > 0
        paths.append(path)
    final = tpch.q1_final(shuffle.shuffle_reader(ctx, paths))
    got = helpers.concat(helpers.collect_product(final))
    rows = g["rows"]
    assert list(zip(got["l_returnflag"].values, got["l_linestatus"].values)) == [(r["l_returnflag"], r["l_linestatus"]) for r in rows]
    for i, r in enumerate(rows):
        assert int(got["count_order"].values[i]) == r["count_order"]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert abs(got[k].values[i] - r[k]) <= 1e-6 * abs(r[k]), k
