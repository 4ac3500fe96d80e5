"""The library's own Arrow IPC file writer / reader (ballista_amd/csrc/host/ipc.cpp) against pyarrow, both directions:
what `utils::write_stream_to_disk` writes (rust/core/src/utils.rs:49-84) must be readable by any Arrow reader (the next
stage's executor may be a CPU one), and files written by arrow-rs / pyarrow must be readable here (ShuffleReaderExec's
payload).  The format code is host-only, so this tier needs no GPU; `-m gpu` adds the device round trip."""
import os

import numpy as np
import pytest

import ballista_amd as ba

pa = pytest.importorskip("pyarrow")


def table(n, seed=0):
    rng = np.random.default_rng(seed)
    def maybe(vals, p=0.15):
        return [None if rng.random() < p else v for v in vals]
    return pa.table({
        "i32": pa.array(maybe(rng.integers(-2 ** 31, 2 ** 31 - 1, n).tolist()), pa.int32()),
        "i64": pa.array(rng.integers(-2 ** 62, 2 ** 62, n), pa.int64()),
        "u8": pa.array(maybe(rng.integers(0, 255, n).tolist()), pa.uint8()),
        "u64": pa.array(rng.integers(0, 2 ** 63, n).astype(np.uint64), pa.uint64()),
        "f64": pa.array(maybe(rng.random(n).tolist()), pa.float64()),
        "d32": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "s": pa.array(maybe([("x" * int(k)) + str(i) for i, k in enumerate(rng.integers(0, 12, n))]), pa.string()),
        "b": pa.array(maybe((rng.random(n) > 0.5).tolist()), pa.bool_()),
        "i8": pa.array(rng.integers(-128, 127, n), pa.int8()), "i16": pa.array(rng.integers(-2 ** 15, 2 ** 15, n), pa.int16()),
        "u16": pa.array(rng.integers(0, 2 ** 16, n), pa.uint16()), "u32": pa.array(rng.integers(0, 2 ** 32, n), pa.uint32()),
        "f32": pa.array(rng.random(n).astype(np.float32), pa.float32()),
        "d64": pa.array(rng.integers(0, 10 ** 12, n), pa.int64()).cast(pa.date64()),
        "ts": pa.array(rng.integers(0, 10 ** 15, n), pa.int64()).cast(pa.timestamp("us")),
    })


@pytest.mark.parametrize("sizes", [[0], [1], [7, 0, 1000], [64, 65, 4096]])
def test_files_written_here_are_read_by_pyarrow(tmp_path, sizes):
    batches = [pa.RecordBatch.from_pylist([], schema=table(1).schema) if n == 0 else table(n, seed=i).to_batches()[0] for i, n in enumerate(sizes)]
    schema = batches[0].schema
    path = str(tmp_path / "data.arrow")
    stats = ba.plan.ipc_write_file(pa.RecordBatchReader.from_batches(schema, batches), path)
    assert stats["num_rows"] == sum(sizes) and stats["num_batches"] == len(sizes) and stats["num_bytes"] > 0 or sum(sizes) == 0
    with pa.ipc.open_file(path) as f:
        assert f.schema.equals(schema) and f.num_record_batches == len(sizes)
        for i, b in enumerate(batches):
            assert f.get_batch(i).equals(b)
    # the stream reader inside the file agrees too (schema message + batches + end-of-stream marker)
    raw = open(path, "rb").read()
    assert raw[:6] == b"ARROW1" and raw[-6:] == b"ARROW1"
    again = pa.ipc.open_stream(raw[8:]).read_all()
    assert again.equals(pa.Table.from_batches(batches, schema=schema))


def test_sliced_arrays_are_rebased(tmp_path):
    t = table(500, seed=3)
    sl = t.slice(123, 201)                       # non-zero offsets in every buffer, bit offsets not a multiple of 8
    path = str(tmp_path / "sliced.arrow")
    ba.plan.ipc_write_file(pa.RecordBatchReader.from_batches(sl.schema, sl.to_batches()), path)
    assert pa.ipc.open_file(path).read_all().equals(sl)


@pytest.mark.parametrize("version", ["V5", "V4"])
def test_files_written_by_pyarrow_are_read_here(tmp_path, version):
    t = table(3000, seed=9)
    batches = t.to_batches(max_chunksize=700)
    path = str(tmp_path / "theirs.arrow")
    opts = pa.ipc.IpcWriteOptions(metadata_version=getattr(pa.ipc.MetadataVersion, version))
    with pa.ipc.new_file(path, t.schema, options=opts) as w:
        for b in batches:
            w.write_batch(b)
    got = ba.plan.ipc_open_file(path)
    assert got.schema.equals(t.schema)
    back = got.read_all()
    assert back.equals(t) and back.num_rows == 3000


def test_round_trip_through_both_own_ends(tmp_path):
    t = table(2500, seed=4)
    path = str(tmp_path / "rt.arrow")
    ba.plan.ipc_write_file(pa.RecordBatchReader.from_batches(t.schema, t.to_batches(max_chunksize=999)), path)
    assert ba.plan.ipc_open_file(path).read_all().equals(t)


def test_refusals_are_errors(tmp_path):
    bad = tmp_path / "bad.arrow"
    bad.write_bytes(b"not an arrow file at all")
    with pytest.raises(ba.BallistaError, match="not an Arrow file"):
        ba.plan.ipc_open_file(str(bad))
    with pytest.raises(ba.BallistaError, match="cannot open"):
        ba.plan.ipc_open_file(str(tmp_path / "missing.arrow"))
    # dictionary-encoded and compressed files are outside the path: refused, not misread
    t = pa.table({"s": pa.array(["a", "b", "a"]).dictionary_encode()})
    p = str(tmp_path / "dict.arrow")
    with pa.ipc.new_file(p, t.schema) as w:
        w.write_table(t)
    with pytest.raises(ba.NotImplementedOnGpu):
        ba.plan.ipc_open_file(p)
    t2 = pa.table({"x": pa.array(np.arange(10000))})
    p2 = str(tmp_path / "lz4.arrow")
    try:
        with pa.ipc.new_file(p2, t2.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
            w.write_table(t2)
    except (pa.ArrowNotImplementedError, pa.ArrowInvalid):
        return
    with pytest.raises(OSError, match="compressed"):               # raised while pulling a batch: pyarrow reports the stream's error
        ba.plan.ipc_open_file(p2).read_all()
    # truncated file
    raw = open(p2, "rb").read()
    (tmp_path / "cut.arrow").write_bytes(raw[:len(raw) // 2])
    with pytest.raises((ba.BallistaError, OSError)):
        ba.plan.ipc_open_file(str(tmp_path / "cut.arrow")).read_all()


@pytest.mark.gpu
def test_stage_output_written_from_the_device_and_read_back(ctx, tmp_path):
    """write_stream_to_disk on a GPU stage (Q1 stage 1), the file read by pyarrow, then served to a Final stage through
    bhip_plan_ipc_files — a stage boundary through the library's own files"""
    import helpers
    from ballista_amd import tpch
    from oracle import gen
    li = gen.lineitem(0.01)
    n = len(li["l_quantity"].values)
    parts = [[helpers.slice_batch(li, 0, n // 2)], [helpers.slice_batch(li, n // 2, n)]]
    scan = helpers.memory_exec(ctx, parts)
    stage1 = tpch.q1_stage1(scan)
    paths = []
    for p in range(2):
        path = str(tmp_path / f"part{p}.arrow")
        stats = stage1.execute(p).write_ipc(path)
        assert stats["num_batches"] == 1 and 1 <= stats["num_rows"] <= 4 and stats["num_bytes"] > 0
        t = pa.ipc.open_file(path).read_all()                        # any Arrow reader can take it from here
        assert t.num_rows == stats["num_rows"] and t.schema.names[:2] == ["l_returnflag", "l_linestatus"]
        paths.append(path)
    final = tpch.q1_final(ba.plan.IpcFileExec(paths, ctx))
    got = helpers.concat(helpers.collect_product(final))
    want = helpers.concat(helpers.collect_product(tpch.q1_plan(scan)))
    helpers.assert_rows_equal(got, want, ordered=True, float_rtol=1e-12)
