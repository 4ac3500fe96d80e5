"""ParquetExec (ballista_amd/csrc/host/parquet.cpp; reference: the ParquetScan leaf of rust/core/src/serde/physical_plan/from_proto.rs:111-121,
`--format parquet` of rust/benchmarks/tpch/src/main.rs:147-150) against pyarrow: files are written HERE with pyarrow.parquet in the variants
a Parquet writer chooses between — Snappy (the reference benchmark's default, main.rs:84-86) / uncompressed, dictionary on / off, data
pages V1 / V2, small pages and row groups, required / optional columns with NULLs — and every value read back on the device must equal
what pyarrow reads.  Nothing of the reference travels; no reference Parquet file exists in its tree."""
import json
import os

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col

import helpers

pa = pytest.importorskip("pyarrow")
pq = pytest.importorskip("pyarrow.parquet")
pytestmark = pytest.mark.gpu


def table(n, seed=0, nulls=True):
    rng = np.random.default_rng(seed)

    def maybe(vals, p=0.2):
        return [None if (nulls and rng.random() < p) else v for v in vals]
    words = ["", "a", "MAIL", "SHIP", "REG AIR", "a longer string with spaces", "é日本", "x" * 40]
    return pa.table({
        "i32": pa.array(maybe(rng.integers(-2 ** 31, 2 ** 31 - 1, n).tolist()), pa.int32()),
        "low": pa.array(maybe(rng.integers(0, 7, n).tolist()), pa.int32()),                   # low cardinality: long RLE runs
        "i64": pa.array(maybe(rng.integers(-2 ** 62, 2 ** 62, n).tolist()), pa.int64()),
        "f64": pa.array(maybe(np.round(rng.random(n) * 1e5, 2).tolist()), pa.float64()),
        "d32": pa.array(maybe(rng.integers(8000, 11000, n).tolist()), pa.int32()).cast(pa.date32()),
        "s": pa.array(maybe([words[k] for k in rng.integers(0, len(words), n)]), pa.string()),
        "u": pa.array(maybe([f"unique-{i}-{'z' * int(k)}" for i, k in enumerate(rng.integers(0, 9, n))]), pa.string()),   # falls back to PLAIN
        "b": pa.array(maybe((rng.random(n) > 0.5).tolist()), pa.bool_()),
        "flag": pa.array(rng.integers(0, 2, n), pa.int64()),                                   # required-looking: no NULLs at all
    })


def read_back(ctx, paths, **kw):
    plan = ba.ParquetExec(paths, ctx, **kw)
    out = []
    for p in range(plan.output_partitioning().partition_count()):
        out.extend(b.to_pyarrow() for b in plan.execute(p))
    return plan, (pa.Table.from_batches(out) if out else None)


def same(got, want):
    assert got.schema.names == want.schema.names
    for name in want.schema.names:
        g, w = got[name].combine_chunks(), want[name].combine_chunks()
        if pa.types.is_floating(w.type):
            assert g.to_pylist() == w.to_pylist(), name          # bit exact: values are copied, never recomputed
        else:
            assert g.cast(w.type).to_pylist() == w.to_pylist(), name


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY"])
@pytest.mark.parametrize("use_dictionary", [True, False])
@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
def test_writer_variants(ctx, tmp_path, compression, use_dictionary, page_version):
    t = table(20_000, seed=5)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, use_dictionary=use_dictionary, data_page_version=page_version, row_group_size=7_000,
                   data_page_size=16 * 1024)
    plan, got = read_back(ctx, [path])
    assert plan.as_any() == "ParquetExec" and [n for n, _, _ in plan.schema()] == t.schema.names
    assert [ty for _, ty, _ in plan.schema()] == ["Int32", "Int32", "Int64", "Float64", "Date32", "Utf8", "Utf8", "Boolean", "Int64"]
    same(got, t)


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY"])
@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
def test_many_pages_per_chunk_without_nulls_go_through_one_launch(ctx, tmp_path, compression, page_version):
    """a NULL-free column chunk of many pages is decoded as ONE set of buffers (parquet.cpp upload_chunk): the pages' run tables are
    re-based into one, every run carries its own bit width — `grow`'s dictionary grows from page to page, so its index width does —
    PLAIN pages (the dictionary of `f64` and `u` overflows) are copied to their rows of the column, a string column waits for the
    host once per chunk"""
    n = 60_000
    t = table(n, seed=21, nulls=False)
    t = t.append_column("grow", pa.array((np.arange(n) // 7).astype(np.int32)))           # distinct values keep appearing: widths 1 .. 14 bits
    t = t.append_column("one", pa.array(np.full(n, 42, np.int64)))                        # a one-entry dictionary: index width 0
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, use_dictionary=True, data_page_version=page_version, row_group_size=25_000, data_page_size=8 * 1024,
                   dictionary_pagesize_limit=64 * 1024)
    md = pq.ParquetFile(path).metadata
    assert md.num_row_groups == 3
    plan, got = read_back(ctx, [path])
    same(got, t)


def test_projection_partitions_and_no_nulls(ctx, tmp_path):
    paths = []
    parts = []
    for i in range(5):
        t = table(3_000 + 17 * i, seed=10 + i, nulls=False)
        p = str(tmp_path / f"part-{i}.parquet")
        pq.write_table(t, p, compression="SNAPPY", row_group_size=1_000)
        paths.append(p)
        parts.append(t)
    plan, got = read_back(ctx, paths, projection=[5, 0, 3], num_partitions=2)
    assert plan.output_partitioning().partition_count() == 2
    assert [n for n, _, _ in plan.schema()] == ["s", "i32", "f64"]
    same(got, pa.concat_tables(parts).select(["s", "i32", "f64"]))
    # a leaf like any other: Filter + aggregate above it
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("s"), "s")], [E.Sum(col("f64"), "sf"), E.Count(E.lit(1, E.UINT8), "n")],
                               ba.FilterExec(E.coerce(col("i32") > E.lit(0, E.INT32), {"i32": E.INT32}), plan))
    fin = ba.HashAggregateExec(ba.plan.FINAL, [(col("s"), "s")], [E.Sum(col("f64"), "sf"), E.Count(E.lit(1, E.UINT8), "n")], ba.MergeExec(agg))
    res = pa.Table.from_batches([b.to_pyarrow() for b in fin.collect()])
    whole = pa.concat_tables(parts)
    ref = whole.filter(pa.compute.greater(whole["i32"], 0)).group_by("s").aggregate([("f64", "sum"), ("f64", "count")])
    want = {k: (v, c) for k, v, c in zip(ref["s"].to_pylist(), ref["f64_sum"].to_pylist(), ref["f64_count"].to_pylist())}
    gotd = {k: (v, c) for k, v, c in zip(res["s"].to_pylist(), res["sf"].to_pylist(), res["n"].to_pylist())}
    assert set(gotd) == set(want)
    for k in want:
        assert gotd[k][1] == want[k][1] and abs(gotd[k][0] - want[k][0]) <= 1e-9 * abs(want[k][0])


def test_q1_over_parquet_lineitem_through_the_wire_plan(ctx, tmp_path, monkeypatch):
    """the benchmark's `--format parquet` shape: a ParquetScanExecNode leaf in the protobuf plan, no resolver — the library reads the
    files itself; Q1 == the committed golden of the same seeded table"""
    import plan_nodes as N
    import proto_encode as pe
    from oracle import gen
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
    a = gen.lineitem_arrays(g["sf"])
    n = len(a["l_quantity"])
    cols = {k: pa.array(a[k]) for k in ("l_orderkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax")}
    for k in ("l_returnflag", "l_linestatus"):
        cols[k] = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(a[k + ".off"]), pa.py_buffer(a[k + ".data"])])
    cols["l_shipdate"] = pa.array(a["l_shipdate"], pa.int32()).cast(pa.date32())
    t = pa.table(cols)
    files = []
    for i, lo in enumerate(range(0, n, 25_000)):
        p = str(tmp_path / f"lineitem-{i}.parquet")
        pq.write_table(t.slice(lo, 25_000), p, compression="SNAPPY", row_group_size=10_000)
        files.append(p)
    li = N.MemoryExec([[gen.lineitem(0.001)]])             # a stand-in leaf with lineitem's schema, replaced by the Parquet scan bytes
    li.name = "mem://x"
    monkeypatch.setattr(tpch, "P", N)
    q1 = tpch.q1_plan(li)
    monkeypatch.undo()
    scan_bytes = pe.parquet_scan(files, list(range(9)), num_partitions=2)
    orig = pe.plan
    monkeypatch.setattr(pe, "plan", lambda p: scan_bytes if p is li else orig(p))
    data = orig(q1)
    monkeypatch.undo()
    plan = ba.ExecutionPlan.from_proto(ctx, data)
    assert "ParquetExec: files=3" in plan.display()
    got = helpers.concat([helpers.from_device(b) for b in plan.collect()])
    rows = g["rows"]
    assert list(zip(got["l_returnflag"].to_pylist(), got["l_linestatus"].to_pylist())) == [(r["l_returnflag"], r["l_linestatus"]) for r in rows]
    assert got["count_order"].to_pylist() == [r["count_order"] for r in rows]
    for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
        assert np.allclose(got[k].to_pylist(), [r[k] for r in rows], rtol=1e-9, atol=0)


@pytest.mark.parametrize("use_dictionary", [True, False])
def test_float32_uint32_and_timestamp_columns(ctx, tmp_path, use_dictionary):
    """FLOAT, INT32/UINT_32, INT64/TIMESTAMP_MILLIS|MICROS pages: the same fixed-width decode as INT32 / INT64 / DOUBLE"""
    rng = np.random.default_rng(5)
    n = 20000
    mask = rng.random(n) < 0.15
    t = pa.table({"f32": pa.array(np.round(rng.normal(0, 100, n), 2).astype(np.float32), mask=mask),
                  "u32": pa.array(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)),
                  "tms": pa.array(rng.integers(0, 2 ** 41, n), pa.int64()).cast(pa.timestamp("ms")),
                  "tus": pa.array(rng.integers(0, 2 ** 51, n), mask=mask, type=pa.int64()).cast(pa.timestamp("us")),
                  "few": pa.array(rng.integers(0, 5, n).astype(np.float32)),
                  "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=mask), "u8": pa.array(rng.integers(0, 256, n).astype(np.uint8)),
                  "i16": pa.array(rng.integers(-2 ** 15, 2 ** 15, n).astype(np.int16)), "u16": pa.array(rng.integers(0, 2 ** 16, n).astype(np.uint16), mask=mask)})
    p = str(tmp_path / "more_types.parquet")
    pq.write_table(t, p, use_dictionary=use_dictionary, row_group_size=7000, data_page_size=4096)
    plan, got = read_back(ctx, [p])
    assert [ty for _, ty, _ in plan.schema()] == ["Float32", "UInt32", "Timestamp(Millisecond)", "Timestamp(Microsecond)", "Float32", "Int8", "UInt8", "Int16", "UInt16"]
    same(got, t)


def test_what_is_outside_the_path_is_refused(ctx, tmp_path):
    t = pa.table({"x": pa.array(np.arange(1000, dtype=np.float16))})
    p = str(tmp_path / "f16.parquet")
    pq.write_table(t, p)
    with pytest.raises(ba.NotImplementedOnGpu, match="type outside the GPU path"):
        ba.ParquetExec([p], ctx)
    t = pa.table({"x": pa.array([[1, 2], [3]], pa.list_(pa.int32()))})
    p = str(tmp_path / "nested.parquet")
    pq.write_table(t, p)
    with pytest.raises(ba.NotImplementedOnGpu, match="nested"):
        ba.ParquetExec([p], ctx)
    t = pa.table({"x": pa.array(np.arange(5000))})
    p = str(tmp_path / "zstd.parquet")
    pq.write_table(t, p, compression="ZSTD")
    plan = ba.ParquetExec([p], ctx)                         # the codec is a property of the column chunk: found when it is read
    with pytest.raises(ba.NotImplementedOnGpu, match="compression codec"):
        plan.collect()
    p = str(tmp_path / "delta.parquet")
    pq.write_table(t, p, use_dictionary=False, column_encoding={"x": "DELTA_BINARY_PACKED"})
    with pytest.raises(ba.NotImplementedOnGpu, match="encoding"):
        ba.ParquetExec([p], ctx).collect()
    bad = tmp_path / "bad.parquet"
    bad.write_bytes(b"PAR1 this is not parquet PAR0")
    with pytest.raises(ba.BallistaError):
        ba.ParquetExec([str(bad)], ctx)
    with pytest.raises(ba.BallistaError, match="cannot open"):
        ba.ParquetExec([str(tmp_path / "missing.parquet")], ctx)
