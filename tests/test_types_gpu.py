"""Every primitive type the serde ships (rust/core/proto/ballista.proto:755-790: Int8/16/32/64, UInt8/16/32/64, Float32/64,
Date32/64, Timestamp(s/ms/us/ns), Boolean, Utf8) through every operator, against the oracle (numpy arithmetic of the same width:
integer arithmetic wraps, Float32 nodes round to float, a cast the target cannot hold is NULL)."""
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E
from ballista_amd.expr import col, lit
from oracle import engine as og, plan_eval
from oracle.engine import OCol
from tests import helpers

pytestmark = pytest.mark.gpu

INTS = ["Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64"]
FLOATS = ["Float32", "Float64"]
TEMPORAL = ["Date32", "Date64", "Timestamp(Second)", "Timestamp(Millisecond)", "Timestamp(Microsecond)", "Timestamp(Nanosecond)"]
NUMERIC = INTS + FLOATS
FIXED = NUMERIC + TEMPORAL


@pytest.fixture(scope="module")
def ctx():
    return ba.Context(0)


def values_of(dtype, n, rng, small=False):
    """small: values whose sums / products stay inside every type's range"""
    if dtype == "Boolean":
        return rng.random(n) > 0.5
    if dtype in FLOATS:
        v = np.round(rng.normal(0, 50, n), 3)
        return v.astype(og.NP_TYPES[dtype])
    if dtype in TEMPORAL:
        per_day = og.TEMPORAL_UNITS[dtype]
        return (rng.integers(8000, 12000, n) * per_day + (rng.integers(0, per_day, n) if per_day > 1 else 0)).astype(og.NP_TYPES[dtype])
    info = np.iinfo(og.NP_TYPES[dtype])
    if small:
        return rng.integers(max(info.min, -10), min(info.max, 10) + 1, n).astype(og.NP_TYPES[dtype])
    lo, hi = max(info.min, -2**62), min(info.max, 2**62)
    # the whole range, with the extremes present
    v = rng.integers(lo, hi, n, dtype=np.int64 if info.min < 0 else np.uint64).astype(og.NP_TYPES[dtype])
    if n >= 4:
        v[0], v[1] = info.min, info.max
    return v


def batch_of(dtypes, n, seed=1, nulls=True, small=False):
    rng = np.random.default_rng(seed)
    b = OrderedDict()
    for i, t in enumerate(dtypes):
        b[f"c{i}"] = OCol(t, values_of(t, n, rng, small), (rng.random(n) > 0.12) if nulls else None)
    return b


def run_both(plan, ordered=True, float_rtol=0.0, key_cols=None):
    got = helpers.concat(helpers.collect_product(plan))
    want = plan_eval.collect(plan)
    helpers.assert_rows_equal(got, want, ordered=ordered, float_rtol=float_rtol, key_cols=key_cols)
    return got


@pytest.mark.parametrize("n", [0, 1, 1000, 4099])
def test_round_trip_every_type(ctx, n):
    b = batch_of(FIXED + ["Boolean"], n, seed=n)
    got = helpers.from_device(helpers.to_device(ctx, b))
    helpers.assert_rows_equal(got, b, ordered=True)


@pytest.mark.parametrize("dtype", NUMERIC)
def test_arithmetic_wraps_and_rounds_per_node(ctx, dtype):
    """a op b for two columns of one type: the integer types wrap at their width (arrow's kernels do), every Float32 node
    rounds to float (bit-exact against numpy float32); chains of two nodes check that the intermediate is narrowed too"""
    n = 3000
    b = batch_of([dtype, dtype, dtype], n, seed=7)
    nz = b["c1"].values == 0
    b["c1"].values[nz] = 1                                              # no division by zero here (tested elsewhere)
    m = helpers.memory_exec(ctx, [[b]])
    exprs = [(col("c0") + col("c1"), "add"), (col("c0") - col("c1"), "sub"), (col("c0") * col("c1"), "mul"),
             ((col("c0") + col("c1")) * col("c2"), "chain"), (col("c0") < col("c1"), "lt"), (col("c0").eq(col("c1")), "eq")]
    if dtype not in ("Int8", "Int16", "Int32", "Int64"):
        exprs.append((col("c0") / col("c1"), "div"))                   # signed MIN / -1 is left out: overflow is unspecified
    if not dtype.startswith("UInt"):
        exprs.append((-col("c0"), "neg"))
    run_both(ba.ProjectionExec(exprs, m), ordered=True)


@pytest.mark.parametrize("dtype", ["Int8", "Int16", "Int32", "Int64"])
def test_signed_division_truncates(ctx, dtype):
    rng = np.random.default_rng(3)
    a = rng.integers(-100, 100, 500).astype(og.NP_TYPES[dtype])
    d = rng.integers(1, 9, 500).astype(og.NP_TYPES[dtype]) * rng.choice([-1, 1], 500).astype(og.NP_TYPES[dtype])
    b = OrderedDict([("a", OCol(dtype, a)), ("d", OCol(dtype, d, rng.random(500) > 0.1))])
    run_both(ba.ProjectionExec([(col("a") / col("d"), "q")], helpers.memory_exec(ctx, [[b]])), ordered=True)


@pytest.mark.parametrize("src", NUMERIC)
def test_cast_matrix(ctx, src):
    """CAST between every pair of numeric types: a value the target cannot hold becomes NULL, int -> Float32 rounds once"""
    n = 2000
    b = batch_of([src], n, seed=11)
    if src in FLOATS:
        b["c0"].values[:6] = np.array([0.5, -0.5, 1e10, -1e10, 255.9, -128.9], og.NP_TYPES[src])
    m = helpers.memory_exec(ctx, [[b]])
    exprs = [(E.CastExpr(col("c0"), t), f"as_{t}") for t in NUMERIC if t != src]
    run_both(ba.ProjectionExec(exprs, m), ordered=True)


def test_temporal_casts(ctx):
    b = batch_of(TEMPORAL, 1500, seed=13)
    m = helpers.memory_exec(ctx, [[b]])
    exprs = []
    for i, s in enumerate(TEMPORAL):
        for t in TEMPORAL:
            if s != t and not (t == "Timestamp(Nanosecond)" and s == "Date32" and False):
                exprs.append((E.CastExpr(col(f"c{i}"), t), f"{i}_as_{t}"))
    for k in range(0, len(exprs), 10):                                    # at most 16 computed columns per projection
        run_both(ba.ProjectionExec(exprs[k:k + 10], m), ordered=True)
    run_both(ba.ProjectionExec([(E.CastExpr(col("c0"), "Int32"), "d_i32"), (E.CastExpr(col("c1"), "Int64"), "d64_i64"),
                                (E.CastExpr(col("c3"), "Int64"), "ts_i64")], m), ordered=True)


@pytest.mark.parametrize("dtype", FIXED)
def test_filter_and_literal_coercion(ctx, dtype):
    """column vs literal of another numeric type: coerced the way DataFusion's planner does (numerical_coercion order)"""
    n = 5000
    b = batch_of([dtype, "Int32"], n, seed=17)
    m = helpers.memory_exec(ctx, [[b]])
    schema = {"c0": dtype, "c1": "Int32"}
    mid = b["c0"].values[n // 2]
    literal = lit(float(mid), "Float64") if dtype in FLOATS else lit(int(mid), "Int64")
    for op in ("Lt", "GtEq", "Eq"):
        pred = E.coerce(E.BinaryExpr(col("c0"), op, literal), schema)
        run_both(ba.FilterExec(pred, m), ordered=True)


@pytest.mark.parametrize("dtype", FIXED)
def test_group_by_key_of_every_type(ctx, dtype):
    n = 6000
    rng = np.random.default_rng(19)
    pool = values_of(dtype, 37, rng)
    b = OrderedDict([("k", OCol(dtype, pool[rng.integers(0, 37, n)], rng.random(n) > 0.1)), ("x", OCol("Float64", rng.random(n))),
                     ("y", OCol("Int32", rng.integers(-5, 5, n)))])
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 2500)], [helpers.slice_batch(b, 2500, n)]])
    aggs = [E.Sum(col("x"), "sx"), E.Count(col("y"), "cy"), E.Min(col("y"), "mn")]
    partial = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("k"), "k")], aggs, m)
    final = ba.HashAggregateExec(ba.plan.FINAL, [(col("k"), "k")], aggs, ba.MergeExec(partial))
    run_both(final, ordered=False, float_rtol=1e-9, key_cols=["k"])


@pytest.mark.parametrize("dtype", FIXED)
def test_aggregates_over_every_type(ctx, dtype):
    """SUM -> Int64 / UInt64 / Float32 / Float64 (sum_return_type), AVG -> Float64, MIN / MAX keep the type"""
    n = 5000
    rng = np.random.default_rng(23)
    small = dtype not in TEMPORAL
    v = values_of(dtype, n, rng, small=small)
    if dtype in FLOATS:
        v = (np.abs(v) + 1).astype(og.NP_TYPES[dtype])                     # no cancellation: the tolerance below is relative
    b = OrderedDict([("g", OCol("Int32", rng.integers(0, 5, n))), ("v", OCol(dtype, v, rng.random(n) > 0.1))])
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 1700)], [helpers.slice_batch(b, 1700, n)]])
    aggs = [E.Min(col("v"), "mn"), E.Max(col("v"), "mx"), E.Count(col("v"), "c")]
    if dtype not in TEMPORAL:
        aggs += [E.Sum(col("v"), "s"), E.Avg(col("v"), "a")]
    partial = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("g"), "g")], aggs, m)
    final = ba.HashAggregateExec(ba.plan.FINAL, [(col("g"), "g")], aggs, ba.MergeExec(partial))
    # SUM(Float32): added in double and rounded once on both sides, but in a different order -> one float ulp
    run_both(final, ordered=False, float_rtol=2e-7 if dtype == "Float32" else 1e-9, key_cols=["g"])


@pytest.mark.parametrize("n", [200, 3000])
@pytest.mark.parametrize("dtype", FIXED)
def test_sort_by_every_type(ctx, dtype, n):
    b = batch_of([dtype, "Int32"], n, seed=29)
    b["c1"] = OCol("Int32", np.arange(n))
    m = helpers.memory_exec(ctx, [[b]])
    for desc in (False, True):
        run_both(ba.SortExec([E.PhysicalSortExpr(col("c0"), descending=desc, nulls_first=not desc)], m), ordered=True)


@pytest.mark.parametrize("dtype", INTS + TEMPORAL)
def test_join_on_key_of_every_integer_type(ctx, dtype):
    rng = np.random.default_rng(31)
    pool = np.unique(values_of(dtype, 300, rng))
    left = OrderedDict([("k", OCol(dtype, pool[rng.integers(0, len(pool), 400)], rng.random(400) > 0.05)), ("x", OCol("Float32", rng.random(400).astype(np.float32)))])
    right = OrderedDict([("rk", OCol(dtype, pool[rng.integers(0, len(pool), 2500)], rng.random(2500) > 0.05)), ("y", OCol("Int16", rng.integers(-9, 9, 2500)))])
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]])
    for jt in (ba.plan.INNER, ba.plan.LEFT, ba.plan.RIGHT):
        run_both(ba.HashJoinExec(lm, rm, [("k", "rk")], jt), ordered=False)


@pytest.mark.parametrize("dtype", FIXED)
def test_hash_repartition_by_every_type(ctx, dtype):
    b = batch_of([dtype, "UInt16", "Float32"], 3000, seed=37)
    m = helpers.memory_exec(ctx, [[b]])
    plan = ba.RepartitionExec(m, ba.Partitioning.Hash([col("c0")], 4))
    total = 0
    for p in range(4):
        got = helpers.concat([helpers.from_device(x) for x in plan.execute(p)])
        want = helpers.concat(plan_eval.execute(plan, p))
        helpers.assert_rows_equal(got, want, ordered=True)
        total += og.batch_len(got)
    assert total == 3000


def test_arrow_and_ipc_edges_carry_every_type(ctx, tmp_path):
    """pyarrow RecordBatchReader -> ArrowStreamExec -> GPU sort -> IPC file written by the library -> pyarrow reads it back"""
    import pyarrow as pa
    rng = np.random.default_rng(41)
    n = 700
    arrow_types = {"Int8": pa.int8(), "Int16": pa.int16(), "Int32": pa.int32(), "Int64": pa.int64(), "UInt8": pa.uint8(), "UInt16": pa.uint16(),
                   "UInt32": pa.uint32(), "UInt64": pa.uint64(), "Float32": pa.float32(), "Float64": pa.float64(), "Date32": pa.date32(),
                   "Date64": pa.date64(), "Timestamp(Second)": pa.timestamp("s"), "Timestamp(Millisecond)": pa.timestamp("ms"),
                   "Timestamp(Microsecond)": pa.timestamp("us"), "Timestamp(Nanosecond)": pa.timestamp("ns")}
    arrays, names = [pa.array(np.arange(n, dtype=np.int32)[::-1].copy())], ["ord"]
    for i, (t, at) in enumerate(arrow_types.items()):
        v = values_of(t, n, rng)
        mask = rng.random(n) < 0.1
        if t == "Date64":
            v = (v // 86400000) * 86400000                                  # pyarrow validates whole days
        arrays.append(pa.array(v, mask=mask).cast(at) if t not in ("Date32",) else pa.array(v.astype(np.int32), mask=mask, type=pa.int32()).cast(at))
        names.append(f"c{i}")
    rb = pa.RecordBatch.from_arrays(arrays, names=names)
    leaf = ba.ArrowStreamExec(pa.RecordBatchReader.from_batches(rb.schema, [rb]), ctx)
    assert [t for _, t, _ in leaf.schema()][1:] == list(arrow_types.keys())
    plan = ba.SortExec([E.PhysicalSortExpr(col("ord"))], leaf)
    path = str(tmp_path / "all_types.arrow")
    stats = plan.execute(0).write_ipc(path)
    assert stats["num_rows"] == n
    back = pa.ipc.open_file(path).read_all()
    assert back.schema.types == rb.schema.types
    want = pa.Table.from_batches([rb]).sort_by("ord")
    assert back.equals(want)


def test_large_utf8_at_the_boundary(ctx, tmp_path):
    """LargeUtf8 (64-bit offsets) comes in through the Arrow C stream, is an ordinary Utf8 column on the device, keeps its schema
    type through Filter / Sort / Projection / GROUP BY / MIN, and leaves as LargeUtf8 again — as an Arrow array and in the IPC file"""
    import pyarrow as pa
    import pyarrow.compute as pc
    rng = np.random.default_rng(43)
    n = 3000
    words = ["alpha", "", "Beta ", "gamma-gamma", "é日本", "z"]
    s = pa.array([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n)], pa.large_string())
    t = pa.table({"s": s, "k": pa.array(rng.integers(0, 9, n).astype(np.int32)), "u": pa.array([f"u{i % 5}" for i in range(n)], pa.string())})
    leaf = ba.ArrowStreamExec(pa.RecordBatchReader.from_batches(t.schema, t.to_batches(max_chunksize=1100)), ctx)
    assert [ty for _, ty, _ in leaf.schema()] == ["LargeUtf8", "Int32", "Utf8"]
    flt = ba.FilterExec(E.coerce(col("k") < lit(5), {"k": "Int32"}), leaf)
    plan = ba.SortExec([E.PhysicalSortExpr(col("s"), nulls_first=False), E.PhysicalSortExpr(col("k")), E.PhysicalSortExpr(col("u"))],
                       ba.ProjectionExec([(col("s"), "s"), (col("k"), "k"), (col("u"), "u"), (E.ScalarFunctionExpr("trim", [col("s")]), "ts")], flt))
    assert [ty for _, ty, _ in plan.schema()] == ["LargeUtf8", "Int32", "Utf8", "LargeUtf8"]
    got = pa.Table.from_batches([b.to_pyarrow() for b in plan.collect()])
    assert got.schema.field("s").type == pa.large_string() and got.schema.field("ts").type == pa.large_string()
    want = t.filter(pc.less(t["k"], 5)).sort_by([("s", "ascending"), ("k", "ascending"), ("u", "ascending")])
    # (pyarrow sorts NULLs last: nulls_first=False above)
    assert got["s"].to_pylist() == want["s"].to_pylist() and got["k"].to_pylist() == want["k"].to_pylist()
    assert got["ts"].to_pylist() == [None if v is None else v.strip() for v in want["s"].to_pylist()]
    # the stage file carries the type, and reads back the same way
    path = str(tmp_path / "large.arrow")
    plan.execute(0).write_ipc(path)
    back = pa.ipc.open_file(path).read_all()
    assert back.schema.field("s").type == pa.large_string() and back.equals(got)
    again = ba.IpcFileExec([path], ctx)
    assert [ty for _, ty, _ in again.schema()][:1] == ["LargeUtf8"]
    assert pa.Table.from_batches([b.to_pyarrow() for b in again.collect()]).equals(got)
    # GROUP BY and MIN keep it too
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("s"), "s")], [E.Min(col("s"), "m"), E.Count(col("k"), "c")], leaf)
    assert [ty for _, ty, _ in agg.schema()][:2] == ["LargeUtf8", "LargeUtf8"]
    out = pa.Table.from_batches([b.to_pyarrow() for b in agg.collect()])
    assert out.schema.field("s").type == pa.large_string() and out.num_rows == len(words) + 1


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 15, 16, 17, 63, 65, 255, 257, 1023, 1025])
def test_narrow_columns_of_odd_lengths_through_the_vm(ctx, n):
    """1- / 2- / 4-byte values and Booleans are read as the aligned 4-byte word that holds them (vm_device.h): lengths that end
    inside such a word, with and without NULLs, through a projection (the VM) and a filter (VM predicate: the Boolean column)"""
    types = ["Int8", "UInt8", "Int16", "UInt16", "Int32", "UInt32", "Float32", "Date32", "Boolean", "Int64"]
    for nulls in (False, True):
        b = batch_of(types, n, seed=100 + n, nulls=nulls, small=True)
        m = helpers.memory_exec(ctx, [[b]])
        schema = {f"c{i}": t for i, t in enumerate(types)}
        exprs = [(E.coerce(col("c0") + col("c2"), schema), "a"), (E.coerce(col("c1") + col("c3"), schema), "b"),
                 (E.coerce(col("c4") + col("c5"), schema), "c"), (E.coerce(col("c6") * lit(2.0, "Float64"), schema), "d"),
                 (col("c7"), "e"), (E.NotExpr(col("c8")), "f"), (col("c9"), "g")]
        run_both(ba.ProjectionExec(exprs, m), ordered=True)
        run_both(ba.FilterExec(col("c8"), m), ordered=True)
