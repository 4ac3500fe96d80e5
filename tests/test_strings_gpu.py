"""Expressions that produce Utf8 values — lower / upper / trim / ltrim / rtrim (rust/core/src/serde/logical_plan/from_proto.rs:910-918),
CASE with string branches, string literals as output columns, octet_length — and MIN / MAX over Utf8, in every operator that can
hold them, against the oracle (Python str methods with Rust's White_Space set)."""
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


@pytest.fixture(scope="module")
def ctx():
    return ba.Context(0)


WORDS = ["  Building ", "BUILDING", "building\t\n", " Auto Mobile ", "", " ", "x", "MiXeD Case 42", "　wide　", "tail  ", "  head",
         "a much Longer String value with  inner  spaces  "]


def fn(name, e):
    return E.ScalarFunctionExpr(name, [e])


def string_batch(n, seed=1):
    rng = np.random.default_rng(seed)
    return OrderedDict([("s", OCol("Utf8", [WORDS[k] for k in rng.integers(0, len(WORDS), n)], rng.random(n) > 0.1)),
                        ("t", OCol("Utf8", [["ab", "AB", " ab", "Ab "][k] for k in rng.integers(0, 4, n)])),
                        ("k", OCol("Int32", rng.integers(0, 6, n))), ("x", OCol("Float64", rng.random(n), rng.random(n) > 0.05))])


def run_both(plan, ordered=True, float_rtol=0.0, key_cols=None):
    got = helpers.concat(helpers.collect_product(plan))
    want = plan_eval.collect(plan)
    helpers.assert_rows_equal(got, want, ordered=ordered, float_rtol=float_rtol, key_cols=key_cols)
    return got


@pytest.mark.parametrize("n", [1, 300, 5000])
def test_trims_and_octet_length(ctx, n):
    """Unicode White_Space at both ends (U+00A0, U+2003, U+3000 next to the ASCII ones), NULLs, empty and all-space strings"""
    m = helpers.memory_exec(ctx, [[string_batch(n, seed=n)]])
    exprs = [(fn("trim", col("s")), "tr"), (fn("ltrim", col("s")), "lt"), (fn("rtrim", col("s")), "rt"), (fn("octet_length", col("s")), "len"),
             (fn("trim", fn("rtrim", col("s"))), "nested"), (col("k"), "k")]
    got = run_both(ba.ProjectionExec(exprs, m), ordered=True)
    assert got["len"].dtype == "Int32"


@pytest.mark.parametrize("n", [300, 5000])
def test_lower_upper_ascii(ctx, n):
    b = string_batch(n, seed=n + 1)
    b["s"] = OCol("Utf8", [s.encode("ascii", "ignore").decode() for s in b["s"].values], b["s"].valid)
    m = helpers.memory_exec(ctx, [[b]])
    run_both(ba.ProjectionExec([(fn("lower", col("s")), "lo"), (fn("upper", col("s")), "up"), (fn("upper", fn("trim", col("t"))), "ut")], m), ordered=True)


def test_lower_of_non_ascii_text_is_declined(ctx):
    """Rust's to_lowercase is Unicode aware ('É' -> 'é', some mappings change the byte length): not on the GPU path"""
    b = OrderedDict([("s", OCol("Utf8", ["abc", "Été"]))])
    with pytest.raises(ba.NotImplementedOnGpu, match="non-ASCII"):
        ba.ProjectionExec([(fn("lower", col("s")), "lo")], helpers.memory_exec(ctx, [[b]])).collect()


def test_case_with_string_branches_and_literal_columns(ctx):
    m = helpers.memory_exec(ctx, [[string_batch(4000, seed=5)]])
    case = E.CaseExpr(None, [(col("k").eq(lit(0, "Int32")), lit("zero")), (col("x") > lit(0.5), fn("trim", col("s"))), (col("k").eq(lit(3, "Int32")), col("t"))],
                      lit("other"))
    no_else = E.CaseExpr(None, [(col("k") < lit(2, "Int32"), col("s"))], None)
    based = E.CaseExpr(col("k"), [(lit(1, "Int32"), lit("one")), (lit(2, "Int32"), lit("two"))], col("t"))
    run_both(ba.ProjectionExec([(case, "c"), (no_else, "n"), (based, "b"), (lit("const"), "lit"), (col("k"), "k")], m), ordered=True)


def test_filter_group_and_sort_on_string_expressions(ctx):
    b = string_batch(6000, seed=9)
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 2500)], [helpers.slice_batch(b, 2500, 6000)]])
    # WHERE upper(trim(t)) = 'AB'
    run_both(ba.FilterExec(fn("upper", fn("trim", col("t"))).eq(lit("AB")), m), ordered=True)
    # GROUP BY trim(t), CASE ... ; SUM(x), COUNT(*)
    bucket = E.CaseExpr(None, [(col("k") < lit(3, "Int32"), lit("low"))], lit("high"))
    aggs = [E.Sum(col("x"), "sx"), E.Count(lit(1, "Int64"), "n")]
    keys = [(fn("trim", col("t")), "tt"), (bucket, "bucket")]
    partial = ba.HashAggregateExec(ba.plan.PARTIAL, keys, aggs, m)
    final = ba.HashAggregateExec(ba.plan.FINAL, [(col("tt"), "tt"), (col("bucket"), "bucket")], aggs, ba.MergeExec(partial))
    run_both(final, ordered=False, float_rtol=1e-9, key_cols=["tt", "bucket"])
    # ORDER BY rtrim(s) DESC, k
    one = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 1500)]])
    run_both(ba.SortExec([E.PhysicalSortExpr(fn("rtrim", col("s")), descending=True), E.PhysicalSortExpr(col("k")), E.PhysicalSortExpr(col("x"))], one), ordered=True)


@pytest.mark.parametrize("grouped", [False, True])
def test_min_max_over_utf8(ctx, grouped):
    b = string_batch(7000, seed=13)
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 3000)], [helpers.slice_batch(b, 3000, 7000)]])
    group = [(col("k"), "k")] if grouped else []
    aggs = [E.Min(col("s"), "mn"), E.Max(col("s"), "mx"), E.Max(fn("trim", col("t")), "mt"), E.Sum(col("x"), "sx"), E.Avg(col("x"), "ax")]
    partial = ba.HashAggregateExec(ba.plan.PARTIAL, group, aggs, m)
    assert [t for _, t, _ in partial.schema()][len(group):len(group) + 3] == ["Utf8"] * 3
    final = ba.HashAggregateExec(ba.plan.FINAL, group, aggs, ba.MergeExec(partial))
    run_both(final, ordered=False, float_rtol=1e-9, key_cols=["k"] if grouped else None)
    # a group without any non-NULL string: MIN is NULL
    e = OrderedDict([("k", OCol("Int32", [1, 1, 2])), ("s", OCol("Utf8", ["b", "a", ""], [True, True, False]))])
    run_both(ba.HashAggregateExec(ba.plan.PARTIAL, [(col("k"), "k")], [E.Min(col("s"), "mn")], helpers.memory_exec(ctx, [[e]])), ordered=False, key_cols=["k"])


def test_string_functions_through_the_wire_plan(ctx):
    """ScalarFunctionNode LOWER / TRIM / OCTETLENGTH decoded by bhip_plan_from_proto run as the ctypes-built plan does"""
    from tests import plan_nodes as N, proto_encode as pe
    b = string_batch(500, seed=21)
    b["s"] = OCol("Utf8", [s.encode("ascii", "ignore").decode() for s in b["s"].values], b["s"].valid)
    m = helpers.memory_exec(ctx, [[b]])
    exprs = [(fn("lower", fn("trim", col("s"))), "lt"), (fn("octet_length", col("t")), "n")]
    stand_in = N.MemoryExec([[b]])
    stand_in.name = "mem://strings"
    data = pe.plan(N.ProjectionExec(exprs, stand_in))
    decoded = ba.ExecutionPlan.from_proto(ctx, data, lambda leaf: m)
    direct = ba.ProjectionExec(exprs, m)
    assert decoded.display() == direct.display()
    got = helpers.concat([helpers.from_device(x) for x in decoded.collect()])
    helpers.assert_rows_equal(got, plan_eval.collect(direct), ordered=True)


@pytest.mark.parametrize("n", [1, 777])
def test_sha2_digests(ctx, tmp_path, n):
    """sha224 / sha256 / sha384 / sha512 (from_proto.rs:924-927) against hashlib: messages of every padding class (0, 55, 56, 63, 64,
    111, 112, 119, 120, 128 bytes and longer), NULLs, non-ASCII bytes; the result is a Binary column at the boundary"""
    import hashlib
    import pyarrow as pa
    rng = np.random.default_rng(n)
    lens = [0, 1, 55, 56, 57, 63, 64, 65, 111, 112, 113, 119, 120, 127, 128, 129, 300]
    vals = [("é" * 200 + "x" * 200)[:lens[i % len(lens)]] if i % 2 else "abc" * (lens[i % len(lens)] // 3 + 1) for i in range(n)]
    vals = [v[:lens[i % len(lens)]] for i, v in enumerate(vals)]
    b = OrderedDict([("s", OCol("Utf8", vals, (rng.random(n) > 0.15) if n > 1 else None)), ("k", OCol("Int32", np.arange(n, dtype=np.int32)))])
    m = helpers.memory_exec(ctx, [[b]])
    fns = ["sha224", "sha256", "sha384", "sha512"]
    plan = ba.ProjectionExec([(fn(f, col("s")), f) for f in fns] + [(col("k"), "k")], m)
    assert [t for _, t, _ in plan.schema()] == ["Binary"] * 4 + ["Int32"]
    got = pa.Table.from_batches([x.to_pyarrow() for x in plan.collect()])
    ok = b["s"].is_valid()
    for f in fns:
        assert got.schema.field(f).type == pa.binary()
        want = [getattr(hashlib, f)(v.encode()).digest() if k else None for v, k in zip(vals, ok)]
        assert got[f].to_pylist() == want, f
    # the digests travel through a stage file as Binary
    path = str(tmp_path / "digests.arrow")
    plan.execute(0).write_ipc(path)
    back = pa.ipc.open_file(path).read_all()
    assert back.schema.field("sha256").type == pa.binary() and back.equals(got)
    # and a filter can compare them (byte order, as Utf8 columns compare)
    flt = ba.FilterExec(E.IsNotNullExpr(col("sha256")), ba.IpcFileExec([path], ctx))
    assert sum(x.num_rows for x in flt.collect()) == int(np.sum(ok))
