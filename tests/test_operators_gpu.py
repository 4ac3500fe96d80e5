"""GPU parity of every operator of SURVEY.md §8(a) rows a4-a12 against the CPU oracle, on seeded
random batches with NULLs, ragged sizes, empty inputs and the reference's .tbl fixtures.
Bit-exact for integers / strings / per-row Float64 expression values (separately rounded ops);
row order: preserved where the reference preserves it (Filter, Projection, Limit), else compared
as multisets (Join, Repartition) or by sort keys (Sort)."""
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit, coerce
from oracle import engine as og, gen, plan_eval
from oracle.engine import OCol

import helpers

pytestmark = pytest.mark.gpu


def random_batch(n, seed=1, nulls=True, long_strings=True):
    rng = np.random.default_rng(seed)
    def valid(p=0.15):
        return (rng.random(n) > p) if nulls else None
    words = ["AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD", "", "x", "a much longer string value"]
    if not long_strings:
        words = words[:-1]
    from collections import OrderedDict
    return OrderedDict([
        ("i32", OCol("Int32", rng.integers(-50, 50, n), valid())),
        ("i64", OCol("Int64", rng.integers(-10**12, 10**12, n), valid())),
        ("u8", OCol("UInt8", rng.integers(0, 256, n), None)),
        ("u64", OCol("UInt64", rng.integers(0, 2**63, n, dtype=np.uint64) * 2, valid())),
        ("f", OCol("Float64", np.round(rng.normal(0, 100, n), 2), valid())),
        ("g", OCol("Float64", rng.random(n), None)),
        ("d", OCol("Date32", rng.integers(8000, 11000, n), valid())),
        ("b", OCol("Boolean", rng.random(n) > 0.5, valid())),
        ("s", OCol("Utf8", [words[k] for k in rng.integers(0, len(words), n)], valid())),
        ("k", OCol("Int32", rng.integers(0, 7, n), None)),
    ])


SCHEMA = {"i32": E.INT32, "i64": E.INT64, "u8": E.UINT8, "u64": E.UINT64, "f": E.FLOAT64, "g": E.FLOAT64,
          "d": E.DATE32, "b": E.BOOLEAN, "s": E.UTF8, "k": E.INT32}


def run_both(plan, ordered=True, float_rtol=0.0, key_cols=None):
    got = helpers.concat(helpers.collect_product(plan))
    want = plan_eval.collect(plan)
    helpers.assert_rows_equal(got, want, ordered=ordered, float_rtol=float_rtol, key_cols=key_cols)
    return got


PREDICATES = [
    col("i32") > lit(0, E.INT32),
    (col("f") <= lit(12.5)).and_(col("d") >= E.date32("1995-01-01")),
    (col("f") < lit(0.0)).or_(col("b")),
    E.NotExpr(col("b")),
    E.IsNullExpr(col("f")),
    E.IsNotNullExpr(col("s")),
    col("s").eq(lit("BUILDING")),
    col("s").ne(lit("BUILDING")),
    col("s") < lit("FURNITURE"),
    E.BinaryExpr(col("s"), "Like", lit("%UI%")),
    E.BinaryExpr(col("s"), "Like", lit("MACH%")),
    E.BinaryExpr(col("s"), "NotLike", lit("%E")),
    E.InListExpr(col("k"), [lit(1, E.INT32), lit(3, E.INT32), lit(5, E.INT32)]),
    E.InListExpr(col("i32"), [lit(1, E.INT32), lit(2, E.INT32)], negated=True),
    E.NotExpr((col("i32") > lit(0, E.INT32)).and_(E.InListExpr(col("k"), [lit(1, E.INT32), lit(2, E.INT32)]))),
    coerce(col("i64") * lit(2) > col("i32"), SCHEMA),
    col("u64") > lit(2**63 + 5, E.UINT64),
]


@pytest.mark.parametrize("pi", range(len(PREDICATES)))
def test_filter_predicates(ctx, pi):
    b = random_batch(3000, seed=pi)
    plan = ba.FilterExec(PREDICATES[pi], helpers.memory_exec(ctx, [[b]]))
    run_both(plan)


@pytest.mark.parametrize("n", [0, 1, 64, 1023, 1024, 1025, 5000])
def test_filter_sizes_and_empty(ctx, n):
    b = random_batch(max(n, 1), seed=n)
    if n == 0:
        b = helpers.slice_batch(b, 0, 0)
    plan = ba.FilterExec(col("g") < lit(0.3), helpers.memory_exec(ctx, [[b]]))
    run_both(plan)


def test_filter_all_and_none(ctx):
    b = random_batch(2000, seed=3, nulls=False)
    run_both(ba.FilterExec(col("g") < lit(2.0), helpers.memory_exec(ctx, [[b]])))
    got = helpers.collect_product(ba.FilterExec(col("g") < lit(-1.0), helpers.memory_exec(ctx, [[b]])))
    assert sum(len(next(iter(x.values()))) for x in got) == 0


PROJECTIONS = [
    [(col("f") * col("g"), "p"), (col("s"), "s"), (col("i32"), "i32")],
    [(coerce(col("f") * (lit(1) - col("g")) * (lit(1) + col("g")), SCHEMA), "charge")],
    [(col("i32") + col("k"), "a"), (col("i32") - col("k"), "b"), (col("i32") * col("k"), "c")],
    [(E.CastExpr(col("i32"), E.FLOAT64), "cf"), (E.CastExpr(col("f"), E.INT64), "ci"), (E.CastExpr(col("i64"), E.INT32), "cn"),
     (E.CastExpr(col("u8"), E.INT64), "cu"), (E.CastExpr(col("b"), E.INT32), "cb"), (E.CastExpr(col("d"), E.INT32), "cd")],
    [(E.NegativeExpr(col("f")), "nf"), (E.NegativeExpr(col("i64")), "ni")],
    [(E.CaseExpr(None, [(col("g") < lit(0.3), col("f")), (col("g") < lit(0.6), col("g"))], lit(-1.0)), "c1"),
     (E.CaseExpr(None, [(col("b"), col("i32"))], None), "c2"),
     (E.CaseExpr(col("k"), [(lit(1, E.INT32), lit(10.0)), (lit(2, E.INT32), lit(20.0))], lit(0.0)), "c3")],
    [(E.ScalarFunctionExpr("sqrt", [col("g")]), "sq"), (E.ScalarFunctionExpr("abs", [col("f")]), "ab"),
     (E.ScalarFunctionExpr("floor", [col("f")]), "fl"), (E.ScalarFunctionExpr("ceil", [col("f")]), "ce"),
     (E.ScalarFunctionExpr("round", [col("f")]), "ro"), (E.ScalarFunctionExpr("trunc", [col("f")]), "tr"),
     (E.ScalarFunctionExpr("signum", [col("f")]), "sg")],
    [(col("f") / col("g"), "q"), (col("b").and_(col("g") > lit(0.5)), "bb"), (E.IsNullExpr(col("b")), "nb"),
     (col("g") > lit(0.5), "cmp")],
    [(col("f") / lit(0.0), "inf"), (lit(7.5) + lit(1.0), "konst"), (E.Literal(None, E.FLOAT64), "nul")],
]


@pytest.mark.parametrize("pi", range(len(PROJECTIONS)))
def test_projection_expressions_bit_exact(ctx, pi):
    b = random_batch(2500, seed=100 + pi)
    plan = ba.ProjectionExec(PROJECTIONS[pi], helpers.memory_exec(ctx, [[b]]))
    run_both(plan)     # float_rtol=0: per-row values are bit-identical (no FMA contraction)


def test_transcendental_functions_close(ctx):
    b = random_batch(2000, seed=9, nulls=False)
    fns = ["exp", "ln", "log2", "log10", "sin", "cos", "tan", "asin", "acos", "atan"]
    plan = ba.ProjectionExec([(E.ScalarFunctionExpr(f, [col("g")]), f) for f in fns], helpers.memory_exec(ctx, [[b]]))
    run_both(plan, float_rtol=1e-14)   # libm implementations differ in the last ulp


def test_integer_divide(ctx):
    b = random_batch(1500, seed=11)
    b["k1"] = OCol("Int32", b["k"].values + 1)
    plan = ba.ProjectionExec([(col("i32") / col("k1"), "q")], helpers.memory_exec(ctx, [[b]]))
    run_both(plan)
    bad = ba.ProjectionExec([(col("i32") / col("k"), "q")], helpers.memory_exec(ctx, [[b]]))
    with pytest.raises(ba.ExecutionError, match="Divide by zero"):
        bad.collect()
    # rows the fused predicate rejects are never evaluated by the reference either
    guarded = ba.HashAggregateExec(ba.plan.PARTIAL, [], [E.Sum(col("i32") / col("k"), "s")],
                                   ba.FilterExec(col("k") > lit(0, E.INT32), helpers.memory_exec(ctx, [[b]])))
    run_both(guarded)


def test_plan_errors(ctx):
    b = random_batch(10)
    m = helpers.memory_exec(ctx, [[b]])
    with pytest.raises(ba.PlanError, match="No field named"):
        ba.FilterExec(col("nope") > lit(1), m)
    with pytest.raises(ba.PlanError, match="boolean"):
        ba.FilterExec(col("f"), m)
    with pytest.raises(ba.PlanError, match="Cannot evaluate binary expression"):
        ba.ProjectionExec([(col("f") + col("i32"), "x")], m)
    with pytest.raises(ba.PlanError, match="invalid partition"):
        ba.FilterExec(col("b"), m).execute(3)
    with pytest.raises(ba.NotImplementedOnGpu):
        ba.FilterExec(E.BinaryExpr(col("s"), "Like", lit("a_c" + "x" * 300)), m)      # pattern longer than 255 bytes


def test_trait_methods(ctx):
    b = random_batch(100)
    m = helpers.memory_exec(ctx, [[b], [b]])
    f = ba.FilterExec(col("b"), m)
    assert f.as_any() == "FilterExec"
    assert [n for n, _, _ in f.schema()] == list(b.keys())
    assert f.output_partitioning().partition_count() == 2
    assert f.children() == [m]
    m2 = helpers.memory_exec(ctx, [[b]])
    f2 = f.with_new_children([m2])
    assert f2.output_partitioning().partition_count() == 1 and f2.as_any() == "FilterExec"
    assert ba.MergeExec(f).output_partitioning().partition_count() == 1
    assert "FilterExec" in f.display() and "MemoryExec" in f.display()
    stats = f.execute(0).drain()
    assert stats["num_batches"] == 1 and stats["num_rows"] == int(np.sum(b["b"].values & b["b"].is_valid()))
    assert stats["num_bytes"] > 0


def test_limit_coalesce_merge(ctx):
    bs = [random_batch(700, seed=s) for s in range(4)]
    m = helpers.memory_exec(ctx, [[bs[0], bs[1]], [bs[2], bs[3]]])
    run_both(ba.GlobalLimitExec(ba.MergeExec(m), 1000))
    run_both(ba.GlobalLimitExec(ba.MergeExec(m), 10 ** 6))
    run_both(ba.GlobalLimitExec(ba.MergeExec(m), 0))
    run_both(ba.LocalLimitExec(m, 701))
    co = ba.CoalesceBatchesExec(ba.FilterExec(col("g") < lit(0.1), m), 100)
    run_both(co)
    assert len(list(co.execute(0))) == 1          # two small filtered batches were concatenated


GROUPINGS = [
    ([("k", "k")], 7),
    ([("s", "s")], 9),
    ([("k", "k"), ("b", "b")], 21),
    ([("d", "d")], 3000),
    ([("i32", "i32"), ("k", "k"), ("u8", "u8")], 5000),
]


@pytest.mark.parametrize("gi", range(len(GROUPINGS)))
def test_hash_aggregate_group_by(ctx, gi):
    names, _ = GROUPINGS[gi]
    bs = [random_batch(4000, seed=40 + gi, long_strings=False), random_batch(1500, seed=50 + gi, long_strings=False)]
    m = helpers.memory_exec(ctx, [[bs[0]], [bs[1]]])
    group = [(col(a), n) for a, n in names]
    aggs = [E.Sum(col("f"), "sf"), E.Avg(col("f"), "af"), E.Count(col("f"), "cf"), E.Count(lit(1, E.UINT8), "n"),
            E.Sum(col("i32"), "si"), E.Avg(col("i32"), "ai"), E.Min(col("f"), "mn"), E.Max(col("i64"), "mx"),
            E.Sum(col("u8"), "su")]
    part = ba.HashAggregateExec(ba.plan.PARTIAL, group, aggs, m)
    run_both(part, ordered=False, float_rtol=1e-9, key_cols=[n for _, n in names])
    fin = ba.HashAggregateExec(ba.plan.FINAL, group, aggs, ba.MergeExec(part))
    run_both(fin, ordered=False, float_rtol=1e-9, key_cols=[n for _, n in names])


@pytest.mark.parametrize("shape", ["clustered", "clustered_two_keys", "comes_back", "descending", "with_predicate", "clustered_1.3M", "comes_back_1.3M"])
def test_hash_aggregate_over_input_clustered_by_group(ctx, shape):
    """many groups whose rows are consecutive (lineitem joined to orders, grouped by the order key): every run of equal keys is a
    group and no table is needed — IF no key comes back later, which the same pass checks (every key change an increase of the
    packed image).  'comes_back' / 'descending' fail that check and take the table; a fused predicate does too.  Several input
    batches, NULLs in the summed column, runs that straddle 1024-row tiles and batch boundaries."""
    from collections import OrderedDict
    rng = np.random.default_rng(7)
    # (beyond 2^20 rows the run structure is decided on the leading rows and confirmed — run count, "first key part ascending" — by
    # a second read after the full pass: ops_agg.cpp hash_aggregate)
    n = 1_300_000 if shape.endswith("1.3M") else 40_000
    shape = shape.split("_1.3M")[0]
    sizes = rng.integers(1, 9, n)
    sizes[100] = 5000                                               # one run longer than several tiles
    key = np.repeat(np.arange(len(sizes), dtype=np.int64) * 3 + 10, sizes)[:n]
    if shape == "comes_back":
        key[n - 50:] = key[5]                                       # an early key again at the very end
    if shape == "descending":
        key = key[::-1].copy()
    b = OrderedDict([("k", OCol("Int64", key)), ("k2", OCol("Int32", (key % 7).astype(np.int32))),
                     # (the large shapes sum positive values: among 300 K groups of signed ones some sum cancels to ~1e-3 and a different
                     # but equally valid order of addition then differs by more than any relative gate)
                     ("x", OCol("Float64", np.round(np.abs(rng.normal(0, 10, n)) if n > 100_000 else rng.normal(0, 10, n), 3), rng.random(n) > 0.1)),
                     ("q", OCol("Int32", rng.integers(0, 50, n)))])
    parts = [[helpers.slice_batch(b, 0, 13_000)], [helpers.slice_batch(b, 13_000, n)]]
    m = ba.MergeExec(helpers.memory_exec(ctx, parts))               # ONE partition, two batches: runs cross the batch boundary
    src = ba.FilterExec(col("q") < lit(40, E.INT32), m) if shape == "with_predicate" else m
    group = [(col("k"), "k")] + ([(col("k2"), "k2")] if shape == "clustered_two_keys" else [])
    aggs = [E.Sum(col("x"), "sx"), E.Count(col("x"), "cx"), E.Count(lit(1, E.UINT8), "n"), E.Min(col("q"), "mn"), E.Avg(col("q"), "aq")]
    plan = ba.HashAggregateExec(ba.plan.PARTIAL, group, aggs, src)
    for _ in range(2):                                              # the second run uses what the first one learned about the input
        got = run_both(plan, ordered=False, float_rtol=1e-12, key_cols=[nm for _, nm in group])
    assert len(got["k"].values) > 5000


def test_aggregate_over_projection_and_filter_fuses(ctx):
    b = random_batch(5000, seed=77)
    m = helpers.memory_exec(ctx, [[b]])
    proj = ba.ProjectionExec([(col("f") * col("g"), "fg"), (col("k"), "kk"), (col("g"), "g")], m)
    flt = ba.FilterExec(col("g") > lit(0.25), proj)
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("kk"), "kk")], [E.Sum(col("fg"), "s"), E.Count(col("fg"), "c")],
                               ba.CoalesceBatchesExec(flt, 4096))
    run_both(agg, ordered=False, float_rtol=1e-9, key_cols=["kk"])


def test_group_key_longer_than_the_packed_key(ctx):
    """ "a much longer string value" does not fit the 16-byte packed key: the run falls over to the wide-key path
    (representative rows, ops_agg_wide.cpp) and the operator remembers it"""
    b = random_batch(3000, seed=5)
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 1000)], [helpers.slice_batch(b, 1000, 3000)]])
    aggs = [E.Count(lit(1, E.UINT8), "n"), E.Sum(col("f"), "sf"), E.Avg(col("i32"), "ai"), E.Min(col("d"), "md")]
    part = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("s"), "s")], aggs, m)
    for _ in range(2):
        run_both(part, ordered=False, float_rtol=1e-9, key_cols=["s"])
    fin = ba.HashAggregateExec(ba.plan.FINAL, [(col("s"), "s")], aggs, ba.MergeExec(part))
    got = run_both(fin, ordered=False, float_rtol=1e-9, key_cols=["s"])
    assert len(got["s"].values) == 9                       # 8 words + NULL


WIDE_GROUPINGS = [
    [("s", "s"), ("k", "k"), ("i64", "i64"), ("d", "d")],               # Utf8 + 16 fixed bytes
    [("i64", "i64"), ("u64", "u64"), ("f", "f")],                       # 24 fixed bytes, NULLs in every part
    [("s", "s1"), ("s", "s2"), ("b", "b"), ("u8", "u8"), ("i32", "i32"), ("d", "d"), ("k", "k")],
]


@pytest.mark.parametrize("gi", range(len(WIDE_GROUPINGS)))
def test_hash_aggregate_wide_group_keys(ctx, gi):
    names = WIDE_GROUPINGS[gi]
    rng = np.random.default_rng(gi)
    bs = [random_batch(6000, seed=140 + gi), random_batch(2500, seed=150 + gi)]
    for b in bs:                                           # few distinct values per column, so that groups repeat
        for name, mod in (("i64", 5), ("u64", 3), ("d", 4)):
            c = b[name]
            b[name] = OCol(c.dtype, (c.values % mod).astype(c.values.dtype), c.valid)
        b["f"] = OCol("Float64", np.round(b["f"].values / 100) + 0.0, b["f"].valid)       # + 0.0: no -0.0 among the keys
    m = helpers.memory_exec(ctx, [[bs[0]], [bs[1]]])
    group = [(col(a), n) for a, n in names]
    aggs = [E.Sum(col("g"), "sg"), E.Avg(col("f"), "af"), E.Count(col("f"), "cf"), E.Count(lit(1, E.UINT8), "n"),
            E.Max(col("i32"), "mx"), E.Min(col("g"), "mn")]
    keys = [n for _, n in names]
    part = ba.HashAggregateExec(ba.plan.PARTIAL, group, aggs, m)
    run_both(part, ordered=False, float_rtol=1e-9, key_cols=keys)
    fin = ba.HashAggregateExec(ba.plan.FINAL, [(col(n), n) for n in keys], aggs, ba.MergeExec(part))
    run_both(fin, ordered=False, float_rtol=1e-9, key_cols=keys)


def test_wide_group_keys_q10_shape(ctx):
    """TPC-H Q10's GROUP BY: c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment"""
    from collections import OrderedDict
    rng = np.random.default_rng(10)
    n_cust, n = 3000, 40_000
    def text(lo, hi):
        return ["".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(lo, hi)))) for _ in range(n_cust)]
    cust = dict(name=[f"Customer#{i:09d}" for i in range(n_cust)], acctbal=np.round(rng.uniform(-999, 9999, n_cust), 2),
                phone=[f"{i % 25 + 10}-{i % 900 + 100}-{i % 9000 + 1000}" for i in range(n_cust)],
                nation=[["FRANCE", "GERMANY", "UNITED STATES", "UNITED KINGDOM"][i % 4] for i in range(n_cust)],
                address=text(10, 41), comment=text(29, 117))
    pick = rng.integers(0, n_cust, n)
    b = OrderedDict([("c_custkey", OCol("Int32", pick.astype(np.int32))),
                     ("c_name", OCol("Utf8", [cust["name"][i] for i in pick])),
                     ("c_acctbal", OCol("Float64", cust["acctbal"][pick])),
                     ("c_phone", OCol("Utf8", [cust["phone"][i] for i in pick])),
                     ("n_name", OCol("Utf8", [cust["nation"][i] for i in pick])),
                     ("c_address", OCol("Utf8", [cust["address"][i] for i in pick])),
                     ("c_comment", OCol("Utf8", [cust["comment"][i] for i in pick])),
                     ("l_extendedprice", OCol("Float64", np.round(rng.uniform(900, 100000, n), 2))),
                     ("l_discount", OCol("Float64", rng.integers(0, 11, n) / 100.0))])
    keys = list(b)[:7]
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 15_000), helpers.slice_batch(b, 15_000, 25_000)], [helpers.slice_batch(b, 25_000, n)]])
    aggs = [E.Sum(col("l_extendedprice") * (lit(1.0) - col("l_discount")), "revenue")]
    part = ba.HashAggregateExec(ba.plan.PARTIAL, [(col(k), k) for k in keys], aggs, m)
    fin = ba.HashAggregateExec(ba.plan.FINAL, [(col(k), k) for k in keys], aggs, ba.MergeExec(part))
    got = run_both(fin, ordered=False, float_rtol=1e-9, key_cols=["c_custkey"])
    assert len(got["c_custkey"].values) == len(np.unique(pick))


def test_wide_group_keys_over_an_empty_input(ctx):
    b = random_batch(100, seed=5)
    flt = ba.FilterExec(col("g") > lit(2.0), helpers.memory_exec(ctx, [[b]]))
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("i64"), "i64"), (col("u64"), "u64"), (col("f"), "f")], [E.Count(lit(1, E.UINT8), "n")], flt)
    assert sum(x.num_rows for x in agg.collect()) == 0


@pytest.mark.parametrize("jt", [ba.plan.INNER, ba.plan.LEFT, ba.plan.RIGHT])
def test_hash_join(ctx, jt):
    from collections import OrderedDict
    rng = np.random.default_rng(3)
    nl, nr = 700, 3000
    left = OrderedDict([("lk", OCol("Int32", rng.integers(0, 400, nl), rng.random(nl) > 0.05)),
                        ("lv", OCol("Float64", rng.random(nl))),
                        ("ls", OCol("Utf8", [f"L{i % 13}" for i in range(nl)]))])
    def right(seed):
        r = np.random.default_rng(seed)
        return OrderedDict([("rk", OCol("Int32", r.integers(0, 600, nr), r.random(nr) > 0.05)),
                            ("rv", OCol("Int64", r.integers(0, 10 ** 9, nr))),
                            ("rs", OCol("Utf8", [f"R{i % 7}" for i in range(nr)], r.random(nr) > 0.1))])
    lm = helpers.memory_exec(ctx, [[helpers.slice_batch(left, 0, 300)], [helpers.slice_batch(left, 300, nl)]])
    rm = helpers.memory_exec(ctx, [[right(1), right(2)], [right(3)]])
    plan = ba.HashJoinExec(lm, rm, [("lk", "rk")], jt)
    assert [n for n, _, _ in plan.schema()] == ["lk", "lv", "ls", "rk", "rv", "rs"]
    assert plan.output_partitioning().partition_count() == 2
    for p in range(2):
        got = helpers.concat([helpers.from_device(b) for b in plan.execute(p)])
        want = helpers.concat(plan_eval.execute(plan, p))
        helpers.assert_rows_equal(got, want, ordered=False, key_cols=["lk", "rk", "rv", "lv"])


def test_hash_join_same_name_key_and_multi_key(ctx):
    from collections import OrderedDict
    rng = np.random.default_rng(8)
    left = OrderedDict([("key", OCol("Int64", rng.integers(0, 50, 200))), ("g", OCol("Utf8", [f"n{i % 5}" for i in range(200)])),
                        ("x", OCol("Float64", rng.random(200)))])
    right = OrderedDict([("key", OCol("Int64", rng.integers(0, 60, 900))), ("g2", OCol("Utf8", [f"n{i % 6}" for i in range(900)])),
                         ("y", OCol("Int32", rng.integers(0, 9, 900)))])
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]])
    plan = ba.HashJoinExec(lm, rm, [("key", "key"), ("g", "g2")], ba.plan.INNER)
    assert [n for n, _, _ in plan.schema()] == ["key", "g", "x", "g2", "y"]     # right `key` dropped
    run_both(plan, ordered=False, key_cols=["key", "g", "y", "x"])
    with pytest.raises(ba.PlanError):
        ba.HashJoinExec(lm, rm, [("key", "y")], ba.plan.INNER)                   # Int64 vs Int32


def test_hash_join_empty_sides(ctx):
    b = random_batch(50, seed=2, nulls=False)
    e = helpers.slice_batch(b, 0, 0)
    b2 = {("r_" + k): v for k, v in b.items()}
    from collections import OrderedDict
    b2 = OrderedDict(b2)
    e2 = helpers.slice_batch(b2, 0, 0)
    for l, r, jt in [(e, b2, ba.plan.INNER), (b, e2, ba.plan.INNER), (b, e2, ba.plan.LEFT), (e, b2, ba.plan.RIGHT)]:
        plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[l]]), helpers.memory_exec(ctx, [[r]]), [("k", "r_k")], jt)
        run_both(plan, ordered=False)


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 70_000])
def test_filter_utf8_column_against_a_literal(ctx, n):
    """`Utf8 column = / != literal` has its own kernel (offsets, the length test, then the bytes 8 at a time with an overlapping last
    load): literals of 0, 1, 7, 8, 9, 16, 17 and 64 bytes, 65 bytes (longer than the kernel's inline literal: the expression VM answers),
    the literal on the left, values that share the literal's length or its first / last 8 bytes, NULL rows (dropped under either
    operator)"""
    from collections import OrderedDict
    rng = np.random.default_rng(n)
    lits = ["", "B", "BUILDIN", "BUILDING", "MACHINERY", "0123456789abcdef", "0123456789abcdefg", "x" * 64, "y" * 65]
    pool = lits + ["BUILDINH", "AUILDING", "MACHINERZ", "0123456789abcdeX", "X123456789abcdefg", "x" * 63 + "y", "y" * 64 + "z", "b", "BUILDING "]
    vals = [pool[int(i)] for i in rng.integers(0, len(pool), n)]
    b = OrderedDict([("s", OCol("Utf8", vals, rng.random(n) > 0.15)), ("t", OCol("Utf8", vals)), ("i", OCol("Int64", np.arange(n)))])
    m = helpers.memory_exec(ctx, [[b]])
    for lit_s in lits:
        for column in ("s", "t"):
            run_both(ba.FilterExec(col(column).eq(lit(lit_s)), m))
            run_both(ba.FilterExec(col(column).ne(lit(lit_s)), m))
    run_both(ba.FilterExec(E.BinaryExpr(lit("BUILDING"), "Eq", col("s")), m))


SORTS = [
    [E.PhysicalSortExpr(col("k")), E.PhysicalSortExpr(col("g"), descending=True)],
    [E.PhysicalSortExpr(col("f"), descending=True, nulls_first=False), E.PhysicalSortExpr(col("d"))],
    [E.PhysicalSortExpr(col("s")), E.PhysicalSortExpr(col("i64"), descending=True)],
    [E.PhysicalSortExpr(col("s"), descending=True, nulls_first=False), E.PhysicalSortExpr(col("u8"))],
    [E.PhysicalSortExpr(col("b")), E.PhysicalSortExpr(col("u64"), nulls_first=False), E.PhysicalSortExpr(col("i32"))],
    [E.PhysicalSortExpr(col("f") * col("g"))],
]


@pytest.mark.parametrize("si", range(len(SORTS)))
def test_sort(ctx, si):
    bs = [random_batch(3000, seed=60 + si), random_batch(1234, seed=70 + si)]
    plan = ba.SortExec(SORTS[si], ba.MergeExec(helpers.memory_exec(ctx, [[bs[0]], [bs[1]]])))
    run_both(plan, ordered=True)      # both sorts are stable, so even ties line up


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 200, 256, 257])
@pytest.mark.parametrize("si", range(len(SORTS)))
def test_sort_few_rows(ctx, si, n):
    """at most 256 rows: ranks by row comparison and the gather of all ten columns in ONE launch (rowsort_kernel); 257: the
    LSD passes.  Same stable order either way: NULL placement per key, DESC, Boolean/Utf8/validity output by ballot + scan"""
    b = random_batch(n, seed=90 + si)
    run_both(ba.SortExec(SORTS[si], helpers.memory_exec(ctx, [[b]])), ordered=True)


def test_sort_few_rows_edge_values(ctx):
    """strings that are prefixes of each other with trailing NUL bytes, empty strings, NaN / infinities"""
    from collections import OrderedDict
    svals = ["ab", "ab\0", "a", "", "ab\0c", "abc", "a\0", "b", "", "ab"]
    fvals = np.array([0.0, 2.5, np.nan, np.inf, -np.inf, 1.5, -1.5, 2.5, 0.0, -3.0])     # (one NaN, no -0.0: their tie order is unspecified)
    b = OrderedDict([("s", OCol("Utf8", svals, np.array([1, 1, 1, 1, 1, 1, 1, 0, 1, 1], bool))), ("f", OCol("Float64", fvals)),
                     ("i", OCol("Int64", np.arange(10)))])
    m = helpers.memory_exec(ctx, [[b]])
    for desc in (False, True):
        for nf in (False, True):
            run_both(ba.SortExec([E.PhysicalSortExpr(col("s"), descending=desc, nulls_first=nf)], m), ordered=True)
            run_both(ba.SortExec([E.PhysicalSortExpr(col("f"), descending=desc, nulls_first=nf), E.PhysicalSortExpr(col("s"))], m), ordered=True)


@pytest.mark.parametrize("n", [4097, 150_000])
def test_sort_mid_size(ctx, n):
    """several workgroups per radix pass (37 at 150 K rows): Float64 keys DESC with NULLs, ties broken by a second key, then by
    input position — the same stable order as the oracle's sort"""
    from collections import OrderedDict
    rng = np.random.default_rng(n)
    b = OrderedDict([("f", OCol("Float64", np.round(rng.normal(0, 1000, n), 1) + 0.0, rng.random(n) > 0.05)),     # (+ 0.0: no -0.0, whose tie with 0.0 is unspecified)
                     ("d", OCol("Date32", rng.integers(8000, 10400, n))), ("i", OCol("Int64", rng.integers(-2**40, 2**40, n))),
                     ("s", OCol("Utf8", [f"k{v % 97}" for v in range(n)]))])
    m = helpers.memory_exec(ctx, [[b]])
    run_both(ba.SortExec([E.PhysicalSortExpr(col("f"), descending=True, nulls_first=False), E.PhysicalSortExpr(col("d"))], m), ordered=True)
    run_both(ba.SortExec([E.PhysicalSortExpr(col("s")), E.PhysicalSortExpr(col("i"), descending=True)], m), ordered=True)


@pytest.mark.parametrize("n", [1025, 4097, 150_000, 600_000])
def test_sort_bucket_path(n, monkeypatch):
    """1 K - 4 M rows under fixed-width keys: one split on the 16 highest differing bits of the composite key + an LDS sort per
    segment (kernels_sort.hip: bucket_sort).  The Q3 shape (Float64 DESC with NULLs, then a date), a low-cardinality first key
    (the digit takes its few bits and continues in the next word), two nullable keys (four words); the kernel list says the
    bucket path ran.  Heavily repeated keys overflow a bin and all-equal keys have no differing bit: both take the LSD passes."""
    from collections import OrderedDict
    monkeypatch.setenv("BHIP_KERNEL_TIMING", "2")             # every launch, however small (read when a context is created)
    ctx = ba.Context(0)
    rng = np.random.default_rng(n)
    b = OrderedDict([("f", OCol("Float64", np.round(rng.lognormal(10, 1, n), 2), rng.random(n) > 0.05)),
                     ("r", OCol("Float64", np.round(rng.lognormal(10, 1, n), 2))),
                     ("d", OCol("Date32", rng.integers(8000, 10400, n))),
                     ("g", OCol("Int32", rng.integers(0, 5, n))),
                     ("u", OCol("Float64", rng.random(n))),
                     ("i", OCol("Int64", rng.integers(-2**40, 2**40, n), rng.random(n) > 0.5)),
                     ("few", OCol("Int64", rng.integers(0, 3, n) * 1000)),
                     ("one", OCol("Int32", np.full(n, 7)))])
    m = helpers.memory_exec(ctx, [[b]])
    small = n <= 4097              # the NULL rows of a key are one bin: more than 512 of them overflow it
    cases = [
        ([E.PhysicalSortExpr(col("r"), descending=True, nulls_first=False), E.PhysicalSortExpr(col("d"))], True),
        ([E.PhysicalSortExpr(col("g")), E.PhysicalSortExpr(col("u"), descending=True)], True),
        ([E.PhysicalSortExpr(col("one")), E.PhysicalSortExpr(col("u"))], True),
        ([E.PhysicalSortExpr(col("f"), descending=True, nulls_first=False), E.PhysicalSortExpr(col("d"))], small),
        ([E.PhysicalSortExpr(col("f"), nulls_first=True), E.PhysicalSortExpr(col("i"), descending=True, nulls_first=False)], small),
        ([E.PhysicalSortExpr(col("few")), E.PhysicalSortExpr(col("g"))], small),                        # fifteen value pairs: a fifteenth of the rows per bin
        ([E.PhysicalSortExpr(col("one"))], False),                                                      # no differing bit at all
        ([E.PhysicalSortExpr(col("g")), E.PhysicalSortExpr(col("f")), E.PhysicalSortExpr(col("i"))], None),   # five words: not eligible
    ]
    for keys, buckets in cases:
        ctx.kernel_stats(reset=True)
        run_both(ba.SortExec(keys, m), ordered=True)
        ks = ctx.kernel_stats(reset=True)
        assert ("bucket_sort" in ks) == (buckets is not None), (keys, ks)
        assert ("sort_key_fixed" in ks) == (not buckets), (keys, ks)      # the LSD passes ran iff the bucket path did not, or gave up


@pytest.mark.parametrize("n", [700, 5000])
def test_sort_by_long_strings(ctx, n):
    """Utf8 sort keys of any length (here up to 90 bytes, sharing long prefixes, with NULLs and empty strings): one
    stable pass per 8-byte chunk from the last to the first, then the length; small (one-launch) and radix paths"""
    from collections import OrderedDict
    rng = np.random.default_rng(n)
    stems = ["", "x", "Customer#000000", "carefully final deposits detect slyly agai", "carefully final deposits detect slyly agai" * 2]
    vals = [stems[int(rng.integers(0, len(stems)))] + "".join(chr(int(c)) for c in rng.integers(97, 100, int(rng.integers(0, 7)))) for _ in range(n)]
    b = OrderedDict([("s", OCol("Utf8", vals, rng.random(n) > 0.1)), ("k", OCol("Int32", rng.integers(0, 5, n))),
                     ("v", OCol("Float64", rng.random(n)))])
    m = helpers.memory_exec(ctx, [[b]])
    for desc in (False, True):
        run_both(ba.SortExec([E.PhysicalSortExpr(col("s"), descending=desc, nulls_first=desc)], m), ordered=True)
    run_both(ba.SortExec([E.PhysicalSortExpr(col("k")), E.PhysicalSortExpr(col("s"), descending=True)], m), ordered=True)


def test_sort_requires_single_partition(ctx):
    b = random_batch(10)
    with pytest.raises(ba.PlanError, match="single input partition"):
        ba.SortExec([E.PhysicalSortExpr(col("k"))], helpers.memory_exec(ctx, [[b], [b]])).collect()


@pytest.mark.parametrize("nparts", [2, 4, 8])
def test_hash_repartition(ctx, nparts):
    bs = [random_batch(2500, seed=80), random_batch(900, seed=81)]
    m = helpers.memory_exec(ctx, [[bs[0]], [bs[1]]])
    exprs = [col("i32"), col("s")]
    plan = ba.RepartitionExec(m, ba.Partitioning.Hash(exprs, nparts))
    assert plan.output_partitioning().partition_count() == nparts
    total = 0
    for p in range(nparts):
        got = helpers.concat([helpers.from_device(b) for b in plan.execute(p)])
        want = helpers.concat(plan_eval.execute(plan, p))
        helpers.assert_rows_equal(got, want, ordered=True)      # same hash, input order kept inside a partition
        total += og.batch_len(got)
    assert total == 3400


def test_round_robin_repartition(ctx):
    bs = [random_batch(100, seed=s) for s in range(5)]
    m = helpers.memory_exec(ctx, [[bs[0], bs[1], bs[2]], [bs[3], bs[4]]])
    plan = ba.RepartitionExec(m, ba.Partitioning.RoundRobinBatch(3))
    for p in range(3):
        got = helpers.concat([helpers.from_device(b) for b in plan.execute(p)])
        helpers.assert_rows_equal(got, helpers.concat(plan_eval.execute(plan, p)), ordered=True)


def test_arrow_c_data_and_stream_interface(ctx):
    import pyarrow as pa
    t = pa.table({"a": pa.array([1, 2, None, 4], pa.int32()), "s": pa.array(["x", None, "zz", ""]),
                  "f": pa.array([0.5, 1.5, 2.5, None]), "b": pa.array([True, False, None, True]),
                  "d": pa.array([8035, 9000, None, 10000], pa.date32())})
    rb = ba.RecordBatch.from_pyarrow(ctx, t.to_batches()[0])
    back = rb.to_pyarrow()
    assert back.to_pydict() == t.to_pydict()
    # sliced arrays (non-zero offset) cross the boundary too
    sl = t.slice(1, 3).to_batches()[0]
    assert ba.RecordBatch.from_pyarrow(ctx, sl).to_pyarrow().to_pydict() == sl.to_pydict()
    plan = ba.FilterExec(col("a") > lit(1, E.INT32), ba.MemoryExec([[rb]], ctx))
    reader = plan.execute(0).to_arrow_reader()
    out = reader.read_all()
    assert out.to_pydict() == t.filter(pa.compute.greater(t["a"], 1)).to_pydict()


def test_tpch_q3_q5_on_reference_fixtures(ctx):
    """the reference's own 10-row table fixtures through the full Q3 / Q5 plans"""
    li = helpers.concat([helpers.lineitem_fixture("lineitem_partition0"), helpers.lineitem_fixture("lineitem_partition1")])
    # make the joins hit: fixture order keys 1-3 exist in orders; customers 1-10
    m = lambda b: helpers.memory_exec(ctx, [[b]])
    q3 = tpch.q3_plan(m(helpers.customer_fixture()), m(helpers.orders_fixture()), m(li))
    run_both(q3, ordered=False, float_rtol=1e-9)
    q5 = tpch.q5_plan(m(helpers.customer_fixture()), m(helpers.orders_fixture()), m(li), m(helpers.supplier_fixture()),
                      m(helpers.nation_fixture()), m(helpers.region_fixture()))
    run_both(q5, ordered=False, float_rtol=1e-9)


def test_tpch_q3_synthetic(ctx):
    sf = 0.002
    m = lambda b, k=1: helpers.memory_exec(ctx, [[b]])
    li = gen.lineitem(sf)
    half = og.batch_len(li) // 2
    lim = helpers.memory_exec(ctx, [[helpers.slice_batch(li, 0, half)], [helpers.slice_batch(li, half, 10 ** 9)]])
    q3 = tpch.q3_plan(m(gen.customer(sf)), m(gen.orders(sf)), lim)
    got = run_both(q3, ordered=False, float_rtol=1e-9, key_cols=["l_orderkey"])
    assert og.batch_len(got) > 10
    rev = got["revenue"].values
    assert np.all(rev[:-1] >= rev[1:])       # ORDER BY revenue DESC


def test_tpch_q5_synthetic(ctx):
    sf = 0.01
    m = lambda b: helpers.memory_exec(ctx, [[b]])
    q5 = tpch.q5_plan(m(gen.customer(sf)), m(gen.orders(sf)), m(gen.lineitem(sf)), m(gen.supplier(sf)),
                      m(gen.nation()), m(gen.region()))
    got = run_both(q5, ordered=False, float_rtol=1e-9, key_cols=["n_name"])
    assert 1 <= og.batch_len(got) <= 5
    assert set(got["n_name"].values) <= {"CHINA", "INDIA", "INDONESIA", "JAPAN", "VIETNAM"}      # r_name = 'ASIA'


def test_full_q1_plan_sorted(ctx):
    li = gen.lineitem(0.005)
    n = og.batch_len(li)
    parts = [[helpers.slice_batch(li, i * 7000, (i + 1) * 7000)] for i in range((n + 6999) // 7000)]
    plan = tpch.q1_plan(helpers.memory_exec(ctx, parts))
    got = run_both(plan, ordered=True, float_rtol=1e-6)
    assert list(zip(got["l_returnflag"].values, got["l_linestatus"].values)) == [("A", "F"), ("N", "F"), ("N", "O"), ("R", "F")]


def test_arrow_stream_leaf(ctx):
    """bhip_plan_arrow_stream: a pyarrow RecordBatchReader (Arrow C stream) as the input of GPU operators"""
    import pyarrow as pa
    rng = np.random.default_rng(17)
    batches = []
    for n in (1000, 1, 2500):
        batches.append(pa.RecordBatch.from_arrays(
            [pa.array(rng.integers(0, 5, n).astype(np.int32)), pa.array([None if i % 5 == 0 else float(v) for i, v in enumerate(rng.random(n))], type=pa.float64()),
             pa.array([None if i % 7 == 0 else f"s{i % 3}" for i in range(n)], type=pa.string())], names=["k", "x", "s"]))
    schema = batches[0].schema
    leaf = ba.ArrowStreamExec(pa.RecordBatchReader.from_batches(schema, batches), ctx)
    assert [(n, t) for n, t, _ in leaf.schema()] == [("k", "Int32"), ("x", "Float64"), ("s", "Utf8")]
    assert leaf.output_partitioning().partition_count() == 1
    plan = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("k"), "k")], [E.Sum(col("x"), "sx"), E.Count(col("x"), "cx")],
                                ba.FilterExec(E.IsNotNullExpr(col("s")), leaf))
    for _ in range(2):                                     # the second execute replays the drained stream
        got = helpers.concat(helpers.collect_product(plan))
        t = pa.Table.from_batches(batches)
        tab = t.filter(__import__("pyarrow.compute").compute.is_valid(t["s"])).group_by("k").aggregate([("x", "sum"), ("x", "count")]).sort_by("k").to_pydict()
        order = np.argsort(got["k"].values)
        assert [int(got["k"].values[i]) for i in order] == tab["k"]
        assert [int(got["cx[count]"].values[i]) for i in order] == tab["x_count"]
        assert np.allclose([got["sx[sum]"].values[i] for i in order], tab["x_sum"], rtol=1e-12)
    # a producer whose batch does not match the stream's schema is reported, not read out of bounds
    liar = pa.RecordBatch.from_arrays([pa.array([1], type=pa.int32()), pa.array([None]), pa.array([None])], names=["k", "x", "s"])
    with pytest.raises(ba.BallistaError):
        ba.ArrowStreamExec(pa.RecordBatchReader.from_batches(schema, [batches[0], liar]), ctx).collect()
    # an unsupported column type is refused at plan time
    bad = pa.RecordBatchReader.from_batches(pa.schema([("f", pa.float16())]), [])
    with pytest.raises(ba.NotImplementedOnGpu):
        ba.ArrowStreamExec(bad, ctx)


@pytest.mark.parametrize("grouped", [False, True])
def test_count_of_a_nullable_utf8_column(ctx, grouped):
    """COUNT(s) = number of non-NULL strings (compiled as COUNT(CASE WHEN s IS NOT NULL THEN 1 END))"""
    b = random_batch(5000, seed=77, long_strings=False)
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 3000)], [helpers.slice_batch(b, 3000, 5000)]])
    group = [(col("i32"), "i32")] if grouped else []
    aggs = [E.Count(col("s"), "cs"), E.Count(lit(1, E.UINT8), "n"), E.Sum(col("g"), "sg")]
    part = ba.HashAggregateExec(ba.plan.PARTIAL, group, aggs, m)
    fin = ba.HashAggregateExec(ba.plan.FINAL, group, aggs, ba.MergeExec(part))
    got = run_both(fin, ordered=False, float_rtol=1e-9, key_cols=[n for _, n in group])
    assert sum(int(v) for v in got["cs"].values) == int(np.sum(b["s"].valid))


@pytest.mark.parametrize("pattern", ["a_c", "_", "__", "%", "%%", "a%c", "%a%c%", "_b%", "%b_", "a%b%c_d", "%é_", "_é%", "日_語", "%語",
                                      "ab", "", "a__%__z", "%ab%ab%"])
def test_like_general_patterns(ctx, pattern):
    """'%' = any sequence, '_' = exactly one character (UTF-8 aware), anywhere in the pattern"""
    from collections import OrderedDict
    words = ["", "a", "ab", "abc", "aXc", "ac", "abbc", "abcabc", "aébc", "é", "日本語", "日x語", "bb", "ab\ncd", "a1b2c3d", "xaby", "abab",
             "a__z", "aqwertz", "a12z"]
    rng = np.random.default_rng(len(pattern))
    n = 3000
    vals = [words[k] for k in rng.integers(0, len(words), n)]
    b = OrderedDict([("s", OCol("Utf8", vals, rng.random(n) > 0.1)), ("i", OCol("Int32", np.arange(n, dtype=np.int32)))])
    m = helpers.memory_exec(ctx, [[b]])
    for op in ("Like", "NotLike"):
        run_both(ba.FilterExec(E.BinaryExpr(col("s"), op, lit(pattern)), m), ordered=True)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 5000])
@pytest.mark.parametrize("density", [0.0, 0.02, 0.5, 1.0])
def test_filter_selection_sizes_and_densities(ctx, n, density):
    """select_indices takes 16 rows per lane and writes one set bit per step: row counts around its pieces (16), words (64) and
    tiles (1024), from nothing selected to everything"""
    rng = np.random.default_rng(n)
    keep = rng.random(n) < density
    b = OrderedDict([("k", OCol("Int32", np.where(keep, 1, 0).astype(np.int32))), ("v", OCol("Int64", np.arange(n, dtype=np.int64))),
                     ("s", OCol("Utf8", [f"s{i % 11}" for i in range(n)]))])
    plan = ba.FilterExec(E.coerce(col("k").eq(lit(1)), {"k": "Int32"}), helpers.memory_exec(ctx, [[b]]))
    helpers.assert_rows_equal(helpers.concat(helpers.collect_product(plan)), plan_eval.collect(plan), ordered=True)


@pytest.mark.parametrize("ng", [9000, 150])
@pytest.mark.parametrize("clustered", [True, False])
def test_hash_aggregate_counts_rows_of_filtered_runs(ctx, clustered, ng):
    """many groups (the hash path), Float64 sums (fixed order: the segment kernel counts a group's rows run by run) and a
    predicate that drops rows in the middle of runs: COUNT / AVG need every kept row counted exactly once.  9000 groups: a few
    runs per group, combined through the per-group lists; 150 unclustered groups: hundreds of runs each, the sorted combine"""
    rng = np.random.default_rng(21)
    n = 60000
    g = np.sort(rng.integers(0, ng, n)) if clustered else rng.integers(0, ng, n)
    b = OrderedDict([("g", OCol("Int32", g.astype(np.int32))), ("x", OCol("Float64", np.round(rng.random(n) * 100, 2))),
                     ("keep", OCol("Int32", (rng.random(n) > 0.3).astype(np.int32)))])
    m = helpers.memory_exec(ctx, [[helpers.slice_batch(b, 0, 25000), helpers.slice_batch(b, 25000, n)]])
    flt = ba.FilterExec(E.coerce(col("keep").eq(lit(1)), {"keep": "Int32"}), m)
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("g"), "g")], [E.Sum(col("x"), "s"), E.Count(col("x"), "c"), E.Avg(col("x"), "a")], flt)
    helpers.assert_rows_equal(helpers.concat(helpers.collect_product(agg)), plan_eval.collect(agg), ordered=False, float_rtol=1e-12, key_cols=["g"])
