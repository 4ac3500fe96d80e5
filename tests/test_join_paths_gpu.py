"""GPU parity of HashJoinExec's table variants against the CPU oracle (reference operator:
rust/core/src/serde/physical_plan/from_proto.rs:253-276 — Inner / Left / Right, equi-keys by column name):

  narrow   one Int32 / Date32 key, unique build side: key + build row in one 8-byte slot, probe reads the key column
  general  packed 16-byte keys: duplicates on the build side, Int64 / multi-column / Utf8 keys
  late materialisation  probe side = column projection over a filter: only its key columns are gathered before the probe

Row multisets must match exactly (join output order is unspecified)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E
from ballista_amd.expr import col, lit
from oracle import plan_eval
from oracle.engine import OCol

import helpers

pytestmark = pytest.mark.gpu
JOIN_TYPES = [ba.plan.INNER, ba.plan.LEFT, ba.plan.RIGHT]


def sides(n_left, n_right, key_type="Int32", unique=True, nulls=False, seed=0, key_range=None):
    rng = np.random.default_rng(seed)
    key_range = key_range or 2 * n_left
    lk = rng.permutation(key_range)[:n_left] if unique else rng.integers(0, max(2, n_left // 3), n_left)
    rk = rng.integers(0, key_range + 50, n_right)
    np_t = np.int64 if key_type == "Int64" else np.int32
    lv = (rng.random(n_left) > 0.1) if nulls else None
    rv = (rng.random(n_right) > 0.1) if nulls else None
    left = OrderedDict([("lk", OCol(key_type, lk.astype(np_t) - 7, lv)), ("lx", OCol("Float64", rng.random(n_left))),
                        ("ls", OCol("Utf8", [f"L{i % 11}" for i in range(n_left)]))])
    right = OrderedDict([("rk", OCol(key_type, rk.astype(np_t) - 7, rv)), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right))),
                         ("rd", OCol("Date32", rng.integers(9000, 10000, n_right).astype(np.int32)))])
    return left, right


def check(plan, keys):
    got = helpers.concat(helpers.collect_product(plan))
    helpers.assert_rows_equal(got, plan_eval.collect(plan), ordered=False, key_cols=keys)
    return got


@pytest.mark.parametrize("jt", JOIN_TYPES)
@pytest.mark.parametrize("key_type", ["Int32", "Date32", "Int64"])
@pytest.mark.parametrize("unique", [True, False])
@pytest.mark.parametrize("nulls", [False, True])
def test_table_variants(ctx, jt, key_type, unique, nulls):
    left, right = sides(900, 5000, key_type, unique, nulls, seed=11)
    lm = helpers.memory_exec(ctx, [[helpers.slice_batch(left, 0, 400)], [helpers.slice_batch(left, 400, 900)]])
    rm = helpers.memory_exec(ctx, [[helpers.slice_batch(right, 0, 1025), helpers.slice_batch(right, 1025, 3000)], [helpers.slice_batch(right, 3000, 5000)]])
    plan = ba.HashJoinExec(lm, rm, [("lk", "rk")], jt)
    for p in range(2):
        got = helpers.concat([helpers.from_device(b) for b in plan.execute(p)])
        want = helpers.concat(plan_eval.execute(plan, p))
        helpers.assert_rows_equal(got, want, ordered=False, key_cols=["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("jt", JOIN_TYPES)
@pytest.mark.parametrize("unique", ["first_column", "pairs", "no"])
@pytest.mark.parametrize("nulls", [False, True])
@pytest.mark.parametrize("filtered", [False, True])
def test_two_four_byte_keys(ctx, jt, unique, nulls, filtered):
    """ON (a, b) = (c, d) over Int32 / Date32 columns (TPC-H Q5's supplier join: suppkey, nationkey).  Build side unique on `a`
    alone: the join goes by `a` and every match is confirmed on the second column inside the probe kernel; unique as pairs only:
    both columns packed into one 8-byte key, CAS table; duplicate pairs: the general tables.  A NULL in either part never matches"""
    rng = np.random.default_rng(3)
    nl, nr = 700, 6000
    la, lb = rng.integers(-20, 60, nl).astype(np.int32), rng.integers(9000, 9012, nl).astype(np.int32)
    if unique == "pairs":
        pairs = sorted({(int(a), int(b)) for a, b in zip(la, lb)})
        rng.shuffle(pairs)
        la, lb = np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32)
        nl = len(la)
    if unique == "first_column":
        la = (rng.permutation(90)[:80] - 22).astype(np.int32)
        nl = len(la)
        lb = rng.integers(9000, 9012, nl).astype(np.int32)
    ra, rb = rng.integers(-25, 65, nr).astype(np.int32), rng.integers(8998, 9014, nr).astype(np.int32)
    v = lambda n: (rng.random(n) > 0.1) if nulls else None
    left = OrderedDict([("la", OCol("Int32", la, v(nl))), ("lb", OCol("Date32", lb, v(nl))), ("lx", OCol("Float64", rng.random(nl)))])
    right = OrderedDict([("ra", OCol("Int32", ra, v(nr))), ("rb", OCol("Date32", rb, v(nr))), ("ry", OCol("Int64", rng.integers(0, 100, nr))),
                         ("rs", OCol("Utf8", [f"r{i % 7}" for i in range(nr)]))])
    lm = helpers.memory_exec(ctx, [[left]])
    rm = helpers.memory_exec(ctx, [[helpers.slice_batch(right, 0, 2500), helpers.slice_batch(right, 2500, nr)]])
    probe = ba.FilterExec(E.coerce(col("ry") < lit(60), {"ry": "Int64"}), rm) if filtered else rm
    check(ba.HashJoinExec(lm, probe, [("la", "ra"), ("lb", "rb")], jt), ["la", "lb", "ra", "rb", "ry", "lx"])


@pytest.mark.parametrize("jt_top", JOIN_TYPES)
@pytest.mark.parametrize("jt_mid", JOIN_TYPES)
@pytest.mark.parametrize("shape", ["probe_chain", "probe_direct", "build_chain"])
def test_payload_columns_pass_through_joins_as_views(ctx, jt_top, jt_mid, shape):
    """three joins in a row with column projections between them (the shape of TPC-H Q3 / Q5): a column a join only passes on
    travels as (source column, row indices) and is gathered once, by the last join — through Inner / Left / Right joins (NULL
    partners), Utf8 / Boolean / nullable payloads, on the build and on the probe side of the next join"""
    rng = np.random.default_rng(5)
    na, nb, nc, nd = 60, 900, 4000, 300
    a = OrderedDict([("ak", OCol("Int32", np.arange(na, dtype=np.int32))), ("aname", OCol("Utf8", [f"name-{i % 13}" for i in range(na)], rng.random(na) > 0.1))])
    b = OrderedDict([("bk", OCol("Int32", np.arange(nb, dtype=np.int32) + 5)), ("b_ak", OCol("Int32", rng.integers(-3, na + 3, nb).astype(np.int32), rng.random(nb) > 0.05)),
                     ("bflag", OCol("Boolean", rng.random(nb) > 0.5, rng.random(nb) > 0.2))])
    c = OrderedDict([("c_bk", OCol("Int32", rng.integers(0, nb + 20, nc).astype(np.int32))), ("cx", OCol("Float64", rng.random(nc), rng.random(nc) > 0.1)),
                     ("cs", OCol("Utf8", [f"c{i % 29}" for i in range(nc)])), ("c_dk", OCol("Int32", rng.integers(0, nd + 10, nc).astype(np.int32)))])
    d = OrderedDict([("dk", OCol("Int32", np.arange(nd, dtype=np.int32))), ("dy", OCol("Int64", rng.integers(0, 1000, nd)))])
    A, B, C, D = (helpers.memory_exec(ctx, [[t]]) for t in (a, b, c, d))
    proj = lambda names, p: ba.ProjectionExec([(col(n), n) for n in names], p)
    j1 = proj(["bk", "aname", "bflag"], ba.HashJoinExec(A, B, [("ak", "b_ak")], jt_mid))              # a |x| b
    if shape == "probe_chain":
        j2 = proj(["aname", "bflag", "cx", "cs", "c_dk"], ba.HashJoinExec(j1, C, [("bk", "c_bk")], jt_mid))    # (a b) |x| c: j1 is the BUILD side
        top = ba.HashJoinExec(D, j2, [("dk", "c_dk")], jt_top)                                          # d |x| (a b c): j2 is the PROBE side
        keys = ["dk", "c_dk", "cx", "cs", "aname"]
    elif shape == "probe_direct":
        top = ba.HashJoinExec(D, ba.HashJoinExec(j1, C, [("bk", "c_bk")], jt_mid), [("dk", "c_dk")], jt_top)    # Q5's supplier join: no projection between
        keys = ["dk", "c_dk", "cx", "cs", "aname", "bk"]
    else:
        j2 = proj(["c_dk", "aname", "cx", "bflag"], ba.HashJoinExec(j1, C, [("bk", "c_bk")], jt_mid))
        top = ba.HashJoinExec(j2, D, [("c_dk", "dk")], jt_top)                                          # (a b c) |x| d: j2 is the BUILD side
        keys = ["c_dk", "dk", "cx", "aname", "dy"]
    check(top, keys)
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("aname"), "aname")], [E.Sum(col("cx"), "s"), E.Count(col("bflag"), "c")], top)
    helpers.assert_rows_equal(helpers.concat(helpers.collect_product(agg)), plan_eval.collect(agg), ordered=False, float_rtol=1e-9, key_cols=["aname"])


@pytest.mark.parametrize("jt", JOIN_TYPES)
def test_probe_side_projection_over_filter(ctx, jt):
    """the Q3 shape: HashJoin(build, Projection(Filter(probe))) with a renamed and a dropped column"""
    left, right = sides(600, 4000, "Int32", True, False, seed=5)
    schema = {"rk": "Int32", "ry": "Int64", "rd": "Date32"}
    flt = ba.FilterExec(E.coerce(col("rd") > E.date32("1996-01-01"), schema), helpers.memory_exec(ctx, [[right]]))
    proj = ba.ProjectionExec([(col("ry"), "payload"), (col("rk"), "rk")], ba.CoalesceBatchesExec(flt, 4096))
    plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), proj, [("lk", "rk")], jt)
    assert [n for n, _, _ in plan.schema()] == ["lk", "lx", "ls", "payload", "rk"]
    check(plan, ["lk", "rk", "payload", "lx"])


def test_probe_side_filter_selects_nothing_or_everything(ctx):
    left, right = sides(300, 2000, "Int32", True, False, seed=9)
    schema = {"rk": "Int32", "ry": "Int64", "rd": "Date32"}
    for pred in (col("rd") > E.date32("2030-01-01"), col("rd") > E.date32("1970-01-01")):
        flt = ba.FilterExec(E.coerce(pred, schema), helpers.memory_exec(ctx, [[right]]))
        proj = ba.ProjectionExec([(col("rk"), "rk"), (col("rd"), "rd")], flt)
        check(ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), proj, [("lk", "rk")], ba.plan.INNER), ["lk", "rk", "rd", "lx"])


def test_probe_sizes_around_the_selection_tile(ctx):
    for n in (1, 63, 64, 65, 1023, 1024, 1025, 4097):
        left, right = sides(200, n, "Int32", True, False, seed=n)
        plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]]), [("lk", "rk")], ba.plan.INNER)
        check(plan, ["lk", "rk", "ry", "lx"])


def test_negative_and_extreme_keys(ctx):
    lk = np.array([-2 ** 31, -1, 0, 1, 2 ** 31 - 1, 123456789], dtype=np.int32)
    left = OrderedDict([("lk", OCol("Int32", lk)), ("lx", OCol("Float64", np.arange(6, dtype=np.float64)))])
    rk = np.array([2 ** 31 - 1, -2 ** 31, 5, 0, 0, -1, 123456789, 7], dtype=np.int32)
    right = OrderedDict([("rk", OCol("Int32", rk)), ("ry", OCol("Int64", np.arange(8)))])
    plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]]), [("lk", "rk")], ba.plan.INNER)
    got = check(plan, ["lk", "rk", "ry", "lx"])
    assert len(got["lk"].values) == 6


@pytest.mark.parametrize("jt", JOIN_TYPES)
@pytest.mark.parametrize("shape", ["dense", "sparse", "full_range"])
@pytest.mark.parametrize("key_type", ["Int32", "Int64"])
def test_build_key_set_bitmap(ctx, jt, shape, key_type):
    """a unique integer build side of >= 2^18 rows gets its exact key set as a bitmap in front of the table (ops_join.cpp);
    dense: keys in a window around 0 with NULLs; sparse: a window of 2^29 values; full_range: the window exceeds the
    bitmap budget and the table alone answers. Probe keys fall inside, below and above the window."""
    rng = np.random.default_rng({"dense": 1, "sparse": 2, "full_range": 3}[shape])
    n_left, n_right = 300_000, 500_000
    top = 63 if key_type == "Int64" else 31                  # full_range spans the whole key type
    base = 10 ** 12 if key_type == "Int64" else 0            # Int64 windows sit far outside the Int32 values
    if shape == "dense":
        lk = rng.permutation(400_000)[:n_left].astype(np.int64) - 150_000
    elif shape == "sparse":
        lk = rng.choice(np.unique(rng.integers(-2 ** 28, 2 ** 28, 2 * n_left)), n_left, replace=False)
    else:
        lk = rng.choice(np.unique(rng.integers(-2 ** top + 1, 2 ** top - 1, 2 * n_left)), n_left - 2, replace=False)
        lk = np.concatenate([lk[:1000], [-2 ** top, 2 ** top - 1], lk[1000:]])
        base = 0
    lk = lk.astype(np.int64) + base
    n_left = len(lk)
    lo, hi = int(lk.min()), int(lk.max())
    rk = np.where(rng.random(n_right) < 0.5, lk[rng.integers(0, n_left, n_right)],
                  rng.integers(max(lo - 1000, -2 ** top), min(hi + 1000, 2 ** top - 1), n_right))
    rk[:4] = [max(lo - 1, -2 ** top), min(hi + 1, 2 ** top - 1), lo, hi]
    lv = (rng.random(n_left) > 0.05) if shape == "dense" else None
    np_t = np.int64 if key_type == "Int64" else np.int32
    left = OrderedDict([("lk", OCol(key_type, lk.astype(np_t), lv)), ("lx", OCol("Float64", rng.random(n_left)))])
    right = OrderedDict([("rk", OCol(key_type, rk.astype(np_t))), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right)))])
    plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]]), [("lk", "rk")], jt)
    check(plan, ["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("jt", JOIN_TYPES)
@pytest.mark.parametrize("order", ["sorted", "shuffled", "sorted_with_gap_words"])
@pytest.mark.parametrize("key_type", ["Int32", "Int64"])
def test_rank_map_build_orders(ctx, jt, order, key_type):
    """the rank map (kernels_join.hip): a build side sorted by key is written with plain stores and rank = row; any other order
    goes through atomicOr + perm[].  Keys cross many 64-key words, some words hold one key, some 64, runs straddle waves."""
    rng = np.random.default_rng(21)
    if order == "sorted_with_gap_words":
        lk = np.concatenate([np.arange(0, 200), np.arange(1000, 1064), [5000, 5001, 70000], np.arange(70064, 72000, 3)]).astype(np.int64)
    else:
        lk = np.sort(rng.choice(40_000, 9_000, replace=False)).astype(np.int64)
    if order == "shuffled":
        lk = rng.permutation(lk)
    lk = lk - 3000 + (10 ** 11 if key_type == "Int64" else 0)
    n_left, n_right = len(lk), 30_000
    rk = np.where(rng.random(n_right) < 0.4, lk[rng.integers(0, n_left, n_right)], rng.integers(lk.min() - 50, lk.max() + 50, n_right))
    np_t = np.int64 if key_type == "Int64" else np.int32
    left = OrderedDict([("lk", OCol(key_type, lk.astype(np_t))), ("lx", OCol("Float64", rng.random(n_left))), ("ls", OCol("Utf8", [f"L{i % 7}" for i in range(n_left)]))])
    right = OrderedDict([("rk", OCol(key_type, rk.astype(np_t))), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right))),
                         ("rd", OCol("Date32", rng.integers(9000, 10000, n_right).astype(np.int32)))])
    schema = {"rk": key_type, "ry": "Int64", "rd": "Date32"}
    rm = helpers.memory_exec(ctx, [[helpers.slice_batch(right, 0, 12_345), helpers.slice_batch(right, 12_345, n_right)]])
    lm = helpers.memory_exec(ctx, [[left]])
    check(ba.HashJoinExec(lm, rm, [("lk", "rk")], jt), ["lk", "rk", "ry", "lx"])
    # the same under a range filter on the probe side (the fused filter + probe pass), two ranges on two columns
    pred = E.coerce((col("rd") >= E.date32("1995-01-01")).and_(col("rd") < E.date32("1996-06-01")).and_(col("rk") > lit(int(lk.min()) + 10, key_type)), schema)
    if key_type == "Int32":
        flt = ba.FilterExec(pred, rm)
    else:
        flt = ba.FilterExec(E.coerce((col("rd") >= E.date32("1995-01-01")).and_(col("rd") < E.date32("1996-06-01")), schema), rm)
    check(ba.HashJoinExec(lm, flt, [("lk", "rk")], jt), ["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("jt", [ba.plan.INNER, ba.plan.LEFT])
@pytest.mark.parametrize("order", ["sorted", "shuffled"])
def test_rank_map_window_beyond_2_to_the_30(jt, order, monkeypatch):
    """TPC-H SF1000 order keys: Int64, a window of 6 x 10^9 values (dbgen's sparse layout), far more than 2^30 and more than 2^32
    — the rank map's granule index is (offset >> 5) of a 64-bit offset (ops_join.cpp: windows up to 2^36).  1.6 M build keys
    over [base, base + 6e9): ~1 KiB of map per build row, the sparsest a rank map is built for; probe keys inside, at both ends
    of and outside the window, and 2^32 apart from build keys (an offset truncated to 32 bits would match them)."""
    monkeypatch.setenv("BHIP_KERNEL_TIMING", "1")             # (read when a context is created: this test names the kernels that ran)
    ctx = ba.Context(0)
    rng = np.random.default_rng(33)
    # (a rank map is built when the window is not sparser than 4096 key values per build row: 6e9 / 4096 = 1.47 M rows)
    n_left, n_right, span, base = 1_600_000, 2_000_000, 6_000_000_000, 7_000_000_123
    lk = np.unique(rng.integers(0, span, n_left + 4096))[:n_left] + base
    lk[0], lk[-1] = base, base + span - 1
    if order == "shuffled":
        lk = rng.permutation(lk)
    n_left = len(lk)
    pick = lk[rng.integers(0, n_left, n_right)]
    rk = np.where(rng.random(n_right) < 0.4, pick, rng.integers(base - 1000, base + span + 1000, n_right))
    alias = rng.random(n_right) < 0.1
    rk = np.where(alias, pick + (1 << 32), rk)                                     # same low 32 bits as a build key
    rk[:4] = [base - 1, base + span, base, base + span - 1]
    left = OrderedDict([("lk", OCol("Int64", lk.astype(np.int64))), ("lx", OCol("Float64", rng.random(n_left)))])
    right = OrderedDict([("rk", OCol("Int64", rk.astype(np.int64))), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right))),
                         ("rd", OCol("Date32", rng.integers(9000, 10000, n_right).astype(np.int32)))])
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]])
    ctx.kernel_stats(reset=True)
    check(ba.HashJoinExec(lm, rm, [("lk", "rk")], jt), ["lk", "rk", "ry", "lx"])
    ks = ctx.kernel_stats(reset=True)                         # the rank map it is, not the CAS table
    assert any(k.startswith("rank_bits") for k in ks) and "join_build_narrow" not in ks, ks
    flt = ba.FilterExec(E.coerce((col("rd") >= E.date32("1995-01-01")).and_(col("rd") < E.date32("1996-06-01")), {"rk": "Int64", "ry": "Int64", "rd": "Date32"}), rm)
    check(ba.HashJoinExec(lm, flt, [("lk", "rk")], jt), ["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("key_type", ["Int32", "Int64"])
@pytest.mark.parametrize("n_left,span", [(1, 1), (5, 5), (25, 65535), (25, 65536), (25, 65537), (1024, 65536), (1025, 5000), (700, 10 ** 7)])
@pytest.mark.parametrize("shape", ["sorted", "shuffled", "nulls", "duplicates"])
def test_dimension_table_build_in_one_launch(ctx, key_type, n_left, span, shape):
    """A build side of at most 1024 keys spanning at most 2^16 values is built by ONE launch (statistics, key set in LDS, packed
    map, permutation): both sides of the row limit and of the span limit, one key, negative keys, unsorted keys, NULL keys,
    duplicate keys (which send the join to the general table) — inner and left joins, with probe keys below, inside and above."""
    rng = np.random.default_rng(n_left * 7 + span % 1000 + len(shape))
    np_t = np.int64 if key_type == "Int64" else np.int32
    base = -40_000 if key_type == "Int32" else -(2 ** 40)
    if n_left == 1:
        lk = np.array([0])
    else:
        inner = rng.permutation(max(span - 2, 0))[: max(n_left - 2, 0)] + 1 if span > 2 else np.zeros(0, dtype=np.int64)
        lk = np.sort(np.concatenate([[0, span - 1], inner]))[:n_left] if span >= 2 else np.arange(n_left)
        lk[-1] = span - 1                                                            # the keys span exactly `span` values
    n_left = len(lk)
    lv = None
    if shape == "shuffled":
        lk = rng.permutation(lk)
    elif shape == "nulls":
        lv = rng.random(n_left) > 0.2
    elif shape == "duplicates" and n_left > 1:
        lk[n_left // 2] = lk[0]
    n_right = 5000
    rk = rng.integers(-50, span + 50, n_right)
    rk[:4] = [0, span - 1, -1, span]
    left = OrderedDict([("lk", OCol(key_type, (lk + base).astype(np_t), lv)), ("lx", OCol("Float64", rng.random(n_left)))])
    right = OrderedDict([("rk", OCol(key_type, (rk + base).astype(np_t))), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right)))])
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]])
    for jt in (ba.plan.INNER, ba.plan.LEFT):
        check(ba.HashJoinExec(lm, rm, [("lk", "rk")], jt), ["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("key_type", ["Int32", "Int64"])
@pytest.mark.parametrize("probe_order", ["ascending", "descending", "runs", "random"])
@pytest.mark.parametrize("semi", [False, True])
@pytest.mark.parametrize("build", ["sorted", "shuffled"])
def test_probe_map_reads_by_key_clustering(ctx, key_type, probe_order, semi, build):
    """The rank-map probe reads its map words through the scalar cache when every 64-row slot of a pass spans at most two
    neighbouring granules counted from its first live row (keys clustered like the probe order), and gathers otherwise — decided
    per pass, so one input mixes both.  Probe keys ascending (TPC-H lineitem by order key), descending (never narrow: the first
    row holds the largest key), ascending runs with jumps, random; keys below, at both ends of and beyond the window (a scalar
    read has no range check: such a slot must gather); `semi`: no build column is read, so the key-set words alone are probed
    (4 bytes per granule) and nothing is staged.  A date filter on the probe side drops rows inside the slots."""
    rng = np.random.default_rng(sum(map(ord, key_type + probe_order + build)) + int(semi))
    np_t = np.int64 if key_type == "Int64" else np.int32
    n_left, n_right, base = 40_000, 300_000, 1_000_003
    lk = np.sort(rng.permutation(4 * n_left)[:n_left]) + base                      # unique, sorted: rank = row
    lo, hi = int(lk[0]), int(lk[-1])
    if build == "shuffled":
        lk = rng.permutation(lk)                                                    # rank -> row through the permutation (PERM)
    if probe_order == "random":
        rk = rng.integers(lo - 500, hi + 500, n_right)
    else:
        rk = np.sort(rng.integers(lo - 500, hi + 500, n_right))                  # ~7 probe rows per 4 key values: a slot spans ~1 granule
        if probe_order == "descending":
            rk = rk[::-1].copy()
        elif probe_order == "runs":
            rk = np.concatenate([rk[i::7] for i in range(7)])                    # seven ascending runs: jumps in the middle of passes
    rk[:6] = [lo, hi, lo - 1, hi + 1, lo - 40, hi + 40]
    rk[-3:] = [hi, hi + 1, hi + 33]                                               # the last slot ends at and beyond the window's last granule
    # whole passes of NARROW slots that straddle the window's ends: eight rows per key value from 20 below the first key (offsets
    # that wrap to just under 2^32: out of range like a dropped row, and never a match) and up to 27 above the last
    edge = np.arange(1024) // 8
    rk = np.concatenate([lo - 20 + edge, rk, hi - 100 + edge, lo - 31 + edge])
    left = OrderedDict([("lk", OCol(key_type, lk.astype(np_t))), ("lx", OCol("Float64", rng.random(n_left)))])
    n_right = len(rk)
    right = OrderedDict([("rk", OCol(key_type, rk.astype(np_t))), ("ry", OCol("Int64", rng.integers(0, 10 ** 9, n_right))),
                         ("rd", OCol("Date32", rng.integers(9000, 10000, n_right).astype(np.int32)))])
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[right]])
    schema = {"rk": key_type, "ry": "Int64", "rd": "Date32"}
    flt = ba.FilterExec(E.coerce((col("rd") >= E.date32("1995-01-01")).and_(col("rd") < E.date32("1996-06-01")), schema), rm)
    for probe_side in (rm, flt):
        j = ba.HashJoinExec(lm, probe_side, [("lk", "rk")], ba.plan.INNER)
        if semi:
            check(ba.ProjectionExec([(col("rk"), "rk"), (col("ry"), "ry")], j), ["rk", "ry"])
        else:
            check(j, ["lk", "rk", "ry", "lx"])


@pytest.mark.parametrize("jt", JOIN_TYPES)
def test_parents_that_read_only_some_join_columns(ctx, jt):
    """ProjectionExec / HashAggregateExec above a join ask it for the columns they read only (HashJoinExec::execute_needed):
    no left column at all (a semi-join: no partners are staged), no right column, and an aggregate over one of each"""
    left, right = sides(700, 6000, "Int32", True, False, seed=31)
    lm, rm = helpers.memory_exec(ctx, [[left]]), helpers.memory_exec(ctx, [[helpers.slice_batch(right, 0, 2500)], [helpers.slice_batch(right, 2500, 6000)]])
    j = ba.HashJoinExec(lm, rm, [("lk", "rk")], jt)
    check(ba.ProjectionExec([(col("ry"), "ry"), (col("rd"), "rd")], j), ["ry", "rd"])
    check(ba.ProjectionExec([(col("ls"), "ls"), (col("lx"), "lx")], j), ["ls", "lx"])
    check(ba.ProjectionExec([(col("ry") + lit(1), "y1"), (col("lk"), "k")], j), ["k", "y1"])
    agg = ba.HashAggregateExec(ba.plan.PARTIAL, [(col("ls"), "ls")], [E.Sum(col("ry"), "s"), E.Count(lit(1, E.UINT8), "n")], j)
    fin = ba.HashAggregateExec(ba.plan.FINAL, [(col("ls"), "ls")], [E.Sum(col("ry"), "s"), E.Count(lit(1, E.UINT8), "n")], ba.MergeExec(agg))
    check(fin, ["ls"])


@pytest.mark.parametrize("sorted_build", [True, False])
@pytest.mark.parametrize("n_ranges", [2, 3])
def test_probe_side_filter_of_several_ranges(ctx, n_ranges, sorted_build):
    """the probe kernel compiled for JOIN_FILTER_MAX range columns (an AND of two or three Int32 / Date32 ranges under the join),
    over a build side in key order (rank = row) and one that is not (rank -> row through the permutation)"""
    rng = np.random.default_rng(31 + n_ranges)
    nl, nr = 700, 9000
    lk = rng.permutation(3000)[:nl].astype(np.int32)
    if sorted_build:
        lk = np.sort(lk)
    left = OrderedDict([("lk", OCol("Int32", lk)), ("lx", OCol("Float64", rng.random(nl)))])
    right = OrderedDict([("rk", OCol("Int32", rng.integers(-20, 3100, nr).astype(np.int32))), ("ra", OCol("Int32", rng.integers(0, 100, nr).astype(np.int32))),
                         ("rb", OCol("Date32", rng.integers(9000, 9400, nr).astype(np.int32))), ("rc", OCol("Int32", rng.integers(-50, 50, nr).astype(np.int32))),
                         ("ry", OCol("Int64", rng.integers(0, 10 ** 6, nr)))])
    schema = {"rk": "Int32", "ra": "Int32", "rb": "Date32", "rc": "Int32", "ry": "Int64"}
    pred = (col("ra") >= lit(10)).and_(col("ra") < lit(80)).and_(col("rb") > E.date32("1994-12-01"))
    if n_ranges == 3:
        pred = pred.and_(col("rc") <= lit(20))
    rm = helpers.memory_exec(ctx, [[helpers.slice_batch(right, 0, 4100), helpers.slice_batch(right, 4100, nr)]])
    plan = ba.HashJoinExec(helpers.memory_exec(ctx, [[left]]), ba.FilterExec(E.coerce(pred, schema), rm), [("lk", "rk")], ba.plan.INNER)
    check(plan, ["lk", "rk", "ry", "lx"])
