"""RepartitionExec(Hash) on fixed-width NULL-free columns by one integer key takes the scatter pass
(kernels_sort.hip::partition_scatter: histogram + scan + one scatter of every column); anything else the
sort-by-partition-id pass.  Both must give the oracle's partitions: partition id = row hash % n (DESIGN.md §6,
reference operator rust/core/src/serde/physical_plan/from_proto.rs:133-147), input order kept inside a partition."""
from collections import OrderedDict

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd.expr import col
from oracle import engine as og
from oracle.engine import OCol

import helpers

pytestmark = pytest.mark.gpu


def table(n, key_type, seed, with_string=False, nulls=False):
    rng = np.random.default_rng(seed)
    np_t = np.int64 if key_type in ("Int64",) else np.int32
    lo, hi = (-10 ** 12, 10 ** 12) if key_type == "Int64" else (-2 ** 31, 2 ** 31 - 1)
    t = OrderedDict([("k", OCol(key_type, rng.integers(lo, hi, n).astype(np_t), (rng.random(n) > 0.1) if nulls else None)),
                     ("a", OCol("Float64", rng.random(n))), ("b", OCol("Int32", rng.integers(0, 1000, n).astype(np.int32))),
                     ("c", OCol("Int64", rng.integers(0, 10 ** 15, n)))])
    if with_string:
        t["s"] = OCol("Utf8", [f"v{i % 17}" for i in range(n)])
    return t


def check(ctx, t, nparts):
    dev = helpers.to_device(ctx, t)
    got = ba.plan.hash_partition(dev, [col("k")], nparts)
    want = og.repartition_hash(t, [col("k")], nparts)
    assert len(got) == nparts
    for g, w in zip(got, want):
        helpers.assert_rows_equal(helpers.from_device(g), w, ordered=True)


@pytest.mark.parametrize("key_type", ["Int32", "Date32", "Int64"])
@pytest.mark.parametrize("nparts", [1, 2, 8, 256])
def test_scatter_pass(ctx, key_type, nparts):
    check(ctx, table(9000, key_type, 3), nparts)


@pytest.mark.parametrize("n", [1, 255, 256, 4095, 4096, 4097, 8193])
def test_scatter_pass_chunk_boundaries(ctx, n):
    check(ctx, table(n, "Int32", n), 8)


def test_other_inputs_take_the_sort_pass(ctx):
    check(ctx, table(5000, "Int32", 5, with_string=True), 4)       # a Utf8 column
    check(ctx, table(5000, "Int32", 6, nulls=True), 4)             # NULL keys


def test_partitions_feed_downstream_operators(ctx):
    """a partition's columns are slices of the scattered columns: they must work as operator inputs"""
    from ballista_amd import expr as E
    t = table(20000, "Int32", 9)
    parts = ba.plan.hash_partition(helpers.to_device(ctx, t), [col("k")], 3)
    want_parts = og.repartition_hash(t, [col("k")], 3)
    for p, w in zip(parts, want_parts):
        src = ba.MemoryExec([[p]], ctx)
        src._oracle_partitions = [[w]]
        plan = ba.HashAggregateExec(ba.plan.PARTIAL, [], [E.Sum(col("a"), "s"), E.Count(E.lit(1, E.UINT8), "n")], src)
        got = helpers.concat(helpers.collect_product(plan))
        from oracle import plan_eval
        helpers.assert_rows_equal(got, plan_eval.collect(plan), ordered=False, float_rtol=1e-9)
        back = p.to_pyarrow()
        assert back.num_rows == og.batch_len(w)
