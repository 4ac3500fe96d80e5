"""The benchmark's input generator exists twice — oracle/tpch_gen.c (CPU) and ballista_amd/csrc/kernels_gen.hip
(HIP) — written independently from the spec in DESIGN.md §6.  They must agree bit for bit on every column,
at any row offset (the N-rank bench generates rows [rank*R, (rank+1)*R) on each GPU)."""
import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import tpch
from oracle import gen

import helpers

pytestmark = pytest.mark.gpu


def _cmp(dev_batch, cpu_batch):
    got = helpers.from_device(dev_batch)
    assert list(got.keys())[:len(cpu_batch)] == list(cpu_batch.keys()) or set(cpu_batch.keys()) <= set(got.keys())
    for name, want in cpu_batch.items():
        g = got[name]
        assert g.dtype == want.dtype, (name, g.dtype, want.dtype)
        if want.dtype == "Utf8":
            assert list(g.values) == list(want.values), name
        elif want.dtype == "Float64":
            # bit patterns, not values
            assert np.array_equal(np.asarray(g.values, dtype=np.float64).view(np.uint64),
                                  np.asarray(want.values, dtype=np.float64).view(np.uint64)), name
        else:
            assert np.array_equal(np.asarray(g.values), np.asarray(want.values)), name
        assert g.valid is None or bool(np.all(g.valid))


@pytest.mark.parametrize("sf,row0,n", [(0.001, 0, 6001), (0.01, 0, 60000), (0.01, 12345, 20011), (100.0, 599_000_000, 70001),
                                       (100.0, 3_000_000_017, 4099)])
@pytest.mark.parametrize("key64", [False, True])
def test_lineitem_generator_bit_identical(ctx, sf, row0, n, key64):
    card = gen.cardinalities(sf)
    if row0 + n > card["lineitem"] and sf < 100:
        n = card["lineitem"] - row0
    dev = ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, row0, n, key64=key64)
    cpu = gen.lineitem(sf, row0, n, key64=key64)
    assert dev.num_rows == n
    _cmp(dev, cpu)


def test_lineitem_generator_dates(ctx):
    from collections import OrderedDict
    from oracle.engine import OCol
    sf, row0, n = 0.01, 777, 30000
    dev = ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, row0, n, with_dates=True)
    a = gen.lineitem_arrays(sf, row0, n, dates=True)
    cpu = OrderedDict([("l_shipdate", OCol("Date32", a["l_shipdate"])), ("l_commitdate", OCol("Date32", a["l_commitdate"])),
                       ("l_receiptdate", OCol("Date32", a["l_receiptdate"]))])
    _cmp(dev, cpu)
    ship, rec = a["l_shipdate"], a["l_receiptdate"]
    assert np.all(rec > ship) and np.all(rec - ship <= 30)


@pytest.mark.parametrize("key64", [False, True])
def test_orders_generator_bit_identical(ctx, key64):
    from collections import OrderedDict
    from oracle.engine import OCol
    sf = 0.01
    n = gen.cardinalities(sf)["orders"]
    for row0, cnt in ((0, n), (4321, 5000)):
        dev = ba.plan.tpch_orders(ctx, sf, tpch.SEED, row0, cnt, key64=key64)
        a = gen.orders_arrays(sf, row0, cnt, key64=key64)
        cpu = OrderedDict([("o_orderkey", OCol("Int64" if key64 else "Int32", a["o_orderkey"])), ("o_custkey", OCol("Int32", a["o_custkey"])),
                           ("o_orderdate", OCol("Date32", a["o_orderdate"])), ("o_shippriority", OCol("Int32", a["o_shippriority"]))])
        _cmp(dev, cpu)


@pytest.mark.parametrize("sparse,base", [(True, 0), (False, (1 << 32) + 17), (True, 5_000_000_000)])
def test_generator_key_layouts_and_column_subsets(ctx, sparse, base):
    """the order-key layouts config #5 needs (dbgen's sparse keys, keys beyond 2^32) and the column subset option: the HIP
    generator against the oracle's numpy restatement of the layout"""
    from collections import OrderedDict
    from oracle.engine import OCol
    sf, row0, n = 0.01, 1234, 40001
    dev = ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, row0, n, key64=True, sparse_keys=sparse, key_base=base,
                                columns=["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"])
    assert [f[0] for f in dev.schema3()] == ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]
    a = gen.lineitem_arrays(sf, row0, n, key64=True, sparse_keys=sparse, key_base=base)
    cpu = OrderedDict([("l_orderkey", OCol("Int64", a["l_orderkey"])), ("l_suppkey", OCol("Int32", a["l_suppkey"])),
                       ("l_extendedprice", OCol("Float64", a["l_extendedprice"])), ("l_discount", OCol("Float64", a["l_discount"]))])
    _cmp(dev, cpu)
    assert int(a["l_orderkey"].max()) > (1 << 32) or sparse
    if sparse and not base:
        assert np.all(((a["l_orderkey"] - 1) & 31) < 8)                  # 8 of every 32 key values are used
    dev = ba.plan.tpch_orders(ctx, sf, tpch.SEED, 100, 9000, key64=True, sparse_keys=sparse, key_base=base, columns=["o_orderkey", "o_custkey", "o_orderdate"])
    o = gen.orders_arrays(sf, 100, 9000, key64=True, sparse_keys=sparse, key_base=base)
    _cmp(dev, OrderedDict([("o_orderkey", OCol("Int64", o["o_orderkey"])), ("o_custkey", OCol("Int32", o["o_custkey"])),
                           ("o_orderdate", OCol("Date32", o["o_orderdate"]))]))
    assert np.all(np.diff(o["o_orderkey"]) > 0)                          # the layouts keep the table sorted by key


def test_generator_pin(ctx):
    """the committed known-answer rows of the generator (tests/golden/gen_pin.json) hold for the HIP generator too"""
    import json
    import os
    pin = json.load(open(os.path.join(helpers.GOLDEN, "gen_pin.json")))
    sf = pin["sf"]
    assert pin["seed"] == tpch.SEED
    n = pin["lineitem"]["l_quantity"]["n"]
    dev = helpers.from_device(ba.plan.tpch_lineitem(ctx, sf, tpch.SEED, 0, n, with_dates=True))
    for name in ("l_orderkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_shipdate", "l_commitdate",
                 "l_receiptdate"):
        want = pin["lineitem"][name]
        v = np.asarray(dev[name].values)
        assert len(v) == want["n"]
        assert [x.item() for x in v[:8]] == want["first"], name
        v64 = v.view(np.uint64) if v.dtype == np.float64 else v.astype(np.uint64)
        assert int(np.bitwise_xor.reduce(v64)) == want["xor"], name
    flags = "".join(dev["l_returnflag"].values).encode()
    assert list(flags[:8]) == pin["lineitem"]["l_returnflag.data"]["first"]
