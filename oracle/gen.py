"""TEST INFRASTRUCTURE — ctypes front-end for oracle/tpch_gen.c (CPU twin of the HIP
synthetic TPC-H generator) and for the threaded Q1/Q6 stage-1 port in oracle_ops.c
(bench.py's cpu_baseline, kind "port").  See oracle/__init__.py."""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np

from .engine import OCol, lib

SEED = 0x7C4A_0BA1_1157_A001

# TPC-H cardinalities; lineitem counts are dbgen's for SF1 / SF100
_LINEITEM_ROWS = {1: 6_001_215, 100: 600_037_902}


def cardinalities(sf: float):
    n_orders = max(7, int(round(1_500_000 * sf)))
    return {
        "orders": n_orders,
        "lineitem": _LINEITEM_ROWS.get(sf, 4 * n_orders),
        "customer": max(3, int(round(150_000 * sf))),
        "supplier": max(1, int(round(10_000 * sf))),
        "part": max(1, int(round(200_000 * sf))),
    }


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def order_key_layout(keys, sparse_keys=False, key_base=0):
    """the order-key layouts of the HIP generator's options (ballista_amd/csrc/kernels_gen.hip::order_key), restated in numpy on the
    dense keys `order number + 1`: sparse = dbgen's layout (low 3 bits of the order number kept, the rest shifted up by two: 8 of
    every 32 values used, SF1000 keys reach 6 x 10^9); key_base is added to every key.  Only meaningful with Int64 keys when the
    result leaves the Int32 range."""
    if not sparse_keys and not key_base:
        return keys
    o = keys.astype(np.int64) - 1
    if sparse_keys:
        o = ((o >> 3) << 5) | (o & 7)
    return (o + 1 + int(key_base)).astype(keys.dtype)


def lineitem_arrays(sf=0.001, row0=0, n=None, seed=SEED, key64=False, dates=False, sparse_keys=False, key_base=0):
    """raw numpy arrays of lineitem rows [row0, row0+n)"""
    card = cardinalities(sf)
    if n is None:
        n = card["lineitem"] - row0
    a = OrderedDict()
    a["l_orderkey"] = np.empty(n, np.int64 if key64 else np.int32)
    a["l_suppkey"] = np.empty(n, np.int32)
    for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"):
        a[k] = np.empty(n, np.float64)
    a["l_shipdate"] = np.empty(n, np.int32)
    commit = np.empty(n, np.int32) if dates else None
    receipt = np.empty(n, np.int32) if dates else None
    a["l_returnflag.off"] = np.empty(n + 1, np.int32)
    a["l_returnflag.data"] = np.empty(n, np.uint8)
    a["l_linestatus.off"] = np.empty(n + 1, np.int32)
    a["l_linestatus.data"] = np.empty(n, np.uint8)
    lib().tpch_gen_lineitem(
        ctypes.c_uint64(seed), ctypes.c_uint64(row0), ctypes.c_uint64(n),
        ctypes.c_uint64(card["orders"]), ctypes.c_uint64(card["part"]), ctypes.c_uint64(card["supplier"]),
        None if key64 else _p(a["l_orderkey"], ctypes.c_int32),
        _p(a["l_orderkey"], ctypes.c_int64) if key64 else None,
        _p(a["l_suppkey"], ctypes.c_int32),
        _p(a["l_quantity"], ctypes.c_double), _p(a["l_extendedprice"], ctypes.c_double),
        _p(a["l_discount"], ctypes.c_double), _p(a["l_tax"], ctypes.c_double),
        _p(a["l_shipdate"], ctypes.c_int32), _p(commit, ctypes.c_int32), _p(receipt, ctypes.c_int32),
        _p(a["l_returnflag.off"], ctypes.c_int32), _p(a["l_returnflag.data"], ctypes.c_uint8),
        _p(a["l_linestatus.off"], ctypes.c_int32), _p(a["l_linestatus.data"], ctypes.c_uint8))
    if dates:
        a["l_commitdate"] = commit
        a["l_receiptdate"] = receipt
    a["l_orderkey"] = order_key_layout(a["l_orderkey"], sparse_keys, key_base)
    return a


def _utf8_from(off, data):
    b = data.tobytes()
    return [b[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]


def lineitem(sf=0.001, row0=0, n=None, seed=SEED, key64=False, sparse_keys=False, key_base=0):
    a = lineitem_arrays(sf, row0, n, seed, key64, sparse_keys=sparse_keys, key_base=key_base)
    out = OrderedDict()
    out["l_orderkey"] = OCol("Int64" if key64 else "Int32", a["l_orderkey"])
    out["l_suppkey"] = OCol("Int32", a["l_suppkey"])
    for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"):
        out[k] = OCol("Float64", a[k])
    out["l_returnflag"] = OCol("Utf8", _utf8_from(a["l_returnflag.off"], a["l_returnflag.data"]))
    out["l_linestatus"] = OCol("Utf8", _utf8_from(a["l_linestatus.off"], a["l_linestatus.data"]))
    out["l_shipdate"] = OCol("Date32", a["l_shipdate"])
    return out


def orders_arrays(sf=0.001, row0=0, n=None, seed=SEED, key64=False, sparse_keys=False, key_base=0):
    card = cardinalities(sf)
    if n is None:
        n = card["orders"] - row0
    a = OrderedDict()
    a["o_orderkey"] = np.empty(n, np.int64 if key64 else np.int32)
    a["o_custkey"] = np.empty(n, np.int32)
    a["o_orderdate"] = np.empty(n, np.int32)
    a["o_shippriority"] = np.empty(n, np.int32)
    lib().tpch_gen_orders(ctypes.c_uint64(seed), ctypes.c_uint64(row0), ctypes.c_uint64(n),
                          ctypes.c_uint64(card["customer"]),
                          None if key64 else _p(a["o_orderkey"], ctypes.c_int32),
                          _p(a["o_orderkey"], ctypes.c_int64) if key64 else None,
                          _p(a["o_custkey"], ctypes.c_int32), _p(a["o_orderdate"], ctypes.c_int32),
                          _p(a["o_shippriority"], ctypes.c_int32))
    a["o_orderkey"] = order_key_layout(a["o_orderkey"], sparse_keys, key_base)
    return a


def orders(sf=0.001, seed=SEED, key64=False, sparse_keys=False, key_base=0):
    a = orders_arrays(sf, seed=seed, key64=key64, sparse_keys=sparse_keys, key_base=key_base)
    return OrderedDict([
        ("o_orderkey", OCol("Int64" if key64 else "Int32", a["o_orderkey"])),
        ("o_custkey", OCol("Int32", a["o_custkey"])),
        ("o_orderdate", OCol("Date32", a["o_orderdate"])),
        ("o_shippriority", OCol("Int32", a["o_shippriority"]))])


def customer_arrays(sf=0.001, seed=SEED):
    n = cardinalities(sf)["customer"]
    a = OrderedDict()
    a["c_custkey"] = np.empty(n, np.int32)
    a["c_nationkey"] = np.empty(n, np.int32)
    a["c_mktsegment.off"] = np.empty(n + 1, np.int32)
    data = np.empty(10 * n, np.uint8)
    L = lib()
    L.tpch_gen_customer.restype = ctypes.c_uint64
    used = L.tpch_gen_customer(ctypes.c_uint64(seed), ctypes.c_uint64(n),
                               _p(a["c_custkey"], ctypes.c_int32), _p(a["c_nationkey"], ctypes.c_int32),
                               _p(a["c_mktsegment.off"], ctypes.c_int32), _p(data, ctypes.c_uint8))
    a["c_mktsegment.data"] = data[:used].copy()
    return a


def customer(sf=0.001, seed=SEED):
    a = customer_arrays(sf, seed)
    return OrderedDict([
        ("c_custkey", OCol("Int32", a["c_custkey"])),
        ("c_nationkey", OCol("Int32", a["c_nationkey"])),
        ("c_mktsegment", OCol("Utf8", _utf8_from(a["c_mktsegment.off"], a["c_mktsegment.data"])))])


def supplier_arrays(sf=0.001, seed=SEED):
    n = cardinalities(sf)["supplier"]
    a = OrderedDict([("s_suppkey", np.empty(n, np.int32)), ("s_nationkey", np.empty(n, np.int32))])
    lib().tpch_gen_supplier(ctypes.c_uint64(seed), ctypes.c_uint64(n),
                            _p(a["s_suppkey"], ctypes.c_int32), _p(a["s_nationkey"], ctypes.c_int32))
    return a


def supplier(sf=0.001, seed=SEED):
    a = supplier_arrays(sf, seed)
    return OrderedDict([("s_suppkey", OCol("Int32", a["s_suppkey"])),
                        ("s_nationkey", OCol("Int32", a["s_nationkey"]))])


# the 25 nations / 5 regions of the TPC-H spec, in the format of the reference's fixtures
# (rust/scheduler/testdata/nation/nation.tbl, region/region.tbl)
NATIONS = [("ALGERIA", 0), ("ARGENTINA", 1), ("BRAZIL", 1), ("CANADA", 1), ("EGYPT", 4), ("ETHIOPIA", 0),
           ("FRANCE", 3), ("GERMANY", 3), ("INDIA", 2), ("INDONESIA", 2), ("IRAN", 4), ("IRAQ", 4),
           ("JAPAN", 2), ("JORDAN", 4), ("KENYA", 0), ("MOROCCO", 0), ("MOZAMBIQUE", 0), ("PERU", 1),
           ("CHINA", 2), ("ROMANIA", 3), ("SAUDI ARABIA", 4), ("VIETNAM", 2), ("RUSSIA", 3),
           ("UNITED KINGDOM", 3), ("UNITED STATES", 1)]
REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]


def nation():
    return OrderedDict([("n_nationkey", OCol("Int32", np.arange(25))),
                        ("n_name", OCol("Utf8", [n for n, _ in NATIONS])),
                        ("n_regionkey", OCol("Int32", [r for _, r in NATIONS]))])


def region():
    return OrderedDict([("r_regionkey", OCol("Int32", np.arange(5))), ("r_name", OCol("Utf8", REGIONS))])


# ---- threaded stage-1 ports (cpu_baseline) --------------------------------------------

def q1_partial_port(a, n_partitions, n_threads, date_lit=10471):
    """a = lineitem_arrays(...).  Returns (keys u16[P*4], state f64[P*4*8], count u64[P*4])."""
    n = len(a["l_quantity"])
    keys = np.zeros(4 * n_partitions, np.uint16)
    state = np.zeros(32 * n_partitions, np.float64)
    count = np.zeros(4 * n_partitions, np.uint64)
    lib().oracle_q1_partial(
        _p(a["l_quantity"], ctypes.c_double), _p(a["l_extendedprice"], ctypes.c_double),
        _p(a["l_discount"], ctypes.c_double), _p(a["l_tax"], ctypes.c_double),
        _p(a["l_shipdate"], ctypes.c_int32),
        _p(a["l_returnflag.off"], ctypes.c_int32), _p(a["l_returnflag.data"], ctypes.c_uint8),
        _p(a["l_linestatus.off"], ctypes.c_int32), _p(a["l_linestatus.data"], ctypes.c_uint8),
        ctypes.c_int64(n), ctypes.c_int32(date_lit), ctypes.c_int32(n_partitions), ctypes.c_int32(n_threads),
        _p(keys, ctypes.c_uint16), _p(state, ctypes.c_double), _p(count, ctypes.c_uint64))
    return keys, state, count


def q1_final_from_port(keys, state, count):
    """merge the port's per-partition states like HashAggregateExec(Final) and evaluate.
    Returns {(flag, status): dict(sum_qty, ..., count_order)}"""
    groups = OrderedDict()
    for slot in range(len(keys)):
        k = int(keys[slot])
        if k == 0 or count[slot] == 0:
            continue
        key = (chr(k & 0xFF), chr(k >> 8))
        s = state[slot * 8: slot * 8 + 8]
        g = groups.get(key)
        if g is None:
            groups[key] = [s.copy(), int(count[slot])]
        else:
            g[0] = g[0] + s
            g[1] += int(count[slot])
    out = OrderedDict()
    for key, (s, c) in groups.items():
        out[key] = dict(sum_qty=s[0], sum_base_price=s[1], sum_disc_price=s[2], sum_charge=s[3],
                        avg_qty=s[4] / float(c), avg_price=s[5] / float(c), avg_disc=s[6] / float(c),
                        count_order=c)
    return out


def q6_partial_port(a, n_partitions, n_threads):
    n = len(a["l_quantity"])
    s = np.zeros(n_partitions, np.float64)
    c = np.zeros(n_partitions, np.uint64)
    lib().oracle_q6_partial(
        _p(a["l_quantity"], ctypes.c_double), _p(a["l_extendedprice"], ctypes.c_double),
        _p(a["l_discount"], ctypes.c_double), _p(a["l_shipdate"], ctypes.c_int32),
        ctypes.c_int64(n), ctypes.c_int32(8766), ctypes.c_int32(9131),
        ctypes.c_double(0.06 - 0.01), ctypes.c_double(0.06 + 0.01), ctypes.c_double(24.0),
        ctypes.c_int32(n_partitions), ctypes.c_int32(n_threads),
        _p(s, ctypes.c_double), _p(c, ctypes.c_uint64))
    return s, c


class JoinQueryPort:
    """bench.py's cpu_baseline for Q3 / Q5: lineitem rows [0, sample_rows) with the matching prefix of orders (the generator
    gives every 7 orders 28 lines) and the whole small tables (`dims` = ballista_amd.tpch.dimension_arrays(sf)), through
    oracle_ops.c::oracle_q3_join_port / oracle_q5_join_port.  Everything, both hash-join builds included, runs inside run()."""

    SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"]

    def __init__(self, query, sf, sample_rows, dims, key64=False):
        self.query, self.key64 = query, key64
        n_ord = min(cardinalities(sf)["orders"], (sample_rows + 27) // 28 * 7 + 7)
        self.li = lineitem_arrays(sf, 0, sample_rows, key64=key64)
        self.od = orders_arrays(sf, 0, n_ord, key64=key64)
        c = dims["customer"]
        self.c_custkey, self.c_nationkey = np.ascontiguousarray(c["c_custkey"]), np.ascontiguousarray(c["c_nationkey"])
        seg = [s.encode() for s in self.SEGMENTS]
        lens = np.array([len(s) for s in seg], np.int32)[c["c_mktsegment"]]
        self.seg_off = np.zeros(len(lens) + 1, np.int32)
        np.cumsum(lens, out=self.seg_off[1:])
        table = np.frombuffer(b"".join(s.ljust(10) for s in seg), np.uint8).reshape(5, 10)
        mask = np.arange(10)[None, :] < lens[:, None]
        self.seg_data = np.ascontiguousarray(table[c["c_mktsegment"]][mask])
        s = dims["supplier"]
        self.s_suppkey, self.s_nationkey = np.ascontiguousarray(s["s_suppkey"]), np.ascontiguousarray(s["s_nationkey"])
        self.nation_region = np.array([r for _, r in NATIONS], np.int32)

    def run(self, n_partitions, n_threads):
        L = lib()
        kt = ctypes.c_void_p
        li, od = self.li, self.od
        if self.query == "q3":
            L.oracle_q3_join_port.restype = ctypes.c_int64
            total = ctypes.c_double()
            n = L.oracle_q3_join_port(
                _p(self.c_custkey, ctypes.c_int32), _p(self.seg_off, ctypes.c_int32), _p(self.seg_data, ctypes.c_uint8), ctypes.c_int64(len(self.c_custkey)),
                kt(od["o_orderkey"].ctypes.data), _p(od["o_custkey"], ctypes.c_int32), _p(od["o_orderdate"], ctypes.c_int32),
                _p(od["o_shippriority"], ctypes.c_int32), ctypes.c_int64(len(od["o_custkey"])),
                kt(li["l_orderkey"].ctypes.data), _p(li["l_extendedprice"], ctypes.c_double), _p(li["l_discount"], ctypes.c_double),
                _p(li["l_shipdate"], ctypes.c_int32), ctypes.c_int64(len(li["l_shipdate"])), ctypes.c_int32(1 if self.key64 else 0),
                ctypes.c_int32(9204), ctypes.c_int32(n_partitions), ctypes.c_int32(n_threads), ctypes.byref(total))
            return n, total.value
        rev = np.zeros(25, np.float64)
        L.oracle_q5_join_port(
            _p(self.nation_region, ctypes.c_int32), ctypes.c_int32(2),
            _p(self.c_custkey, ctypes.c_int32), _p(self.c_nationkey, ctypes.c_int32), ctypes.c_int64(len(self.c_custkey)),
            _p(self.s_suppkey, ctypes.c_int32), _p(self.s_nationkey, ctypes.c_int32), ctypes.c_int64(len(self.s_suppkey)),
            kt(od["o_orderkey"].ctypes.data), _p(od["o_custkey"], ctypes.c_int32), _p(od["o_orderdate"], ctypes.c_int32), ctypes.c_int64(len(od["o_custkey"])),
            kt(li["l_orderkey"].ctypes.data), _p(li["l_suppkey"], ctypes.c_int32), _p(li["l_extendedprice"], ctypes.c_double),
            _p(li["l_discount"], ctypes.c_double), ctypes.c_int64(len(li["l_suppkey"])), ctypes.c_int32(1 if self.key64 else 0),
            ctypes.c_int32(8766), ctypes.c_int32(9131), ctypes.c_int32(n_partitions), ctypes.c_int32(n_threads), _p(rev, ctypes.c_double))
        return rev
