"""`.tbl` text -> device columns (ballista_amd/csrc/kernels_tbl.hip, bhip_batch_from_tbl): the scan leaf the reference
builds as CsvExec(delimiter '|', no header, explicit schema) — rust/benchmarks/tpch/src/main.rs:129-150, schemas :267-360.

Checked against (i) the reference's own fixture files (rust/scheduler/testdata/*, copied as data under tests/golden/tbl/),
parsed here with Python's int / float / date — float() is the correctly rounded decimal conversion, the same value
Rust's str::parse::<f64> gives; (ii) generated text with awkward values; (iii) Q1 over the scanned fixture == golden."""
import datetime
import json
import os

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd._lib import ExecutionError, NotImplementedOnGpu

import helpers

pytestmark = pytest.mark.gpu
TBL = os.path.join(helpers.GOLDEN, "tbl")

LINEITEM = [("l_orderkey", E.INT32), ("l_partkey", E.INT32), ("l_suppkey", E.INT32), ("l_linenumber", E.INT32),
            ("l_quantity", E.FLOAT64), ("l_extendedprice", E.FLOAT64), ("l_discount", E.FLOAT64), ("l_tax", E.FLOAT64),
            ("l_returnflag", E.UTF8), ("l_linestatus", E.UTF8), ("l_shipdate", E.DATE32), ("l_commitdate", E.DATE32),
            ("l_receiptdate", E.DATE32), ("l_shipinstruct", E.UTF8), ("l_shipmode", E.UTF8), ("l_comment", E.UTF8)]
ORDERS = [("o_orderkey", E.INT32), ("o_custkey", E.INT32), ("o_orderstatus", E.UTF8), ("o_totalprice", E.FLOAT64),
          ("o_orderdate", E.DATE32), ("o_orderpriority", E.UTF8), ("o_clerk", E.UTF8), ("o_shippriority", E.INT32), ("o_comment", E.UTF8)]
NATION = [("n_nationkey", E.INT32), ("n_name", E.UTF8), ("n_regionkey", E.INT32), ("n_comment", E.UTF8)]


def days(s):
    return (datetime.date.fromisoformat(s) - datetime.date(1970, 1, 1)).days


def py_parse(text, schema, columns=None):
    names = [n for n, _ in schema]
    rows = [ln.rstrip("\r").split("|") for ln in text.decode().split("\n") if ln != ""]
    out = {}
    for name in (columns or names):
        i = names.index(name)
        t = schema[i][1]
        vals = [r[i] for r in rows]
        if t in (E.INT32, E.INT64):
            out[name] = [int(v) for v in vals]
        elif t == E.FLOAT64:
            out[name] = [float(v) for v in vals]
        elif t == E.DATE32:
            out[name] = [days(v) for v in vals]
        else:
            out[name] = vals
    return out


def check(ctx, text, schema, columns=None):
    rb = ba.RecordBatch.from_tbl(ctx, text, schema, columns)
    want = py_parse(text, schema, columns)
    assert [rb.column_info(i)[0] for i in range(rb.num_columns)] == list(want)
    for i, (name, w) in enumerate(want.items()):
        dtype, vals, valid = rb.column(i)
        assert valid is None
        if dtype == E.UTF8:
            assert list(vals) == w, name
        elif dtype == E.FLOAT64:
            assert np.array_equal(np.asarray(vals, np.float64).view(np.uint64), np.asarray(w, np.float64).view(np.uint64)), name   # bit-exact
        else:
            assert [int(v) for v in vals] == w, name
    assert rb.num_rows == len(next(iter(want.values()))) if want else True
    return rb


@pytest.mark.parametrize("name,schema", [("lineitem_partition0", LINEITEM), ("lineitem_partition1", LINEITEM), ("orders_orders", ORDERS),
                                         ("nation_nation", NATION)])
def test_reference_fixture_files(ctx, name, schema):
    text = open(os.path.join(TBL, name + ".tbl"), "rb").read()
    check(ctx, text, schema)
    check(ctx, text, schema, [schema[-1][0], schema[0][0]])              # projection, reordered
    check(ctx, text.rstrip(b"\n"), schema, [schema[1][0]])               # no newline after the last line


def test_generated_text_with_awkward_values(ctx):
    rng = np.random.default_rng(0)
    lines = []
    for i in range(50_000):
        k = int(rng.integers(-2 ** 31, 2 ** 31 - 1))
        big = int(rng.integers(-2 ** 62, 2 ** 62))
        cents = int(rng.integers(0, 10 ** 9))
        dec = f"{'-' if rng.random() < 0.3 else ''}{cents // 100}.{cents % 100:02d}"
        frac = ["0.1", "0.07", "123456.789012345", "0", "-0.00", "9007199254740991", "1e0"][int(rng.integers(0, 6))]
        d = datetime.date(1970, 1, 1) + datetime.timedelta(days=int(rng.integers(-20000, 40000)))
        s = "".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(0, 40))))
        lines.append(f"{k}|{big}|{dec}|{frac}|{d.isoformat()}|{s}|tail|")
    text = ("\n".join(lines) + "\n").encode()
    schema = [("a", E.INT32), ("b", E.INT64), ("c", E.FLOAT64), ("d", E.FLOAT64), ("e", E.DATE32), ("f", E.UTF8), ("g", E.UTF8)]
    check(ctx, text, schema)
    check(ctx, text.replace(b"\n", b"\r\n"), schema, ["f", "c", "e"])    # CRLF line ends


def test_long_lines_walk_the_text_in_hbm(ctx):
    """256 lines of ~300 bytes do not fit the 48 KB LDS stage of a workgroup: that tile is walked in HBM instead;
    short and long tiles alternate here"""
    rng = np.random.default_rng(5)
    lines = []
    for i in range(3000):
        long_tile = (i // 256) % 2 == 1
        s = "".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(250, 400)) if long_tile else int(rng.integers(0, 20))))
        lines.append(f"{i}|{s}|{i * 0.25:.2f}|{s[::-1][:7]}|")
    text = ("\n".join(lines) + "\n").encode()
    schema = [("a", E.INT32), ("s", E.UTF8), ("x", E.FLOAT64), ("t", E.UTF8)]
    check(ctx, text, schema)
    check(ctx, text, schema, ["x", "t"])


def test_empty_and_single_line(ctx):
    schema = [("a", E.INT32), ("s", E.UTF8)]
    rb = ba.RecordBatch.from_tbl(ctx, b"", schema)
    assert rb.num_rows == 0 and rb.num_columns == 2
    check(ctx, b"7|x|", schema)
    check(ctx, b"7||\n", schema)                                          # empty string field


def test_malformed_text_is_reported_not_guessed(ctx):
    schema = [("a", E.INT32), ("x", E.FLOAT64), ("d", E.DATE32)]
    ba.RecordBatch.from_tbl(ctx, b"1|2.5|1996-01-02|\n", schema)
    for bad in (b"1|2.5|\n",                    # a field is missing
                b"1|abc|1996-01-02|\n",         # not a number
                b"1|2.5|1996-13-02|\n",         # not a date
                b"99999999999|2.5|1996-01-02|\n",   # out of Int32 range
                b"1|2.5|1996-01-02|\n\n2|1.0|1996-01-03|\n"):   # blank line
        with pytest.raises(ExecutionError):
            ba.RecordBatch.from_tbl(ctx, bad, schema)
    with pytest.raises(NotImplementedOnGpu):
        ba.RecordBatch.from_tbl(ctx, b"1|0.12345678901234567890|1996-01-02|\n", schema)     # cannot be converted exactly here


def test_q1_over_the_scanned_fixture_equals_golden(ctx):
    cols = list(tpch.LINEITEM_SCHEMA)
    parts = [[ba.RecordBatch.from_tbl(ctx, os.path.join(TBL, f"lineitem_partition{p}.tbl"), LINEITEM, cols)] for p in range(2)]
    got = tpch.q1_plan(ba.MemoryExec(parts, ctx)).collect()
    rows = json.load(open(os.path.join(helpers.GOLDEN, "q1_fixture.json")))["rows"]
    d = got[0].to_pydict()
    assert list(zip(d["l_returnflag"], d["l_linestatus"])) == [(r["l_returnflag"], r["l_linestatus"]) for r in rows]
    for i, r in enumerate(rows):
        assert d["count_order"][i] == r["count_order"]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert abs(d[k][i] - r[k]) <= 1e-9 * abs(r[k]), k
