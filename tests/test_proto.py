"""bhip_plan_from_proto: the executor's wire plan (protobuf PhysicalPlanNode, rust/core/proto/ballista.proto:294-422) decoded by
the library's own proto3 reader (ballista_amd/csrc/host/proto.cpp) — the C image of
rust/core/src/serde/physical_plan/from_proto.rs:58-364.

The bytes come from tests/proto_encode.py (an encoder written from the .proto's field numbers).  Without a GPU (`-m "not gpu"`)
plans are decoded with ctx = NULL and inspected: operator names, the schema of every node against tests/plan_nodes.py's
independent statement of the schema rules, expression renderings, error behaviour.  With a GPU the decoded plans run: their
rendering equals the ctypes-built plan's and their results equal the committed goldens."""
import json
import os

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit
from oracle import gen

import helpers
import plan_nodes as N
import proto_encode as pe


@pytest.fixture
def nodes_tpch(monkeypatch):
    """tpch's plan builders over the GPU-free plan descriptions"""
    monkeypatch.setattr(tpch, "P", N)
    return tpch


def leaf(name, batch):
    m = N.MemoryExec([[batch]])
    m.name = "mem://" + name
    return m


def tables(sf=0.001):
    return dict(lineitem=gen.lineitem(sf), orders=gen.orders(sf), customer=gen.customer(sf), supplier=gen.supplier(sf),
                nation=gen.nation(), region=gen.region())


def build(nodes_tpch, query, t):
    L = {k: leaf(k, v) for k, v in t.items()}
    if query == "q1":
        return nodes_tpch.q1_plan(L["lineitem"])
    if query == "q6":
        return nodes_tpch.q6_plan(L["lineitem"])
    if query == "q3":
        return nodes_tpch.q3_plan(L["customer"], L["orders"], L["lineitem"])
    return nodes_tpch.q5_plan(L["customer"], L["orders"], L["lineitem"], L["supplier"], L["nation"], L["region"])


NAME_OF = {"MemoryExec": "CsvExec", "GlobalLimitExec": "GlobalLimitExec", "LocalLimitExec": "LocalLimitExec"}


def same_tree(decoded, want):
    """operator names and the schema of every node"""
    assert decoded.as_any() == NAME_OF.get(type(want).__name__, type(want).__name__)
    got = [(n, t) for n, t, _ in decoded.schema()]
    assert got == [(n, t) for n, t, _ in want.schema()], (decoded.as_any(), got, want.schema())
    kids = decoded.children()
    assert len(kids) == len(want.children())
    for k, w in zip(kids, want.children()):
        same_tree(k, w)


@pytest.mark.parametrize("query", ["q1", "q6", "q3", "q5"])
def test_tpch_plans_decode_to_the_same_tree(nodes_tpch, query):
    want = build(nodes_tpch, query, tables())
    data = pe.plan(want)
    got = ba.ExecutionPlan.from_proto(None, data)
    same_tree(got, want)
    text = got.display()
    assert "CsvExec: path=mem://lineitem" in text
    if query == "q3":
        assert "HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(o_orderkey, l_orderkey)]" in text
        assert "SortExec: [revenue DESC NULLS FIRST, o_orderdate ASC NULLS FIRST]" in text
    # a plan without a device context describes itself but does not run
    with pytest.raises(ba.ExecutionError, match="unresolved leaves"):
        got.execute(0)


def expr_text(e):
    import ctypes as C
    from ballista_amd import _lib as L
    data = pe.expr(e)
    buf = C.create_string_buffer(4096)
    L.check(L.lib().bhip_expr_from_proto_display(data, len(data), buf, len(buf)))
    return buf.value.decode()


def test_every_expression_kind_of_the_serde():
    """LogicalExprNode kinds of ballista.proto:14-45 as to_proto.rs:380-511 writes them"""
    x, s = col("x"), col("s")
    cases = [
        (x, "x"),
        (lit(7, E.INT32), "Int32(7)"), (lit(-3), "Int64(-3)"), (lit(2.5), "2.5"), (lit(True), "true"), (lit("it's"), "'it's'"),
        (lit(200, E.UINT8), "UInt8(200)"), (lit(2 ** 63 + 5, E.UINT64), f"UInt64({2 ** 63 + 5 - 2 ** 64})"),
        (E.date32("1998-09-02"), "Date32(10471)"), (E.Literal(None, E.FLOAT64), "NULL:Float64"), (E.Literal(None, E.UTF8), "NULL:Utf8"),
        ((x + lit(1)) * (x - lit(2)) / x, "(((x Plus Int64(1)) Multiply (x Minus Int64(2))) Divide x)"),
        ((x < lit(1)).and_(x >= lit(0)).or_(x.eq(lit(5))).and_(x.ne(lit(6))).and_(x <= lit(9)).and_(x > lit(-1)),
         "((((((x Lt Int64(1)) And (x GtEq Int64(0))) Or (x Eq Int64(5))) And (x NotEq Int64(6))) And (x LtEq Int64(9))) And (x Gt Int64(-1)))"),
        (E.BinaryExpr(s, "Like", lit("a%")), "(s Like 'a%')"), (E.BinaryExpr(s, "NotLike", lit("_b")), "(s NotLike '_b')"),
        (E.CastExpr(x, E.FLOAT64), "CAST(x AS Float64)"), (E.CastExpr(x, E.DATE32), "CAST(x AS Date32)"),
        (E.NotExpr(x.eq(lit(1))), "NOT (x Eq Int64(1))"), (E.IsNullExpr(x), "x IS NULL"), (E.IsNotNullExpr(s), "s IS NOT NULL"),
        (E.NegativeExpr(x), "(- x)"),
        (E.InListExpr(s, [lit("MAIL"), lit("SHIP")]), "s IN ('MAIL', 'SHIP')"), (E.InListExpr(x, [lit(1)], negated=True), "x NOT IN (Int64(1))"),
        (E.CaseExpr(None, [(x.eq(lit(1)), lit(10)), (x.eq(lit(2)), lit(20))], lit(0)),
         "CASE WHEN (x Eq Int64(1)) THEN Int64(10) WHEN (x Eq Int64(2)) THEN Int64(20) ELSE Int64(0) END"),
        (E.CaseExpr(x, [(lit(1), lit("a"))], None), "CASE x WHEN Int64(1) THEN 'a' END"),
        (E.ScalarFunctionExpr("sqrt", [x]), "sqrt(x)"), (E.ScalarFunctionExpr("abs", [x]), "abs(x)"), (E.ScalarFunctionExpr("log10", [x]), "log10(x)"),
        (pe.Between(x, lit(1), lit(5)), "((x GtEq Int64(1)) And (x LtEq Int64(5)))"),
        (pe.Between(x, lit(1), lit(5), negated=True), "NOT ((x GtEq Int64(1)) And (x LtEq Int64(5)))"),
        (pe.Alias(x + lit(1), "y"), "(x Plus Int64(1))"),
    ]
    for e, want in cases:
        assert expr_text(e) == want, (e, expr_text(e), want)
    # Float64 bit patterns survive (0.06 - 0.01 is not 0.05)
    assert expr_text(lit(0.06 - 0.01)) == "0.049999999999999996"


def test_coercion_matches_datafusions_planner(nodes_tpch):
    """compile_expr plans every expression against its input schema: numeric literals take the column's type, mixed columns are
    cast to the wider type (the host mirror's `coerce` is the independent statement of the same rules)"""
    from oracle.engine import OCol
    from collections import OrderedDict
    b = OrderedDict([("i", OCol("Int32", np.arange(3, dtype=np.int32))), ("f", OCol("Float64", np.ones(3))), ("d", OCol("Date32", np.arange(3, dtype=np.int32))),
                     ("l", OCol("Int64", np.arange(3, dtype=np.int64)))])
    m = leaf("t", b)
    raw = (col("i") < lit(5)).and_(col("f") * (lit(1) - col("f")) > col("i")).and_(col("d") <= lit(10471, E.INT32)).and_(col("l").eq(col("i")))
    plan = N.FilterExec(raw, m)                                   # UNcoerced on the wire
    got = ba.ExecutionPlan.from_proto(None, pe.plan(plan)).display().splitlines()[0]
    # Int32 column vs Int64 literal: the COLUMN is cast up (DataFusion's numerical_coercion), a numeric literal beside a wider
    # column is re-typed
    assert got == ("FilterExec: ((((CAST(i AS Int64) Lt Int64(5)) And ((f Multiply (1 Minus f)) Gt CAST(i AS Float64))) And (d LtEq Date32(10471))) "
                   "And (l Eq CAST(i AS Int64)))")
    # and what the Python mirror inserts is the same tree
    schema = {"i": E.INT32, "f": E.FLOAT64, "d": E.DATE32, "l": E.INT64}
    again = ba.ExecutionPlan.from_proto(None, pe.plan(N.FilterExec(E.coerce(raw, schema), m))).display().splitlines()[0]
    assert again == got


def test_leaf_kinds_and_their_descriptions():
    fields = [("k", "Int32", False), ("v", "Float64", True), ("s", "Utf8", True)]
    sr = ba.ExecutionPlan.from_proto(None, pe.shuffle_reader([("job1", 2, 0, "ex-a", "10.0.0.1", 50051), ("job1", 2, 1, "ex-b", "10.0.0.2", 50052)], fields))
    assert sr.as_any() == "ShuffleReaderExec"
    assert sr.display().strip() == "ShuffleReaderExec: partition_locations=[job1/2/0@10.0.0.1:50051, job1/2/1@10.0.0.2:50052], schema=[k, v, s]"
    assert sr.schema() == fields
    un = ba.ExecutionPlan.from_proto(None, pe.unresolved_shuffle([3, 4], fields, 8))
    assert un.as_any() == "UnresolvedShuffleExec" and un.output_partitioning().partition_count() == 8
    assert "query_stage_ids=[3, 4], partition_count=8" in un.display()
    with pytest.raises(ba.NotImplementedOnGpu, match="ParquetExec"):
        ba.ExecutionPlan.from_proto(None, pe.parquet_scan(["/data/a.parquet"], [0, 2]))
    # the resolver sees every field of the leaf
    seen = []

    def resolver(leaf):
        seen.append(leaf)
        return None

    with pytest.raises(ba.NotImplementedOnGpu):          # the resolver declined: a Parquet leaf has no schema on the wire
        ba.ExecutionPlan.from_proto(None, pe.parquet_scan(["/data/a.parquet", "/data/b.parquet"], [0, 2], num_partitions=4), resolver)
    assert seen[0]["kind"] == "ParquetScan" and seen[0]["filenames"] == ["/data/a.parquet", "/data/b.parquet"]
    assert seen[0]["projection"] == [0, 2] and seen[0]["num_partitions"] == 4
    seen.clear()
    ba.ExecutionPlan.from_proto(None, pe.shuffle_reader([("j", 1, 7, "e", "h", 1234)], fields), resolver)
    assert seen[0]["kind"] == "ShuffleReader" and seen[0]["fields"] == fields
    assert seen[0]["locations"][0] == dict(job_id="j", stage_id=1, partition_id=7, executor_id="e", host="h", port=1234, num_rows=-1,
                                           num_batches=-1, num_bytes=-1)


def test_types_of_the_whole_serde_surface():
    """ArrowType variants of ballista.proto:755-790 the library carries"""
    names = ["Boolean", "UInt8", "Int8", "UInt16", "Int16", "UInt32", "Int32", "UInt64", "Int64", "Float32", "Float64", "Utf8", "Date32", "Date64",
             "Timestamp(Second)", "Timestamp(Millisecond)", "Timestamp(Microsecond)", "Timestamp(Nanosecond)", "LargeUtf8", "Binary"]
    fields = [(f"c{i}", t, i % 2 == 0) for i, t in enumerate(names)]
    p = ba.ExecutionPlan.from_proto(None, pe.unresolved_shuffle([1], fields, 1))
    assert p.schema() == fields
    with pytest.raises(ba.NotImplementedOnGpu, match="Float16"):
        ba.ExecutionPlan.from_proto(None, pe.unresolved_shuffle([1], [("x", "Float16", True)], 1))


def test_malformed_and_unsupported_plans_are_errors_not_crashes(nodes_tpch):
    good = pe.plan(build(nodes_tpch, "q3", tables()))
    for cut in (1, 2, 7, len(good) // 3, len(good) // 2, len(good) - 1):
        with pytest.raises(ba.BallistaError):
            ba.ExecutionPlan.from_proto(None, good[:cut])
    with pytest.raises(ba.PlanError, match="Unsupported physical plan"):
        ba.ExecutionPlan.from_proto(None, b"")
    m = leaf("t", gen.orders(0.001))
    # a node without its input (convert_box_required!)
    with pytest.raises(ba.PlanError, match="without an input"):
        ba.ExecutionPlan.from_proto(None, pe.f_bytes(13, pe.f_bytes(2, pe.expr(col("o_orderkey") > lit(1)))))
    with pytest.raises(ba.PlanError, match="filter .FilterExecNode. in PhysicalPlanNode is missing"):
        ba.ExecutionPlan.from_proto(None, pe.f_bytes(13, pe.f_bytes(1, pe.plan(m))))
    # unknown enum values / operators
    with pytest.raises(ba.PlanError, match="unknown JoinType 7"):
        ba.ExecutionPlan.from_proto(None, pe.f_bytes(9, pe.f_bytes(1, pe.plan(m)) + pe.f_bytes(2, pe.plan(m)) +
                                                     pe.f_bytes(3, pe.f_str(1, "o_orderkey") + pe.f_str(2, "o_orderkey")) + pe.f_varint(4, 7)))
    bad_op = pe.f_bytes(4, pe.f_bytes(1, pe.expr(col("o_orderkey"))) + pe.f_bytes(2, pe.expr(lit(1))) + pe.f_str(3, "BitwiseAnd"))
    with pytest.raises(ba.PlanError, match="Unsupported binary operator 'BitwiseAnd'"):
        ba.ExecutionPlan.from_proto(None, pe.f_bytes(13, pe.f_bytes(1, pe.plan(m)) + pe.f_bytes(2, bad_op)))
    with pytest.raises(ba.PlanError, match="No field named 'nope'"):
        ba.ExecutionPlan.from_proto(None, pe.plan(N.FilterExec(col("nope") > lit(1), m)))
    with pytest.raises(ba.NotImplementedOnGpu, match="md5"):
        ba.ExecutionPlan.from_proto(None, pe.plan(N.ProjectionExec([(pe_fn("md5", col("o_orderkey")), "h")], m)))
    # unknown fields are skipped (proto3), here field 99 inside a FilterExecNode
    ok = pe.f_bytes(13, pe.f_bytes(1, pe.plan(m)) + pe.f_bytes(2, pe.expr(col("o_orderkey") > lit(1))) + pe.f_varint(99, 5) + pe.f_str(98, "x"))
    assert ba.ExecutionPlan.from_proto(None, ok).as_any() == "FilterExec"


def test_deeply_nested_messages_are_refused_not_recursed_into():
    """a NOT inside a NOT inside ... 10 000 deep, and 5 000 nested LocalLimit nodes: the decoder refuses at 128 levels
    instead of walking the C++ stack down with the message"""
    import ctypes as C
    from ballista_amd import _lib as L
    e = pe.expr(col("x"))
    for _ in range(10_000):
        e = pe.f_bytes(8, pe.f_bytes(1, e))                      # LogicalExprNode.not_expr = 8 { expr = 1 } (ballista.proto:34)
    buf = C.create_string_buffer(256)
    assert L.lib().bhip_expr_from_proto_display(e, len(e), buf, len(buf)) == L.EINVAL
    assert b"nested deeper" in L.lib().bhip_last_error()
    p = pe.plan(leaf("t", gen.orders(0.001)))
    for _ in range(5_000):
        p = pe.f_bytes(7, pe.f_bytes(1, p) + pe.f_varint(2, 3))   # LocalLimitExecNode { input, limit }
    with pytest.raises(ba.PlanError, match="nested deeper"):
        ba.ExecutionPlan.from_proto(None, p)
    # 100 levels are fine
    p = pe.plan(leaf("t", gen.orders(0.001)))
    for _ in range(100):
        p = pe.f_bytes(7, pe.f_bytes(1, p) + pe.f_varint(2, 3))
    assert ba.ExecutionPlan.from_proto(None, p).as_any() == "LocalLimitExec"


def pe_fn(fun, *args):
    e = object.__new__(E.ScalarFunctionExpr)          # bypass the host mirror's "supported functions" check: the wire can carry any
    e.fun, e.args = fun, list(args)
    return e


def test_repartition_limit_and_empty_nodes():
    m = leaf("t", gen.orders(0.001))
    for part, text in ((N.Partitioning.Hash([col("o_custkey")], 8), "Hash([o_custkey], 8)"), (N.Partitioning.RoundRobinBatch(3), "RoundRobinBatch(3)"),
                       (N.Partitioning.UnknownPartitioning(2), "UnknownPartitioning(2)")):
        p = ba.ExecutionPlan.from_proto(None, pe.plan(N.RepartitionExec(m, part)))
        assert p.as_any() == "RepartitionExec" and p.output_partitioning().partition_count() == part.count
        assert text in p.display()
    p = ba.ExecutionPlan.from_proto(None, pe.plan(N.GlobalLimitExec(N.LocalLimitExec(m, 10), 5)))
    assert p.display().splitlines()[0] == "GlobalLimitExec: limit=5" and p.children()[0].as_any() == "LocalLimitExec"
    empty = pe.f_bytes(3, pe.f_varint(1, 1) + pe.f_bytes(2, pe.schema([("a", "Int64", True)])))
    e = ba.ExecutionPlan.from_proto(None, empty)
    assert e.as_any() == "EmptyExec" and e.schema() == [("a", "Int64", True)]


# ---- with a GPU: decoded plans run ------------------------------------------------------------------------------------------

def _resolver(ctx, t):
    made = {}

    def resolve(leaf):
        assert leaf["kind"] == "CsvScan" and leaf["path"].startswith("mem://")
        name = leaf["path"][6:]
        if name not in made:
            made[name] = helpers.memory_exec(ctx, [[t[name]]])
        return made[name]
    return resolve, made


@pytest.mark.gpu
@pytest.mark.parametrize("query", ["q1", "q6", "q3", "q5"])
def test_decoded_plan_runs_and_matches_the_ctypes_built_plan(ctx, monkeypatch, query):
    sf = 0.01
    t = tables(sf)
    monkeypatch.setattr(tpch, "P", N)
    data = pe.plan(build(tpch, query, t))
    monkeypatch.undo()
    resolve, made = _resolver(ctx, t)
    decoded = ba.ExecutionPlan.from_proto(ctx, data, resolve)
    m = lambda k: made[k]
    direct = {"q1": lambda: tpch.q1_plan(m("lineitem")), "q6": lambda: tpch.q6_plan(m("lineitem")),
              "q3": lambda: tpch.q3_plan(m("customer"), m("orders"), m("lineitem")),
              "q5": lambda: tpch.q5_plan(m("customer"), m("orders"), m("lineitem"), m("supplier"), m("nation"), m("region"))}[query]()
    assert decoded.display() == direct.display()
    got = helpers.concat([helpers.from_device(b) for b in decoded.collect()])
    if query in ("q3", "q5"):
        import test_goldens as tg
        g = tg.load(f"{query}_synth.json")
        assert g["sf"] == sf
        (tg.check_q3 if query == "q3" else tg.check_q5)(got, g)
    elif query == "q1":
        g = json.load(open(os.path.join(helpers.GOLDEN, "q1_synth.json")))
        rows = g["rows"]
        assert [(a, b) for a, b in zip(got["l_returnflag"].to_pylist(), got["l_linestatus"].to_pylist())] == [(r["l_returnflag"], r["l_linestatus"]) for r in rows]
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert np.allclose(got[k].to_pylist(), [r[k] for r in rows], rtol=1e-9, atol=0)
        assert got["count_order"].to_pylist() == [r["count_order"] for r in rows]
    else:
        g = json.load(open(os.path.join(helpers.GOLDEN, "q6_synth.json")))
        assert abs(got["revenue"].to_pylist()[0] - g["revenue"]) <= 1e-9 * abs(g["revenue"])


@pytest.mark.gpu
def test_csv_scan_leaf_runs_on_the_device_tbl_reader(ctx, monkeypatch):
    """no resolver: a CsvScanExecNode over '|'-separated files is the library's own device scan — Q1 over the reference's
    lineitem fixture files straight from the wire plan"""
    tbl = os.path.join(helpers.GOLDEN, "tbl")
    # the 16 fields of lineitem.tbl (rust/benchmarks/tpch/src/main.rs:267-360)
    file_schema = [("l_orderkey", "Int32"), ("l_partkey", "Int32"), ("l_suppkey", "Int32"), ("l_linenumber", "Int32"), ("l_quantity", "Float64"),
                   ("l_extendedprice", "Float64"), ("l_discount", "Float64"), ("l_tax", "Float64"), ("l_returnflag", "Utf8"), ("l_linestatus", "Utf8"),
                   ("l_shipdate", "Date32"), ("l_commitdate", "Date32"), ("l_receiptdate", "Date32"), ("l_shipinstruct", "Utf8"),
                   ("l_shipmode", "Utf8"), ("l_comment", "Utf8")]
    proj = [0, 2, 4, 5, 6, 7, 8, 9, 10]
    # a CsvScanExecNode with partition filenames
    body = (pe.f_str(1, tbl) + pe.f_packed(2, proj) + pe.f_bytes(3, pe.schema([(n, t, False) for n, t in file_schema])) + pe.f_str(4, ".tbl") +
            pe.f_varint(6, 32768) + pe.f_str(7, "|") + pe.f_str(8, os.path.join(tbl, "lineitem_partition0.tbl")) +
            pe.f_str(8, os.path.join(tbl, "lineitem_partition1.tbl")))
    scan_bytes = pe.f_bytes(2, body)
    scan = ba.ExecutionPlan.from_proto(ctx, scan_bytes)
    assert scan.as_any() == "CsvExec" and scan.output_partitioning().partition_count() == 2
    assert [n for n, _, _ in scan.schema()] == [file_schema[i][0] for i in proj]
    # Q1 above it, on the wire as well: encode the operators over a stand-in leaf, then splice the real scan bytes in
    li = N.MemoryExec([[helpers.lineitem_fixture()]])
    li.name = "mem://x"
    li._schema = [(file_schema[i][0], file_schema[i][1], False) for i in proj]
    monkeypatch.setattr(tpch, "P", N)
    q1 = tpch.q1_plan(li)
    monkeypatch.undo()
    orig = pe.plan

    def plan_with_scan(p):
        return scan_bytes if p is li else orig(p)
    monkeypatch.setattr(pe, "plan", plan_with_scan)
    data = orig(q1)
    monkeypatch.undo()
    got = helpers.concat([helpers.from_device(b) for b in ba.ExecutionPlan.from_proto(ctx, data).collect()])
    g = json.load(open(os.path.join(helpers.GOLDEN, "q1_fixture.json")))["rows"]
    assert list(zip(got["l_returnflag"].to_pylist(), got["l_linestatus"].to_pylist())) == [(r["l_returnflag"], r["l_linestatus"]) for r in g]
    assert got["count_order"].to_pylist() == [r["count_order"] for r in g]
    assert np.allclose(got["sum_charge"].to_pylist(), [r["sum_charge"] for r in g], rtol=1e-12, atol=0)
