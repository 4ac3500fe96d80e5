"""CPU tests of the boundary: the C-ABI library loads and exports every symbol that
include/ballista_hip.h declares, reports errors as status + message (never aborts), and the
host-side mirror (expressions, coercion, plan lowering) behaves without a GPU."""
import ctypes as C
import os
import re

import pytest

import ballista_amd as ba
from ballista_amd import _lib as L, expr as E
from ballista_amd.expr import col, lit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ballista_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(bhip_[a-z0-9_]+)\s*\(", text))
    return sorted(names - {"bhip_status"})      # `bhip_status (*bhip_batch_sink)(...)` is a typedef


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 45
    lib = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ballista_hip.h but not exported"
    missing = [n for n in names if n not in L.SYMBOLS]
    assert not missing, f"symbols without a ctypes binding: {missing}"
    extra = [n for n in L.SYMBOLS if n not in names]
    assert not extra, f"bindings for undeclared symbols: {extra}"


def test_library_is_in_tree_and_native():
    assert os.path.realpath(L.LIB_PATH).startswith(os.path.realpath(ROOT))
    with open(L.LIB_PATH, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"\x7fELF"
    assert b"gfx950" in blob            # carries a gfx950 code object
    assert L.lib().bhip_version().startswith(b"ballista_hip")


def test_errors_are_values_not_aborts():
    lib = L.lib()
    h = C.c_void_p()
    st = lib.bhip_ctx_create(9999, C.byref(h))
    assert st in (L.EINVAL, L.EHIP)          # no such device / no device at all
    assert lib.bhip_last_error()
    # null arguments are reported, not dereferenced
    assert lib.bhip_plan_filter(None, None, C.byref(h)) == L.EINVAL
    assert b"null" in lib.bhip_last_error()
    assert lib.bhip_stream_next(None, C.byref(h)) == L.EINVAL
    assert lib.bhip_batch_num_rows(None) == -1
    lib.bhip_batch_release(None)
    lib.bhip_plan_release(None)
    lib.bhip_stream_release(None)


def test_product_has_no_cpu_fallback():
    """the product package never imports the oracle, and without the HIP library it raises"""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import ballista_amd, ballista_amd.plan, ballista_amd.tpch; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imported the oracle'") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['BHIP_LIB_PATH'] = '/nonexistent/libballista_hip.so'; "
            "import ballista_amd\n"
            "try:\n    ballista_amd.Context(0)\n    raise SystemExit('no error without the library')\n"
            "except RuntimeError as e:\n    assert 'no CPU fallback' in str(e)") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
    for root, _, files in os.walk(os.path.join(ROOT, "ballista_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_expression_mirror_and_coercion():
    s = {"l_extendedprice": E.FLOAT64, "l_discount": E.FLOAT64, "l_shipdate": E.DATE32, "k": E.INT32}
    e = E.coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), s)
    assert isinstance(e.right.left, E.Literal) and e.right.left.dtype == E.FLOAT64 and e.right.left.value == 1.0
    assert E.expr_type(e, s) == E.FLOAT64
    p = E.coerce(col("l_shipdate") <= E.date32("1998-09-02"), s)
    assert p.right.value == 10471 and E.expr_type(p, s) == E.BOOLEAN
    c = E.coerce(col("k") + lit(1), s)           # Int32 + Int64 literal -> Int64 + Int64
    assert isinstance(c.left, E.CastExpr) and c.left.dtype == E.INT64
    with pytest.raises(ValueError, match="Unsupported binary operator"):
        E.BinaryExpr(col("k"), "Modulus", lit(2))
    with pytest.raises(NotImplementedError):
        E.ScalarFunctionExpr("md5", [col("k")])
    with pytest.raises(KeyError):
        E.expr_type(col("nope"), s)


def test_expression_lowering_is_postfix():
    from ballista_amd.plan import _Lowered
    lw = _Lowered()
    ex = lw.expr(E.NotExpr((col("a") > lit(1)).and_(E.InListExpr(col("b"), [lit(1), lit(2)]))))
    kinds = [ex.nodes[i].kind for i in range(ex.n_nodes)]
    # a 1 Gt b 1 2 IN And NOT
    assert kinds == [1, 2, 3, 1, 2, 2, 9, 3, 5]
    assert ex.nodes[2].name == b"Gt" and ex.nodes[7].name == b"And" and ex.nodes[6].n_args == 2
    case = lw.expr(E.CaseExpr(col("k"), [(lit(1), lit(1.0)), (lit(2), lit(2.0))], lit(0.0)))
    last = case.nodes[case.n_nodes - 1]
    assert last.kind == 10 and last.n_args == 2 and last.flags == 3


def test_tpch_plan_expressions():
    from ballista_amd import tpch
    q1 = tpch.q1_parts(tpch.LINEITEM_SCHEMA)
    assert [a.fun for a in q1["aggs"]] == tpch.Q1_AGG_FUNS and [a.name for a in q1["aggs"]] == tpch.Q1_AGG_NAMES
    assert tpch.Q1_BYTES_PER_ROW == 46 and tpch.Q6_BYTES_PER_ROW == 28
    p = tpch.q6_parts(tpch.LINEITEM_SCHEMA)["predicate"]
    text = repr(p)
    assert "0.049999999999999996" in text and "0.06999999999999999" in text     # f64 results of 0.06 -/+ 0.01
