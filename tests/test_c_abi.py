"""The boundary is a C ABI: tests/c/abi_client.c is a plain-C11 client (no Python, no C++) that feeds host columns,
builds Filter -> HashAggregate(Partial) -> Merge -> HashAggregate(Final) -> Sort from postfix expressions, reads the
result back and checks it against its own scalar loops — what a cgo / JNI / Rust-FFI binding would do.

CPU tier: the header compiles as C and the client links against the library.  GPU tier: it runs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_client.c")
LIBDIR = os.path.join(ROOT, "ballista_amd", "lib")


def build(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "abi_client")
    cmd = [cc, "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR, "-lballista_hip",
           f"-Wl,-rpath,{LIBDIR}", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c_and_client_links(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, "libballista_hip.so")), "build the library first (__graft_entry__.build())"
    build(tmp_path)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C ABI OK" in r.stdout, r.stdout + r.stderr


# ---- the Rust shim's call sequence, replayed in C (tests/c/shim_sequence.c) ------------------------------------------------------------

SHIM_SRC = os.path.join(ROOT, "tests", "c", "shim_sequence.c")


def build_shim(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "shim_sequence")
    cmd = [cc, "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SHIM_SRC, "-L", LIBDIR, "-lballista_hip",
           f"-Wl,-rpath,{LIBDIR}", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_shim_sequence_client_links_and_plan_fixture_is_current(tmp_path):
    """the committed wire-plan fixture is what tests/golden/make_plan_fixture.py writes today (the encoder and the plan builders
    have not drifted from it), and the C replay compiles against the header"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import plan_nodes as N
    import proto_encode as pe
    from ballista_amd import tpch
    build_shim(tmp_path)
    old = tpch.P
    tpch.P = N
    try:
        leaf = N.MemoryExec([[helpers.lineitem_fixture()]])
        leaf.name = "mem://lineitem"
        want = pe.plan(tpch.q1_plan(leaf))
    finally:
        tpch.P = old
    with open(os.path.join(ROOT, "tests", "golden", "plans", "q1_fixture.plan.bin"), "rb") as f:
        assert f.read() == want


@pytest.mark.gpu
def test_shim_sequence_runs(tmp_path):
    """dry run -> resolver answering with caller-implemented Arrow C streams (moved, released once) -> execute / next / export ->
    write_ipc without a release, against the exact-rational Q1 golden of the reference's lineitem fixture"""
    import json
    exe = build_shim(tmp_path)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "q1_fixture.json")))["rows"]
    expected = tmp_path / "expected.txt"
    with open(expected, "w") as f:
        for r in g:
            f.write(" ".join([r["l_returnflag"], r["l_linestatus"], str(r["count_order"])] +
                             [repr(float(r[k])) for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc")]) + "\n")
    tbl = os.path.join(ROOT, "tests", "golden", "tbl")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "plans", "q1_fixture.plan.bin"), os.path.join(tbl, "lineitem_partition0.tbl"),
                        os.path.join(tbl, "lineitem_partition1.tbl"), str(expected), str(tmp_path / "stage.arrow")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SHIM SEQUENCE OK" in r.stdout, r.stdout + r.stderr
    # the stage file the C client wrote is an ordinary Arrow IPC file
    import pyarrow as pa
    t = pa.ipc.open_file(str(tmp_path / "stage.arrow")).read_all()
    assert t.num_rows == 3 and t.column("count_order").to_pylist() == [x["count_order"] for x in g]
