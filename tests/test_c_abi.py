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
