"""CPU tier: the host parsers of the library under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C ballista_amd/csrc host-asan` compiles EVERY host source with -fsanitize=address,undefined and links
tests/c/host_fuzz.cpp against them (no GPU is touched: NULL device context, host-only entry points).  The harness gets valid
fixtures — wire plans of TPC-H Q1 / Q3 / Q5 / Q6 and an expression (tests/proto_encode.py), Arrow IPC files written by pyarrow
and by the library's own writer, Parquet files in every writer variant the scan supports — and feeds the parsers every
truncation and bit flip of them (sampled for the larger files, dense over headers and footers) plus length / offset words
pushed to extremes.  A mutant may be accepted or refused; the test fails on a crash, a sanitizer report or a VALID fixture
that is refused.

Parsers covered: host/proto.cpp (reference: rust/core/src/serde/physical_plan/from_proto.rs:58-364), host/ipc.cpp
(rust/core/src/utils.rs:49-84, execution_plans/shuffle_reader.rs:77-99), host/parquet_host.cpp (from_proto.rs:111-121)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import ballista_amd as ba
from ballista_amd import expr as E, tpch
from ballista_amd.expr import col, lit

import plan_nodes as N
import proto_encode as pe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ballista_amd", "csrc")
EXE = os.path.join(CSRC, "build_asan", "host_fuzz")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang for the sanitizer build")
    r = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), "host-asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return EXE


def write_fixtures(d, monkeypatch):
    import pyarrow as pa
    import pyarrow.parquet as pq
    from oracle import gen
    import test_proto as tp
    # ---- wire plans and an expression --------------------------------------------------------------------------------------
    monkeypatch.setattr(tpch, "P", N)
    t = tp.tables(0.001)
    for q in ("q1", "q6", "q3", "q5"):
        with open(os.path.join(d, f"{q}.plan.bin"), "wb") as f:
            f.write(pe.plan(tp.build(tpch, q, t)))
    monkeypatch.undo()
    x, s = col("x"), col("s")
    e = E.CaseExpr(None, [(E.InListExpr(s, [lit("MAIL"), lit("SHIP")]), (x + lit(1)) * lit(2.5))], E.CastExpr(E.NegativeExpr(x), E.FLOAT64))
    with open(os.path.join(d, "case.expr.bin"), "wb") as f:
        f.write(pe.expr(e))
    with open(os.path.join(d, "shuffle.plan.bin"), "wb") as f:
        f.write(pe.shuffle_reader([("j", 1, p, "e", "h", 50051) for p in range(3)],
                                  [("a", "Int32", False), ("s", "Utf8", True)]))
    # ---- Arrow IPC files: pyarrow's writer and the library's own -------------------------------------------------------------------
    rng = np.random.default_rng(5)
    n = 300
    tbl = pa.table({
        "i": pa.array(rng.integers(-1000, 1000, n).astype(np.int32)),
        "l": pa.array([None if k % 7 == 0 else int(k) * 10 ** 9 for k in range(n)], pa.int64()),
        "f": pa.array(rng.random(n)),
        "s": pa.array([None if k % 5 == 0 else f"s{k % 13}-{'x' * (k % 4)}" for k in range(n)], pa.string()),
        "ls": pa.array([f"L{k}" for k in range(n)], pa.large_string()),
        "d": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "b": pa.array([None if k % 11 == 0 else bool(k & 1) for k in range(n)], pa.bool_()),
        "u": pa.array(rng.integers(0, 2 ** 40, n).astype(np.uint64)),
    })
    with pa.OSFile(os.path.join(d, "pyarrow.arrow"), "wb") as sink, pa.ipc.new_file(sink, tbl.schema) as w:
        for b in tbl.to_batches(max_chunksize=128):
            w.write_batch(b)
    ba.plan.ipc_write_file(pa.RecordBatchReader.from_batches(tbl.schema, tbl.to_batches(max_chunksize=100)), os.path.join(d, "own.arrow"))
    with pa.OSFile(os.path.join(d, "empty.arrow"), "wb") as sink, pa.ipc.new_file(sink, tbl.schema):
        pass
    # ---- Parquet: every writer variant the scan reads ---------------------------------------------------------------------------------
    li = gen.lineitem(0.001, 0, 1500)
    cols = {k: (pa.array(list(c.values)) if c.dtype == "Utf8" else pa.array(c.values)) for k, c in li.items()}
    cols["l_shipdate"] = cols["l_shipdate"].cast(pa.date32())
    cols["opt_f"] = pa.array([None if k % 6 == 0 else float(k) for k in range(1500)], pa.float64())
    cols["opt_s"] = pa.array([None if k % 4 == 0 else f"c{k % 9}" for k in range(1500)], pa.string())
    cols["flag"] = pa.array([None if k % 10 == 0 else bool(k % 3) for k in range(1500)], pa.bool_())
    pt = pa.table(cols)
    variants = [("snappy_dict_v1", dict(compression="snappy", use_dictionary=True, data_page_version="1.0")),
                ("plain_uncompressed_v1", dict(compression="none", use_dictionary=False, data_page_version="1.0")),
                ("snappy_dict_v2", dict(compression="snappy", use_dictionary=True, data_page_version="2.0")),
                ("snappy_plain_v2_pages", dict(compression="snappy", use_dictionary=False, data_page_version="2.0", data_page_size=2048))]
    for name, kw in variants:
        pq.write_table(pt, os.path.join(d, name + ".parquet"), row_group_size=700, **kw)


def test_host_parsers_survive_truncations_and_bit_flips_under_asan(harness, tmp_path, monkeypatch):
    fixtures, scratch = tmp_path / "fixtures", tmp_path / "scratch"
    fixtures.mkdir()
    scratch.mkdir()
    write_fixtures(str(fixtures), monkeypatch)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else str(scratch)          # mutants are rewritten thousands of times: keep them in memory
    work = os.path.join(shm, f"bhip_fuzz_{os.getpid()}")
    os.makedirs(work, exist_ok=True)
    try:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
        r = subprocess.run([harness, str(fixtures), work], capture_output=True, text=True, timeout=1500, env=env)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert r.returncode == 0, tail
    assert "host_fuzz OK" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    # every fixture kind was exercised
    for name in ("q5.plan.bin", "case.expr.bin", "pyarrow.arrow", "own.arrow", "snappy_dict_v2.parquet"):
        assert name in r.stdout, tail
