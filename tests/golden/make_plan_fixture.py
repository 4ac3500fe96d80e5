#!/usr/bin/env python3
"""Writes tests/golden/plans/q1_fixture.plan.bin: the protobuf bytes of TPC-H Q1's PhysicalPlanNode
(rust/core/proto/ballista.proto:294-312) over a CsvScan leaf `mem://lineitem` with the nine projected lineitem columns of the
reference's fixture (rust/scheduler/testdata/lineitem/partition{0,1}.tbl) — what a Ballista executor receives in its task
(rust/executor/src/flight_service.rs:87).  The encoder is tests/proto_encode.py (written from the .proto's field numbers);
tests/c/shim_sequence.c replays the Rust shim's call sequence over these bytes.  Run from the repo root:
    python tests/golden/make_plan_fixture.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
import plan_nodes as N  # noqa: E402
import proto_encode as pe  # noqa: E402
from ballista_amd import tpch  # noqa: E402

tpch.P = N
leaf = N.MemoryExec([[helpers.lineitem_fixture()]])
leaf.name = "mem://lineitem"
out = os.path.join(ROOT, "tests", "golden", "plans", "q1_fixture.plan.bin")
with open(out, "wb") as f:
    f.write(pe.plan(tpch.q1_plan(leaf)))
print(out, os.path.getsize(out), "bytes")
