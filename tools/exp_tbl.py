#!/usr/bin/env python3
"""`.tbl` scan throughput (run on the GPU box): the reference's lineitem fixture lines repeated to ~N MB of text,
scanned with the Q1 projection (7 of 16 fields) and with all fields; the text starts in host memory (PCIe included)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ballista_amd as ba
from ballista_amd import expr as E, tpch

LINEITEM = [("l_orderkey", E.INT32), ("l_partkey", E.INT32), ("l_suppkey", E.INT32), ("l_linenumber", E.INT32),
            ("l_quantity", E.FLOAT64), ("l_extendedprice", E.FLOAT64), ("l_discount", E.FLOAT64), ("l_tax", E.FLOAT64),
            ("l_returnflag", E.UTF8), ("l_linestatus", E.UTF8), ("l_shipdate", E.DATE32), ("l_commitdate", E.DATE32),
            ("l_receiptdate", E.DATE32), ("l_shipinstruct", E.UTF8), ("l_shipmode", E.UTF8), ("l_comment", E.UTF8)]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = open(os.path.join(root, "tests", "golden", "tbl", "lineitem_partition0.tbl"), "rb").read()
mb = int(os.environ.get("MB", 1024))
text = unit * max(1, mb * (1 << 20) // len(unit))
ctx = ba.Context(0)
for name, cols in (("q1 projection (7 of 16 fields)", list(tpch.LINEITEM_SCHEMA)), ("all 16 fields", None)):
    for it in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        rb = ba.RecordBatch.from_tbl(ctx, text, LINEITEM, cols)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps(dict(scan=name, text_mb=len(text) / 2 ** 20, rows=rb.num_rows, ms=dt * 1e3, text_gbs=len(text) / dt / 1e9,
                          rows_per_s=rb.num_rows / dt)), flush=True)

# ---- end to end: text in host memory -> scan (Q1 projection) -> Q1 plan, at the size of an SF1 lineitem.tbl (~724 MB)
sf1 = unit * (724 * (1 << 20) // len(unit))
cols = list(tpch.LINEITEM_SCHEMA)
for it in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    rb = ba.RecordBatch.from_tbl(ctx, sf1, LINEITEM, cols)
    t1 = time.perf_counter()
    res = tpch.q1_plan(ba.MemoryExec([[rb]], ctx)).collect()
    ctx.synchronize()
    t2 = time.perf_counter()
print(json.dumps(dict(run="Q1 from .tbl text in host memory", text_mb=len(sf1) / 2 ** 20, rows=rb.num_rows, scan_ms=(t1 - t0) * 1e3,
                      q1_ms=(t2 - t1) * 1e3, total_ms=(t2 - t0) * 1e3, groups=res[0].num_rows)), flush=True)
