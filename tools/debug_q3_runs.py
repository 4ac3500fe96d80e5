import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BHIP_KERNEL_TIMING"] = "2"
import numpy as np
import ballista_amd as ba
from ballista_amd import tpch
P = ba.plan
ctx = ba.Context(0)
sf = 1.0
n = tpch.table_rows(sf)
li = P.MemoryExec([[P.tpch_lineitem(ctx, sf, tpch.SEED, 0, n["lineitem"])]], ctx)
od = P.MemoryExec([[P.tpch_orders(ctx, sf, tpch.SEED, 0, n["orders"])]], ctx)
dims = {k: P.MemoryExec([[b]], ctx) for k, b in tpch.dimension_tables(ctx, sf, "q3").items()}
j = tpch.q3_partial(tpch.q3_build_side(dims["customer"], od), tpch.q3_probe_side(li))
join = j.children()[0]
out = join.collect()
print("join batches", len(out), [b.num_rows for b in out], [c[0] for c in out[0].schema3()])
names = [c[0] for c in out[0].schema3()]
k = out[0].column(names.index("l_orderkey"))[1]
print("l_orderkey ascending:", bool(np.all(np.diff(k.astype(np.int64)) >= 0)), k[:10])
ctx.kernel_stats(reset=True)
r = tpch.fresh(j).collect()
print(sorted(ctx.kernel_stats(reset=True).keys()))
