"""TPC-H workload definitions: schemas and the physical plans of Q1 / Q6 / Q3 / Q5.

Schemas: rust/benchmarks/tpch/src/main.rs:267-360 (keys Int32, money/qty Float64, dates Date32,
flags/names Utf8, all non-nullable).  Queries: rust/benchmarks/tpch/queries/q{1,3,5,6}.sql.
The plans are built in the shape DataFusion's planner + Ballista's DistributedPlanner produce
(rust/scheduler/src/planner.rs:136-171, expected plan :412-426):

    stage 1  HashAggregateExec(Partial) <- CoalesceBatchesExec <- FilterExec <- scan     (N partitions)
    stage 2  MergeExec
    stage 3  SortExec <- ProjectionExec <- HashAggregateExec(Final)

with casts inserted where DataFusion's type coercion would (`coerce`).
"""
from __future__ import annotations

from typing import Sequence

from . import expr as E
from .expr import col, lit, date32, coerce, Sum, Avg, Count, PhysicalSortExpr
from . import plan as P

SEED = 0x7C4A0BA11157A001

LINEITEM_SCHEMA = {"l_orderkey": E.INT32, "l_suppkey": E.INT32, "l_quantity": E.FLOAT64, "l_extendedprice": E.FLOAT64,
                   "l_discount": E.FLOAT64, "l_tax": E.FLOAT64, "l_returnflag": E.UTF8, "l_linestatus": E.UTF8,
                   "l_shipdate": E.DATE32}
ORDERS_SCHEMA = {"o_orderkey": E.INT32, "o_custkey": E.INT32, "o_orderdate": E.DATE32, "o_shippriority": E.INT32}
CUSTOMER_SCHEMA = {"c_custkey": E.INT32, "c_nationkey": E.INT32, "c_mktsegment": E.UTF8}
SUPPLIER_SCHEMA = {"s_suppkey": E.INT32, "s_nationkey": E.INT32}
NATION_SCHEMA = {"n_nationkey": E.INT32, "n_name": E.UTF8, "n_regionkey": E.INT32}
REGION_SCHEMA = {"r_regionkey": E.INT32, "r_name": E.UTF8}

# the 25 nations / 5 regions of the TPC-H spec (format of rust/scheduler/testdata/nation/nation.tbl, region/region.tbl)
NATIONS = [("ALGERIA", 0), ("ARGENTINA", 1), ("BRAZIL", 1), ("CANADA", 1), ("EGYPT", 4), ("ETHIOPIA", 0), ("FRANCE", 3),
           ("GERMANY", 3), ("INDIA", 2), ("INDONESIA", 2), ("IRAN", 4), ("IRAQ", 4), ("JAPAN", 2), ("JORDAN", 4), ("KENYA", 0),
           ("MOROCCO", 0), ("MOZAMBIQUE", 0), ("PERU", 1), ("CHINA", 2), ("ROMANIA", 3), ("SAUDI ARABIA", 4), ("VIETNAM", 2),
           ("RUSSIA", 3), ("UNITED KINGDOM", 3), ("UNITED STATES", 1)]
REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]
SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"]
LINEITEM_ROWS_SF100 = 600_037_902


def table_rows(sf: float):
    """row counts of the synthetic tables at scale factor sf"""
    n_li = LINEITEM_ROWS_SF100 if sf == 100 else int(round(6_000_379.02 * sf))
    return dict(lineitem=n_li, orders=int(1_500_000 * sf), customer=int(150_000 * sf), supplier=int(10_000 * sf))


def dimension_arrays(sf: float, segments=True):
    """numpy columns of the small tables (customer, supplier, nation, region): dense keys from 1, uniform nation keys
    and market segments from fixed seeds — the same on every rank, so the small sides of the joins are replicated.
    segments=False skips c_mktsegment (only Q3 reads it; at SF1000 it is 150 M strings)."""
    import numpy as np
    n = table_rows(sf)
    rc, rs = np.random.default_rng(7), np.random.default_rng(11)
    customer = dict(c_custkey=np.arange(1, n["customer"] + 1, dtype=np.int32),
                    c_nationkey=rs.integers(0, 25, n["customer"]).astype(np.int32))
    if segments:
        customer["c_mktsegment"] = rc.integers(0, 5, n["customer"]).astype(np.int32)       # index into SEGMENTS
    return dict(customer=customer,
                supplier=dict(s_suppkey=np.arange(1, n["supplier"] + 1, dtype=np.int32),
                              s_nationkey=rs.integers(0, 25, n["supplier"]).astype(np.int32)))


def dimension_tables(ctx, sf: float, query=None):
    """the small tables as device batches (through the Arrow C Data Interface); query "q5": customer without c_mktsegment,
    "q3": customer only"""
    import numpy as np
    import pyarrow as pa
    a = dimension_arrays(sf, segments=query != "q5")

    def dev(names, arrays):
        return P.RecordBatch.from_pyarrow(ctx, pa.RecordBatch.from_arrays([x if isinstance(x, pa.Array) else pa.array(x) for x in arrays], names=names))

    c, s_ = a["customer"], a["supplier"]
    if query == "q5":
        out = dict(customer=dev(["c_custkey", "c_nationkey"], [c["c_custkey"], c["c_nationkey"]]))
    else:
        seg = pa.DictionaryArray.from_arrays(pa.array(c["c_mktsegment"]), pa.array(SEGMENTS)).cast(pa.string())
        out = dict(customer=dev(["c_custkey", "c_nationkey", "c_mktsegment"], [c["c_custkey"], c["c_nationkey"], seg]))
    if query == "q3":
        return out
    out.update(supplier=dev(["s_suppkey", "s_nationkey"], [s_["s_suppkey"], s_["s_nationkey"]]),
               nation=dev(["n_nationkey", "n_name", "n_regionkey"],
                          [np.arange(25, dtype=np.int32), [n for n, _ in NATIONS], np.array([r for _, r in NATIONS], np.int32)]),
               region=dev(["r_regionkey", "r_name"], [np.arange(5, dtype=np.int32), REGIONS]))
    return out


def fresh(plan: "P.ExecutionPlan") -> "P.ExecutionPlan":
    """a clone of the whole tree through with_new_children: new operator objects, hence empty join build caches and no
    remembered path choices — what a task that has just decoded its plan from the wire starts from
    (rust/executor/src/flight_service.rs:87-121).  Leaves (MemoryExec) are shared: the tables stay in HBM."""
    kids = plan.children()
    if not kids:
        return plan
    return plan.with_new_children([fresh(k) for k in kids])


# Algorithmic HBM bytes per input row: every needed column read once.
Q1_BYTES_PER_ROW = 4 * 8 + 4 + 2 * (4 + 1)      # 46: 4 x f64, Date32, 2 x Utf8 (offset + 1 char)
Q6_BYTES_PER_ROW = 4 + 3 * 8                    # 28


# algorithmic bytes of ONE launch of a kernel over n lineitem rows with kb-byte order keys: what the
# launch must read and write at least.  Used by bench.py for the roofline of whichever kernel dominates a query.
KERNEL_BYTES = {}


def q3_algorithmic_bytes(n_li, n_ord, n_cust, key_bytes=4):
    """every input column once (orderkey of `key_bytes`)"""
    return n_li * (key_bytes + 8 + 8 + 4) + n_ord * (key_bytes + 12) + n_cust * 17


def q5_algorithmic_bytes(n_li, n_ord, n_cust, n_supp, key_bytes=4):
    """lineitem orderkey + suppkey + 2 x f64, orders orderkey + custkey + orderdate,
    customer / supplier 8 B"""
    return n_li * (key_bytes + 4 + 16) + n_ord * (key_bytes + 8) + n_cust * 8 + n_supp * 8


def _schema_of(plan):
    return {n: t for n, t, _ in plan.schema()}


def q1_parts(schema):
    """the expressions of Q1 stage 1 over `schema` (name -> type): predicate, group, aggregates"""
    c = lambda e: coerce(e, schema)
    disc_price = col("l_extendedprice") * (lit(1) - col("l_discount"))
    charge = disc_price * (lit(1) + col("l_tax"))
    aggs = [Sum(c(col("l_quantity")), "sum_qty"), Sum(c(col("l_extendedprice")), "sum_base_price"),
            Sum(c(disc_price), "sum_disc_price"), Sum(c(charge), "sum_charge"),
            Avg(c(col("l_quantity")), "avg_qty"), Avg(c(col("l_extendedprice")), "avg_price"),
            Avg(c(col("l_discount")), "avg_disc"), Count(lit(1, E.UINT8), "count_order")]
    group = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]
    return dict(predicate=c(col("l_shipdate") <= date32("1998-09-02")), group=group, aggs=aggs)


def q1_stage1(scan: P.ExecutionPlan, target_batch_size=4096) -> P.ExecutionPlan:
    """TPC-H Q1 stage 1: scan -> Filter -> CoalesceBatches -> HashAggregate(Partial)."""
    q = q1_parts(_schema_of(scan))
    flt = P.FilterExec(q["predicate"], scan)
    co = P.CoalesceBatchesExec(flt, target_batch_size)
    return P.HashAggregateExec(P.PARTIAL, q["group"], q["aggs"], co)


Q1_AGG_NAMES = ["sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc",
                "count_order"]
Q1_AGG_FUNS = ["SUM", "SUM", "SUM", "SUM", "AVG", "AVG", "AVG", "COUNT"]


def q1_final_aggs():
    # Final mode reads the state columns by position; the argument expressions are placeholders
    return [E.AggregateExpr(f, col("l_returnflag"), n) for f, n in zip(Q1_AGG_FUNS, Q1_AGG_NAMES)]


def q1_final(partial: P.ExecutionPlan) -> P.ExecutionPlan:
    """stages 2+3: Merge -> HashAggregate(Final) -> Projection -> Sort."""
    merged = P.MergeExec(partial)
    group = [(col("l_returnflag"), "l_returnflag"), (col("l_linestatus"), "l_linestatus")]
    aggs = q1_final_aggs()
    fin = P.HashAggregateExec(P.FINAL, group, aggs, merged)
    proj = P.ProjectionExec([(col(n), n) for n in ["l_returnflag", "l_linestatus"] + Q1_AGG_NAMES], fin)
    return P.SortExec([PhysicalSortExpr(col("l_returnflag")), PhysicalSortExpr(col("l_linestatus"))], proj)


def q1_plan(scan: P.ExecutionPlan) -> P.ExecutionPlan:
    return q1_final(q1_stage1(scan))


def q6_predicate(schema):
    c = lambda e: coerce(e, schema)
    # the BETWEEN bounds are the f64 results of 0.06 - 0.01 and 0.06 + 0.01
    lo, hi = 0.06 - 0.01, 0.06 + 0.01
    p = (col("l_shipdate") >= date32("1994-01-01")).and_(col("l_shipdate") < date32("1995-01-01"))
    p = p.and_((col("l_discount") >= lit(lo)).and_(col("l_discount") <= lit(hi)))
    p = p.and_(col("l_quantity") < lit(24))
    return c(p)


def q6_parts(schema):
    return dict(predicate=q6_predicate(schema), group=[],
                aggs=[Sum(coerce(col("l_extendedprice") * col("l_discount"), schema), "revenue")])


def q6_stage1(scan: P.ExecutionPlan) -> P.ExecutionPlan:
    q = q6_parts(_schema_of(scan))
    flt = P.FilterExec(q["predicate"], scan)
    co = P.CoalesceBatchesExec(flt, 4096)
    return P.HashAggregateExec(P.PARTIAL, [], q["aggs"], co)


def q6_final(partial: P.ExecutionPlan) -> P.ExecutionPlan:
    fin = P.HashAggregateExec(P.FINAL, [], [E.AggregateExpr("SUM", col("revenue[sum]"), "revenue")], P.MergeExec(partial))
    return P.ProjectionExec([(col("revenue"), "revenue")], fin)


def q6_plan(scan: P.ExecutionPlan) -> P.ExecutionPlan:
    return q6_final(q6_stage1(scan))


def q3_build_side(customer: P.ExecutionPlan, orders: P.ExecutionPlan) -> P.ExecutionPlan:
    """customer(BUILDING) |x| orders(< 1995-03-15) -> (o_orderkey, o_orderdate, o_shippriority): the build side of Q3's
    order-key join.  `orders` may be one rank's row block (the customer side is small and replicated)."""
    cs, os_ = _schema_of(customer), _schema_of(orders)
    cust = P.FilterExec(coerce(col("c_mktsegment").eq(lit("BUILDING")), cs), customer)
    cust = P.ProjectionExec([(col("c_custkey"), "c_custkey")], cust)
    ords = P.FilterExec(coerce(col("o_orderdate") < date32("1995-03-15"), os_), orders)
    j1 = P.HashJoinExec(cust, ords, [("c_custkey", "o_custkey")], P.INNER)
    return P.ProjectionExec([(col(n), n) for n in ["o_orderkey", "o_orderdate", "o_shippriority"]], j1)


def q3_probe_side(lineitem: P.ExecutionPlan) -> P.ExecutionPlan:
    """lineitem(> 1995-03-15) -> (l_orderkey, l_extendedprice, l_discount)"""
    li = P.FilterExec(coerce(col("l_shipdate") > date32("1995-03-15"), _schema_of(lineitem)), lineitem)
    return P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "l_extendedprice", "l_discount"]], li)


Q3_GROUP = ["l_orderkey", "o_orderdate", "o_shippriority"]


def q3_partial(j1: P.ExecutionPlan, li: P.ExecutionPlan) -> P.ExecutionPlan:
    """order-key join + HashAggregate(Partial)"""
    j2 = P.HashJoinExec(j1, li, [("o_orderkey", "l_orderkey")], P.INNER)
    revenue = coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), _schema_of(j2))
    return P.HashAggregateExec(P.PARTIAL, [(col(n), n) for n in Q3_GROUP], [Sum(revenue, "revenue")], j2)


def q3_final(partial: P.ExecutionPlan) -> P.ExecutionPlan:
    """Merge -> HashAggregate(Final) -> Projection -> Sort(revenue DESC, o_orderdate)"""
    group = [(col(n), n) for n in Q3_GROUP]
    fin = P.HashAggregateExec(P.FINAL, group, [E.AggregateExpr("SUM", col("l_orderkey"), "revenue")], P.MergeExec(partial))
    proj = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]], fin)
    return P.SortExec([PhysicalSortExpr(col("revenue"), descending=True), PhysicalSortExpr(col("o_orderdate"))], proj)


def q3_plan(customer: P.ExecutionPlan, orders: P.ExecutionPlan, lineitem: P.ExecutionPlan) -> P.ExecutionPlan:
    """customer(BUILDING) |x| orders(< 1995-03-15) |x| lineitem(> 1995-03-15); build side = left
    (collect-left hash join, from_proto.rs:253-276)."""
    return q3_final(q3_partial(q3_build_side(customer, orders), q3_probe_side(lineitem)))


def q5_build_side(customer, orders, nation, region) -> P.ExecutionPlan:
    """region(ASIA) |x| nation |x| customer |x| orders(1994) -> (o_orderkey, n_nationkey, n_name): the build side of Q5's
    order-key join.  `orders` may be one rank's row block."""
    rs, os_ = _schema_of(region), _schema_of(orders)
    reg = P.FilterExec(coerce(col("r_name").eq(lit("ASIA")), rs), region)
    reg = P.ProjectionExec([(col("r_regionkey"), "r_regionkey")], reg)
    nat = P.HashJoinExec(reg, nation, [("r_regionkey", "n_regionkey")], P.INNER)
    nat = P.ProjectionExec([(col("n_nationkey"), "n_nationkey"), (col("n_name"), "n_name")], nat)
    cust = P.HashJoinExec(nat, customer, [("n_nationkey", "c_nationkey")], P.INNER)
    cust = P.ProjectionExec([(col("c_custkey"), "c_custkey"), (col("n_nationkey"), "n_nationkey"), (col("n_name"), "n_name")], cust)
    ords = P.FilterExec(coerce((col("o_orderdate") >= date32("1994-01-01")).and_(col("o_orderdate") < date32("1995-01-01")), os_), orders)
    ords = P.ProjectionExec([(col("o_orderkey"), "o_orderkey"), (col("o_custkey"), "o_custkey")], ords)
    co = P.HashJoinExec(cust, ords, [("c_custkey", "o_custkey")], P.INNER)
    return P.ProjectionExec([(col("o_orderkey"), "o_orderkey"), (col("n_nationkey"), "n_nationkey"), (col("n_name"), "n_name")], co)


def q5_probe_side(lineitem) -> P.ExecutionPlan:
    return P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]], lineitem)


def q5_partial(co, li, supplier) -> P.ExecutionPlan:
    """order-key join, supplier join on (suppkey, nationkey), HashAggregate(Partial) by n_name"""
    col_ = P.HashJoinExec(co, li, [("o_orderkey", "l_orderkey")], P.INNER)
    # supplier joins on (suppkey, nationkey): the residual c_nationkey = s_nationkey is a second key pair
    sup = P.HashJoinExec(supplier, col_, [("s_suppkey", "l_suppkey"), ("s_nationkey", "n_nationkey")], P.INNER)
    revenue = coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), _schema_of(sup))
    return P.HashAggregateExec(P.PARTIAL, [(col("n_name"), "n_name")], [Sum(revenue, "revenue")], sup)


def q5_final(partial) -> P.ExecutionPlan:
    fin = P.HashAggregateExec(P.FINAL, [(col("n_name"), "n_name")], [E.AggregateExpr("SUM", col("n_name"), "revenue")], P.MergeExec(partial))
    return P.SortExec([PhysicalSortExpr(col("revenue"), descending=True)], fin)


def q5_plan(customer, orders, lineitem, supplier, nation, region) -> P.ExecutionPlan:
    """region(ASIA) |x| nation |x| customer |x| orders(1994) |x| lineitem |x| supplier (on suppkey and
    c_nationkey = s_nationkey)."""
    return q5_final(q5_partial(q5_build_side(customer, orders, nation, region), q5_probe_side(lineitem), supplier))


def q12_plan(orders: P.ExecutionPlan, lineitem: P.ExecutionPlan) -> P.ExecutionPlan:
    """TPC-H Q12 (rust/benchmarks/tpch/queries/q12.sql): lineitem(IN-list on l_shipmode, two column-vs-column date
    comparisons, a receipt-date year) |x| orders, GROUP BY l_shipmode with two SUM(CASE WHEN ... THEN 1 ELSE 0 END).
    The filtered lineitem side is the (small, non-unique) build side, as `FROM lineitem JOIN orders` plans it."""
    ls, os_ = _schema_of(lineitem), _schema_of(orders)
    pred = E.InListExpr(col("l_shipmode"), [lit("MAIL"), lit("SHIP")])
    pred = pred.and_(col("l_commitdate") < col("l_receiptdate")).and_(col("l_shipdate") < col("l_commitdate"))
    pred = pred.and_(col("l_receiptdate") >= date32("1994-01-01")).and_(col("l_receiptdate") < date32("1995-01-01"))
    li = P.FilterExec(coerce(pred, ls), lineitem)
    li = P.ProjectionExec([(col("l_orderkey"), "l_orderkey"), (col("l_shipmode"), "l_shipmode")], li)
    od = P.ProjectionExec([(col("o_orderkey"), "o_orderkey"), (col("o_orderpriority"), "o_orderpriority")], orders)
    j = P.HashJoinExec(li, od, [("l_orderkey", "o_orderkey")], P.INNER)
    sj = _schema_of(j)
    urgent = col("o_orderpriority").eq(lit("1-URGENT")).or_(col("o_orderpriority").eq(lit("2-HIGH")))
    other = col("o_orderpriority").ne(lit("1-URGENT")).and_(col("o_orderpriority").ne(lit("2-HIGH")))
    one, zero = lit(1, E.INT64), lit(0, E.INT64)
    aggs = [Sum(coerce(E.CaseExpr(None, [(urgent, one)], zero), sj), "high_line_count"),
            Sum(coerce(E.CaseExpr(None, [(other, one)], zero), sj), "low_line_count")]
    group = [(col("l_shipmode"), "l_shipmode")]
    part = P.HashAggregateExec(P.PARTIAL, group, aggs, j)
    fin = P.HashAggregateExec(P.FINAL, group, [E.AggregateExpr("SUM", col("l_shipmode"), "high_line_count"),
                                               E.AggregateExpr("SUM", col("l_shipmode"), "low_line_count")], P.MergeExec(part))
    return P.SortExec([PhysicalSortExpr(col("l_shipmode"))], fin)
