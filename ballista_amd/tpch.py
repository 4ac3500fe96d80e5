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

# Algorithmic HBM bytes per input row: every needed column read once.
Q1_BYTES_PER_ROW = 4 * 8 + 4 + 2 * (4 + 1)      # 46: 4 x f64, Date32, 2 x Utf8 (offset + 1 char)
Q6_BYTES_PER_ROW = 4 + 3 * 8                    # 28


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


def q6_plan(scan: P.ExecutionPlan) -> P.ExecutionPlan:
    part = q6_stage1(scan)
    fin = P.HashAggregateExec(P.FINAL, [], [E.AggregateExpr("SUM", col("revenue[sum]"), "revenue")], P.MergeExec(part))
    return P.ProjectionExec([(col("revenue"), "revenue")], fin)


def q3_plan(customer: P.ExecutionPlan, orders: P.ExecutionPlan, lineitem: P.ExecutionPlan) -> P.ExecutionPlan:
    """customer(BUILDING) |x| orders(< 1995-03-15) |x| lineitem(> 1995-03-15); build side = left
    (collect-left hash join, from_proto.rs:253-276)."""
    cs, os_, ls = _schema_of(customer), _schema_of(orders), _schema_of(lineitem)
    cust = P.FilterExec(coerce(col("c_mktsegment").eq(lit("BUILDING")), cs), customer)
    cust = P.ProjectionExec([(col("c_custkey"), "c_custkey")], cust)
    ords = P.FilterExec(coerce(col("o_orderdate") < date32("1995-03-15"), os_), orders)
    j1 = P.HashJoinExec(cust, ords, [("c_custkey", "o_custkey")], P.INNER)
    j1 = P.ProjectionExec([(col(n), n) for n in ["o_orderkey", "o_orderdate", "o_shippriority"]], j1)
    li = P.FilterExec(coerce(col("l_shipdate") > date32("1995-03-15"), ls), lineitem)
    li = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "l_extendedprice", "l_discount"]], li)
    j2 = P.HashJoinExec(j1, li, [("o_orderkey", "l_orderkey")], P.INNER)
    s2 = _schema_of(j2)
    revenue = coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), s2)
    group = [(col("l_orderkey"), "l_orderkey"), (col("o_orderdate"), "o_orderdate"), (col("o_shippriority"), "o_shippriority")]
    part = P.HashAggregateExec(P.PARTIAL, group, [Sum(revenue, "revenue")], j2)
    fin = P.HashAggregateExec(P.FINAL, group, [E.AggregateExpr("SUM", col("l_orderkey"), "revenue")], P.MergeExec(part))
    proj = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]], fin)
    return P.SortExec([PhysicalSortExpr(col("revenue"), descending=True), PhysicalSortExpr(col("o_orderdate"))], proj)


def q5_plan(customer, orders, lineitem, supplier, nation, region) -> P.ExecutionPlan:
    """region(ASIA) |x| nation |x| customer |x| orders(1994) |x| lineitem |x| supplier (on suppkey and
    c_nationkey = s_nationkey)."""
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
    co = P.ProjectionExec([(col("o_orderkey"), "o_orderkey"), (col("n_nationkey"), "n_nationkey"), (col("n_name"), "n_name")], co)
    li = P.ProjectionExec([(col(n), n) for n in ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]], lineitem)
    col_ = P.HashJoinExec(co, li, [("o_orderkey", "l_orderkey")], P.INNER)
    # supplier joins on (suppkey, nationkey): the residual c_nationkey = s_nationkey is a second key pair
    sup = P.HashJoinExec(supplier, col_, [("s_suppkey", "l_suppkey"), ("s_nationkey", "n_nationkey")], P.INNER)
    s5 = _schema_of(sup)
    revenue = coerce(col("l_extendedprice") * (lit(1) - col("l_discount")), s5)
    group = [(col("n_name"), "n_name")]
    part = P.HashAggregateExec(P.PARTIAL, group, [Sum(revenue, "revenue")], sup)
    fin = P.HashAggregateExec(P.FINAL, group, [E.AggregateExpr("SUM", col("n_name"), "revenue")], P.MergeExec(part))
    return P.SortExec([PhysicalSortExpr(col("revenue"), descending=True)], fin)


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
