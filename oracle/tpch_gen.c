/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU generator for synthetic TPC-H-shaped columns (SURVEY.md Appendix B, §8(d)
 * "Synthetic inputs").  It is the checker-side twin of the HIP generator in
 * ballista_amd/csrc/kernels_gen.hip: both are written independently from the same
 * written spec (DESIGN.md "Synthetic data spec") and tests/test_generator.py
 * asserts they are bit-identical.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Schema follows the reference's TPC-H table definitions
 * (rust/benchmarks/tpch/src/main.rs:267-360): keys Int32 (Int64 variant for
 * SF1000), money/qty Float64, dates Date32, flags/names Utf8.
 *
 * Every value is a pure function of (seed, stream id, row index) through a
 * splitmix64-style counter hash, so any row range can be generated on its own.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* stream ids (one per generated attribute) */
enum {
    ST_QTY = 1, ST_PART = 2, ST_DISC = 3, ST_TAX = 4, ST_SHIP = 5, ST_COMMIT = 6,
    ST_RECEIPT = 7, ST_RFLAG = 8, ST_SUPP = 9, ST_ODATE = 10, ST_OCUST = 11,
    ST_CSEG = 12, ST_CNATION = 13, ST_SNATION = 14, ST_ROT = 15
};

static inline uint64_t rnd(uint64_t seed, uint32_t stream, uint64_t idx) {
    return mix64(mix64(seed + 0xD1342543DE82EF95ull * (uint64_t)stream) ^ (idx * 0x2545F4914F6CDD1Dull));
}

/* lineitem row -> order index: orders come in groups of 7 holding 1..7 lines
 * (28 lines per group, mean 4 per order as in TPC-H), the sequence rotated by a
 * per-group hash. */
static inline uint64_t line_to_order(uint64_t seed, uint64_t row, uint64_t n_orders, uint32_t *line_no) {
    uint64_t g = row / 28u;
    uint32_t w = (uint32_t)(row % 28u);
    uint32_t rot = (uint32_t)(rnd(seed, ST_ROT, g) % 7u);
    uint32_t acc = 0, j = 0;
    for (j = 0; j < 7; j++) {
        uint32_t cnt = 1u + ((j + rot) % 7u);
        if (w < acc + cnt) break;
        acc += cnt;
    }
    if (line_no) *line_no = w - acc;
    return (g * 7u + j) % n_orders;
}

static inline int32_t order_date(uint64_t seed, uint64_t o) {
    /* uniform in [1992-01-01 = 8035, 1998-08-02 = 10440] */
    return 8035 + (int32_t)(rnd(seed, ST_ODATE, o) % 2406u);
}

/*
 * Generate lineitem rows [row0, row0+n).  Any output pointer may be NULL.
 * key64 != 0 writes l_orderkey / l_suppkey keys as int64 into *_i64 pointers.
 * flag_off / status_off receive n+1 Arrow Utf8 offsets each (1 char per row),
 * relative to row0 (offset[0] = 0).
 */
void tpch_gen_lineitem(uint64_t seed, uint64_t row0, uint64_t n,
                       uint64_t n_orders, uint64_t n_parts, uint64_t n_supp,
                       int32_t *l_orderkey, int64_t *l_orderkey_i64,
                       int32_t *l_suppkey,
                       double *l_quantity, double *l_extendedprice,
                       double *l_discount, double *l_tax,
                       int32_t *l_shipdate, int32_t *l_commitdate, int32_t *l_receiptdate,
                       int32_t *flag_off, uint8_t *flag_data,
                       int32_t *status_off, uint8_t *status_data)
{
    /* rows are independent (counter hash): generate them on all host cores */
#pragma omp parallel for schedule(static)
    for (uint64_t k = 0; k < n; k++) {
        uint64_t row = row0 + k;
        uint64_t o = line_to_order(seed, row, n_orders, NULL);
        int32_t odate = order_date(seed, o);
        uint32_t qty = 1u + (uint32_t)(rnd(seed, ST_QTY, row) % 50u);
        uint64_t partkey = 1u + rnd(seed, ST_PART, row) % n_parts;
        uint64_t retail_cents = 90000u + ((partkey / 10u) % 20001u) + 100u * (partkey % 1000u);
        int32_t ship = odate + 1 + (int32_t)(rnd(seed, ST_SHIP, row) % 121u);
        int32_t commit = odate + 30 + (int32_t)(rnd(seed, ST_COMMIT, row) % 61u);
        int32_t receipt = ship + 1 + (int32_t)(rnd(seed, ST_RECEIPT, row) % 30u);
        if (l_orderkey) l_orderkey[k] = (int32_t)(o + 1u);
        if (l_orderkey_i64) l_orderkey_i64[k] = (int64_t)(o + 1u);
        if (l_suppkey) l_suppkey[k] = (int32_t)(1u + rnd(seed, ST_SUPP, row) % n_supp);
        if (l_quantity) l_quantity[k] = (double)qty;
        if (l_extendedprice) l_extendedprice[k] = (double)((uint64_t)qty * retail_cents) / 100.0;
        if (l_discount) l_discount[k] = (double)(rnd(seed, ST_DISC, row) % 11u) / 100.0;
        if (l_tax) l_tax[k] = (double)(rnd(seed, ST_TAX, row) % 9u) / 100.0;
        if (l_shipdate) l_shipdate[k] = ship;
        if (l_commitdate) l_commitdate[k] = commit;
        if (l_receiptdate) l_receiptdate[k] = receipt;
        if (flag_data) {
            uint8_t f = 'N';
            if (receipt <= 9298) f = (rnd(seed, ST_RFLAG, row) & 1u) ? 'R' : 'A';
            flag_data[k] = f;
        }
        if (status_data) status_data[k] = (ship > 9298) ? 'O' : 'F';
        if (flag_off) flag_off[k] = (int32_t)k;
        if (status_off) status_off[k] = (int32_t)k;
    }
    if (flag_off) flag_off[n] = (int32_t)n;
    if (status_off) status_off[n] = (int32_t)n;
}

/* orders rows [row0,row0+n): o_orderkey = row+1, o_custkey never = 0 mod 3 */
void tpch_gen_orders(uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_cust,
                     int32_t *o_orderkey, int64_t *o_orderkey_i64, int32_t *o_custkey,
                     int32_t *o_orderdate, int32_t *o_shippriority)
{
    for (uint64_t k = 0; k < n; k++) {
        uint64_t o = row0 + k;
        uint64_t c = 1u + rnd(seed, ST_OCUST, o) % n_cust;
        if (c % 3u == 0) c -= 1u;
        if (o_orderkey) o_orderkey[k] = (int32_t)(o + 1u);
        if (o_orderkey_i64) o_orderkey_i64[k] = (int64_t)(o + 1u);
        if (o_custkey) o_custkey[k] = (int32_t)c;
        if (o_orderdate) o_orderdate[k] = order_date(seed, o);
        if (o_shippriority) o_shippriority[k] = 0;
    }
}

static const char *SEGMENTS[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"};

/* customer rows [0,n) (whole table: Utf8 offsets are a running sum).
 * seg_data must hold 10*n bytes; returns bytes used. */
uint64_t tpch_gen_customer(uint64_t seed, uint64_t n,
                           int32_t *c_custkey, int32_t *c_nationkey,
                           int32_t *seg_off, uint8_t *seg_data)
{
    uint64_t pos = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (c_custkey) c_custkey[i] = (int32_t)(i + 1u);
        if (c_nationkey) c_nationkey[i] = (int32_t)(rnd(seed, ST_CNATION, i) % 25u);
        const char *s = SEGMENTS[rnd(seed, ST_CSEG, i) % 5u];
        size_t len = strlen(s);
        if (seg_off) seg_off[i] = (int32_t)pos;
        if (seg_data) memcpy(seg_data + pos, s, len);
        pos += len;
    }
    if (seg_off) seg_off[n] = (int32_t)pos;
    return pos;
}

void tpch_gen_supplier(uint64_t seed, uint64_t n, int32_t *s_suppkey, int32_t *s_nationkey)
{
    for (uint64_t i = 0; i < n; i++) {
        if (s_suppkey) s_suppkey[i] = (int32_t)(i + 1u);
        if (s_nationkey) s_nationkey[i] = (int32_t)(rnd(seed, ST_SNATION, i) % 25u);
    }
}
