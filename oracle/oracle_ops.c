/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  PARITY UNPINNED BY THE REFERENCE.
 *
 * CPU restatement of the order-sensitive parts of the reference's hot path.
 * The reference (kyprifog/ballista) owns no compute kernels: an executor task
 * calls `plan.execute(partition)` (rust/executor/src/flight_service.rs:117-121)
 * on DataFusion operators built at rust/core/src/serde/physical_plan/
 * from_proto.rs:58-346, and those operators live in the un-vendored dependency
 * `datafusion`/`arrow` 4.0.0-SNAPSHOT, git rev 46161d2 (rust/Cargo.lock:77-80,
 * 497-500).  That source is absent, no reference test executes a plan
 * (SURVEY.md §4, §8(c)), so this file restates the published algorithm
 *  and is pinned only by (i) the reference's 20-row
 * lineitem fixture (rust/scheduler/testdata/lineitem/partition{0,1}.tbl) with
 * expected values computed by two independent engines (math.fsum and
 * pyarrow/Acero) and (ii) Acero cross-checks on seeded synthetic data
 * (tests/test_oracle.py).
 *
 * What is restated here:
 *   - oracle_batched_group_sum_*: HashAggregateExec accumulation order — per
 *     input batch, per group: sequential fold over that group's rows of the
 *     batch (arrow `sum` kernel, no SIMD feature), then add to the running
 *     state (from_proto.rs:173-252; Appendix A "Float summation order").
 *   - oracle_q1_partial / oracle_q6_partial: the whole stage-1 pipeline
 *     Scan -> FilterExec -> HashAggregateExec(Partial) of TPC-H Q1 / Q6
 *     (rust/benchmarks/tpch/queries/q1.sql, q6.sql; stage shape
 *     rust/scheduler/src/planner.rs:136-171) in DataFusion's structure: 32768-row
 *     batches (from_proto.rs:102), a materialised boolean array per predicate
 *     node, filter copying every projected column, one materialised f64 array
 *     per arithmetic node, per-group take + sum.  One partition per thread.
 *     This is also bench.py's `cpu_baseline` ("port").
 *
 * Compile with -ffp-contract=off: each arithmetic node rounds separately.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define BATCH 32768

/* ---- generic: batched, order-faithful group sums ------------------------- */

/* values[n] f64, valid[n] (NULL = all valid), gid[n] (dense group id, <0 = row
 * filtered out).  sum[g], cnt[g] (non-null rows), has[g] (0 => SUM is NULL).
 * Rows are consumed in `batch`-sized slices as HashAggregateExec sees them. */
void oracle_batched_group_sum_f64(const double *values, const uint8_t *valid,
                                  const int32_t *gid, int64_t n, int64_t batch,
                                  int32_t ngroups, double *sum, uint64_t *cnt, uint8_t *has)
{
    double *bsum = (double *)malloc(sizeof(double) * (size_t)(ngroups > 0 ? ngroups : 1));
    uint8_t *bhas = (uint8_t *)malloc((size_t)(ngroups > 0 ? ngroups : 1));
    for (int32_t g = 0; g < ngroups; g++) { sum[g] = 0.0; cnt[g] = 0; has[g] = 0; }
    for (int64_t b0 = 0; b0 < n; b0 += batch) {
        int64_t b1 = b0 + batch < n ? b0 + batch : n;
        memset(bhas, 0, (size_t)ngroups);
        for (int64_t i = b0; i < b1; i++) {
            int32_t g = gid[i];
            if (g < 0) continue;
            if (valid && !valid[i]) continue;
            if (!bhas[g]) { bsum[g] = values[i]; bhas[g] = 1; }
            else bsum[g] = bsum[g] + values[i];
            cnt[g]++;
        }
        for (int32_t g = 0; g < ngroups; g++) {
            if (!bhas[g]) continue;
            if (!has[g]) { sum[g] = bsum[g]; has[g] = 1; }
            else sum[g] = sum[g] + bsum[g];
        }
    }
    free(bsum); free(bhas);
}

void oracle_batched_group_sum_i64(const int64_t *values, const uint8_t *valid,
                                  const int32_t *gid, int64_t n,
                                  int32_t ngroups, int64_t *sum, uint64_t *cnt, uint8_t *has)
{
    for (int32_t g = 0; g < ngroups; g++) { sum[g] = 0; cnt[g] = 0; has[g] = 0; }
    for (int64_t i = 0; i < n; i++) {
        int32_t g = gid[i];
        if (g < 0) continue;
        if (valid && !valid[i]) continue;
        sum[g] = (int64_t)((uint64_t)sum[g] + (uint64_t)values[i]);
        cnt[g]++; has[g] = 1;
    }
}

/* ---- TPC-H Q1 stage 1 in DataFusion's structure -------------------------- */

/* Output layout per partition p, group slot g in [0,4):
 *   keys[p*4+g]   = returnflag | linestatus<<8   (0 = unused slot)
 *   state[(p*4+g)*8 + k], k = sum_qty, sum_base, sum_disc_price, sum_charge,
 *                              avg_qty.sum, avg_price.sum, avg_disc.sum, (unused)
 *   count[p*4+g]  = COUNT(*) = every AVG count (columns are non-null)
 * Partitions are contiguous row ranges, one per thread. */
typedef struct {
    double qty[BATCH], price[BATCH], disc[BATCH], tax[BATCH];
    uint8_t flag[BATCH], status[BATCH];
    uint8_t pred[BATCH];
    double one_minus_disc[BATCH], disc_price[BATCH], one_plus_tax[BATCH], charge[BATCH];
    int32_t idx[4][BATCH];
} q1_scratch;

static double take_sum(const double *v, const int32_t *idx, int32_t n)
{
    double s = v[idx[0]];
    for (int32_t i = 1; i < n; i++) s = s + v[idx[i]];
    return s;
}

void oracle_q1_partial(const double *l_quantity, const double *l_extendedprice,
                       const double *l_discount, const double *l_tax,
                       const int32_t *l_shipdate,
                       const int32_t *flag_off, const uint8_t *flag_data,
                       const int32_t *status_off, const uint8_t *status_data,
                       int64_t n, int32_t date_lit, int32_t n_partitions, int32_t n_threads,
                       uint16_t *keys, double *state, uint64_t *count)
{
    memset(keys, 0, sizeof(uint16_t) * 4 * (size_t)n_partitions);
    memset(state, 0, sizeof(double) * 32 * (size_t)n_partitions);
    memset(count, 0, sizeof(uint64_t) * 4 * (size_t)n_partitions);
    int64_t per = (n + n_partitions - 1) / n_partitions;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        q1_scratch *s = (q1_scratch *)malloc(sizeof(q1_scratch));
        uint16_t *pk = keys + p * 4;
        double *ps = state + p * 32;
        uint64_t *pc = count + p * 4;
        int64_t r0 = per * p, r1 = r0 + per < n ? r0 + per : n;
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0);
            /* FilterExec: predicate array, then copy surviving rows of every column */
            for (int32_t i = 0; i < bn; i++) s->pred[i] = l_shipdate[b0 + i] <= date_lit;
            int32_t m = 0;
            for (int32_t i = 0; i < bn; i++) {
                if (!s->pred[i]) continue;
                int64_t r = b0 + i;
                s->qty[m] = l_quantity[r]; s->price[m] = l_extendedprice[r];
                s->disc[m] = l_discount[r]; s->tax[m] = l_tax[r];
                s->flag[m] = flag_data[flag_off[r]];     /* 1-char Utf8 values */
                s->status[m] = status_data[status_off[r]];
                m++;
            }
            if (m == 0) continue;
            /* aggregate input expressions, one array per BinaryExpr node */
            for (int32_t i = 0; i < m; i++) s->one_minus_disc[i] = 1.0 - s->disc[i];
            for (int32_t i = 0; i < m; i++) s->disc_price[i] = s->price[i] * s->one_minus_disc[i];
            for (int32_t i = 0; i < m; i++) s->one_plus_tax[i] = 1.0 + s->tax[i];
            for (int32_t i = 0; i < m; i++) s->charge[i] = s->disc_price[i] * s->one_plus_tax[i];
            /* group rows of this batch */
            int32_t gn[4] = {0, 0, 0, 0};
            for (int32_t i = 0; i < m; i++) {
                uint16_t k = (uint16_t)(s->flag[i] | (s->status[i] << 8));
                int g = 0;
                for (; g < 4; g++) { if (pk[g] == k) break; if (pk[g] == 0) { pk[g] = k; break; } }
                s->idx[g][gn[g]++] = i;
            }
            /* per group: take + sum, then update running state */
            for (int g = 0; g < 4; g++) {
                if (!gn[g]) continue;
                double d[7];
                d[0] = take_sum(s->qty, s->idx[g], gn[g]);
                d[1] = take_sum(s->price, s->idx[g], gn[g]);
                d[2] = take_sum(s->disc_price, s->idx[g], gn[g]);
                d[3] = take_sum(s->charge, s->idx[g], gn[g]);
                d[4] = d[0]; d[5] = d[1];
                d[6] = take_sum(s->disc, s->idx[g], gn[g]);
                if (pc[g] == 0) for (int k = 0; k < 7; k++) ps[g * 8 + k] = d[k];
                else for (int k = 0; k < 7; k++) ps[g * 8 + k] = ps[g * 8 + k] + d[k];
                pc[g] += (uint64_t)gn[g];
            }
        }
        free(s);
    }
}

/* ---- TPC-H Q6 stage 1 ------------------------------------------------------ */
/* sum(l_extendedprice * l_discount) where shipdate in [d0,d1) and
 * discount between lo and hi and quantity < qmax.  Per partition: sum, count
 * of selected rows (has = count > 0). */
void oracle_q6_partial(const double *l_quantity, const double *l_extendedprice,
                       const double *l_discount, const int32_t *l_shipdate,
                       int64_t n, int32_t d0, int32_t d1, double lo, double hi, double qmax,
                       int32_t n_partitions, int32_t n_threads,
                       double *sum, uint64_t *count)
{
    int64_t per = (n + n_partitions - 1) / n_partitions;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        uint8_t *a = (uint8_t *)malloc(BATCH), *b = (uint8_t *)malloc(BATCH);
        double *price = (double *)malloc(sizeof(double) * BATCH);
        double *disc = (double *)malloc(sizeof(double) * BATCH);
        double *prod = (double *)malloc(sizeof(double) * BATCH);
        double acc = 0.0; uint64_t cnt = 0;
        int64_t r0 = per * p, r1 = r0 + per < n ? r0 + per : n;
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0);
            /* one boolean array per comparison, AND-ed pairwise */
            for (int32_t i = 0; i < bn; i++) a[i] = l_shipdate[b0 + i] >= d0;
            for (int32_t i = 0; i < bn; i++) b[i] = l_shipdate[b0 + i] < d1;
            for (int32_t i = 0; i < bn; i++) a[i] = a[i] & b[i];
            for (int32_t i = 0; i < bn; i++) b[i] = l_discount[b0 + i] >= lo;
            for (int32_t i = 0; i < bn; i++) a[i] = a[i] & b[i];
            for (int32_t i = 0; i < bn; i++) b[i] = l_discount[b0 + i] <= hi;
            for (int32_t i = 0; i < bn; i++) a[i] = a[i] & b[i];
            for (int32_t i = 0; i < bn; i++) b[i] = l_quantity[b0 + i] < qmax;
            for (int32_t i = 0; i < bn; i++) a[i] = a[i] & b[i];
            int32_t m = 0;
            for (int32_t i = 0; i < bn; i++) {
                if (!a[i]) continue;
                price[m] = l_extendedprice[b0 + i]; disc[m] = l_discount[b0 + i]; m++;
            }
            if (!m) continue;
            for (int32_t i = 0; i < m; i++) prod[i] = price[i] * disc[i];
            double s = prod[0];
            for (int32_t i = 1; i < m; i++) s = s + prod[i];
            acc = cnt ? acc + s : s;
            cnt += (uint64_t)m;
        }
        sum[p] = acc; count[p] = cnt;
        free(a); free(b); free(price); free(disc); free(prod);
    }
}

/* ---- TPC-H Q3 / Q5 in DataFusion's structure (bench.py cpu_baseline, kind "port") --------------------
 * rust/benchmarks/tpch/queries/q3.sql, q5.sql.  Collect-left hash joins (from_proto.rs:253-276): the build side
 * is drained into one hash map (key -> row), every probe partition (one per thread at a time) walks its rows in
 * 32768-row batches: predicate array, filter copy of the projected columns, probe, gather of both sides, then the
 * partial aggregate by hash map; the partial states are merged at the end (Final).  Unique build keys (every
 * TPC-H join here), so the maps hold one row per key. */
typedef struct { int64_t *key; int32_t *row; uint64_t mask; } imap;

static uint64_t mix64u(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void imap_init(imap *m, int64_t n) {
    uint64_t cap = 1024;
    while (cap < 2ull * (uint64_t)n) cap <<= 1;
    m->mask = cap - 1;
    m->key = (int64_t *)malloc(sizeof(int64_t) * cap);
    m->row = (int32_t *)malloc(sizeof(int32_t) * cap);
    for (uint64_t i = 0; i < cap; i++) m->row[i] = -1;
}
static void imap_free(imap *m) { free(m->key); free(m->row); }
static void imap_put(imap *m, int64_t k, int32_t row) {
    uint64_t s = mix64u((uint64_t)k) & m->mask;
    while (m->row[s] >= 0 && m->key[s] != k) s = (s + 1) & m->mask;
    m->key[s] = k; m->row[s] = row;
}
static int32_t imap_get(const imap *m, int64_t k) {
    uint64_t s = mix64u((uint64_t)k) & m->mask;
    while (m->row[s] >= 0) { if (m->key[s] == k) return m->row[s]; s = (s + 1) & m->mask; }
    return -1;
}
static int64_t key_at(const void *keys, int32_t key64, int64_t i) {
    return key64 ? ((const int64_t *)keys)[i] : (int64_t)((const int32_t *)keys)[i];
}

/* Q3: customer(BUILDING) |x| orders(< cutoff) |x| lineitem(> cutoff), GROUP BY l_orderkey, o_orderdate, o_shippriority.
 * Returns the number of groups; *revenue_total = sum of the group sums (a checksum for the caller). */
int64_t oracle_q3_join_port(const int32_t *c_custkey, const int32_t *seg_off, const uint8_t *seg_data, int64_t n_cust,
                            const void *o_orderkey, const int32_t *o_custkey, const int32_t *o_orderdate,
                            const int32_t *o_shippriority, int64_t n_ord,
                            const void *l_orderkey, const double *l_extendedprice, const double *l_discount,
                            const int32_t *l_shipdate, int64_t n_li, int32_t key64, int32_t cutoff,
                            int32_t n_partitions, int32_t n_threads, double *revenue_total)
{
    /* build 1: FilterExec(c_mktsegment = 'BUILDING') -> projection c_custkey */
    imap cust;
    int64_t n_b = 0;
    for (int64_t i = 0; i < n_cust; i++)
        n_b += (seg_off[i + 1] - seg_off[i] == 8 && memcmp(seg_data + seg_off[i], "BUILDING", 8) == 0);
    imap_init(&cust, n_b);
    for (int64_t i = 0; i < n_cust; i++)
        if (seg_off[i + 1] - seg_off[i] == 8 && memcmp(seg_data + seg_off[i], "BUILDING", 8) == 0) imap_put(&cust, c_custkey[i], (int32_t)i);
    /* probe 1 (orders partitions) -> j1 rows (o_orderkey, o_orderdate, o_shippriority), per partition, then concatenated */
    int64_t per = (n_ord + n_partitions - 1) / n_partitions;
    int64_t **j1_key = (int64_t **)calloc((size_t)n_partitions, sizeof(void *));
    int32_t **j1_date = (int32_t **)calloc((size_t)n_partitions, sizeof(void *));
    int32_t **j1_prio = (int32_t **)calloc((size_t)n_partitions, sizeof(void *));
    int64_t *j1_n = (int64_t *)calloc((size_t)n_partitions, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        int64_t r0 = per * p, r1 = r0 + per < n_ord ? r0 + per : n_ord;
        if (r1 <= r0) continue;
        int64_t cap = r1 - r0, m = 0;
        int64_t *k = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        int32_t *d = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap), *s = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        uint8_t pred[BATCH];
        int64_t fk[BATCH]; int32_t fc[BATCH], fd[BATCH], fs[BATCH];
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0), fm = 0;
            for (int32_t i = 0; i < bn; i++) pred[i] = o_orderdate[b0 + i] < cutoff;
            for (int32_t i = 0; i < bn; i++) {                          /* filter copies every column */
                if (!pred[i]) continue;
                fk[fm] = key_at(o_orderkey, key64, b0 + i); fc[fm] = o_custkey[b0 + i]; fd[fm] = o_orderdate[b0 + i]; fs[fm] = o_shippriority[b0 + i]; fm++;
            }
            for (int32_t i = 0; i < fm; i++)
                if (imap_get(&cust, fc[i]) >= 0) { k[m] = fk[i]; d[m] = fd[i]; s[m] = fs[i]; m++; }
        }
        j1_key[p] = k; j1_date[p] = d; j1_prio[p] = s; j1_n[p] = m;
    }
    imap_free(&cust);
    int64_t n_j1 = 0;
    for (int32_t p = 0; p < n_partitions; p++) n_j1 += j1_n[p];
    int64_t *bk = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_j1 + 1));
    int32_t *bd = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_j1 + 1)), *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_j1 + 1));
    int64_t at = 0;
    for (int32_t p = 0; p < n_partitions; p++) {
        if (!j1_n[p]) { free(j1_key[p]); free(j1_date[p]); free(j1_prio[p]); continue; }
        memcpy(bk + at, j1_key[p], sizeof(int64_t) * (size_t)j1_n[p]);
        memcpy(bd + at, j1_date[p], sizeof(int32_t) * (size_t)j1_n[p]);
        memcpy(bp + at, j1_prio[p], sizeof(int32_t) * (size_t)j1_n[p]);
        at += j1_n[p];
        free(j1_key[p]); free(j1_date[p]); free(j1_prio[p]);
    }
    free(j1_key); free(j1_date); free(j1_prio); free(j1_n);
    /* build 2 */
    imap ords;
    imap_init(&ords, n_j1);
    for (int64_t i = 0; i < n_j1; i++) imap_put(&ords, bk[i], (int32_t)i);
    /* probe 2 + partial aggregate per lineitem partition: the group is the build row (unique order key) */
    double *gsum = (double *)calloc((size_t)(n_j1 + 1), sizeof(double));
    uint8_t *ghas = (uint8_t *)calloc((size_t)(n_j1 + 1), 1);
    per = (n_li + n_partitions - 1) / n_partitions;
    double **psum = (double **)calloc((size_t)n_partitions, sizeof(void *));
    int32_t **prow = (int32_t **)calloc((size_t)n_partitions, sizeof(void *));
    int64_t *pn = (int64_t *)calloc((size_t)n_partitions, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        int64_t r0 = per * p, r1 = r0 + per < n_li ? r0 + per : n_li;
        if (r1 <= r0) continue;
        /* partial state of this partition: hash map build row -> slot */
        imap part;
        imap_init(&part, (r1 - r0) / 2 + 16);
        int64_t cap = (r1 - r0) / 2 + 16, ng = 0;
        double *sum = (double *)malloc(sizeof(double) * (size_t)cap);
        int32_t *row = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        uint8_t pred[BATCH];
        int64_t fk[BATCH]; double fp[BATCH], fdisc[BATCH], one_minus[BATCH], rev[BATCH];
        int32_t hit[BATCH];
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0), fm = 0, hm = 0;
            for (int32_t i = 0; i < bn; i++) pred[i] = l_shipdate[b0 + i] > cutoff;
            for (int32_t i = 0; i < bn; i++) {
                if (!pred[i]) continue;
                fk[fm] = key_at(l_orderkey, key64, b0 + i); fp[fm] = l_extendedprice[b0 + i]; fdisc[fm] = l_discount[b0 + i]; fm++;
            }
            for (int32_t i = 0; i < fm; i++) {                          /* probe + gather of the matching rows */
                int32_t r = imap_get(&ords, fk[i]);
                if (r < 0) continue;
                hit[hm] = r; fp[hm] = fp[i]; fdisc[hm] = fdisc[i]; hm++;
            }
            for (int32_t i = 0; i < hm; i++) one_minus[i] = 1.0 - fdisc[i];
            for (int32_t i = 0; i < hm; i++) rev[i] = fp[i] * one_minus[i];
            for (int32_t i = 0; i < hm; i++) {
                int32_t s = imap_get(&part, hit[i]);
                if (s < 0) {
                    if (ng == cap) { cap *= 2; sum = (double *)realloc(sum, sizeof(double) * (size_t)cap); row = (int32_t *)realloc(row, sizeof(int32_t) * (size_t)cap); }
                    s = (int32_t)ng++; imap_put(&part, hit[i], s); sum[s] = rev[i]; row[s] = hit[i];
                } else sum[s] = sum[s] + rev[i];
            }
        }
        imap_free(&part);
        psum[p] = sum; prow[p] = row; pn[p] = ng;
    }
    /* Final: merge the partial states in partition order */
    int64_t n_groups = 0;
    double total = 0.0;
    for (int32_t p = 0; p < n_partitions; p++) {
        for (int64_t i = 0; i < pn[p]; i++) {
            int32_t r = prow[p][i];
            if (!ghas[r]) { ghas[r] = 1; gsum[r] = psum[p][i]; n_groups++; } else gsum[r] = gsum[r] + psum[p][i];
        }
        free(psum[p]); free(prow[p]);
    }
    for (int64_t i = 0; i < n_j1; i++) if (ghas[i]) total += gsum[i];
    free(psum); free(prow); free(pn); free(gsum); free(ghas); free(bk); free(bd); free(bp);
    imap_free(&ords);
    if (revenue_total) *revenue_total = total;
    return n_groups;
}

/* Q5: region(ASIA=2) |x| nation |x| customer |x| orders(in [d0,d1)) |x| lineitem |x| supplier(on suppkey, nationkey),
 * GROUP BY nation.  nation_region[25]; revenue[25] out (0 for nations outside the region). */
void oracle_q5_join_port(const int32_t *nation_region, int32_t region,
                         const int32_t *c_custkey, const int32_t *c_nationkey, int64_t n_cust,
                         const int32_t *s_suppkey, const int32_t *s_nationkey, int64_t n_supp,
                         const void *o_orderkey, const int32_t *o_custkey, const int32_t *o_orderdate, int64_t n_ord,
                         const void *l_orderkey, const int32_t *l_suppkey, const double *l_extendedprice, const double *l_discount,
                         int64_t n_li, int32_t key64, int32_t d0, int32_t d1, int32_t n_partitions, int32_t n_threads, double *revenue)
{
    /* nation |x| region, then customer probe (build = the region's nations) */
    imap cust;
    int64_t n_c = 0;
    for (int64_t i = 0; i < n_cust; i++) n_c += nation_region[c_nationkey[i]] == region;
    imap_init(&cust, n_c);
    for (int64_t i = 0; i < n_cust; i++)
        if (nation_region[c_nationkey[i]] == region) imap_put(&cust, c_custkey[i], c_nationkey[i]);     /* value = nation */
    imap supp;
    imap_init(&supp, n_supp);
    for (int64_t i = 0; i < n_supp; i++) imap_put(&supp, s_suppkey[i], s_nationkey[i]);
    /* orders probe -> (o_orderkey, nation) */
    int64_t per = (n_ord + n_partitions - 1) / n_partitions;
    int64_t **ok = (int64_t **)calloc((size_t)n_partitions, sizeof(void *));
    int32_t **on = (int32_t **)calloc((size_t)n_partitions, sizeof(void *));
    int64_t *cnt = (int64_t *)calloc((size_t)n_partitions, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        int64_t r0 = per * p, r1 = r0 + per < n_ord ? r0 + per : n_ord, m = 0;
        if (r1 <= r0) continue;
        int64_t *k = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r1 - r0));
        int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(r1 - r0));
        uint8_t a[BATCH], b[BATCH];
        int64_t fk[BATCH]; int32_t fc[BATCH];
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0), fm = 0;
            for (int32_t i = 0; i < bn; i++) a[i] = o_orderdate[b0 + i] >= d0;
            for (int32_t i = 0; i < bn; i++) b[i] = o_orderdate[b0 + i] < d1;
            for (int32_t i = 0; i < bn; i++) a[i] = a[i] & b[i];
            for (int32_t i = 0; i < bn; i++) { if (!a[i]) continue; fk[fm] = key_at(o_orderkey, key64, b0 + i); fc[fm] = o_custkey[b0 + i]; fm++; }
            for (int32_t i = 0; i < fm; i++) { int32_t nat = imap_get(&cust, fc[i]); if (nat >= 0) { k[m] = fk[i]; nn[m] = nat; m++; } }
        }
        ok[p] = k; on[p] = nn; cnt[p] = m;
    }
    int64_t n_co = 0;
    for (int32_t p = 0; p < n_partitions; p++) n_co += cnt[p];
    imap ords;
    imap_init(&ords, n_co);
    for (int32_t p = 0; p < n_partitions; p++) {
        for (int64_t i = 0; i < cnt[p]; i++) imap_put(&ords, ok[p][i], on[p][i]);
        free(ok[p]); free(on[p]);
    }
    free(ok); free(on); free(cnt);
    imap_free(&cust);
    /* lineitem probe, supplier probe on (suppkey, nation), partial aggregate by nation */
    per = (n_li + n_partitions - 1) / n_partitions;
    double *part = (double *)calloc((size_t)n_partitions * 25, sizeof(double));
    uint8_t *phas = (uint8_t *)calloc((size_t)n_partitions * 25, 1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int32_t p = 0; p < n_partitions; p++) {
        int64_t r0 = per * p, r1 = r0 + per < n_li ? r0 + per : n_li;
        double fp[BATCH], fdisc[BATCH], one_minus[BATCH], rev[BATCH];
        int32_t fs[BATCH], fnat[BATCH];
        for (int64_t b0 = r0; b0 < r1; b0 += BATCH) {
            int32_t bn = (int32_t)(b0 + BATCH < r1 ? BATCH : r1 - b0), hm = 0, sm = 0;
            for (int32_t i = 0; i < bn; i++) {
                int32_t nat = imap_get(&ords, key_at(l_orderkey, key64, b0 + i));
                if (nat < 0) continue;
                fnat[hm] = nat; fs[hm] = l_suppkey[b0 + i]; fp[hm] = l_extendedprice[b0 + i]; fdisc[hm] = l_discount[b0 + i]; hm++;
            }
            for (int32_t i = 0; i < hm; i++) {
                if (imap_get(&supp, fs[i]) != fnat[i]) continue;
                fnat[sm] = fnat[i]; fp[sm] = fp[i]; fdisc[sm] = fdisc[i]; sm++;
            }
            for (int32_t i = 0; i < sm; i++) one_minus[i] = 1.0 - fdisc[i];
            for (int32_t i = 0; i < sm; i++) rev[i] = fp[i] * one_minus[i];
            double bs[25]; uint8_t bh[25];
            memset(bh, 0, 25);
            for (int32_t i = 0; i < sm; i++) { int g = fnat[i]; if (!bh[g]) { bh[g] = 1; bs[g] = rev[i]; } else bs[g] = bs[g] + rev[i]; }
            for (int g = 0; g < 25; g++) {
                if (!bh[g]) continue;
                if (!phas[p * 25 + g]) { phas[p * 25 + g] = 1; part[p * 25 + g] = bs[g]; } else part[p * 25 + g] = part[p * 25 + g] + bs[g];
            }
        }
    }
    for (int g = 0; g < 25; g++) {
        revenue[g] = 0.0;
        int has = 0;
        for (int32_t p = 0; p < n_partitions; p++) {
            if (!phas[p * 25 + g]) continue;
            revenue[g] = has ? revenue[g] + part[p * 25 + g] : part[p * 25 + g];
            has = 1;
        }
    }
    free(part); free(phas);
    imap_free(&ords); imap_free(&supp);
}
