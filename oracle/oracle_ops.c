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
