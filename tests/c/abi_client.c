/* A plain-C client of include/ballista_hip.h — what a cgo / JNI / Rust-FFI binding of the executor would do
 * (INTEGRATION.md): host columns in, a Filter -> HashAggregate(Partial) -> Merge -> HashAggregate(Final) -> Sort plan
 * built from postfix expressions, results read back, errors reported through status codes.  The expected values
 * are computed right here with scalar loops (an independent check; no oracle, no Python).
 *
 *   SELECT k, SUM(x * (1 - d)) AS s, COUNT(*) AS n FROM t WHERE day <= 9500 AND d <= 0.05 GROUP BY k ORDER BY k
 *
 * build: gcc -std=c11 -O1 -I include tests/c/abi_client.c -L ballista_amd/lib -lballista_hip -Wl,-rpath,... -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ballista_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        bhip_status st_ = (call);                                                                     \
        if (st_ != BHIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, bhip_last_error()); return 1; } \
    } while (0)

static bhip_expr_node col_(const char* n) { bhip_expr_node e; memset(&e, 0, sizeof(e)); e.kind = BHIP_EXPR_COLUMN; e.name = n; return e; }
static bhip_expr_node bin_(const char* op) { bhip_expr_node e; memset(&e, 0, sizeof(e)); e.kind = BHIP_EXPR_BINARY; e.name = op; e.n_args = 2; return e; }
static bhip_expr_node f64_(double v) { bhip_expr_node e; memset(&e, 0, sizeof(e)); e.kind = BHIP_EXPR_LITERAL; e.dtype = BHIP_FLOAT64; e.f64 = v; return e; }
static bhip_expr_node i_(int32_t dtype, int64_t v) { bhip_expr_node e; memset(&e, 0, sizeof(e)); e.kind = BHIP_EXPR_LITERAL; e.dtype = dtype; e.i64 = v; return e; }

int main(void) {
    enum { N = 50000, G = 3 };
    static int32_t k[N], day[N];
    static double x[N], d[N];
    uint64_t s = 88172645463325252ull;
    double want_s[G] = {0, 0, 0};
    long want_n[G] = {0, 0, 0};
    for (int i = 0; i < N; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        k[i] = (int32_t)(s % G) * 10 - 10;                       /* -10, 0, 10 */
        day[i] = 9000 + (int32_t)((s >> 8) % 1000);
        x[i] = (double)((s >> 20) % 100000) / 100.0;
        d[i] = (double)((s >> 40) % 11) / 100.0;
    }
    for (int i = 0; i < N; ++i)
        if (day[i] <= 9500 && d[i] <= 0.05) { const int g = k[i] / 10 + 1; want_s[g] += x[i] * (1.0 - d[i]); want_n[g] += 1; }

    bhip_ctx* ctx = NULL;
    CHECK(bhip_ctx_create(0, &ctx));
    bhip_column_desc cols[4];
    memset(cols, 0, sizeof(cols));
    cols[0].name = "k";   cols[0].dtype = BHIP_INT32;   cols[0].data = k;
    cols[1].name = "day"; cols[1].dtype = BHIP_DATE32;  cols[1].data = day;
    cols[2].name = "x";   cols[2].dtype = BHIP_FLOAT64; cols[2].data = x;
    cols[3].name = "d";   cols[3].dtype = BHIP_FLOAT64; cols[3].data = d;
    bhip_batch* halves[2];
    CHECK(bhip_batch_from_host(ctx, 4, cols, N / 2, &halves[0]));
    cols[0].data = k + N / 2; cols[1].data = day + N / 2; cols[2].data = x + N / 2; cols[3].data = d + N / 2;
    CHECK(bhip_batch_from_host(ctx, 4, cols, N - N / 2, &halves[1]));

    /* two partitions of one batch each */
    const int32_t offsets[3] = {0, 1, 2};
    bhip_plan *scan, *flt, *part, *merged, *fin, *sorted;
    CHECK(bhip_plan_memory(ctx, 2, offsets, halves, &scan));

    /* day <= 9500 AND d <= 0.05   (postfix) */
    bhip_expr_node p[] = {col_("day"), i_(BHIP_DATE32, 9500), bin_("LtEq"), col_("d"), f64_(0.05), bin_("LtEq"), bin_("And")};
    bhip_expr pred = {p, (int32_t)(sizeof(p) / sizeof(p[0]))};
    CHECK(bhip_plan_filter(scan, &pred, &flt));

    bhip_expr_node kx[] = {col_("k")};
    bhip_expr key = {kx, 1};
    const char* key_names[] = {"k"};
    bhip_expr_node sx[] = {col_("x"), f64_(1.0), col_("d"), bin_("Minus"), bin_("Multiply")};   /* x * (1 - d) */
    bhip_expr_node one[] = {i_(BHIP_UINT8, 1)};
    bhip_aggregate aggs[2];
    aggs[0].fn = BHIP_AGG_SUM;   aggs[0].arg.nodes = sx;  aggs[0].arg.n_nodes = 5; aggs[0].name = "s";
    aggs[1].fn = BHIP_AGG_COUNT; aggs[1].arg.nodes = one; aggs[1].arg.n_nodes = 1; aggs[1].name = "n";
    CHECK(bhip_plan_hash_aggregate(flt, BHIP_AGG_PARTIAL, 1, &key, key_names, 2, aggs, &part));
    CHECK(bhip_plan_merge(part, &merged));
    CHECK(bhip_plan_hash_aggregate(merged, BHIP_AGG_FINAL, 1, &key, key_names, 2, aggs, &fin));
    bhip_sort_expr se;
    se.expr = key; se.descending = 0; se.nulls_first = 0;
    CHECK(bhip_plan_sort(fin, 1, &se, &sorted));

    char text[2048];
    CHECK(bhip_plan_display(sorted, text, sizeof(text)));
    if (!strstr(text, "SortExec") || !strstr(text, "HashAggregateExec") || !strstr(text, "FilterExec")) { fprintf(stderr, "display:\n%s\n", text); return 1; }
    int32_t scheme = -1, count = -1;
    CHECK(bhip_plan_output_partitioning(part, &scheme, &count));
    if (count != 2) { fprintf(stderr, "partial aggregate should keep 2 partitions, has %d\n", count); return 1; }

    bhip_stream* stream = NULL;
    CHECK(bhip_plan_execute(sorted, 0, &stream));
    int rows_seen = 0;
    for (;;) {
        bhip_batch* b = NULL;
        CHECK(bhip_stream_next(stream, &b));
        if (!b) break;
        const int64_t n = bhip_batch_num_rows(b);
        if (bhip_batch_num_columns(b) != 3 || n != G) { fprintf(stderr, "unexpected result shape %d x %ld\n", bhip_batch_num_columns(b), (long)n); return 1; }
        int32_t gk[G]; double gs[G]; uint64_t gn[G];
        CHECK(bhip_batch_column_to_host(b, 0, gk, NULL, NULL));
        CHECK(bhip_batch_column_to_host(b, 1, gs, NULL, NULL));
        CHECK(bhip_batch_column_to_host(b, 2, gn, NULL, NULL));
        for (int g = 0; g < G; ++g) {
            if (gk[g] != g * 10 - 10 || (long)gn[g] != want_n[g] || fabs(gs[g] - want_s[g]) > 1e-9 * fabs(want_s[g])) {
                fprintf(stderr, "group %d: got (%d, %.6f, %lu) want (%d, %.6f, %ld)\n", g, gk[g], gs[g], (unsigned long)gn[g], g * 10 - 10, want_s[g], want_n[g]);
                return 1;
            }
        }
        rows_seen += (int)n;
        bhip_batch_release(b);
    }
    if (rows_seen != G) { fprintf(stderr, "%d result rows\n", rows_seen); return 1; }
    bhip_stream_release(stream);

    /* errors are values: an unknown column is a status + message, nothing aborts */
    bhip_expr_node bad[] = {col_("no_such_column"), f64_(1.0), bin_("Lt")};
    bhip_expr badp = {bad, 3};
    bhip_plan* nope = NULL;
    if (bhip_plan_filter(scan, &badp, &nope) == BHIP_OK || !strstr(bhip_last_error(), "no_such_column")) { fprintf(stderr, "missing column not reported: %s\n", bhip_last_error()); return 1; }

    bhip_plan_release(sorted); bhip_plan_release(fin); bhip_plan_release(merged); bhip_plan_release(part); bhip_plan_release(flt); bhip_plan_release(scan);
    bhip_batch_release(halves[0]); bhip_batch_release(halves[1]);
    bhip_ctx_release(ctx);
    printf("C ABI OK: %d groups, %ld rows aggregated\n", G, want_n[0] + want_n[1] + want_n[2]);
    return 0;
}
