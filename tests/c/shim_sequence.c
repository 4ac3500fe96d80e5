/* shim_sequence.c — the exact call sequence of the Rust shim (rust/ballista-hip/src/lib.rs), replayed in plain C11.
 *
 * The shim cannot be compiled here (no Rust toolchain), so what it relies on is pinned at the C ABI it binds:
 *
 *   GpuExec::try_new     bhip_plan_from_proto(NULL ctx, bytes)              dry run: is the subtree one the library takes?
 *   GpuExec::gpu_plan    bhip_plan_from_proto(ctx, bytes, resolve_leaf)     the resolver answers every scan leaf with
 *                        bhip_plan_arrow_streams over ArrowArrayStreams the CALLER implements (cstream.rs: a CPU child's
 *                        RecordBatchStream per partition); the library MOVES them (release-on-move of the C stream interface)
 *   GpuExec::execute     bhip_plan_execute -> bhip_stream_next -> bhip_batch_export_arrow   (GpuStream::next_batch)
 *   execute_to_file      bhip_plan_execute -> bhip_stream_write_ipc, which CONSUMES the stream: no bhip_stream_release after it
 *                        (the ownership rule ADVICE r02 found violated in the shim; calling release here would be a double free)
 *
 * Reference call sites: rust/core/src/execution_plans/query_stage.rs:49-85 (the trait), rust/executor/src/flight_service.rs:
 * 117-150 (execute + write_stream_to_disk + PartitionStats).
 *
 * Input: the wire plan of TPC-H Q1 (tests/golden/plans/q1_fixture.plan.bin, CsvScan leaf mem://lineitem), the reference's two
 * lineitem fixture partitions as '|'-separated text, the expected rows (tests/golden/q1_fixture.json rewritten as text by the
 * pytest wrapper).  Output: "SHIM SEQUENCE OK".
 *
 *   shim_sequence <plan.bin> <partition0.tbl> <partition1.tbl> <expected.txt> <out.arrow> */
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
#define REQUIRE(cond)                                                                                 \
    do { if (!(cond)) { fprintf(stderr, "line %d: requirement failed: %s\n", __LINE__, #cond); return 1; } } while (0)

/* ---- one lineitem partition as host columns ------------------------------------------------------------------------------ */
enum { MAX_ROWS = 64 };
typedef struct {
    int64_t n;
    int32_t orderkey[MAX_ROWS], suppkey[MAX_ROWS], shipdate[MAX_ROWS];
    double qty[MAX_ROWS], price[MAX_ROWS], disc[MAX_ROWS], tax[MAX_ROWS];
    int32_t flag_off[MAX_ROWS + 1], status_off[MAX_ROWS + 1];
    char flag[MAX_ROWS * 4], status[MAX_ROWS * 4];
} partition_t;

static int32_t days_from_civil(int y, int m, int d) {      /* days since 1970-01-01 */
    y -= m <= 2;
    const int era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153u * (unsigned)(m + (m > 2 ? -3 : 9)) + 2u) / 5u + (unsigned)d - 1u;
    const unsigned doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
    return era * 146097 + (int32_t)doe - 719468;
}

static int load_partition(const char* path, partition_t* p) {
    FILE* f = fopen(path, "r");
    if (!f) { perror(path); return 1; }
    char line[1024];
    p->n = 0;
    p->flag_off[0] = p->status_off[0] = 0;
    while (fgets(line, sizeof(line), f) && p->n < MAX_ROWS) {
        char* fields[16];
        int nf = 0;
        for (char* s = line; nf < 16; ++nf) {
            fields[nf] = s;
            char* bar = strchr(s, '|');
            if (!bar) break;
            *bar = 0;
            s = bar + 1;
        }
        if (nf < 11) continue;
        const int64_t i = p->n++;
        p->orderkey[i] = atoi(fields[0]);
        p->suppkey[i] = atoi(fields[2]);
        p->qty[i] = strtod(fields[4], NULL);
        p->price[i] = strtod(fields[5], NULL);
        p->disc[i] = strtod(fields[6], NULL);
        p->tax[i] = strtod(fields[7], NULL);
        const size_t lf = strlen(fields[8]), ls = strlen(fields[9]);
        memcpy(p->flag + p->flag_off[i], fields[8], lf);
        p->flag_off[i + 1] = p->flag_off[i] + (int32_t)lf;
        memcpy(p->status + p->status_off[i], fields[9], ls);
        p->status_off[i + 1] = p->status_off[i] + (int32_t)ls;
        int y, m, d;
        if (sscanf(fields[10], "%d-%d-%d", &y, &m, &d) != 3) { fclose(f); return 1; }
        p->shipdate[i] = days_from_civil(y, m, d);
    }
    fclose(f);
    return 0;
}

/* ---- a caller-implemented ArrowArrayStream: one batch, then end of stream (rust/ballista-hip/src/cstream.rs) ------------------------ */
enum { N_COLS = 9 };
static const char* const COL_NAMES[N_COLS] = {"l_orderkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax",
                                              "l_returnflag", "l_linestatus", "l_shipdate"};
static const char* const COL_FORMATS[N_COLS] = {"i", "i", "g", "g", "g", "g", "u", "u", "tdD"};

typedef struct {
    const partition_t* part;
    int served;
    int* released_counter;          /* how often the stream's release ran: must end at exactly 1 per stream */
    int* array_released_counter;
} stream_state_t;

typedef struct {
    struct ArrowArray children[N_COLS];
    struct ArrowArray* child_ptrs[N_COLS];
    const void* buffers[N_COLS][3];
    const void* top_buffers[1];
    int* released_counter;
} array_hold_t;

typedef struct {
    struct ArrowSchema children[N_COLS];
    struct ArrowSchema* child_ptrs[N_COLS];
} schema_hold_t;

static void release_child_array(struct ArrowArray* a) { a->release = NULL; }
static void release_top_array(struct ArrowArray* a) {
    array_hold_t* h = (array_hold_t*)a->private_data;
    ++*h->released_counter;
    free(h);
    a->release = NULL;
}
static void release_child_schema(struct ArrowSchema* s) { s->release = NULL; }
static void release_top_schema(struct ArrowSchema* s) { free(s->private_data); s->release = NULL; }

static int my_get_schema(struct ArrowArrayStream* st, struct ArrowSchema* out) {
    (void)st;
    schema_hold_t* h = (schema_hold_t*)calloc(1, sizeof(*h));
    for (int c = 0; c < N_COLS; ++c) {
        h->children[c].format = COL_FORMATS[c];
        h->children[c].name = COL_NAMES[c];
        h->children[c].flags = 0;                     /* nullable = false, as the TPC-H schema says (main.rs:267-360) */
        h->children[c].release = release_child_schema;
        h->child_ptrs[c] = &h->children[c];
    }
    memset(out, 0, sizeof(*out));
    out->format = "+s";
    out->name = "";
    out->n_children = N_COLS;
    out->children = h->child_ptrs;
    out->release = release_top_schema;
    out->private_data = h;
    return 0;
}

static int my_get_next(struct ArrowArrayStream* st, struct ArrowArray* out) {
    stream_state_t* s = (stream_state_t*)st->private_data;
    memset(out, 0, sizeof(*out));
    if (s->served) return 0;                              /* end of stream: a released array */
    s->served = 1;
    const partition_t* p = s->part;
    array_hold_t* h = (array_hold_t*)calloc(1, sizeof(*h));
    h->released_counter = s->array_released_counter;
    const void* data[N_COLS] = {p->orderkey, p->suppkey, p->qty, p->price, p->disc, p->tax, p->flag, p->status, p->shipdate};
    for (int c = 0; c < N_COLS; ++c) {
        struct ArrowArray* a = &h->children[c];
        a->length = p->n;
        a->null_count = 0;
        h->buffers[c][0] = NULL;                          /* no validity bitmap */
        if (c == 6 || c == 7) {
            h->buffers[c][1] = c == 6 ? p->flag_off : p->status_off;
            h->buffers[c][2] = data[c];
            a->n_buffers = 3;
        } else {
            h->buffers[c][1] = data[c];
            a->n_buffers = 2;
        }
        a->buffers = h->buffers[c];
        a->release = release_child_array;
        h->child_ptrs[c] = a;
    }
    h->top_buffers[0] = NULL;
    out->length = p->n;
    out->n_buffers = 1;
    out->buffers = h->top_buffers;
    out->n_children = N_COLS;
    out->children = h->child_ptrs;
    out->release = release_top_array;
    out->private_data = h;
    return 0;
}
static const char* my_last_error(struct ArrowArrayStream* st) { (void)st; return "no error"; }
static void my_release(struct ArrowArrayStream* st) {
    stream_state_t* s = (stream_state_t*)st->private_data;
    ++*s->released_counter;
    free(s);
    st->release = NULL;
}

/* ---- the leaf resolver (lib.rs::resolve_leaf) ------------------------------------------------------------------------------------ */
typedef struct {
    bhip_ctx* ctx;
    const partition_t* parts;
    int n_parts;
    int calls, moved_ok;
    int stream_released, arrays_released;
} resolver_state_t;

static bhip_status resolve_leaf(void* user, const bhip_leaf_desc* leaf, bhip_plan** out) {
    resolver_state_t* R = (resolver_state_t*)user;
    ++R->calls;
    *out = NULL;
    if (leaf->kind != BHIP_LEAF_CSV_SCAN || !leaf->path || strcmp(leaf->path, "mem://lineitem") != 0) return BHIP_OK;   /* not ours */
    if (leaf->n_fields != N_COLS) return BHIP_EINVAL;
    struct ArrowArrayStream streams[2];
    struct ArrowArrayStream* ptrs[2];
    for (int p = 0; p < R->n_parts; ++p) {
        stream_state_t* s = (stream_state_t*)calloc(1, sizeof(*s));
        s->part = &R->parts[p];
        s->released_counter = &R->stream_released;
        s->array_released_counter = &R->arrays_released;
        streams[p].get_schema = my_get_schema;
        streams[p].get_next = my_get_next;
        streams[p].get_last_error = my_last_error;
        streams[p].release = my_release;
        streams[p].private_data = s;
        ptrs[p] = &streams[p];
    }
    const bhip_status st = bhip_plan_arrow_streams(R->ctx, R->n_parts, ptrs, out);
    /* release-on-move: the library took the streams over and marked OUR structs released; it owns the private data now.
     * (On failure a struct that still has its release callback is still ours.) */
    R->moved_ok = 1;
    for (int p = 0; p < R->n_parts; ++p) {
        if (st == BHIP_OK && streams[p].release != NULL) R->moved_ok = 0;
        if (streams[p].release != NULL) streams[p].release(&streams[p]);
    }
    return st;
}

/* ---- expected rows -------------------------------------------------------------------------------------------------------------- */
typedef struct { char flag[8], status[8]; long count; double v[7]; } expected_row_t;

static int close_enough(double a, double b) { return fabs(a - b) <= 1e-9 * fmax(fabs(a), fabs(b)); }   /* the bar is 1e-6 */

/* compare an exported batch (struct array + schema) with the expected rows */
static int check_result(const struct ArrowArray* arr, const struct ArrowSchema* sch, const expected_row_t* want, int n_want) {
    static const char* const names[10] = {"l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge",
                                          "avg_qty", "avg_price", "avg_disc", "count_order"};
    REQUIRE(arr->length == n_want && arr->n_children == 10 && sch->n_children == 10);
    for (int c = 0; c < 10; ++c) REQUIRE(strcmp(sch->children[c]->name, names[c]) == 0);
    REQUIRE(strcmp(sch->children[0]->format, "u") == 0 && strcmp(sch->children[2]->format, "g") == 0 && strcmp(sch->children[9]->format, "L") == 0);
    for (int r = 0; r < n_want; ++r) {                    /* ORDER BY l_returnflag, l_linestatus: the expected file is in that order */
        for (int c = 0; c < 2; ++c) {
            const struct ArrowArray* a = arr->children[c];
            const int32_t* off = (const int32_t*)a->buffers[1];
            const char* bytes = (const char*)a->buffers[2];
            const char* w = c == 0 ? want[r].flag : want[r].status;
            REQUIRE((size_t)(off[r + 1] - off[r]) == strlen(w) && memcmp(bytes + off[r], w, strlen(w)) == 0);
        }
        for (int c = 0; c < 7; ++c) {
            const double got = ((const double*)arr->children[2 + c]->buffers[1])[r];
            if (!close_enough(got, want[r].v[c])) { fprintf(stderr, "row %d %s: %.17g, expected %.17g\n", r, names[2 + c], got, want[r].v[c]); return 1; }
        }
        REQUIRE(((const uint64_t*)arr->children[9]->buffers[1])[r] == (uint64_t)want[r].count);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: shim_sequence <plan.bin> <partition0.tbl> <partition1.tbl> <expected.txt> <out.arrow>\n"); return 2; }
    /* the task's plan bytes */
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    static unsigned char plan_bytes[1 << 16];
    const size_t plan_len = fread(plan_bytes, 1, sizeof(plan_bytes), f);
    fclose(f);
    REQUIRE(plan_len > 100 && plan_len < sizeof(plan_bytes));
    static partition_t parts[2];
    if (load_partition(argv[2], &parts[0]) || load_partition(argv[3], &parts[1])) return 2;
    REQUIRE(parts[0].n == 10 && parts[1].n == 10);
    expected_row_t want[8];
    int n_want = 0;
    f = fopen(argv[4], "r");
    if (!f) { perror(argv[4]); return 2; }
    while (n_want < 8 && fscanf(f, "%7s %7s %ld %lf %lf %lf %lf %lf %lf %lf", want[n_want].flag, want[n_want].status, &want[n_want].count,
                                &want[n_want].v[0], &want[n_want].v[1], &want[n_want].v[2], &want[n_want].v[3], &want[n_want].v[4],
                                &want[n_want].v[5], &want[n_want].v[6]) == 10)
        ++n_want;
    fclose(f);
    REQUIRE(n_want == 3);

    /* 1. GpuExec::try_new — dry run without a device context: coverage is decided at plan time */
    bhip_plan* probe = NULL;
    CHECK(bhip_plan_from_proto(NULL, plan_bytes, plan_len, NULL, NULL, &probe));
    REQUIRE(strcmp(bhip_plan_name(probe), "SortExec") == 0);
    bhip_stream* none = NULL;
    REQUIRE(bhip_plan_execute(probe, 0, &none) == BHIP_EEXEC && none == NULL);        /* unresolved leaves: an error, not a crash */
    bhip_plan_release(probe);

    /* 2. GpuExec::gpu_plan — the real plan, the scan leaf answered by two caller-implemented C streams */
    bhip_ctx* ctx = NULL;
    CHECK(bhip_ctx_create(0, &ctx));
    resolver_state_t R;
    memset(&R, 0, sizeof(R));
    R.ctx = ctx;
    R.parts = parts;
    R.n_parts = 2;
    bhip_plan* plan = NULL;
    CHECK(bhip_plan_from_proto(ctx, plan_bytes, plan_len, resolve_leaf, &R, &plan));
    REQUIRE(R.calls == 1 && R.moved_ok == 1);
    REQUIRE(R.stream_released == 0);                       /* moved, not released: the plan drains them on its first execute */
    int32_t scheme = 0, n_out = 0;
    CHECK(bhip_plan_output_partitioning(plan, &scheme, &n_out));
    REQUIRE(n_out == 1);

    /* 3. GpuExec::execute -> GpuStream::next_batch */
    bhip_stream* stream = NULL;
    CHECK(bhip_plan_execute(plan, 0, &stream));
    bhip_batch* batch = NULL;
    CHECK(bhip_stream_next(stream, &batch));
    REQUIRE(batch != NULL);
    struct ArrowArray arr;
    struct ArrowSchema sch;
    CHECK(bhip_batch_export_arrow(batch, &arr, &sch));
    if (check_result(&arr, &sch, want, n_want)) return 1;
    arr.release(&arr);
    sch.release(&sch);
    bhip_batch_release(batch);
    CHECK(bhip_stream_next(stream, &batch));
    REQUIRE(batch == NULL);                                /* end of stream */
    bhip_stream_release(stream);
    REQUIRE(R.arrays_released == 2);                       /* both input batches were imported and handed back */

    /* 4. GpuExec::execute_to_file — bhip_stream_write_ipc CONSUMES the stream: nothing is released afterwards */
    CHECK(bhip_plan_execute(plan, 0, &stream));            /* (the leaf replays what it drained the first time) */
    uint64_t rows = 0, batches = 0, bytes = 0;
    CHECK(bhip_stream_write_ipc(stream, argv[5], &rows, &batches, &bytes));
    stream = NULL;                                         /* consumed */
    REQUIRE(rows == 3 && batches == 1 && bytes > 0);
    struct ArrowArrayStream file;
    CHECK(bhip_ipc_open_file(argv[5], &file));
    REQUIRE(file.get_schema(&file, &sch) == 0 && file.get_next(&file, &arr) == 0 && arr.release != NULL);
    if (check_result(&arr, &sch, want, n_want)) return 1;
    arr.release(&arr);
    sch.release(&sch);
    file.release(&file);
    /* error path of the same call: an unwritable path fails with a status, and the stream is consumed all the same */
    CHECK(bhip_plan_execute(plan, 0, &stream));
    REQUIRE(bhip_stream_write_ipc(stream, "/nonexistent-dir/x.arrow", &rows, &batches, &bytes) == BHIP_EEXEC);
    stream = NULL;

    /* 5. drop(GpuExec): the plan releases the streams it took over, each exactly once */
    bhip_plan_release(plan);
    REQUIRE(R.stream_released == 2);
    bhip_ctx_release(ctx);
    printf("SHIM SEQUENCE OK: %d groups, stage file %llu rows / %llu bytes\n", n_want, (unsigned long long)rows, (unsigned long long)bytes);
    return 0;
}
