// capi.cpp — the extern "C" surface declared in include/ballista_hip.h.  Every entry point
// catches all exceptions and converts them to a status + thread-local message: nothing unwinds
// across the ABI (SURVEY.md §8(b) "Errors": `UnresolvedShuffleExec::execute` returning an error
// instead of panicking, rust/core/src/execution_plans/unresolved_shuffle.rs:83-90, is the model).
#include <cstdio>
#include <cstring>

#include "../util_kernels.h"
#include "plan.hpp"

using namespace bhip;

#define BHIP_API_BEGIN try {
#define BHIP_API_END                                              \
    return BHIP_OK;                                               \
    }                                                             \
    catch (const bhip::Error& e) {                                \
        bhip::set_last_error(e.what());                           \
        return e.code;                                            \
    }                                                             \
    catch (const std::bad_alloc&) {                               \
        bhip::set_last_error("host allocation failed");           \
        return BHIP_EOOM;                                         \
    }                                                             \
    catch (const std::exception& e) {                             \
        bhip::set_last_error(std::string("internal error: ") + e.what()); \
        return BHIP_EINVAL;                                       \
    }                                                             \
    catch (...) {                                                 \
        bhip::set_last_error("internal error");                   \
        return BHIP_EINVAL;                                       \
    }

static void need(const void* p, const char* what) {
    if (!p) fail(BHIP_EINVAL, std::string("null argument: ") + what);
}

extern "C" {

const char* bhip_last_error(void) { return bhip::get_last_error(); }
const char* bhip_version(void) { return "ballista_hip 0.1.0 (gfx950)"; }

// ---- context ----------------------------------------------------------------------------------------
bhip_status bhip_ctx_create(int device, bhip_ctx** out) {
    BHIP_API_BEGIN
    need(out, "out");
    auto c = new bhip_ctx();
    try { c->p = std::make_shared<Context>(device); } catch (...) { delete c; throw; }
    *out = c;
    BHIP_API_END
}

void bhip_ctx_release(bhip_ctx* ctx) {
    if (ctx && ctx->rc.fetch_sub(1) == 1) delete ctx;
}

bhip_status bhip_ctx_synchronize(bhip_ctx* ctx) {
    BHIP_API_BEGIN
    need(ctx, "ctx");
    ctx->p->set_device();
    HIP_CHECK(hipDeviceSynchronize());
    BHIP_API_END
}

bhip_status bhip_ctx_memory(bhip_ctx* ctx, uint64_t* in_use, uint64_t* peak) {
    BHIP_API_BEGIN
    need(ctx, "ctx");
    ctx->p->memory(in_use, peak);
    BHIP_API_END
}

bhip_status bhip_ctx_kernel_time(bhip_ctx* ctx, int32_t reset, double* ms, uint64_t* launches) {
    BHIP_API_BEGIN
    need(ctx, "ctx");
    ctx->p->kernel_time(reset != 0, ms, launches);
    BHIP_API_END
}

bhip_status bhip_ctx_kernel_stats(bhip_ctx* ctx, int32_t reset, char* buf, size_t cap) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(buf, "buf");
    ctx->p->set_device();
    std::string text;
    for (auto& kv : ctx->p->kernel_stats(reset != 0)) {
        char line[256];
        snprintf(line, sizeof line, "%s\t%.6f\t%llu\t%llu\n", kv.first.c_str(), kv.second.ms, (unsigned long long)kv.second.launches,
                 (unsigned long long)kv.second.bytes);
        text += line;
    }
    if (text.size() + 1 > cap) fail(BHIP_EINVAL, "bhip_ctx_kernel_stats: buffer too small");
    memcpy(buf, text.c_str(), text.size() + 1);
    BHIP_API_END
}

const char* bhip_ctx_kernel_name(bhip_ctx* ctx) {
    static thread_local std::string name;
    name = ctx ? ctx->p->kernel_name() : std::string();
    return name.c_str();
}

// ---- batches ------------------------------------------------------------------------------------------
static bhip_batch* wrap_batch(BatchPtr b) {
    auto h = new bhip_batch();
    h->p = std::move(b);
    return h;
}

bhip_status bhip_batch_from_host(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* cols, int64_t n_rows, bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    if (n_cols > 0) need(cols, "cols");
    *out = wrap_batch(batch_from_host(ctx->p, n_cols, cols, n_rows, false));
    BHIP_API_END
}

bhip_status bhip_batch_from_device(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* cols, int64_t n_rows, bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    if (n_cols > 0) need(cols, "cols");
    *out = wrap_batch(batch_from_host(ctx->p, n_cols, cols, n_rows, true));
    BHIP_API_END
}

bhip_status bhip_batch_from_tbl(bhip_ctx* ctx, const void* text, int64_t n_bytes, int32_t n_fields, const bhip_column_desc* fields,
                                int32_t n_projection, const int32_t* projection, bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out"); need(fields, "fields");
    *out = wrap_batch(batch_from_tbl(ctx->p, text, n_bytes, n_fields, fields, n_projection, projection));
    BHIP_API_END
}

void bhip_batch_retain(bhip_batch* b) { if (b) b->rc.fetch_add(1); }
void bhip_batch_release(bhip_batch* b) {
    if (b && b->rc.fetch_sub(1) == 1) delete b;
}
int64_t bhip_batch_num_rows(const bhip_batch* b) { return b ? b->p->n_rows : -1; }
int32_t bhip_batch_num_columns(const bhip_batch* b) { return b ? (int32_t)b->p->cols.size() : -1; }
int64_t bhip_batch_memory_size(const bhip_batch* b) { return b ? b->p->memory_size() : -1; }

bhip_status bhip_batch_column_info(const bhip_batch* b, int32_t i, const char** name, int32_t* dtype, int32_t* nullable,
                                   int64_t* data_bytes, int32_t* has_validity) {
    BHIP_API_BEGIN
    need(b, "batch");
    if (i < 0 || i >= (int)b->p->cols.size()) fail(BHIP_EINVAL, "column index out of range");
    const Field& f = b->p->schema->fields[i];
    const Column& c = b->p->cols[i];
    if (name) *name = f.name.c_str();
    if (dtype) *dtype = f.dtype;
    if (nullable) *nullable = f.nullable;
    if (data_bytes) {
        if (c.dtype == DT_UTF8) *data_bytes = c.data_bytes;
        else if (c.dtype == DT_BOOLEAN) *data_bytes = (c.length + 7) / 8;
        else *data_bytes = c.length * dtype_width(c.dtype);
    }
    if (has_validity) *has_validity = c.validity ? 1 : 0;
    BHIP_API_END
}

bhip_status bhip_batch_column_device(const bhip_batch* b, int32_t i, const void** data, const int32_t** offsets,
                                     const uint8_t** validity) {
    BHIP_API_BEGIN
    need(b, "batch");
    if (i < 0 || i >= (int)b->p->cols.size()) fail(BHIP_EINVAL, "column index out of range");
    const Column& c = b->p->cols[i];
    if (data) *data = c.data ? c.data->ptr() : nullptr;
    if (offsets) *offsets = c.offsets ? c.offsets->as<int32_t>() : nullptr;
    if (validity) *validity = c.validity ? c.validity->as<uint8_t>() : nullptr;
    BHIP_API_END
}

bhip_status bhip_batch_column_to_host(const bhip_batch* b, int32_t i, void* data, int32_t* offsets, uint8_t* validity) {
    BHIP_API_BEGIN
    need(b, "batch");
    column_to_host(*b->p, i, data, offsets, validity);
    BHIP_API_END
}

// ---- plans ----------------------------------------------------------------------------------------------
static bhip_plan* wrap_plan(PlanPtr p) {
    auto h = new bhip_plan();
    h->p = std::move(p);
    return h;
}
static PlanPtr plan_of(const bhip_plan* h, const char* what) {
    need(h, what);
    return h->p;
}
static SchemaPtr schema_from_descs(int32_t n_cols, const bhip_column_desc* cols) {
    auto s = std::make_shared<Schema>();
    for (int i = 0; i < n_cols; ++i) {
        need(cols[i].name, "column name");
        s->fields.push_back(Field{cols[i].name, cols[i].dtype, cols[i].nullable != 0});
    }
    return s;
}

bhip_status bhip_plan_memory(bhip_ctx* ctx, int32_t n_partitions, const int32_t* offsets, bhip_batch* const* batches, bhip_plan** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out"); need(offsets, "offsets");
    if (n_partitions < 1) fail(BHIP_EINVAL, "MemoryExec needs at least one partition");
    if (offsets[n_partitions] < 1) fail(BHIP_EINVAL, "MemoryExec needs at least one batch (it carries the schema)");
    need(batches, "batches");
    std::vector<std::vector<BatchPtr>> parts(n_partitions);
    SchemaPtr schema;
    for (int p = 0; p < n_partitions; ++p)
        for (int k = offsets[p]; k < offsets[p + 1]; ++k) {
            need(batches[k], "batch");
            const BatchPtr& b = batches[k]->p;
            if (!schema) schema = b->schema;
            else {
                if (b->schema->fields.size() != schema->fields.size()) fail(BHIP_EINVAL, "MemoryExec: batches with different schemas");
                for (size_t i = 0; i < schema->fields.size(); ++i)
                    if (b->schema->fields[i].name != schema->fields[i].name || b->schema->fields[i].dtype != schema->fields[i].dtype)
                        fail(BHIP_EINVAL, "MemoryExec: batches with different schemas");
            }
            parts[p].push_back(b);
        }
    // schema nullability = union over batches
    auto merged = std::make_shared<Schema>(*schema);
    for (auto& part : parts)
        for (auto& b : part)
            for (size_t i = 0; i < merged->fields.size(); ++i)
                if (b->schema->fields[i].nullable) merged->fields[i].nullable = true;
    *out = wrap_plan(std::make_shared<MemoryExec>(ctx->p, merged, parts));
    BHIP_API_END
}

bhip_status bhip_plan_empty(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* schema, int32_t produce_one_row, bhip_plan** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    *out = wrap_plan(std::make_shared<EmptyExec>(ctx->p, schema_from_descs(n_cols, schema), produce_one_row != 0));
    BHIP_API_END
}

bhip_status bhip_plan_filter(bhip_plan* input, const bhip_expr* predicate, bhip_plan** out) {
    BHIP_API_BEGIN
    need(predicate, "predicate"); need(out, "out");
    *out = wrap_plan(std::make_shared<FilterExec>(parse_expr(*predicate), plan_of(input, "input")));
    BHIP_API_END
}

bhip_status bhip_plan_projection(bhip_plan* input, int32_t n, const bhip_expr* exprs, const char* const* names, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    if (n > 0) { need(exprs, "exprs"); need(names, "names"); }
    std::vector<std::pair<ExprPtr, std::string>> v;
    for (int i = 0; i < n; ++i) { need(names[i], "name"); v.push_back({parse_expr(exprs[i]), names[i]}); }
    *out = wrap_plan(std::make_shared<ProjectionExec>(v, plan_of(input, "input")));
    BHIP_API_END
}

bhip_status bhip_plan_hash_aggregate(bhip_plan* input, int32_t mode, int32_t n_group, const bhip_expr* group_exprs,
                                     const char* const* group_names, int32_t n_aggr, const bhip_aggregate* aggr, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    std::vector<std::pair<ExprPtr, std::string>> g;
    for (int i = 0; i < n_group; ++i) { need(group_names[i], "group name"); g.push_back({parse_expr(group_exprs[i]), group_names[i]}); }
    std::vector<AggregateDesc> a;
    for (int i = 0; i < n_aggr; ++i) { need(aggr[i].name, "aggregate name"); a.push_back(AggregateDesc{aggr[i].fn, parse_expr(aggr[i].arg), aggr[i].name}); }
    *out = wrap_plan(std::make_shared<HashAggregateExec>(mode, g, a, plan_of(input, "input")));
    BHIP_API_END
}

bhip_status bhip_plan_hash_join(bhip_plan* left, bhip_plan* right, int32_t n_on, const char* const* left_keys,
                                const char* const* right_keys, int32_t join_type, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    if (n_on < 1) fail(BHIP_EINVAL, "HashJoinExec needs at least one key pair");
    std::vector<std::pair<std::string, std::string>> on;
    for (int i = 0; i < n_on; ++i) { need(left_keys[i], "left key"); need(right_keys[i], "right key"); on.push_back({left_keys[i], right_keys[i]}); }
    *out = wrap_plan(std::make_shared<HashJoinExec>(plan_of(left, "left"), plan_of(right, "right"), on, join_type));
    BHIP_API_END
}

bhip_status bhip_plan_sort(bhip_plan* input, int32_t n, const bhip_sort_expr* exprs, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    if (n < 1) fail(BHIP_EINVAL, "SortExec needs at least one sort expression");
    std::vector<SortDesc> v;
    for (int i = 0; i < n; ++i) v.push_back(SortDesc{parse_expr(exprs[i].expr), exprs[i].descending != 0, exprs[i].nulls_first != 0});
    *out = wrap_plan(std::make_shared<SortExec>(v, plan_of(input, "input")));
    BHIP_API_END
}

bhip_status bhip_plan_repartition(bhip_plan* input, int32_t scheme, int32_t n_exprs, const bhip_expr* hash_exprs,
                                  int32_t partition_count, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    Partitioning p;
    p.scheme = scheme;
    p.count = partition_count;
    if (partition_count < 1) fail(BHIP_EINVAL, "partition count must be positive");
    if (scheme == BHIP_PART_HASH) {
        if (n_exprs < 1) fail(BHIP_EINVAL, "hash repartition needs key expressions");
        for (int i = 0; i < n_exprs; ++i) p.exprs.push_back(parse_expr(hash_exprs[i]));
    } else if (scheme != BHIP_PART_ROUND_ROBIN && scheme != BHIP_PART_UNKNOWN) {
        fail(BHIP_EINVAL, "unsupported output partitioning for RepartitionExec");
    }
    *out = wrap_plan(std::make_shared<RepartitionExec>(plan_of(input, "input"), p));
    BHIP_API_END
}

bhip_status bhip_plan_coalesce_batches(bhip_plan* input, int64_t target, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    *out = wrap_plan(std::make_shared<CoalesceBatchesExec>(plan_of(input, "input"), target));
    BHIP_API_END
}
bhip_status bhip_plan_merge(bhip_plan* input, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    *out = wrap_plan(std::make_shared<MergeExec>(plan_of(input, "input")));
    BHIP_API_END
}
bhip_status bhip_plan_global_limit(bhip_plan* input, int64_t limit, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    *out = wrap_plan(std::make_shared<LimitExec>(plan_of(input, "input"), limit, true));
    BHIP_API_END
}
bhip_status bhip_plan_local_limit(bhip_plan* input, int64_t limit, bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    *out = wrap_plan(std::make_shared<LimitExec>(plan_of(input, "input"), limit, false));
    BHIP_API_END
}

bhip_status bhip_plan_parquet(bhip_ctx* ctx, int32_t n_files, const char* const* paths, int32_t n_projection, const uint32_t* projection,
                              int32_t num_partitions, bhip_plan** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(paths, "paths"); need(out, "out");
    std::vector<std::string> files;
    for (int i = 0; i < n_files; ++i) { need(paths[i], "path"); files.push_back(paths[i]); }
    std::vector<uint32_t> proj;
    if (n_projection > 0) { need(projection, "projection"); proj.assign(projection, projection + n_projection); }
    *out = wrap_plan(make_parquet_exec(ctx->p, files, proj, n_projection >= 0 && projection != nullptr, num_partitions));
    BHIP_API_END
}

bhip_status bhip_plan_from_proto(bhip_ctx* ctx, const void* bytes, size_t len, bhip_leaf_resolver resolve, void* user,
                                 bhip_plan** out) {
    BHIP_API_BEGIN
    need(out, "out");
    if (len) need(bytes, "bytes");
    *out = wrap_plan(plan_from_proto(ctx ? ctx->p : ContextPtr(), bytes, len, resolve, user));
    BHIP_API_END
}

bhip_status bhip_expr_from_proto_display(const void* bytes, size_t len, char* buf, size_t cap) {
    BHIP_API_BEGIN
    need(buf, "buf");
    if (len) need(bytes, "bytes");
    snprintf(buf, cap, "%s", expr_from_proto(bytes, len)->to_string().c_str());
    BHIP_API_END
}

void bhip_plan_retain(bhip_plan* p) { if (p) p->rc.fetch_add(1); }
void bhip_plan_release(bhip_plan* p) {
    if (p && p->rc.fetch_sub(1) == 1) delete p;
}

const char* bhip_plan_name(const bhip_plan* p) { return p ? p->p->name() : ""; }

static void fill_schema(const Schema& s, std::vector<std::string>* keep, int32_t cap, const char** names, int32_t* dtypes,
                        int32_t* nullable, int32_t* n_cols) {
    (void)keep;
    if (n_cols) *n_cols = (int32_t)s.fields.size();
    for (int i = 0; i < cap && i < (int)s.fields.size(); ++i) {
        if (names) names[i] = s.fields[i].name.c_str();
        if (dtypes) dtypes[i] = s.fields[i].binary ? (int32_t)DT_BINARY : s.fields[i].large ? (int32_t)DT_LARGE_UTF8 : s.fields[i].dtype;
        if (nullable) nullable[i] = s.fields[i].nullable;
    }
}

bhip_status bhip_plan_schema(const bhip_plan* p, int32_t cap, const char** names, int32_t* dtypes, int32_t* nullable, int32_t* n_cols) {
    BHIP_API_BEGIN
    need(p, "plan");
    // the schema object is owned by the plan, so the name pointers live as long as the plan
    fill_schema(*p->p->schema(), nullptr, cap, names, dtypes, nullable, n_cols);
    BHIP_API_END
}

bhip_status bhip_plan_output_partitioning(const bhip_plan* p, int32_t* scheme, int32_t* count) {
    BHIP_API_BEGIN
    need(p, "plan");
    const Partitioning part = p->p->output_partitioning();
    if (scheme) *scheme = part.scheme;
    if (count) *count = part.count;
    BHIP_API_END
}

bhip_status bhip_plan_children(const bhip_plan* p, int32_t cap, bhip_plan** children, int32_t* n_children) {
    BHIP_API_BEGIN
    need(p, "plan");
    auto ch = p->p->children();
    if (n_children) *n_children = (int32_t)ch.size();
    // children are handed out as NEW handles (caller releases): a borrowed C handle has no owner to borrow from
    for (int i = 0; i < cap && i < (int)ch.size(); ++i)
        if (children) children[i] = wrap_plan(ch[i]);
    BHIP_API_END
}

bhip_status bhip_plan_with_new_children(const bhip_plan* p, int32_t n, bhip_plan* const* children, bhip_plan** out) {
    BHIP_API_BEGIN
    need(p, "plan"); need(out, "out");
    std::vector<PlanPtr> ch;
    for (int i = 0; i < n; ++i) ch.push_back(plan_of(children[i], "child"));
    *out = wrap_plan(p->p->with_new_children(ch));
    BHIP_API_END
}

bhip_status bhip_plan_display(const bhip_plan* p, char* buf, size_t cap) {
    BHIP_API_BEGIN
    need(p, "plan"); need(buf, "buf");
    const std::string s = display_plan(p->p);
    snprintf(buf, cap, "%s", s.c_str());
    BHIP_API_END
}

bhip_status bhip_plan_execute(bhip_plan* p, int32_t partition, bhip_stream** out) {
    BHIP_API_BEGIN
    need(p, "plan"); need(out, "out");
    ContextPtr ctx = p->p->context();
    if (!ctx) fail(BHIP_EEXEC, "Ballista Error: this plan has unresolved leaves and no device context; it can be inspected, not executed");
    ctx->set_device();
    trace_point("plan_execute: enter");
    auto h = std::unique_ptr<bhip_stream>(new bhip_stream());
    h->ex = Exec{ctx, ctx->acquire_stream()};
    h->s = p->p->execute(partition, h->ex);
    *out = h.release();
    trace_point("plan_execute: leave");
    BHIP_API_END
}

bhip_status bhip_plan_collect(bhip_plan* p, int32_t cap, bhip_batch** out, int32_t* n_out) {
    BHIP_API_BEGIN
    need(p, "plan"); need(out, "out"); need(n_out, "n_out");
    ContextPtr ctx = p->p->context();
    if (!ctx) fail(BHIP_EEXEC, "Ballista Error: this plan has unresolved leaves and no device context; it can be inspected, not executed");
    ctx->set_device();
    trace_point("plan_collect: enter");
    std::vector<BatchPtr> got;
    {
        const Exec ex{ctx, ctx->acquire_stream()};
        struct Release { const Exec& ex; ~Release() { ex.ctx->release_stream(ex.stream); } } release{ex};
        const int n_parts = p->p->output_partitioning().count;
        for (int part = 0; part < n_parts; ++part) {
            StreamPtr s = p->p->execute(part, ex);
            while (BatchPtr b = s->next()) got.push_back(b);
        }
        ctx->wait_stream(ex.stream);               // batches only leave the library once everything that produces them has finished
    }
    if ((int64_t)got.size() > cap) fail(BHIP_EINVAL, "bhip_plan_collect: the plan yields " + std::to_string(got.size()) + " batches, the caller has room for " + std::to_string(cap));
    for (size_t i = 0; i < got.size(); ++i) out[i] = wrap_batch(got[i]);
    *n_out = (int32_t)got.size();
    trace_point("plan_collect: leave");
    BHIP_API_END
}

// ---- streams ------------------------------------------------------------------------------------------------
bhip_status bhip_stream_next(bhip_stream* s, bhip_batch** out) {
    BHIP_API_BEGIN
    need(s, "stream"); need(out, "out");
    s->ex.ctx->set_device();
    trace_point("stream_next: enter");
    BatchPtr b = s->s->next();
    trace_point("stream_next: operators returned");
    // a batch only leaves the library once everything that produces it has finished
    if (b) s->ex.ctx->wait_stream(s->ex.stream);
    *out = b ? wrap_batch(b) : nullptr;
    trace_point("stream_next: leave");
    BHIP_API_END
}

bhip_status bhip_stream_schema(const bhip_stream* s, int32_t cap, const char** names, int32_t* dtypes, int32_t* nullable, int32_t* n_cols) {
    BHIP_API_BEGIN
    need(s, "stream");
    fill_schema(*s->s->schema(), nullptr, cap, names, dtypes, nullable, n_cols);
    BHIP_API_END
}

void bhip_stream_release(bhip_stream* s) { delete s; }

bhip_status bhip_stream_drain(bhip_stream* s, bhip_batch_sink sink, void* user, uint64_t* num_rows, uint64_t* num_batches, uint64_t* num_bytes) {
    BHIP_API_BEGIN
    need(s, "stream");
    s->ex.ctx->set_device();
    uint64_t rows = 0, batches = 0, bytes = 0;
    while (BatchPtr b = s->s->next()) {
        s->ex.ctx->wait_stream(s->ex.stream);
        rows += (uint64_t)b->n_rows;
        batches += 1;
        bytes += (uint64_t)b->memory_size();
        if (sink) {
            bhip_batch* h = wrap_batch(b);
            const bhip_status st = sink(user, h);
            bhip_batch_release(h);
            if (st != BHIP_OK) fail(st, "batch sink failed");
        }
    }
    if (num_rows) *num_rows = rows;
    if (num_batches) *num_batches = batches;
    if (num_bytes) *num_bytes = bytes;
    BHIP_API_END
}

// ---- hash partition / concat ------------------------------------------------------------------------------------
bhip_status bhip_batch_hash_partition(bhip_batch* batch, int32_t n_exprs, const bhip_expr* hash_exprs, int32_t n, bhip_batch** out) {
    BHIP_API_BEGIN
    need(batch, "batch"); need(out, "out");
    if (n < 1 || n_exprs < 1) fail(BHIP_EINVAL, "hash partition needs keys and a positive partition count");
    std::vector<ExprPtr> exprs;
    for (int i = 0; i < n_exprs; ++i) exprs.push_back(parse_expr(hash_exprs[i]));
    ContextPtr ctx = batch->p->ctx;
    ctx->set_device();
    Exec ex{ctx, ctx->acquire_stream()};
    std::vector<BatchPtr> parts;
    try {
        parts = hash_partition_batch(ex, batch->p, exprs, n);
        HIP_CHECK(hipStreamSynchronize(ex.stream));
    } catch (...) { ctx->release_stream(ex.stream); throw; }
    ctx->release_stream(ex.stream);
    for (int i = 0; i < n; ++i) out[i] = wrap_batch(parts[i]);
    BHIP_API_END
}

bhip_status bhip_batch_concat(bhip_ctx* ctx, int32_t n, bhip_batch* const* batches, bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    if (n < 1) fail(BHIP_EINVAL, "concat of zero batches");
    std::vector<BatchPtr> parts;
    for (int i = 0; i < n; ++i) { need(batches[i], "batch"); parts.push_back(batches[i]->p); }
    ctx->p->set_device();
    Exec ex{ctx->p, ctx->p->acquire_stream()};
    BatchPtr r;
    try {
        r = concat_batches(ex, parts[0]->schema, parts);
        HIP_CHECK(hipStreamSynchronize(ex.stream));
    } catch (...) { ctx->p->release_stream(ex.stream); throw; }
    ctx->p->release_stream(ex.stream);
    *out = wrap_batch(r);
    BHIP_API_END
}

// ---- synthetic TPC-H ---------------------------------------------------------------------------------------------
struct TpchCard { uint64_t orders, lineitem, customer, supplier, part; };
static TpchCard tpch_card(double sf) {
    TpchCard c;
    c.orders = (uint64_t)llround(1500000.0 * sf);
    if (c.orders < 7) c.orders = 7;
    c.lineitem = sf == 1.0 ? 6001215ull : (sf == 100.0 ? 600037902ull : 4 * c.orders);
    c.customer = (uint64_t)llround(150000.0 * sf); if (c.customer < 3) c.customer = 3;
    c.supplier = (uint64_t)llround(10000.0 * sf); if (c.supplier < 1) c.supplier = 1;
    c.part = (uint64_t)llround(200000.0 * sf); if (c.part < 1) c.part = 1;
    return c;
}

static Column gen_col(const Exec& ex, int dtype, int64_t n) {
    Column c;
    c.dtype = dtype;
    c.length = n;
    c.data = make_buffer(ex, (size_t)n * dtype_width(dtype) + 8);
    return c;
}
static Column gen_char_col(const Exec& ex, int64_t n) {
    Column c;
    c.dtype = DT_UTF8;
    c.length = n;
    c.data = make_buffer(ex, (size_t)n + 8);
    c.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
    c.data_bytes = n;
    return c;
}

// columns == NULL: every column; else only the named ones, in the table's own column order
static bool gen_wants(const bhip_tpch_opts* o, const char* name) {
    if (!o || !o->columns || o->n_columns <= 0) return true;
    for (int i = 0; i < o->n_columns; ++i)
        if (o->columns[i] && !strcmp(o->columns[i], name)) return true;
    return false;
}

bhip_status bhip_tpch_lineitem_opts(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, const bhip_tpch_opts* opts,
                                    bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    if (n > 0xFFFFFFF0ull) fail(BHIP_EINVAL, "at most 2^32-16 rows per batch");
    const bool key64 = opts && opts->key64, with_dates = opts && opts->with_dates;
    const TpchCard card = tpch_card(sf);
    ctx->p->set_device();
    Exec ex{ctx->p, nullptr};
    auto b = std::make_shared<Batch>();
    auto s = std::make_shared<Schema>();
    b->ctx = ctx->p;
    b->n_rows = (int64_t)n;
    auto add = [&](const char* name, Column c) { s->fields.push_back(Field{name, c.dtype, false}); b->cols.push_back(std::move(c)); };
    GenLineitemOut o;
    memset(&o, 0, sizeof(o));
    Column c;
    if (gen_wants(opts, "l_orderkey")) {
        c = gen_col(ex, key64 ? DT_INT64 : DT_INT32, n);
        if (key64) o.l_orderkey_i64 = c.data->as<int64_t>(); else o.l_orderkey = c.data->as<int32_t>();
        add("l_orderkey", c);
    }
    if (gen_wants(opts, "l_suppkey")) { c = gen_col(ex, DT_INT32, n); o.l_suppkey = c.data->as<int32_t>(); add("l_suppkey", c); }
    if (gen_wants(opts, "l_quantity")) { c = gen_col(ex, DT_FLOAT64, n); o.l_quantity = c.data->as<double>(); add("l_quantity", c); }
    if (gen_wants(opts, "l_extendedprice")) { c = gen_col(ex, DT_FLOAT64, n); o.l_extendedprice = c.data->as<double>(); add("l_extendedprice", c); }
    if (gen_wants(opts, "l_discount")) { c = gen_col(ex, DT_FLOAT64, n); o.l_discount = c.data->as<double>(); add("l_discount", c); }
    if (gen_wants(opts, "l_tax")) { c = gen_col(ex, DT_FLOAT64, n); o.l_tax = c.data->as<double>(); add("l_tax", c); }
    if (gen_wants(opts, "l_returnflag")) { c = gen_char_col(ex, n); o.flag_off = c.offsets->as<int32_t>(); o.flag_data = c.data->as<uint8_t>(); add("l_returnflag", c); }
    if (gen_wants(opts, "l_linestatus")) { c = gen_char_col(ex, n); o.status_off = c.offsets->as<int32_t>(); o.status_data = c.data->as<uint8_t>(); add("l_linestatus", c); }
    if (gen_wants(opts, "l_shipdate")) { c = gen_col(ex, DT_DATE32, n); o.l_shipdate = c.data->as<int32_t>(); add("l_shipdate", c); }
    if (with_dates) {
        if (gen_wants(opts, "l_commitdate")) { c = gen_col(ex, DT_DATE32, n); o.l_commitdate = c.data->as<int32_t>(); add("l_commitdate", c); }
        if (gen_wants(opts, "l_receiptdate")) { c = gen_col(ex, DT_DATE32, n); o.l_receiptdate = c.data->as<int32_t>(); add("l_receiptdate", c); }
    }
    if (s->fields.empty()) fail(BHIP_EINVAL, "bhip_tpch_lineitem_opts: no such column");
    b->schema = s;
    const GenKeyLayout keys{opts ? opts->key_base : 0, opts && opts->sparse_keys ? 1 : 0};
    HIP_CHECK(launch_gen_lineitem(ex.cfg(), seed, row0, n, card.orders, card.part, card.supplier, o, keys));
    HIP_CHECK(hipDeviceSynchronize());
    *out = wrap_batch(b);
    BHIP_API_END
}

bhip_status bhip_tpch_orders_opts(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, const bhip_tpch_opts* opts, bhip_batch** out) {
    BHIP_API_BEGIN
    need(ctx, "ctx"); need(out, "out");
    if (n > 0xFFFFFFF0ull) fail(BHIP_EINVAL, "at most 2^32-16 rows per batch");
    const bool key64 = opts && opts->key64;
    const TpchCard card = tpch_card(sf);
    ctx->p->set_device();
    Exec ex{ctx->p, nullptr};
    auto b = std::make_shared<Batch>();
    auto s = std::make_shared<Schema>();
    b->ctx = ctx->p;
    b->n_rows = (int64_t)n;
    auto add = [&](const char* name, Column c) { s->fields.push_back(Field{name, c.dtype, false}); b->cols.push_back(std::move(c)); };
    GenOrdersOut o;
    memset(&o, 0, sizeof(o));
    Column c;
    if (gen_wants(opts, "o_orderkey")) {
        c = gen_col(ex, key64 ? DT_INT64 : DT_INT32, n);
        if (key64) o.o_orderkey_i64 = c.data->as<int64_t>(); else o.o_orderkey = c.data->as<int32_t>();
        add("o_orderkey", c);
    }
    if (gen_wants(opts, "o_custkey")) { c = gen_col(ex, DT_INT32, n); o.o_custkey = c.data->as<int32_t>(); add("o_custkey", c); }
    if (gen_wants(opts, "o_orderdate")) { c = gen_col(ex, DT_DATE32, n); o.o_orderdate = c.data->as<int32_t>(); add("o_orderdate", c); }
    if (gen_wants(opts, "o_shippriority")) { c = gen_col(ex, DT_INT32, n); o.o_shippriority = c.data->as<int32_t>(); add("o_shippriority", c); }
    if (s->fields.empty()) fail(BHIP_EINVAL, "bhip_tpch_orders_opts: no such column");
    b->schema = s;
    const GenKeyLayout keys{opts ? opts->key_base : 0, opts && opts->sparse_keys ? 1 : 0};
    HIP_CHECK(launch_gen_orders(ex.cfg(), seed, row0, n, card.customer, o, keys));
    HIP_CHECK(hipDeviceSynchronize());
    *out = wrap_batch(b);
    BHIP_API_END
}

bhip_status bhip_tpch_lineitem(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, int32_t key64,
                               int32_t with_dates, bhip_batch** out) {
    bhip_tpch_opts o;
    memset(&o, 0, sizeof(o));
    o.key64 = key64;
    o.with_dates = with_dates;
    return bhip_tpch_lineitem_opts(ctx, sf, seed, row0, n, &o, out);
}

bhip_status bhip_tpch_orders(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, int32_t key64, bhip_batch** out) {
    bhip_tpch_opts o;
    memset(&o, 0, sizeof(o));
    o.key64 = key64;
    return bhip_tpch_orders_opts(ctx, sf, seed, row0, n, &o, out);
}

}  // extern "C"
