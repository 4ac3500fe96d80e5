// ops_basic.cpp — batch-level primitives (take / concat / slice / filter) and the operators that
// need nothing else: MemoryExec, EmptyExec, FilterExec, ProjectionExec, CoalesceBatchesExec,
// MergeExec, Global/LocalLimitExec (rust/core/src/serde/physical_plan/from_proto.rs:69-92,
// 122-132,165-172,287-290).
#include <sstream>

#include "../util_kernels.h"
#include "plan.hpp"
#include "sop.hpp"

namespace bhip {

// ---- helpers -------------------------------------------------------------------------------------
void check_partition(const ExecutionPlan& p, int partition) {
    const int n = p.output_partitioning().count;
    if (partition < 0 || partition >= n)
        fail(BHIP_EINVAL, std::string(p.name()) + " invalid partition " + std::to_string(partition) + " (plan has " +
                              std::to_string(n) + ")");
}

std::vector<BatchPtr> drain(RecordBatchStream& s) {
    std::vector<BatchPtr> out;
    while (BatchPtr b = s.next()) out.push_back(b);
    return out;
}

static void render(const PlanPtr& p, int depth, std::ostringstream& o) {
    for (int i = 0; i < depth; ++i) o << "  ";
    o << p->describe() << "\n";
    for (auto& c : p->children()) render(c, depth + 1, o);
}
std::string display_plan(const PlanPtr& p) {
    std::ostringstream o;
    render(p, 0, o);
    return o.str();
}

void check_scan_status(const Exec& ex, const ScanStatus* dev_status, ScanStatus* host_out) {
    ScanStatus st = read_device(ex, dev_status);
    if (host_out) *host_out = st;
    check_scan_flags(st);
}

void check_scan_flags(const ScanStatus& st) {
    if (st.flags & SCAN_ERR_DIV_ZERO) fail(BHIP_EEXEC, "Arrow error: Divide by zero error");
    if (st.flags & SCAN_ERR_KEY_TOO_LONG)
        fail(BHIP_ENOTIMPL, "a Utf8 key value is longer than the packed-key path supports");
}

static ScanStatus* new_status(Temp& tmp) {
    ScanStatus* st = tmp.get<ScanStatus>(1);
    HIP_CHECK(hipMemsetAsync(st, 0, sizeof(ScanStatus), tmp.ex.stream));
    return st;
}

// ---- take ------------------------------------------------------------------------------------------
// known_bytes >= 0: the caller knows the Utf8 value bytes of the result (a permutation keeps them) — no
// host round trip for the total
Column take_column(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n, int64_t known_bytes) {
    const LaunchCfg cfg = ex.cfg();
    Column out;
    out.dtype = c.dtype;
    out.length = n;
    if (c.dtype == DT_UTF8) {
        Temp tmp(ex);
        uint32_t* lengths = tmp.get<uint32_t>((size_t)n + 1);
        TIMED_LAUNCH_N(ex, "take_utf8_lengths", n, launch_take_utf8_lengths(cfg, c.offsets->as<int32_t>(), idx, n, lengths));
        out.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
        uint64_t* total = tmp.get<uint64_t>(1);
        void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n));
        HIP_CHECK(exclusive_scan_u32_i32(ex.stream, lengths, n, out.offsets->as<int32_t>(), true, total, scan_tmp));
        const uint64_t bytes = known_bytes >= 0 ? (uint64_t)known_bytes : read_device(ex, total);
        if (bytes > 0x7FFFFFFFull) fail(BHIP_EEXEC, "Utf8 column exceeds 2 GiB of value bytes");
        out.data_bytes = (int64_t)bytes;
        out.data = make_buffer(ex, (size_t)bytes + 8);
        TIMED_LAUNCH_N(ex, "take_utf8_copy", n, launch_take_utf8_copy(cfg, c.offsets->as<int32_t>(), c.data->as<uint8_t>(), c.data_bytes, idx, n,
                                        out.offsets->as<int32_t>(), out.data->as<uint8_t>()));
    } else if (c.dtype == DT_BOOLEAN) {
        out.data = make_buffer(ex, bitmap_bytes(n) + 8);
        TIMED_LAUNCH(ex, "take_bitmap", launch_take_bitmap(cfg, c.data->as<uint64_t>(), idx, n, out.data->as<uint64_t>()));
    } else {
        const int w = dtype_width(c.dtype);
        out.data = make_buffer(ex, (size_t)n * w + 8);
        TIMED_LAUNCH_N(ex, "take_fixed", n, launch_take_fixed(cfg, c.data->ptr(), w, idx, n, out.data->ptr()));
    }
    return out;
}

// `may_null`: indices can hold NULL_INDEX (outer joins) -> always build a validity bitmap
static Column take_column_v(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n, bool may_null,
                            bool permutation = false) {
    Column out = take_column(ex, c, idx, n, permutation && c.dtype == DT_UTF8 ? c.data_bytes : -1);
    if (c.validity || may_null) {
        out.validity = make_buffer(ex, bitmap_bytes(n) + 8);
        TIMED_LAUNCH(ex, "take_bitmap", launch_take_bitmap(ex.cfg(), c.validity ? c.validity->as<uint64_t>() : nullptr, idx, n,
                                     out.validity->as<uint64_t>()));
    }
    return out;
}

// several columns by the same index vector: the fixed-width values and all bitmaps go in ONE launch
// (a Q1 result batch is 10 columns x 4 rows: ten launches of four threads otherwise); Utf8 columns one by one
std::vector<Column> take_columns(const Exec& ex, const std::vector<const Column*>& cols_in, const uint32_t* idx, int64_t n,
                                 bool may_null, bool permutation, bool keep_views, const BufferPtr& idx_owner) {
    // ---- views in, views out (host/core.hpp Column::view_base) ------------------------------------------------------------
    bool any_view = keep_views;
    for (auto* c : cols_in) any_view = any_view || c->is_view();
    if (any_view) {
        std::vector<Column> out(cols_in.size());
        std::map<const Buffer*, BufferPtr> composed;          // a batch's view columns share their index vector: composed once
        BufferPtr own_idx;                                     // `idx` as a buffer the new views can keep
        std::vector<const Column*> plain;
        std::vector<size_t> plain_pos;
        std::map<const Buffer*, std::pair<std::vector<const Column*>, std::vector<size_t>>> through;   // bases to gather, by composed index vector
        std::map<const Buffer*, bool> through_null;
        for (size_t i = 0; i < cols_in.size(); ++i) {
            const Column& c = *cols_in[i];
            if (c.is_view()) {
                BufferPtr& comp = composed[c.view_idx.get()];
                if (!comp) {
                    comp = make_buffer(ex, (size_t)n * 4 + 8);
                    TIMED_LAUNCH_N(ex, "compose_indices", n, launch_compose_indices(ex.cfg(), c.view_idx->as<uint32_t>(), idx, n, comp->as<uint32_t>()));
                }
                if (keep_views) {
                    Column& o = out[i];
                    o.dtype = c.dtype;
                    o.length = n;
                    o.view_base = c.view_base;
                    o.view_idx = comp;
                    o.view_may_null = c.view_may_null || may_null;
                } else {
                    auto& grp = through[comp.get()];
                    grp.first.push_back(c.view_base.get());
                    grp.second.push_back(i);
                    through_null[comp.get()] = through_null[comp.get()] || c.view_may_null || may_null;
                }
            } else if (keep_views) {
                if (!own_idx && idx_owner && idx_owner->ptr() == (const void*)idx) own_idx = idx_owner;     // the caller's own buffer: kept, not copied
                if (!own_idx) {
                    own_idx = make_buffer(ex, (size_t)n * 4 + 8);
                    if (n) HIP_CHECK(hipMemcpyAsync(own_idx->ptr(), idx, (size_t)n * 4, hipMemcpyDeviceToDevice, ex.stream));
                }
                Column& o = out[i];
                o.dtype = c.dtype;
                o.length = n;
                o.view_base = std::make_shared<Column>(c);
                o.view_idx = own_idx;
                o.view_may_null = may_null;
            } else {
                plain.push_back(&c);
                plain_pos.push_back(i);
            }
        }
        for (auto& kv : through) {
            BufferPtr comp;
            for (auto& cc : composed)
                if (cc.second.get() == kv.first) comp = cc.second;
            auto got = take_columns(ex, kv.second.first, comp->as<uint32_t>(), n, through_null[kv.first], false, false);
            for (size_t k = 0; k < got.size(); ++k) out[kv.second.second[k]] = std::move(got[k]);
        }
        if (!plain.empty()) {
            auto got = take_columns(ex, plain, idx, n, may_null, permutation, false);
            for (size_t k = 0; k < got.size(); ++k) out[plain_pos[k]] = std::move(got[k]);
        }
        return out;
    }
    const std::vector<const Column*>& cols = cols_in;
    std::vector<Column> out(cols.size());
    TakeMany tm;
    tm.n = 0;
    auto flush = [&]() {
        if (tm.n) TIMED_LAUNCH_N(ex, "take_many", n, launch_take_many(ex.cfg(), tm, idx, n));
        tm.n = 0;
    };
    auto add = [&](const void* src, void* dst, int width) {
        if (tm.n == TAKE_MANY_MAX) flush();
        tm.src[tm.n] = src; tm.dst[tm.n] = dst; tm.width[tm.n] = width;
        ++tm.n;
    };
    for (size_t i = 0; i < cols.size(); ++i) {
        const Column& c = *cols[i];
        Column& o = out[i];
        if (c.dtype == DT_UTF8) {
            o = take_column(ex, c, idx, n, permutation ? c.data_bytes : -1);
        } else {
            o.dtype = c.dtype;
            o.length = n;
            if (c.dtype == DT_BOOLEAN) {
                o.data = make_buffer(ex, bitmap_bytes(n) + 8);
                add(c.data->ptr(), o.data->ptr(), 0);
            } else {
                const int w = dtype_width(c.dtype);
                o.data = make_buffer(ex, (size_t)n * w + 8);
                add(c.data->ptr(), o.data->ptr(), w);
            }
        }
        if (c.validity || may_null) {
            o.validity = make_buffer(ex, bitmap_bytes(n) + 8);
            add(c.validity ? c.validity->ptr() : nullptr, o.validity->ptr(), 0);
        }
    }
    flush();
    return out;
}

BatchPtr take_batch(const Exec& ex, const Batch& in, const uint32_t* idx, int64_t n_out, SchemaPtr schema, bool permutation) {
    auto out = std::make_shared<Batch>();
    out->schema = schema ? schema : in.schema;
    out->ctx = in.ctx;
    out->n_rows = n_out;
    std::vector<const Column*> cols;
    for (const auto& c : in.cols) cols.push_back(&c);
    out->cols = take_columns(ex, cols, idx, n_out, false, permutation);
    return out;
}

Column materialize_column(const Exec& ex, const Column& c) {
    if (!c.is_view()) return c;
    return take_column_v(ex, *c.view_base, c.view_idx->as<uint32_t>(), c.length, c.view_may_null);
}
BatchPtr materialize_batch(const Exec& ex, const BatchPtr& b) {
    bool any = false;
    for (auto& c : b->cols) any = any || c.is_view();
    if (!any) return b;
    auto out = std::make_shared<Batch>(*b);
    for (auto& c : out->cols) c = materialize_column(ex, c);
    return out;
}

Column take_column_nullable(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n) {
    return take_column_v(ex, c, idx, n, true);
}

Column take_batch_column(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n) {
    return take_column_v(ex, c, idx, n, false);
}

Column null_column(const Exec& ex, int dtype, int64_t n) {
    Column c;
    c.dtype = dtype;
    c.length = n;
    const size_t bytes = dtype == DT_UTF8 ? 8 : (dtype == DT_BOOLEAN ? bitmap_bytes(n) : (size_t)n * dtype_width(dtype));
    c.data = make_buffer(ex, bytes + 8);
    HIP_CHECK(hipMemsetAsync(c.data->ptr(), 0, bytes + 8, ex.stream));
    if (dtype == DT_UTF8) {
        c.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
        HIP_CHECK(hipMemsetAsync(c.offsets->ptr(), 0, (size_t)(n + 1) * 4, ex.stream));
    }
    c.validity = make_buffer(ex, bitmap_bytes(n) + 8);
    HIP_CHECK(hipMemsetAsync(c.validity->ptr(), 0, bitmap_bytes(n) + 8, ex.stream));
    return c;
}

// ---- concat ------------------------------------------------------------------------------------------
BatchPtr concat_batches(const Exec& ex, const SchemaPtr& schema, const std::vector<BatchPtr>& parts) {
    if (parts.size() == 1) return parts[0];
    const LaunchCfg cfg = ex.cfg();
    auto out = std::make_shared<Batch>();
    out->schema = schema;
    out->ctx = ex.ctx;
    int64_t total = 0;
    for (auto& p : parts) total += p->n_rows;
    if (total > 0xFFFFFFF0ll) fail(BHIP_EEXEC, "concatenated batch exceeds 2^32 rows");
    out->n_rows = total;
    const int n_cols = (int)schema->fields.size();
    for (int ci = 0; ci < n_cols; ++ci) {
        Column oc;
        oc.dtype = schema->fields[ci].dtype;
        oc.length = total;
        bool any_validity = false;
        for (auto& p : parts) any_validity |= (bool)p->cols[ci].validity;
        if (any_validity) {
            oc.validity = make_buffer(ex, bitmap_bytes(total) + 8);
            HIP_CHECK(hipMemsetAsync(oc.validity->ptr(), 0, bitmap_bytes(total) + 8, ex.stream));
        }
        if (oc.dtype == DT_UTF8) {
            int64_t bytes = 0;
            for (auto& p : parts) bytes += p->cols[ci].data_bytes;
            if (bytes > 0x7FFFFFFFll) fail(BHIP_EEXEC, "Utf8 column exceeds 2 GiB of value bytes");
            oc.data_bytes = bytes;
            oc.data = make_buffer(ex, (size_t)bytes + 8);
            oc.offsets = make_buffer(ex, (size_t)(total + 1) * 4);
            if (total == 0) HIP_CHECK(hipMemsetAsync(oc.offsets->ptr(), 0, 4, ex.stream));
        } else if (oc.dtype == DT_BOOLEAN) {
            oc.data = make_buffer(ex, bitmap_bytes(total) + 8);
            HIP_CHECK(hipMemsetAsync(oc.data->ptr(), 0, bitmap_bytes(total) + 8, ex.stream));
        } else {
            oc.data = make_buffer(ex, (size_t)total * dtype_width(oc.dtype) + 8);
        }
        int64_t row = 0, byte = 0;
        for (auto& p : parts) {
            const Column& c = p->cols[ci];
            const int64_t n = p->n_rows;
            if (n == 0) continue;
            if (oc.dtype == DT_UTF8) {
                TIMED_LAUNCH(ex, "rebase_offsets", launch_rebase_offsets(cfg, c.offsets->as<int32_t>(), n + 1, (int32_t)byte, oc.offsets->as<int32_t>() + row));
                if (c.data_bytes)
                    HIP_CHECK(hipMemcpyAsync(oc.data->as<uint8_t>() + byte, c.data->ptr(), (size_t)c.data_bytes,
                                             hipMemcpyDeviceToDevice, ex.stream));
                byte += c.data_bytes;
            } else if (oc.dtype == DT_BOOLEAN) {
                TIMED_LAUNCH(ex, "copy_bits", launch_copy_bits(cfg, c.data->as<uint64_t>(), 0, oc.data->as<uint64_t>(), row, n));
            } else {
                const int w = dtype_width(oc.dtype);
                HIP_CHECK(hipMemcpyAsync(oc.data->as<uint8_t>() + row * w, c.data->ptr(), (size_t)n * w,
                                         hipMemcpyDeviceToDevice, ex.stream));
            }
            if (any_validity)
                TIMED_LAUNCH(ex, "copy_bits", launch_copy_bits(cfg, c.validity ? c.validity->as<uint64_t>() : nullptr, 0,
                                           oc.validity->as<uint64_t>(), row, n));
            row += n;
        }
        out->cols.push_back(std::move(oc));
    }
    return out;
}

BatchPtr slice_head(const Exec& ex, const Batch& in, int64_t n) {
    if (n >= in.n_rows) return std::make_shared<Batch>(in);
    auto out = std::make_shared<Batch>(in);
    out->n_rows = n;
    for (auto& c : out->cols) {
        c.length = n;
        if (c.dtype == DT_UTF8) c.data_bytes = read_device(ex, c.offsets->as<int32_t>() + n);
    }
    return out;
}

// ---- filter ------------------------------------------------------------------------------------------
int64_t filter_indices(const Exec& ex, const Batch& in, const ExprPtr& predicate, BufferPtr& indices_out) {
    if (has_utf8_node(predicate, *in.schema)) {
        // lower(s) = 'x', CASE ... THEN 'a' ... : the string nodes become columns first
        Utf8Lowering low(*in.schema);
        const ExprPtr p2 = low.rewrite(predicate);
        const BatchPtr aug = low.apply(ex, in);
        return filter_indices(ex, *aug, p2, indices_out);
    }
    ProgramBuilder pb(*in.schema);
    pb.set_predicate(predicate);
    ScanParams P;
    pb.finish(P);
    ProgramBuilder::bind(P, pb.columns(), in, pb.creates_nulls());
    Temp tmp(ex);
    const int64_t n = in.n_rows;
    const int64_t n_tiles = (n + SEL_TILE - 1) / SEL_TILE;
    uint64_t* bitmap = tmp.get<uint64_t>((size_t)(n + 63) / 64 + 1);
    uint32_t* tile_counts = tmp.get<uint32_t>((size_t)n_tiles + 1);
    ScanStatus* st = nullptr;                       // the expression VM's error flags: only when it is the VM that runs
    if (n == 0) { indices_out = make_buffer(ex, 8); return 0; }
    // AND of column-vs-literal comparisons over NULL-free numeric columns: the wide-load range kernel
    // (kernels_range.hip) writes the same bitmap + tile counts as the expression VM
    static const bool range_disabled = [] { const char* v = getenv("BHIP_NO_RANGE_FILTER"); return v && atoi(v) != 0; }();
    // Utf8 column = / != literal (Q3's c_mktsegment = 'BUILDING'): offsets, the length test, the bytes — no interpreter
    static const bool utf8_eq_disabled = [] { const char* v = getenv("BHIP_NO_UTF8_EQ_FILTER"); return v && atoi(v) != 0; }();
    const Column* str_col = nullptr;
    Utf8Literal str_lit;
    bool str_negate = false;
    if (!utf8_eq_disabled && predicate->kind == BHIP_EXPR_BINARY && (predicate->name == "Eq" || predicate->name == "NotEq") && predicate->args.size() == 2) {
        ExprPtr a = predicate->args[0], b = predicate->args[1];
        if (a->kind == BHIP_EXPR_LITERAL) std::swap(a, b);
        if (a->kind == BHIP_EXPR_COLUMN && b->kind == BHIP_EXPR_LITERAL && b->dtype == DT_UTF8 && !b->is_null && b->name.size() <= sizeof(str_lit.bytes)) {
            const int ci = in.schema->index_of(a->name);
            if (ci >= 0 && in.cols[ci].dtype == DT_UTF8 && !in.cols[ci].is_view()) {
                str_col = &in.cols[ci];
                memset(&str_lit, 0, sizeof(str_lit));
                memcpy(str_lit.bytes, b->name.data(), b->name.size());
                str_lit.len = (int32_t)b->name.size();
                str_negate = predicate->name == "NotEq";
            }
        }
    }
    SopPlan rp;
    if (str_col) {
        TIMED_LAUNCH_N(ex, "utf8_eq_bitmap", n, launch_utf8_eq_bitmap(ex.cfg(), str_col->offsets->as<int32_t>(), str_col->data->ptr(),
                                                                    str_col->validity ? str_col->validity->as<uint64_t>() : nullptr, n, str_lit, str_negate,
                                                                    bitmap, tile_counts));
    } else if (!range_disabled && build_sop(*in.schema, predicate, {}, {}, rp) && rp.prog.n_ranges >= 1 && sop_columns_bindable(rp, in, true) &&
        lean_bindable(rp, in)) {
        bind_sop(rp, in);
        TIMED_LAUNCH_N(ex, "range_bitmap", n, launch_range_bitmap(ex.cfg(), rp.prog, tmp.get<SopProgram>(1), bitmap, tile_counts));
    } else {
        st = new_status(tmp);
        TIMED_LAUNCH_N(ex, "scan_pred_bitmap", n, launch_scan_pred_bitmap(ex.cfg(), P, bitmap, tile_counts, st));
    }
    uint64_t* tile_off = tmp.get<uint64_t>((size_t)n_tiles + 1);
    uint64_t* total = tmp.get<uint64_t>(1);
    void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_tiles));
    HIP_CHECK(exclusive_scan_u32_u64(ex.stream, tile_counts, n_tiles, tile_off, false, total, scan_tmp));
    const uint64_t count = read_device(ex, total);
    if (st && pb.can_raise()) check_scan_status(ex, st);       // (after the wait above: immediate; the specialised kernels raise nothing)
    indices_out = make_buffer(ex, (size_t)count * 4 + 8);
    if (count) TIMED_LAUNCH_N(ex, "select_indices", n, launch_select_indices(ex.cfg(), bitmap, tile_off, n, indices_out->as<uint32_t>()));
    return (int64_t)count;
}

// ---- MemoryExec / EmptyExec --------------------------------------------------------------------------
MemoryExec::MemoryExec(ContextPtr ctx, SchemaPtr schema, std::vector<std::vector<BatchPtr>> partitions)
    : schema_(std::move(schema)), parts_(std::move(partitions)) {
    ctx_ = std::move(ctx);
}
PlanPtr MemoryExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (!c.empty()) fail(BHIP_EINVAL, "MemoryExec has no children");
    return shared_from_this();
}
StreamPtr MemoryExec::execute(int partition, const Exec&) const {
    check_partition(*this, partition);
    return StreamPtr(new VecStream(schema_, parts_[partition]));
}

EmptyExec::EmptyExec(ContextPtr ctx, SchemaPtr schema, bool produce_one_row) : schema_(std::move(schema)), one_row_(produce_one_row) {
    ctx_ = std::move(ctx);
}
PlanPtr EmptyExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (!c.empty()) fail(BHIP_EINVAL, "EmptyExec has no children");
    return shared_from_this();
}
StreamPtr EmptyExec::execute(int partition, const Exec&) const {
    check_partition(*this, partition);
    if (one_row_) fail(BHIP_ENOTIMPL, "EmptyExec with produce_one_row");
    return StreamPtr(new VecStream(schema_, {}));
}

// ---- FilterExec ----------------------------------------------------------------------------------------
FilterExec::FilterExec(ExprPtr predicate, PlanPtr input) : predicate_(std::move(predicate)) {
    input_ = std::move(input);
    ctx_ = input_->context();
    // type-check now, like FilterExec::try_new
    if (expr_type(predicate_, *input_->schema()) != DT_BOOLEAN)
        fail(BHIP_EINVAL, "Filter predicate must return boolean values, not " +
                              std::string(dtype_name(expr_type(predicate_, *input_->schema()))));
    Utf8Lowering low(*input_->schema());
    const ExprPtr lowered = low.rewrite(predicate_);
    low.validate();
    const SchemaPtr aug = low.schema();
    ProgramBuilder pb(*aug);
    pb.set_predicate(lowered);      // surfaces BHIP_ENOTIMPL at plan time
}
PlanPtr FilterExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "FilterExec wrong number of children");
    return std::make_shared<FilterExec>(predicate_, c[0]);
}
StreamPtr FilterExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    auto child = std::shared_ptr<RecordBatchStream>(input_->execute(partition, ex).release());
    ExprPtr pred = predicate_;
    SchemaPtr sch = schema();
    return StreamPtr(new LazyStream(sch, [child, pred, ex, sch]() {
        std::vector<BatchPtr> out;
        while (BatchPtr b = child->next()) {
            BufferPtr idx;
            const int64_t n = filter_indices(ex, *b, pred, idx);
            if (n == b->n_rows) { out.push_back(b); continue; }
            out.push_back(take_batch(ex, *b, idx->as<uint32_t>(), n));
        }
        return out;
    }));
}

// ---- ProjectionExec --------------------------------------------------------------------------------------
ProjectionExec::ProjectionExec(std::vector<std::pair<ExprPtr, std::string>> exprs, PlanPtr input) : exprs_(std::move(exprs)) {
    input_ = std::move(input);
    ctx_ = input_->context();
    auto s = std::make_shared<Schema>();
    const Schema& in = *input_->schema();
    Utf8Lowering low(in);
    std::vector<ExprPtr> lowered;
    for (auto& en : exprs_) {
        const int t = expr_type(en.first, in);
        s->fields.push_back(Field{en.second, t, expr_nullable(en.first, in), t == DT_UTF8 && expr_large(en.first, in), t == DT_UTF8 && expr_binary(en.first, in)});
        lowered.push_back(low.rewrite(en.first, /*output=*/true));
    }
    low.validate();
    const SchemaPtr aug = low.schema();
    ProgramBuilder pb(*aug);                                  // plan-time check of everything the VM will be asked to do
    for (auto& e : lowered)
        if (e->kind != BHIP_EXPR_COLUMN) pb.add_output(e);
    schema_ = s;
}
PlanPtr ProjectionExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "ProjectionExec wrong number of children");
    return std::make_shared<ProjectionExec>(exprs_, c[0]);
}
std::string ProjectionExec::describe() const {
    std::string s = "ProjectionExec: expr=[";
    for (size_t i = 0; i < exprs_.size(); ++i) s += (i ? ", " : "") + exprs_[i].first->to_string() + " as " + exprs_[i].second;
    return s + "]";
}

BatchPtr project_batch(const Exec& ex, const Batch& in, const std::vector<std::pair<ExprPtr, std::string>>& exprs,
                       const SchemaPtr& schema) {
    {
        bool strings = false;
        for (auto& en : exprs) strings = strings || has_utf8_node(en.first, *in.schema) || (en.first->kind == BHIP_EXPR_LITERAL && en.first->dtype == DT_UTF8);
        if (strings) {
            Utf8Lowering low(*in.schema);
            std::vector<std::pair<ExprPtr, std::string>> e2;
            for (auto& en : exprs) e2.push_back({low.rewrite(en.first, /*output=*/true), en.second});
            const BatchPtr aug = low.apply(ex, in);
            return project_batch(ex, *aug, e2, schema);
        }
    }
    auto out = std::make_shared<Batch>();
    out->schema = schema;
    out->ctx = in.ctx;
    out->n_rows = in.n_rows;
    out->cols.resize(exprs.size());
    ProgramBuilder pb(*in.schema);
    std::vector<int> computed;
    for (size_t i = 0; i < exprs.size(); ++i) {
        if (exprs[i].first->kind == BHIP_EXPR_COLUMN) {
            const int ci = in.schema->index_of(exprs[i].first->name);
            if (ci < 0) fail(BHIP_EINVAL, "No field named '" + exprs[i].first->name + "'");
            out->cols[i] = in.cols[ci];      // zero copy: buffers are shared
        } else {
            pb.add_output(exprs[i].first);
            computed.push_back((int)i);
        }
    }
    if (computed.empty() || in.n_rows == 0) {
        for (int i : computed) {
            Column c;
            c.dtype = schema->fields[i].dtype;
            c.data = make_buffer(ex, 8);
            out->cols[i] = c;
        }
        return out;
    }
    ScanParams P;
    pb.finish(P);
    ProgramBuilder::bind(P, pb.columns(), in, pb.creates_nulls());
    ProjectOut po;
    memset(&po, 0, sizeof(po));
    const int64_t n = in.n_rows;
    for (size_t k = 0; k < computed.size(); ++k) {
        Column c;
        c.dtype = schema->fields[computed[k]].dtype;
        c.length = n;
        const size_t bytes = c.dtype == DT_BOOLEAN ? bitmap_bytes(n) : (size_t)n * dtype_width(c.dtype);
        c.data = make_buffer(ex, bytes + 8);
        po.data[k] = c.data->ptr();
        if (P.prog.nullable) {
            c.validity = make_buffer(ex, bitmap_bytes(n) + 8);
            po.validity[k] = c.validity->as<uint64_t>();
        }
        out->cols[computed[k]] = c;
    }
    Temp tmp(ex);
    ScanStatus* st = new_status(tmp);
    TIMED_LAUNCH(ex, "scan_project", launch_scan_project(ex.cfg(), P, po, st));
    // only an integer division can fail; every other projection is handed on without a host round trip (stream order)
    if (pb.can_raise()) check_scan_status(ex, st);
    return out;
}

Column evaluate_column(const Exec& ex, const Batch& in, const ExprPtr& e) {
    auto s = std::make_shared<Schema>();
    s->fields.push_back(Field{"v", expr_type(e, *in.schema), expr_nullable(e, *in.schema)});
    BatchPtr b = project_batch(ex, in, {{e, "v"}}, s);
    return b->cols[0];
}

StreamPtr ProjectionExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    auto exprs = exprs_;
    SchemaPtr sch = schema_;
    // Projection over (CoalesceBatches over) Filter: the reference's filter copies EVERY column of the surviving
    // rows and the projection then drops most of them; here only the columns the projection
    // reads are gathered.  Same rows, same order, same values.
    const ExecutionPlan* below = input_.get();
    while (auto co = dynamic_cast<const CoalesceBatchesExec*>(below)) below = co->children()[0].get();
    if (auto flt = dynamic_cast<const FilterExec*>(below)) {
        PlanPtr src = flt->children()[0];
        const ExprPtr pred = flt->predicate();
        const SchemaPtr in_schema = src->schema();
        std::vector<std::string> used;
        for (auto& en : exprs) collect_columns(en.first, used);
        auto narrow = std::make_shared<Schema>();
        std::vector<int> keep;
        for (size_t i = 0; i < in_schema->fields.size(); ++i)
            for (auto& u : used)
                if (u == in_schema->fields[i].name) { narrow->fields.push_back(in_schema->fields[i]); keep.push_back((int)i); break; }
        if (keep.size() < in_schema->fields.size()) {
            auto child = std::shared_ptr<RecordBatchStream>(src->execute(partition, ex).release());
            SchemaPtr nsch = narrow;
            return StreamPtr(new LazyStream(sch, [child, exprs, ex, sch, pred, nsch, keep]() {
                std::vector<BatchPtr> out;
                while (BatchPtr b = child->next()) {
                    BufferPtr idx;
                    const int64_t n = filter_indices(ex, *b, pred, idx);
                    auto nb = std::make_shared<Batch>();
                    nb->schema = nsch;
                    nb->ctx = b->ctx;
                    nb->n_rows = b->n_rows;
                    for (int ci : keep) nb->cols.push_back(b->cols[ci]);
                    BatchPtr sel = n == b->n_rows ? BatchPtr(nb) : take_batch(ex, *nb, idx->as<uint32_t>(), n);
                    out.push_back(project_batch(ex, *sel, exprs, sch));
                }
                return out;
            }));
        }
    }
    std::shared_ptr<RecordBatchStream> child;
    if (auto hj = dynamic_cast<const HashJoinExec*>(input_.get())) {
        // Projection over a join: the join gathers only the columns read here (DataFusion's join copies all of both sides)
        const Schema& js = *hj->schema();
        std::vector<std::string> used;
        for (auto& en : exprs) collect_columns(en.first, used);
        std::vector<bool> needed(js.fields.size(), false);
        for (auto& u : used) {
            const int i = js.index_of(u);
            if (i >= 0) needed[i] = true;
        }
        child = std::shared_ptr<RecordBatchStream>(hj->execute_needed(partition, ex, needed).release());
    } else {
        child = std::shared_ptr<RecordBatchStream>(input_->execute(partition, ex).release());
    }
    return StreamPtr(new LazyStream(sch, [child, exprs, ex, sch]() {
        std::vector<BatchPtr> out;
        while (BatchPtr b = child->next()) out.push_back(project_batch(ex, *b, exprs, sch));
        return out;
    }));
}

// ---- CoalesceBatchesExec / MergeExec / LimitExec -----------------------------------------------------------
CoalesceBatchesExec::CoalesceBatchesExec(PlanPtr input, int64_t target) : target_(target) {
    input_ = std::move(input);
    ctx_ = input_->context();
}
PlanPtr CoalesceBatchesExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "CoalesceBatchesExec wrong number of children");
    return std::make_shared<CoalesceBatchesExec>(c[0], target_);
}
StreamPtr CoalesceBatchesExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    auto child = std::shared_ptr<RecordBatchStream>(input_->execute(partition, ex).release());
    SchemaPtr sch = schema();
    const int64_t target = target_;
    return StreamPtr(new LazyStream(sch, [child, ex, sch, target]() {
        // buffer small batches until >= target rows, then concatenate; large batches pass through
        std::vector<BatchPtr> out, pending;
        int64_t rows = 0;
        auto flush = [&]() {
            if (pending.empty()) return;
            out.push_back(concat_batches(ex, sch, pending));
            pending.clear();
            rows = 0;
        };
        while (BatchPtr b = child->next()) {
            if (b->n_rows == 0) continue;
            if (b->n_rows >= target && pending.empty()) { out.push_back(b); continue; }
            pending.push_back(b);
            rows += b->n_rows;
            if (rows >= target) flush();
        }
        flush();
        return out;
    }));
}

MergeExec::MergeExec(PlanPtr input) {
    input_ = std::move(input);
    ctx_ = input_->context();
}
PlanPtr MergeExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "MergeExec wrong number of children");
    return std::make_shared<MergeExec>(c[0]);
}
StreamPtr MergeExec::execute(int partition, const Exec& ex) const {
    if (partition != 0) fail(BHIP_EINVAL, "MergeExec invalid partition " + std::to_string(partition));
    PlanPtr in = input_;
    SchemaPtr sch = schema();
    return StreamPtr(new LazyStream(sch, [in, ex]() {
        std::vector<BatchPtr> out;
        const int n = in->output_partitioning().count;
        for (int p = 0; p < n; ++p) {
            auto s = in->execute(p, ex);
            while (BatchPtr b = s->next()) out.push_back(b);
        }
        return out;
    }));
}

LimitExec::LimitExec(PlanPtr input, int64_t limit, bool global) : limit_(limit), global_(global) {
    input_ = std::move(input);
    ctx_ = input_->context();
    if (limit < 0) fail(BHIP_EINVAL, "negative limit");
}
Partitioning LimitExec::output_partitioning() const {
    return global_ ? Partitioning{BHIP_PART_UNKNOWN, 1, {}} : input_->output_partitioning();
}
PlanPtr LimitExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "LimitExec wrong number of children");
    return std::make_shared<LimitExec>(c[0], limit_, global_);
}
StreamPtr LimitExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    if (global_ && input_->output_partitioning().count != 1)
        fail(BHIP_EINVAL, "GlobalLimitExec requires a single input partition");
    auto child = std::shared_ptr<RecordBatchStream>(input_->execute(partition, ex).release());
    SchemaPtr sch = schema();
    const int64_t limit = limit_;
    return StreamPtr(new LazyStream(sch, [child, ex, limit]() {
        std::vector<BatchPtr> out;
        int64_t left = limit;
        while (left > 0) {
            BatchPtr b = child->next();
            if (!b) break;
            if (b->n_rows <= left) { out.push_back(b); left -= b->n_rows; }
            else { out.push_back(slice_head(ex, *b, left)); left = 0; }
        }
        return out;
    }));
}

}  // namespace bhip
