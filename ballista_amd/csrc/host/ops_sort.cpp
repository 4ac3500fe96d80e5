// ops_sort.cpp — SortExec and RepartitionExec over the stable radix pass (kernels_sort.hip).
//
// SortExec: rust/core/src/serde/physical_plan/from_proto.rs:291-331 — single input partition, all
// rows, lexicographic over PhysicalSortExpr{expr, descending, nulls_first}.  Implemented as LSD
// over the sort keys (last key first), each key an order-preserving u64 image (Utf8: 8-byte
// big-endian chunks, then the length), NULL placement as one more 1-bit key.  Ties keep input
// order (the reference leaves tie order unspecified).
//
// RepartitionExec: from_proto.rs:133-164 — Hash(exprs, n): row -> partition row_hash % n (equal
// keys co-locate; rows keep their input order inside a partition); RoundRobinBatch(n): batch i ->
// partition i mod n.
#include "../sort_kernels.h"
#include "../util_kernels.h"
#include "plan.hpp"

namespace bhip {

void radix_sort_pairs(const Exec& ex, BufferPtr& keys, BufferPtr& perm, int64_t n) {
    if (n <= 1) return;
    const LaunchCfg cfg = ex.cfg();
    if (n <= small_sort_max()) {
        HIP_CHECK(small_sort_pairs(cfg, keys->as<uint64_t>(), perm->as<uint32_t>(), n));
        return;
    }
    Temp tmp(ex);
    uint64_t* diff_dev = tmp.get<uint64_t>(1);
    HIP_CHECK(radix_key_diff(cfg, keys->as<uint64_t>(), n, diff_dev));
    const uint64_t diff = read_device(ex, diff_dev);
    if (diff == 0) return;   // all keys equal: already in order
    BufferPtr keys2 = make_buffer(ex, (size_t)n * 8 + 8), perm2 = make_buffer(ex, (size_t)n * 4 + 8);
    void* pass_tmp = tmp.get<uint8_t>(radix_sort_temp_bytes(n));
    for (int byte = 0; byte < 8; ++byte) {
        if (((diff >> (8 * byte)) & 0xFF) == 0) continue;   // this byte is the same in every key
        HIP_CHECK(radix_pass(cfg, keys->as<uint64_t>(), perm->as<uint32_t>(), n, byte, keys2->as<uint64_t>(),
                             perm2->as<uint32_t>(), pass_tmp));
        std::swap(keys, keys2);
        std::swap(perm, perm2);
    }
}

// ---- SortExec ---------------------------------------------------------------------------------------
SortExec::SortExec(std::vector<SortDesc> exprs, PlanPtr input) : exprs_(std::move(exprs)) {
    input_ = std::move(input);
    ctx_ = input_->context();
    Utf8Lowering low(*input_->schema());
    for (auto& s : exprs_) {
        expr_type(s.expr, *input_->schema());
        low.rewrite(s.expr, true);
    }
    low.validate();                        // a key like lower(s) is evaluated as a column first (utf8_exprs.cpp)
}
PlanPtr SortExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "SortExec wrong number of children");
    return std::make_shared<SortExec>(exprs_, c[0]);
}
std::string SortExec::describe() const {
    std::string s = "SortExec: [";
    for (size_t i = 0; i < exprs_.size(); ++i)
        s += (i ? ", " : "") + exprs_[i].expr->to_string() + (exprs_[i].descending ? " DESC" : " ASC") +
             (exprs_[i].nulls_first ? " NULLS FIRST" : " NULLS LAST");
    return s + "]";
}

StreamPtr SortExec::execute(int partition, const Exec& ex) const {
    if (partition != 0) fail(BHIP_EINVAL, "SortExec invalid partition " + std::to_string(partition));
    if (input_->output_partitioning().count != 1) fail(BHIP_EINVAL, "SortExec requires a single input partition");
    auto self = std::static_pointer_cast<const SortExec>(shared_from_this());
    return StreamPtr(new LazyStream(schema(), [self, ex]() -> std::vector<BatchPtr> {
        std::vector<BatchPtr> parts;
        {
            auto s = self->input_->execute(0, ex);
            while (BatchPtr b = s->next())
                if (b->n_rows > 0) parts.push_back(b);
        }
        if (parts.empty()) return {};
        trace_point("sort: input ready");
        BatchPtr in = concat_batches(ex, self->schema(), parts);
        const int64_t n = in->n_rows;
        const LaunchCfg cfg = ex.cfg();
        static const bool no_rowsort = [] { const char* v = getenv("BHIP_NO_ROWSORT"); return v && atoi(v) != 0; }();
        if (!no_rowsort && n <= ROWSORT_MAX_ROWS && self->exprs_.size() <= (size_t)ROWSORT_MAX_KEYS && in->cols.size() <= (size_t)ROWSORT_MAX_COLS) {
            // a handful of rows (the result of a low-cardinality aggregate): ranks by row comparison + the gather of every column, one launch
            RowSortArgs A;
            memset(&A, 0, sizeof(A));
            A.n_rows = (int32_t)n;
            A.n_keys = (int32_t)self->exprs_.size();
            A.n_cols = (int32_t)in->cols.size();
            std::vector<Column> key_cols;                     // keeps computed key columns alive until the launch is queued
            for (size_t k = 0; k < self->exprs_.size(); ++k) {
                const SortDesc& sd = self->exprs_[k];
                key_cols.push_back(evaluate_column(ex, *in, sd.expr));
                A.key[k] = key_cols.back().ref();
                A.desc[k] = sd.descending ? 1 : 0;
                A.nulls_first[k] = sd.nulls_first ? 1 : 0;
            }
            auto out = std::make_shared<Batch>();
            out->schema = in->schema;
            out->ctx = in->ctx;
            out->n_rows = n;
            for (size_t ci = 0; ci < in->cols.size(); ++ci) {
                const Column& c = in->cols[ci];
                Column o;
                o.dtype = c.dtype;
                o.length = n;
                A.col[ci] = c.ref();
                if (c.dtype == DT_UTF8) {
                    o.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
                    o.data = make_buffer(ex, (size_t)c.data_bytes + 8);
                    o.data_bytes = c.data_bytes;               // a permutation keeps the value bytes
                    A.out_offsets[ci] = o.offsets->as<int32_t>();
                } else if (c.dtype == DT_BOOLEAN) {
                    o.data = make_buffer(ex, bitmap_bytes(n) + 8);
                } else {
                    A.width[ci] = (uint8_t)dtype_width(c.dtype);
                    o.data = make_buffer(ex, (size_t)n * dtype_width(c.dtype) + 8);
                }
                A.out_data[ci] = o.data->ptr();
                if (c.validity) {
                    o.validity = make_buffer(ex, bitmap_bytes(n) + 8);
                    A.out_validity[ci] = o.validity->as<uint64_t>();
                }
                out->cols.push_back(std::move(o));
            }
            TIMED_LAUNCH(ex, "rowsort", launch_rowsort(cfg, A));
            trace_point("sort: queued");
            return {out};                                       // stream order: the consumer's work (or its wait) comes after the launch
        }
        BufferPtr perm = sort_permutation(ex, *in, self->exprs_);
        BatchPtr out = take_batch(ex, *in, perm->as<uint32_t>(), n, nullptr, /*permutation=*/true);
        return {out};                                           // (no wait: scratch is released in stream order)
    }));
}

// the stable order of `in`'s rows under the keys (LSD: last key first, each key an order-preserving u64 image)
BufferPtr sort_permutation(const Exec& ex, const Batch& batch, const std::vector<SortDesc>& exprs) {
    const Batch* in = &batch;
    const int64_t n = batch.n_rows;
    const LaunchCfg cfg = ex.cfg();
    // mid-sized inputs under fixed-width keys: one split + ranks inside the bins (kernels_sort.hip: bucket_sort); a bin that overflows
    // (the leading differing bits repeat heavily) falls through to the LSD passes below, which are few exactly then
    static const bool no_bucket = [] { const char* v = getenv("BHIP_NO_BUCKET_SORT"); return v && atoi(v) != 0; }();
    if (!no_bucket && n > small_sort_max() && n <= bucket_sort_max_rows()) {
        BucketSortKeys K;
        memset(&K, 0, sizeof(K));
        std::vector<Column> key_cols;                          // keeps computed key columns alive until the launches are queued
        bool fits = true;
        for (const SortDesc& sd : exprs) {
            key_cols.push_back(evaluate_column(ex, *in, sd.expr));
            const Column& col = key_cols.back();
            const int words = col.validity ? 2 : 1;
            if (col.dtype == DT_UTF8 || K.n_words + words > BSORT_MAX_WORDS) { fits = false; break; }
            if (col.validity) {
                K.col[K.n_words] = col.ref(); K.null_rank[K.n_words] = 1; K.nulls_first[K.n_words] = sd.nulls_first ? 1 : 0;
                ++K.n_words;
            }
            K.col[K.n_words] = col.ref(); K.desc[K.n_words] = sd.descending ? 1 : 0;
            ++K.n_words;
        }
        if (fits) {
            BufferPtr perm = make_buffer(ex, (size_t)n * 4 + 8);
            Temp tmp(ex);
            void* temp = tmp.get<uint8_t>(bucket_sort_temp_bytes(n, K.n_words));
            uint32_t* status_dev = nullptr;
            TIMED_LAUNCH_N(ex, "bucket_sort", n, bucket_sort(cfg, K, n, temp, perm->as<uint32_t>(), &status_dev));
            if (read_device(ex, status_dev) == 0) return perm;
            trace_point("sort: a bucket overflowed, LSD passes instead");
        }
    }
    {
        BufferPtr perm = make_buffer(ex, (size_t)n * 4 + 8);
        BufferPtr keys = make_buffer(ex, (size_t)n * 8 + 8);
        TIMED_LAUNCH(ex, "iota_u32", launch_iota_u32(cfg, perm->as<uint32_t>(), n, 0));
        for (size_t k = exprs.size(); k-- > 0;) {
            const SortDesc& sd = exprs[k];
            const Column col = evaluate_column(ex, *in, sd.expr);
            const ColumnRef cr = col.ref();
            if (col.dtype == DT_UTF8) {
                uint32_t maxlen = (uint32_t)col.data_bytes;      // an upper bound: passes over all-zero chunks keep the order
                if (col.data_bytes > 32) {
                    Temp tmp(ex);
                    uint32_t* maxlen_dev = tmp.get<uint32_t>(1);
                    TIMED_LAUNCH(ex, "utf8_max_len", launch_utf8_max_len(cfg, col.offsets->as<int32_t>(), n, maxlen_dev));
                    maxlen = read_device(ex, maxlen_dev);
                }
                // least significant first: the length, then the 8-byte chunks from the last to the first
                TIMED_LAUNCH(ex, "sort_key_utf8", launch_sort_key_utf8(cfg, cr, perm->as<uint32_t>(), n, -1, sd.descending, keys->as<uint64_t>()));
                radix_sort_pairs(ex, keys, perm, n);
                for (int chunk = (int)((maxlen + 7) / 8) - 1; chunk >= 0; --chunk) {
                    TIMED_LAUNCH(ex, "sort_key_utf8", launch_sort_key_utf8(cfg, cr, perm->as<uint32_t>(), n, chunk, sd.descending, keys->as<uint64_t>()));
                    radix_sort_pairs(ex, keys, perm, n);
                }
            } else {
                TIMED_LAUNCH(ex, "sort_key_fixed", launch_sort_key_fixed(cfg, cr, perm->as<uint32_t>(), n, sd.descending, keys->as<uint64_t>()));
                radix_sort_pairs(ex, keys, perm, n);
            }
            if (col.validity) {
                TIMED_LAUNCH(ex, "sort_key_null", launch_sort_key_null(cfg, col.validity->as<uint64_t>(), perm->as<uint32_t>(), n, sd.nulls_first,
                                               keys->as<uint64_t>()));
                radix_sort_pairs(ex, keys, perm, n);
            }
        }
        return perm;
    }
}

// ---- hash partitioning ----------------------------------------------------------------------------------
std::vector<BatchPtr> hash_partition_batch(const Exec& ex, const BatchPtr& in, const std::vector<ExprPtr>& exprs, int n_parts) {
    const int64_t n = in->n_rows;
    const LaunchCfg cfg = ex.cfg();
    std::vector<BatchPtr> out((size_t)n_parts);
    std::vector<uint32_t> first((size_t)n_parts + 1, 0);
    // ONE NULL-free integer key over fixed-width NULL-free columns (the exchange of the TPC-H joins): scatter pass
    {
        const Column* kc = nullptr;
        if (exprs.size() == 1 && exprs[0]->kind == BHIP_EXPR_COLUMN) {
            const int ci = in->schema->index_of(exprs[0]->name);
            if (ci >= 0) kc = &in->cols[ci];
        }
        const int kw = !kc || kc->validity ? 0 : (kc->dtype == DT_INT32 || kc->dtype == DT_DATE32) ? 4 : (kc->dtype == DT_INT64 || kc->dtype == DT_UINT64) ? 8 : 0;
        bool fixed = kw != 0 && n > 0 && n_parts <= 256 && (int)in->cols.size() <= TAKE_MANY_MAX;
        for (auto& c : in->cols) fixed = fixed && !c.validity && c.dtype != DT_UTF8 && c.dtype != DT_BOOLEAN;
        static const bool disabled = [] { const char* v = getenv("BHIP_NO_PARTITION_SCATTER"); return v && atoi(v) != 0; }();
        if (fixed && !disabled) {
            TakeMany tm;
            tm.n = 0;
            std::vector<BufferPtr> whole;
            for (auto& c : in->cols) {
                const int w = dtype_width(c.dtype);
                whole.push_back(make_buffer(ex, (size_t)n * w + 8));
                tm.src[tm.n] = c.data->ptr(); tm.dst[tm.n] = whole.back()->ptr(); tm.width[tm.n] = w;
                ++tm.n;
            }
            Temp tmp(ex);
            void* temp = tmp.get<uint8_t>(partition_scatter_temp_bytes(n));
            HIP_CHECK(partition_scatter(cfg, kc->data->ptr(), kw, n, (uint32_t)n_parts, tm, temp, first.data()));
            for (int p = 0; p < n_parts; ++p) {
                auto b = std::make_shared<Batch>();
                b->schema = in->schema;
                b->ctx = in->ctx;
                b->n_rows = (int64_t)first[p + 1] - (int64_t)first[p];
                for (size_t ci = 0; ci < in->cols.size(); ++ci) {
                    const int w = dtype_width(in->cols[ci].dtype);
                    Column c;
                    c.dtype = in->cols[ci].dtype;
                    c.length = b->n_rows;
                    // a partition's column is a slice of the scattered column (kept alive by the slice)
                    c.data = std::make_shared<Buffer>(ex.ctx, whole[ci], static_cast<uint8_t*>(whole[ci]->ptr()) + (size_t)first[p] * w,
                                                      (size_t)b->n_rows * w);
                    b->cols.push_back(std::move(c));
                }
                out[p] = b;
            }
            return out;
        }
    }
    BufferPtr perm = make_buffer(ex, (size_t)n * 4 + 8);
    if (n > 0) {
        ProgramBuilder pb(*in->schema);
        pb.set_hash_only();
        for (auto& e : exprs) pb.add_key(e);
        ScanParams P;
        pb.finish(P);
        ProgramBuilder::bind(P, pb.columns(), *in, pb.creates_nulls());
        Temp tmp(ex);
        uint64_t* hashes = tmp.get<uint64_t>((size_t)n);
        ScanStatus* st = tmp.get<ScanStatus>(1);
        HIP_CHECK(hipMemsetAsync(st, 0, sizeof(ScanStatus), ex.stream));
        TIMED_LAUNCH(ex, "scan_keys", launch_scan_keys(cfg, P, nullptr, hashes, nullptr, st));
        BufferPtr keys = make_buffer(ex, (size_t)n * 8 + 8);
        TIMED_LAUNCH(ex, "hash_to_pid", launch_hash_to_pid(cfg, hashes, n, (uint32_t)n_parts, keys->as<uint64_t>()));
        TIMED_LAUNCH(ex, "iota_u32", launch_iota_u32(cfg, perm->as<uint32_t>(), n, 0));
        check_scan_status(ex, st);
        radix_sort_pairs(ex, keys, perm, n);
        uint32_t* first_dev = tmp.get<uint32_t>((size_t)n_parts + 1);
        TIMED_LAUNCH(ex, "partition_bounds", launch_partition_bounds(cfg, keys->as<uint64_t>(), n, (uint32_t)n_parts, first_dev));
        HIP_CHECK(hipMemcpyAsync(first.data(), first_dev, ((size_t)n_parts + 1) * 4, hipMemcpyDeviceToHost, ex.stream));
        stream_wait(ex);
    }
    for (int p = 0; p < n_parts; ++p) {
        const int64_t cnt = (int64_t)first[p + 1] - (int64_t)first[p];
        out[p] = take_batch(ex, *in, perm->as<uint32_t>() + first[p], cnt);
    }
    return out;        // (`perm` is released in stream order; callers that publish the parts to other tasks wait themselves)
}

// ---- RepartitionExec ----------------------------------------------------------------------------------------
RepartitionExec::RepartitionExec(PlanPtr input, Partitioning part) : part_(std::move(part)) {
    input_ = std::move(input);
    ctx_ = input_->context();
    if (part_.scheme == BHIP_PART_HASH) {
        ProgramBuilder pb(*input_->schema());
        pb.set_hash_only();
        for (auto& e : part_.exprs) pb.add_key(e);   // type checks, BHIP_ENOTIMPL at plan time
    }
    cache_ = std::make_shared<SplitCache>();
}
PlanPtr RepartitionExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "RepartitionExec wrong number of children");
    return std::make_shared<RepartitionExec>(c[0], part_);
}
std::string RepartitionExec::describe() const {
    std::string s = "RepartitionExec: partitioning=";
    if (part_.scheme == BHIP_PART_HASH) {
        s += "Hash([";
        for (size_t i = 0; i < part_.exprs.size(); ++i) s += (i ? ", " : "") + part_.exprs[i]->to_string();
        s += "], " + std::to_string(part_.count) + ")";
    } else {
        s += (part_.scheme == BHIP_PART_ROUND_ROBIN ? "RoundRobinBatch(" : "UnknownPartitioning(") + std::to_string(part_.count) + ")";
    }
    return s;
}

StreamPtr RepartitionExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    auto self = std::static_pointer_cast<const RepartitionExec>(shared_from_this());
    return StreamPtr(new LazyStream(schema(), [self, partition, ex]() {
        std::lock_guard<std::mutex> g(self->cache_->mu);
        if (!self->cache_->done) {
            // the first task to arrive splits every input partition; the others take their share
            auto& parts = self->cache_->parts;
            parts.assign((size_t)self->part_.count, {});
            const int n_in = self->input_->output_partitioning().count;
            int64_t batch_no = 0;
            for (int p = 0; p < n_in; ++p) {
                auto s = self->input_->execute(p, ex);
                while (BatchPtr b = s->next()) {
                    if (self->part_.scheme == BHIP_PART_HASH) {
                        auto split = hash_partition_batch(ex, b, self->part_.exprs, self->part_.count);
                        for (int q = 0; q < self->part_.count; ++q)
                            if (split[q]->n_rows > 0) parts[q].push_back(split[q]);
                    } else {
                        parts[(size_t)(batch_no % self->part_.count)].push_back(b);
                    }
                    ++batch_no;
                }
            }
            stream_wait(ex);
            self->cache_->done = true;
        }
        return self->cache_->parts[(size_t)partition];
    }));
}

}  // namespace bhip
