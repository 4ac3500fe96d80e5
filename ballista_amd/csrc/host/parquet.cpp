// parquet.cpp — ParquetExec: the scan leaf the serde builds at rust/core/src/serde/physical_plan/from_proto.rs:111-121
// (`ParquetExec::try_from_files(filenames, projection, None, batch_size, num_partitions)`), what `--format parquet` of the
// reference's TPC-H benchmark reads (rust/benchmarks/tpch/src/main.rs:147-150; its `convert` writes Snappy by default, :84-86).
//
// Split of the work.  HOST (parquet_host.cpp, no device call: it also runs under the CPU sanitizer harness): the footer (Thrift
// compact protocol), page headers, Snappy, definition levels, the run headers of the RLE / bit-packed hybrid, length-prefixed
// strings — everything that is a sequential byte walk; the column chunks of a row group are walked on a pool of host threads,
// one row group ahead of the device.  DEVICE (this file): every per-value
// step: plain values are one copy, dictionary indices are expanded from the run table by one kernel (one thread per value:
// binary search of its run, bit extraction), NULLs are re-inserted by rank (prefix popcounts of the validity words), dictionary
// values — strings included — are gathered with the take kernels FilterExec uses.  One output batch per row group.
//
// Supported: flat schemas; INT32 (Int32 / Date32), INT64, DOUBLE, BOOLEAN, BYTE_ARRAY (Utf8); required and optional fields;
// PLAIN, PLAIN_DICTIONARY / RLE_DICTIONARY; data pages V1 and V2; UNCOMPRESSED and SNAPPY.  Anything else is BHIP_ENOTIMPL at
// plan time (types, nesting) or at the page that needs it (codec, encoding): the caller keeps its CPU ParquetExec.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <future>
#include <map>
#include <set>
#include <thread>

#include "../util_kernels.h"
#include "parquet_host.hpp"
#include "plan.hpp"

namespace bhip {

using namespace pq;

namespace {

// BHIP_PARQUET_TRACE=1: the calling thread's time inside upload_chunk, by kind of work (ms)
struct UploadTrace { double copies = 0, dict = 0, concat = 0, alloc = 0; };
static thread_local UploadTrace g_trace;
static inline double trace_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- pinned host blocks for what the walk produces (parquet_host.hpp: HostVec) ---------------------------------------------------
// Power-of-two size classes from 64 KiB up, kept for the life of the process (bounded: BHIP_PINNED_POOL_MB, default 16 GiB);
// smaller requests are plain malloc.  A block is faulted in and registered with the device once; the async copies from it do
// not go through the runtime's staging buffers.
class PinnedPool {
public:
    static PinnedPool& get() { static PinnedPool p; return p; }
    void* alloc(size_t bytes) {
        if (bytes < MIN) return malloc(bytes ? bytes : 1);
        const size_t cls = class_of(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.find(cls);
            if (it != free_.end() && !it->second.empty()) {
                void* p = it->second.back();
                it->second.pop_back();
                pooled_ -= cls;
                return p;
            }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, cls, hipHostMallocPortable) == hipSuccess && p) {
            std::lock_guard<std::mutex> g(mu_);
            pinned_.insert(p);
            return p;
        }
        (void)hipGetLastError();
        return malloc(bytes);                                       // no pinned memory left: pageable (slower copies, still correct)
    }
    void release(void* p, size_t bytes) {
        if (!p) return;
        if (bytes < MIN) { free(p); return; }
        const size_t cls = class_of(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!pinned_.count(p)) { free(p); return; }
            if (pooled_ + cls <= cap_) { free_[cls].push_back(p); pooled_ += cls; return; }
            pinned_.erase(p);
        }
        hipHostFree(p);
    }
private:
    static constexpr size_t MIN = 1u << 16;
    static size_t class_of(size_t bytes) { size_t c = MIN; while (c < bytes) c <<= 1; return c; }
    PinnedPool() {
        const char* v = getenv("BHIP_PINNED_POOL_MB");
        cap_ = (size_t)(v ? atoll(v) : 16384) << 20;
    }
    std::mutex mu_;
    std::map<size_t, std::vector<void*>> free_;
    std::set<void*> pinned_;
    size_t pooled_ = 0, cap_;
};
void* pinned_alloc(size_t bytes) { return PinnedPool::get().alloc(bytes); }
void pinned_free(void* p, size_t bytes) { PinnedPool::get().release(p, bytes); }
void install_pinned_allocator() {
    static const bool off = [] { const char* v = getenv("BHIP_PARQUET_PAGEABLE"); return v && atoi(v) != 0; }();       // A/B: plain malloc
    static std::once_flag once;
    if (!off) std::call_once(once, [] { pq::set_host_allocator(pinned_alloc, pinned_free); });
}

BufferPtr upload(const Exec& ex, const void* host, size_t bytes) {
    const double t0 = trace_now();
    BufferPtr b = make_buffer(ex, bytes + 16);
    const double t1 = trace_now();
    if (bytes) HIP_CHECK(hipMemcpyAsync(b->ptr(), host, bytes, hipMemcpyHostToDevice, ex.stream));
    g_trace.alloc += t1 - t0;
    g_trace.copies += trace_now() - t1;
    return b;
}

// the device half: one parsed column chunk -> one device column.  `hc` must stay alive until the stream has caught up
// (the caller waits once per row group).
Column upload_chunk(const Exec& ex, const PqColumn& pc, const HostChunk& hc, std::vector<BufferPtr>& keep,
                    std::vector<std::shared_ptr<HostVec<PqRun>>>& held_runs) {
    const bool optional = pc.repetition == 1;
    const size_t width = (pc.phys == PQ_INT32 || pc.phys == PQ_FLOAT) ? 4 : (pc.phys == PQ_INT64 || pc.phys == PQ_DOUBLE) ? 8 : 0;
    const LaunchCfg cfg = ex.cfg();
    Column dict;
    if (hc.dict.present) {
        dict.dtype = pc.dtype;
        dict.length = hc.dict.n;
        if (pc.phys == PQ_BYTE_ARRAY) {
            dict.offsets = upload(ex, hc.dict.offsets.data(), hc.dict.offsets.size() * 4);
            dict.data = upload(ex, hc.dict.bytes.data(), hc.dict.bytes.size());
            dict.data_bytes = (int64_t)hc.dict.bytes.size();
        } else {
            dict.data = upload(ex, hc.dict.bytes.data(), hc.dict.bytes.size());
        }
    }
    // ---- the whole chunk at once (the common case: no NULLs, every page of one kind) --------------------------------------------------
    // One page at a time, a lineitem row group costs ~100 pages x (2-4 copy calls + 2-3 launches + a host wait for every string page)
    // and a concatenation pass per column: the calling thread spent 0.9 s issuing for 0.4 s of kernels (profiles/r03_parquet_scan.txt).
    // Here a chunk is one set of buffers: page bytes copied behind each other, ONE run table (out_start and byte offsets re-based,
    // the bit width travels with each run), ONE expansion, ONE gather — one host wait per string column chunk — and PLAIN pages are
    // copied straight to their rows of the column.
    if (hc.staged == pq::STAGED_FIXED) {                      // the host walk left ONE buffer: one copy
        Column c;
        c.dtype = pc.dtype;
        c.length = hc.rows;
        const double t0 = trace_now();
        c.data = make_buffer(ex, hc.staged_bytes.size() + 16);
        if (!hc.staged_bytes.empty()) HIP_CHECK(hipMemcpyAsync(c.data->ptr(), hc.staged_bytes.data(), hc.staged_bytes.size(), hipMemcpyHostToDevice, ex.stream));
        g_trace.copies += trace_now() - t0;
        return c;
    }
    if (hc.staged == pq::STAGED_DICT) {                       // ... one buffer of index bytes and one run table: two copies, one expansion, one gather
        const int64_t rows_total = hc.rows;
        BufferPtr dbytes = upload(ex, hc.staged_bytes.data(), hc.staged_bytes.size());
        BufferPtr druns = upload(ex, hc.staged_runs.data(), hc.staged_runs.size() * sizeof(PqRun));
        BufferPtr dense = make_buffer(ex, (size_t)rows_total * 4 + 16);
        keep.push_back(dbytes); keep.push_back(druns); keep.push_back(dense);
        const double t1 = trace_now();
        TIMED_LAUNCH_N(ex, "pq_expand_runs", rows_total, launch_pq_expand_runs(cfg, druns->as<PqRun>(), (uint32_t)hc.staged_runs.size(), dbytes->as<uint8_t>(), 0,
                                                                               (uint32_t)rows_total, (uint32_t)dict.length, dense->as<uint32_t>()));
        Column g = take_column(ex, dict, dense->as<uint32_t>(), rows_total);
        g.dtype = pc.dtype;
        g.validity = nullptr;
        g.length = rows_total;
        g_trace.dict += trace_now() - t1;
        return g;
    }
    bool any_nulls = false, all_dict = !hc.pages.empty(), all_fixed = !hc.pages.empty();
    int64_t rows_total = 0;
    size_t bytes_total = 0, runs_total = 0;
    for (const HostPage& pg : hc.pages) {
        any_nulls = any_nulls || pg.has_nulls;
        all_dict = all_dict && pg.kind == PG_DICT;
        all_fixed = all_fixed && pg.kind == PG_FIXED;
        rows_total += pg.n;
        bytes_total += pg.bytes.size();
        runs_total += pg.runs.size();
    }
    static const bool per_page = [] { const char* v = getenv("BHIP_PARQUET_PER_PAGE"); return v && atoi(v) != 0; }();           // A/B: the page-at-a-time path
    if (!per_page && !any_nulls && hc.pages.size() > 1 && all_fixed && width) {
        Column c;
        c.dtype = pc.dtype;
        c.length = rows_total;
        const double t0 = trace_now();
        c.data = make_buffer(ex, width * (size_t)rows_total + 16);
        int64_t row = 0;
        for (const HostPage& pg : hc.pages) {
            if (pg.n) HIP_CHECK(hipMemcpyAsync(c.data->as<uint8_t>() + width * (size_t)row, pg.bytes.data(), width * (size_t)pg.n, hipMemcpyHostToDevice, ex.stream));
            row += pg.n;
        }
        g_trace.copies += trace_now() - t0;
        return c;
    }
    if (!per_page && !any_nulls && hc.pages.size() > 1 && all_dict && bytes_total < 0xFFFFFFF0u && rows_total < 0xFFFFFFF0ll) {
        const double t0 = trace_now();
        BufferPtr dbytes = make_buffer(ex, bytes_total + 16);
        // the run tables, re-based: `runs_all` is pinned (the copy is asynchronous) and kept by the caller's `held_runs` until the stream caught up
        auto runs_all = std::make_shared<HostVec<PqRun>>();
        runs_all->reserve(runs_total);
        size_t byte_base = 0;
        int64_t row = 0;
        for (const HostPage& pg : hc.pages) {
            if (!pg.bytes.empty()) HIP_CHECK(hipMemcpyAsync(dbytes->as<uint8_t>() + byte_base, pg.bytes.data(), pg.bytes.size(), hipMemcpyHostToDevice, ex.stream));
            for (PqRun r : pg.runs) {
                r.out_start += (uint32_t)row;
                if (r.packed) r.value += (uint32_t)byte_base;
                runs_all->push_back(r);
            }
            byte_base += pg.bytes.size();
            row += pg.n;
        }
        held_runs.push_back(runs_all);
        BufferPtr druns = upload(ex, runs_all->data(), runs_all->size() * sizeof(PqRun));
        BufferPtr dense = make_buffer(ex, (size_t)rows_total * 4 + 16);
        keep.push_back(dbytes); keep.push_back(druns); keep.push_back(dense);
        g_trace.copies += trace_now() - t0;
        const double t1 = trace_now();
        TIMED_LAUNCH_N(ex, "pq_expand_runs", rows_total, launch_pq_expand_runs(cfg, druns->as<PqRun>(), (uint32_t)runs_all->size(), dbytes->as<uint8_t>(), 0,
                                                                               (uint32_t)rows_total, (uint32_t)dict.length, dense->as<uint32_t>()));
        Column g = take_column(ex, dict, dense->as<uint32_t>(), rows_total);
        g.dtype = pc.dtype;
        g.validity = nullptr;
        g.length = rows_total;
        g_trace.dict += trace_now() - t1;
        return g;
    }
    std::vector<Column> pieces;
    for (const HostPage& pg : hc.pages) {
        const int64_t n = pg.n, n_valid = pg.n_valid;
        BufferPtr dvalid, dprefix;
        if (pg.has_nulls) {
            dvalid = upload(ex, pg.validity.data(), pg.validity.size());
            dprefix = upload(ex, pg.prefix.data(), pg.prefix.size() * 4);
            keep.push_back(dprefix);
        }
        Column c;
        c.dtype = pc.dtype;
        c.length = n;
        if (pg.has_nulls) c.validity = dvalid;
        if (pg.kind == PG_DICT) {
            BufferPtr dbytes = upload(ex, pg.bytes.data(), pg.bytes.size()), druns = upload(ex, pg.runs.data(), pg.runs.size() * sizeof(PqRun));
            const double td0 = trace_now();
            BufferPtr dense = make_buffer(ex, (size_t)n_valid * 4 + 16);
            keep.push_back(dbytes); keep.push_back(druns); keep.push_back(dense);
            if (n_valid) TIMED_LAUNCH_N(ex, "pq_expand_runs", n_valid, launch_pq_expand_runs(cfg, druns->as<PqRun>(), (uint32_t)pg.runs.size(), dbytes->as<uint8_t>(), pg.bit_width,
                                                                                            (uint32_t)n_valid, (uint32_t)dict.length, dense->as<uint32_t>()));
            const uint32_t* idx = dense->as<uint32_t>();
            if (pg.has_nulls) {
                BufferPtr full = make_buffer(ex, (size_t)n * 4 + 16);
                keep.push_back(full);
                TIMED_LAUNCH_N(ex, "pq_scatter_valid", n, launch_pq_scatter_valid(cfg, dvalid->as<uint64_t>(), dprefix->as<uint32_t>(), dense->ptr(), 4, n, full->ptr(), 1));
                idx = full->as<uint32_t>();
            }
            Column g = pg.has_nulls ? take_column_nullable(ex, dict, idx, n) : take_column(ex, dict, idx, n);
            g.dtype = pc.dtype;
            if (pg.has_nulls) g.validity = dvalid;          // the definition levels ARE the validity
            else g.validity = nullptr;
            c = g;
            c.length = n;
            g_trace.dict += trace_now() - td0;
        } else if (pg.kind == PG_STRINGS) {
            c.offsets = upload(ex, pg.offsets.data(), pg.offsets.size() * 4);
            c.data = upload(ex, pg.bytes.data(), pg.bytes.size());
            c.data_bytes = (int64_t)pg.bytes.size();
        } else if (pg.kind == PG_BOOL) {
            c.data = upload(ex, pg.bytes.data(), pg.bytes.size());
        } else {
            BufferPtr dense = upload(ex, pg.bytes.data(), width * (size_t)n_valid);
            if (pg.has_nulls) {
                c.data = make_buffer(ex, width * (size_t)n + 16);
                keep.push_back(dense);
                TIMED_LAUNCH_N(ex, "pq_scatter_valid", n, launch_pq_scatter_valid(cfg, dvalid->as<uint64_t>(), dprefix->as<uint32_t>(), dense->ptr(), (int)width, n,
                                                                                 c.data->ptr(), 0));
            } else
                c.data = dense;
        }
        pieces.push_back(c);
    }
    if (pieces.empty()) fail(BHIP_EEXEC, "Parquet: column chunk of '" + pc.name + "' holds no data page");
    if (pieces.size() == 1) return pieces[0];
    // several pages: concatenate them as one-column batches
    auto s = std::make_shared<Schema>();
    s->fields.push_back(Field{pc.name, pc.dtype, optional});
    std::vector<BatchPtr> parts;
    for (auto& c : pieces) {
        auto b = std::make_shared<Batch>();
        b->schema = s;
        b->ctx = ex.ctx;
        b->n_rows = c.length;
        b->cols.push_back(c);
        parts.push_back(b);
    }
    const double tc0 = trace_now();
    Column out = concat_batches(ex, s, parts)->cols[0];
    g_trace.concat += trace_now() - tc0;
    return out;
}

// at most `limit` host walks at once (a row group's chunks + the next row group's, each on its own std::async thread)
class Gate {
public:
    explicit Gate(int limit) : free_(limit) {}
    void enter() { std::unique_lock<std::mutex> g(mu_); cv_.wait(g, [&] { return free_ > 0; }); --free_; }
    void leave() { { std::lock_guard<std::mutex> g(mu_); ++free_; } cv_.notify_one(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    int free_;
};

int decode_threads() {
    static const int n = [] {
        const char* v = getenv("BHIP_PARQUET_THREADS");
        int t = v ? atoi(v) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        return t < 1 ? 1 : t;
    }();
    return n;
}

}  // namespace

class ParquetExec : public ExecutionPlan {
public:
    ParquetExec(ContextPtr ctx, std::vector<std::string> files, std::vector<uint32_t> projection, bool has_projection, int num_partitions) {
        ctx_ = std::move(ctx);
        if (files.empty()) fail(BHIP_EINVAL, "ParquetExec needs at least one file");
        for (auto& f : files) files_.push_back(read_footer(f));
        const PqFile& first = files_[0];
        if (!has_projection) for (uint32_t i = 0; i < first.cols.size(); ++i) projection.push_back(i);
        proj_ = projection;
        auto s = std::make_shared<Schema>();
        for (uint32_t i : proj_) {
            if (i >= first.cols.size()) fail(BHIP_EINVAL, "Parquet projection index " + std::to_string(i) + " is out of range");
            const PqColumn& c = first.cols[i];
            if (!c.dtype) fail(BHIP_ENOTIMPL, "Parquet: column '" + c.name + "' has a type outside the GPU path (physical " + std::to_string(c.phys) + ")");
            s->fields.push_back(Field{c.name, c.out_dtype, c.repetition == 1});
        }
        for (auto& f : files_)
            for (uint32_t i : proj_)
                if (i >= f.cols.size() || f.cols[i].name != first.cols[i].name || f.cols[i].out_dtype != first.cols[i].out_dtype)
                    fail(BHIP_EINVAL, "Parquet: " + f.path + " does not have the schema of " + first.path);
        schema_ = s;
        // files are dealt out in chunks, as ParquetExec::try_from_files splits them over max_concurrency partitions
        const int n = (int)files_.size();
        int parts = num_partitions < 1 ? 1 : num_partitions;
        if (parts > n) parts = n;
        const int chunk = (n + parts - 1) / parts;
        for (int i = 0; i < n; i += chunk) part_files_.push_back({i, std::min(n, i + chunk)});
    }
    const char* name() const override { return "ParquetExec"; }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, (int)part_files_.size(), {}}; }
    std::vector<PlanPtr> children() const override { return {}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override {
        if (!c.empty()) fail(BHIP_EINVAL, "ParquetExec has no children");
        return shared_from_this();
    }
    std::string describe() const override {
        std::string s = "ParquetExec: files=" + std::to_string(files_.size()) + ", projection=[";
        for (size_t i = 0; i < schema_->fields.size(); ++i) s += (i ? ", " : "") + schema_->fields[i].name;
        return s + "], device value decode";
    }
    StreamPtr execute(int partition, const Exec& ex) const override {
        check_partition(*this, partition);
        auto self = std::static_pointer_cast<const ParquetExec>(shared_from_this());
        return StreamPtr(new LazyStream(schema_, [self, partition, ex]() {
            // every (file, row group) of the partition, in order; the host walk of group k + 1 runs while the device takes group k
            struct Unit { const PqFile* F; const PqRowGroup* g; };
            std::vector<Unit> units;
            for (int fi = self->part_files_[partition].first; fi < self->part_files_[partition].second; ++fi)
                for (auto& g : self->files_[fi].groups) {
                    if (g.num_rows == 0) continue;
                    if (g.num_rows > 0xFFFFFFF0ll) fail(BHIP_ENOTIMPL, "Parquet: row group of more than 2^32 rows");
                    units.push_back(Unit{&self->files_[fi], &g});
                }
            install_pinned_allocator();
            auto gate = std::make_shared<Gate>(decode_threads());
            using Parsed = std::vector<std::future<HostChunk>>;
            auto start = [&](const Unit& u) {
                Parsed fs;
                for (uint32_t ci : self->proj_) {
                    if (ci >= u.g->cols.size()) fail(BHIP_EEXEC, "Parquet: row group without column " + std::to_string(ci));
                    const PqFile* F = u.F;
                    const PqColumn* pc = &u.F->cols[ci];
                    const PqChunk* ch = &u.g->cols[ci];
                    const int64_t rows = u.g->num_rows;
                    fs.push_back(std::async(std::launch::async, [gate, F, pc, ch, rows]() {
                        gate->enter();
                        struct Leave { Gate& g; ~Leave() { g.leave(); } } leave{*gate};
                        const std::vector<uint8_t> raw = read_chunk_bytes(*F, *ch, pc->name);
                        return parse_chunk(raw.data(), raw.size(), *pc, *ch, rows);
                    }));
                }
                return fs;
            };
            std::vector<BatchPtr> out;
            // BHIP_PARQUET_TRACE=1: where the calling thread's time goes, per partition (stderr)
            static const bool trace = [] { const char* v = getenv("BHIP_PARQUET_TRACE"); return v && atoi(v) != 0; }();
            double t_walk = 0, t_issue = 0, t_dev = 0;
            auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            // the host walks run AHEAD row groups in front of the device half (default 4: a row group has about as many chunks as
            // columns and its string chunks take several times as long as the others — with the device half down to ~60 ms of issue
            // work per GiB the calling thread otherwise waits for the slowest walker of every group: 215 / 265 / 271 M rows/s at
            // 2 / 3 / 4; every group in flight holds its decoded chunks in pinned memory, ~0.3 GB each for lineitem)
            static const size_t ahead = [] { const char* v = getenv("BHIP_PARQUET_AHEAD"); const int a = v ? atoi(v) : 4; return (size_t)(a < 1 ? 1 : a > 8 ? 8 : a); }();
            std::deque<Parsed> inflight;
            size_t started = 0;
            for (; started < units.size() && started < ahead; ++started) inflight.push_back(start(units[started]));
            for (size_t k = 0; k < units.size(); ++k) {
                Parsed cur = std::move(inflight.front());
                inflight.pop_front();
                if (started < units.size()) inflight.push_back(start(units[started++]));
                const Unit& u = units[k];
                auto b = std::make_shared<Batch>();
                b->schema = self->schema_;
                b->ctx = ex.ctx;
                b->n_rows = u.g->num_rows;
                std::vector<HostChunk> held;                 // host bytes the queued copies read from
                std::vector<std::shared_ptr<HostVec<PqRun>>> held_runs;
                std::vector<BufferPtr> keep;                 // device scratch that must outlive the kernels of this row group
                held.reserve(cur.size());
                try {
                    for (size_t j = 0; j < cur.size(); ++j) {
                        const PqColumn& pc = u.F->cols[self->proj_[j]];
                        double t0 = now();
                        held.push_back(cur[j].get());
                        double t1 = now();
                        t_walk += t1 - t0;
                        Column col = upload_chunk(ex, pc, held.back(), keep, held_runs);
                        t_issue += now() - t1;
                        if (pc.out_dtype != pc.dtype) {                                   // Int32 values of an INT_8 .. UINT_16 column
                            Column narrow = col;
                            narrow.dtype = pc.out_dtype;
                            narrow.data = make_buffer(ex, (size_t)col.length * dtype_width(narrow.dtype) + 8);
                            TIMED_LAUNCH_N(ex, "narrow_i32", col.length, launch_narrow_i32(ex.cfg(), col.data->as<int32_t>(), col.length, dtype_width(narrow.dtype), narrow.data->ptr()));
                            col = narrow;
                        }
                        b->cols.push_back(col);
                    }
                } catch (...) {
                    // the walks still running hold pointers into this partition's footers: let them finish before unwinding
                    for (auto& f : cur) if (f.valid()) f.wait();
                    for (auto& fs : inflight) for (auto& f : fs) if (f.valid()) f.wait();
                    hipStreamSynchronize(ex.stream);
                    throw;
                }
                {
                    const double t0 = now();
                    stream_wait(ex);                         // the copies have left `held`
                    t_dev += now() - t0;
                }
                out.push_back(b);
            }
            if (trace) {
                fprintf(stderr, "[bhip-parquet] partition %d: %zu row groups; calling thread waited %.1f ms for host walks, %.1f ms issuing copies and launches "
                                "(buffer allocation %.1f, copy calls %.1f, dictionary expand + gather %.1f, page concatenation %.1f), %.1f ms for the device\n",
                        partition, units.size(), t_walk, t_issue, g_trace.alloc, g_trace.copies, g_trace.dict, g_trace.concat, t_dev);
                g_trace = UploadTrace{};
            }
            return out;
        }));
    }
private:
    std::vector<PqFile> files_;
    std::vector<uint32_t> proj_;
    SchemaPtr schema_;
    std::vector<std::pair<int, int>> part_files_;
};

PlanPtr make_parquet_exec(const ContextPtr& ctx, const std::vector<std::string>& files, const std::vector<uint32_t>& projection, bool has_projection,
                          int num_partitions) {
    return std::make_shared<ParquetExec>(ctx, files, projection, has_projection, num_partitions);
}

}  // namespace bhip
