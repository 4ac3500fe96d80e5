// core.cpp — context (allocator, streams), buffers, batches.
#include "core.hpp"

#include <immintrin.h>

#include <chrono>
#include <cstdlib>

#include "../util_kernels.h"

// Kernel arguments in device memory (a ROCm runtime option; unset, the argument block of every launch lives in host memory the GPU
// reads across the bus: ~2-3 us more per launch, and the expression VM's kernels — which read their program out of a by-value
// argument, scalar load by scalar load — wait on those reads).  Measured on the bench: Q3 4.69 -> 4.52 ms, Q5 3.05 -> 2.98 ms,
// Q1 5.07 -> 4.99 ms.  Set when the library is loaded, before the process's first HIP call, unless the user has decided otherwise.
__attribute__((constructor)) static void bhip_runtime_defaults() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

namespace bhip {

static thread_local std::string t_last_error;
void set_last_error(const std::string& m) { t_last_error = m; }
const char* get_last_error() { return t_last_error.c_str(); }

const char* dtype_name(int dt) {
    switch (dt) {
        case DT_INT32: return "Int32";
        case DT_INT64: return "Int64";
        case DT_UINT8: return "UInt8";
        case DT_UINT64: return "UInt64";
        case DT_FLOAT64: return "Float64";
        case DT_DATE32: return "Date32";
        case DT_BOOLEAN: return "Boolean";
        case DT_UTF8: return "Utf8";
        case DT_INT8: return "Int8";
        case DT_INT16: return "Int16";
        case DT_UINT16: return "UInt16";
        case DT_UINT32: return "UInt32";
        case DT_FLOAT32: return "Float32";
        case DT_DATE64: return "Date64";
        case DT_TIMESTAMP_S: return "Timestamp(Second)";
        case DT_TIMESTAMP_MS: return "Timestamp(Millisecond)";
        case DT_TIMESTAMP_US: return "Timestamp(Microsecond)";
        case DT_TIMESTAMP_NS: return "Timestamp(Nanosecond)";
        default: return "?";
    }
}

int dtype_width(int dt) {
    switch (dt) {
        case DT_INT32:
        case DT_DATE32:
        case DT_UINT32:
        case DT_FLOAT32: return 4;
        case DT_INT64:
        case DT_UINT64:
        case DT_FLOAT64:
        case DT_DATE64:
        case DT_TIMESTAMP_S:
        case DT_TIMESTAMP_MS:
        case DT_TIMESTAMP_US:
        case DT_TIMESTAMP_NS: return 8;
        case DT_UINT8:
        case DT_INT8: return 1;
        case DT_INT16:
        case DT_UINT16: return 2;
        default: return 0;
    }
}

// ---- context -----------------------------------------------------------------------------------
Context::Context(int device) : device_(device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        fail(BHIP_EHIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= count) fail(BHIP_EINVAL, "device index out of range");
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const char* t = getenv("BHIP_KERNEL_TIMING");
    timing_ = t ? atoi(t) : 0;
    const char* sp = getenv("BHIP_SPIN_WAIT");
    spin_wait_ = !(sp && atoi(sp) == 0);
    if (spin_wait_) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(HostSlot) * N_SLOTS, hipHostMallocDefault) == hipSuccess) {
            memset(p, 0, sizeof(HostSlot) * N_SLOTS);
            slots_ = static_cast<HostSlot*>(p);
        } else {
            spin_wait_ = false;
        }
    }
}

void trace_point(const char* what) {
    static const bool on = [] { const char* v = getenv("BHIP_TRACE_HOST"); return v && atoi(v) != 0; }();
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[bhip-host] %10.1f %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
}

void Context::wait_stream(hipStream_t stream, const void* dev_src, void* host_dst, size_t bytes) {
    trace_point("wait: enter");
    struct Leave { ~Leave() { trace_point("wait: leave"); } } leave;
    if (!spin_wait_ || bytes > 56 || (bytes & 3)) {
        if (bytes) HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    const uint64_t seq = slot_seq_.fetch_add(1) + 1;
    HostSlot* slot = &slots_[seq % N_SLOTS];
    HIP_CHECK(launch_publish(stream, dev_src, (int)(bytes / 4), (void*)slot, seq));
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(&slot->w[0], __ATOMIC_ACQUIRE) == seq) break;
        _mm_pause();
        if ((spins & 0x3FF) == 0x3FF && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            HIP_CHECK(hipStreamSynchronize(stream));                      // long kernels: sleep instead of burning the core
            if (__atomic_load_n(&slot->w[0], __ATOMIC_ACQUIRE) != seq) fail(BHIP_EHIP, "stream finished without publishing its result");
            break;
        }
    }
    if (bytes) memcpy(host_dst, (const void*)&slot->w[1], bytes);
}

Context::~Context() {
    hipSetDevice(device_);
    hipDeviceSynchronize();
    for (auto& kv : free_blocks_) {
        hipFree(kv.second.ptr);
        if (kv.second.ready) hipEventDestroy(kv.second.ready);
    }
    for (auto& kv : live_) hipFree(kv.second.ptr);
    for (auto s : stream_pool_) hipStreamDestroy(s);
    if (slots_) hipHostFree(slots_);
    for (auto& p : pending_timed_) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : event_pool_) hipEventDestroy(e);
}

// Size classes: eight per octave (1, 1.125, ... 1.875 x 2^k; at most 12.5 % over the request), 512 B the smallest.  A request
// takes a cached block of ITS class only.  (The first allocator took the best fit within 25 %: a request then sometimes took the
// block a slightly larger request of the same step was going to need, that one missed and went to hipMalloc — 1-2 ms for tens
// of MB — and a query step could settle into a cycle of four to six hipMallocs: Q3 9.2 ms per step instead of 4.3 ms in about
// half of the processes.  With classes a step's requests find the blocks the same requests released a step earlier.)
static size_t round_size(size_t bytes) {
    if (bytes <= 512) return 512;
    int k = 63 - __builtin_clzll((unsigned long long)(bytes - 1));         // 2^k < bytes <= 2^(k+1)
    const size_t step = (size_t)1 << (k - 3);                               // an eighth of 2^k: eight classes in (2^k, 2^(k+1)]
    return (bytes + step - 1) & ~(step - 1);
}

void* Context::alloc(size_t bytes, hipStream_t stream) {
    const size_t want = round_size(bytes);
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_blocks_.find(want);                         // a block of the request's size class
        if (it != free_blocks_.end()) {
            Block b = it->second;
            free_blocks_.erase(it);
            cached_ -= b.bytes;
            if (b.last_stream != stream) {
                // the block's last user was another task's stream: everything queued there so far comes first.  The event is
                // recorded HERE, not at release — a release then costs no stream operation (a task frees hundreds of scratch
                // buffers, and each record is a packet the command processor serialises), and reuse on the same stream, the
                // common case, is ordered by the stream itself
                if (!b.ready) HIP_CHECK(hipEventCreateWithFlags(&b.ready, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(b.ready, b.last_stream));
                HIP_CHECK(hipStreamWaitEvent(stream, b.ready, 0));
            }
            b.last_stream = stream;
            live_[b.ptr] = b;
            in_use_ += b.bytes;
            if (in_use_ > peak_) peak_ = in_use_;
            return b.ptr;
        }
    }
    set_device();
    void* p = nullptr;
    {
        char msg[160];
        size_t nearest = 0;
        { std::lock_guard<std::mutex> g(mu_); auto it = free_blocks_.lower_bound(want); if (it != free_blocks_.end()) nearest = it->first; }
        snprintf(msg, sizeof(msg), "alloc: no cached block fits %zu bytes (smallest cached block that holds it: %zu; %zu cached in %zu blocks), hipMalloc",
                 want, nearest, cached_, free_blocks_.size());
        trace_point(msg);
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        trace_point("alloc: device memory exhausted, cached blocks released");
        trim();   // give cached blocks back and retry once
        e = hipMalloc(&p, want);
    }
    trace_point("alloc: hipMalloc done");
    if (e != hipSuccess) fail(BHIP_EOOM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    Block b;
    b.ptr = p;
    b.bytes = want;
    b.last_stream = stream;
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = b;
    in_use_ += want;
    if (in_use_ > peak_) peak_ = in_use_;
    return p;
}

void Context::free(void* ptr, hipStream_t stream) {
    if (!ptr) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = live_.find(ptr);
    if (it == live_.end()) return;
    Block b = it->second;
    live_.erase(it);
    in_use_ -= b.bytes;
    // work that used the block was enqueued on `stream` (the owning task's stream)
    b.last_stream = stream;
    cached_ += b.bytes;
    free_blocks_.emplace(b.bytes, b);
}

void Context::trim() {
    std::lock_guard<std::mutex> g(mu_);
    hipSetDevice(device_);
    hipDeviceSynchronize();
    for (auto& kv : free_blocks_) {
        hipFree(kv.second.ptr);
        if (kv.second.ready) hipEventDestroy(kv.second.ready);
    }
    free_blocks_.clear();
    cached_ = 0;
}

void Context::memory(uint64_t* in_use, uint64_t* peak) {
    std::lock_guard<std::mutex> g(mu_);
    if (in_use) *in_use = in_use_;
    if (peak) *peak = peak_;
}

hipStream_t Context::acquire_stream() {
    {
        std::lock_guard<std::mutex> g(mu_);
        if (!stream_pool_.empty()) {
            hipStream_t s = stream_pool_.back();
            stream_pool_.pop_back();
            return s;
        }
    }
    set_device();
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void Context::release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> g(mu_);
    stream_pool_.push_back(s);
}

void Context::add_kernel_time(double ms, uint64_t launches, const char* kernel) {
    std::lock_guard<std::mutex> g(mu_);
    k_ms_ += ms;
    k_launches_ += launches;
    if (kernel && launches) {
        k_name_ = kernel;
        auto& st = k_stats_[kernel];
        st.ms += ms;
        st.launches += launches;
    }
}

hipEvent_t Context::timing_event() {
    {
        std::lock_guard<std::mutex> g(mu_);
        if (!event_pool_.empty()) {
            hipEvent_t e = event_pool_.back();
            event_pool_.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

void Context::push_timed(hipEvent_t a, hipEvent_t b, const char* name, uint64_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    pending_timed_.push_back({a, b, name, bytes});
}

std::map<std::string, Context::KernelStat> Context::kernel_stats(bool reset) {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<PendingTimed> keep;
    for (auto& p : pending_timed_) {
        float t = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            auto& st = k_stats_[p.name];
            st.ms += t;
            st.launches += 1;
            st.bytes += p.bytes;
        }
        event_pool_.push_back(p.a);
        event_pool_.push_back(p.b);
    }
    pending_timed_.clear();
    auto out = k_stats_;
    if (reset) k_stats_.clear();
    return out;
}

std::string Context::kernel_name() {
    std::lock_guard<std::mutex> g(mu_);
    return k_name_;
}

void Context::kernel_time(bool reset, double* ms, uint64_t* launches) {
    std::lock_guard<std::mutex> g(mu_);
    if (ms) *ms = k_ms_;
    if (launches) *launches = k_launches_;
    if (reset) { k_ms_ = 0; k_launches_ = 0; }
}

// ---- buffers ------------------------------------------------------------------------------------
Buffer::Buffer(ContextPtr ctx, size_t bytes, hipStream_t stream)
    : ctx_(std::move(ctx)), ptr_(nullptr), bytes_(bytes), owned_(true), stream_(stream) {
    ptr_ = ctx_->alloc((bytes ? bytes : 8) + BUFFER_SLACK, stream);
}
Buffer::Buffer(ContextPtr ctx, void* borrowed, size_t bytes)
    : ctx_(std::move(ctx)), ptr_(borrowed), bytes_(bytes), owned_(false), stream_(nullptr) {}
Buffer::Buffer(ContextPtr ctx, std::shared_ptr<Buffer> parent, void* ptr, size_t bytes)
    : ctx_(std::move(ctx)), ptr_(ptr), bytes_(bytes), owned_(false), stream_(nullptr), parent_(std::move(parent)) {}
Buffer::~Buffer() {
    if (owned_ && ptr_) ctx_->free(ptr_, stream_);
}

BufferPtr make_buffer(const Exec& ex, size_t bytes) { return std::make_shared<Buffer>(ex.ctx, bytes, ex.stream); }

int64_t Column::memory_size() const {
    int64_t b = 0;
    if (dtype == DT_UTF8) b += data_bytes + (length + 1) * 4;
    else if (dtype == DT_BOOLEAN) b += (length + 7) / 8;
    else b += length * dtype_width(dtype);
    if (validity) b += (length + 7) / 8;
    return b;
}

int64_t Batch::memory_size() const {
    int64_t b = 0;
    for (const auto& c : cols) b += c.memory_size();
    return b;
}

// ---- host <-> device -------------------------------------------------------------------------------
BatchPtr batch_from_host(const ContextPtr& ctx, int n_cols, const bhip_column_desc* cols, int64_t n_rows, bool device_ptrs) {
    if (n_rows < 0 || n_rows > 0xFFFFFFF0ll) fail(BHIP_EINVAL, "batch row count out of range (max 2^32-16 rows per batch)");
    ctx->set_device();
    auto schema = std::make_shared<Schema>();
    auto batch = std::make_shared<Batch>();
    batch->ctx = ctx;
    batch->n_rows = n_rows;
    hipStream_t st = nullptr;   // null stream: synchronous wrt the caller
    Exec ex{ctx, st};
    // small host batches (stage inputs of a Final aggregate, dimension tables): all buffers travel in ONE block
    // and ONE copy; the columns are slices of it (256-byte aligned, zero padded to whole words + slack)
    std::vector<uint8_t> staging;
    std::vector<size_t> slice_off;
    BufferPtr block;
    // a validity bitmap with every bit set says "no NULLs" (Arrow producers often attach one): the column is
    // imported without it, so the NULL-free kernels serve it; the schema keeps its nullability
    std::vector<const uint8_t*> validity((size_t)n_cols, nullptr);
    for (int i = 0; i < n_cols; ++i) {
        validity[i] = cols[i].validity;
        if (device_ptrs || !cols[i].validity) continue;
        const uint8_t* v = cols[i].validity;
        const int64_t whole = n_rows / 8;
        bool all = true;
        for (int64_t k = 0; k < whole && all; ++k) all = v[k] == 0xFF;
        if (all && (n_rows & 7)) all = (v[whole] & ((1u << (n_rows & 7)) - 1u)) == ((1u << (n_rows & 7)) - 1u);
        if (all) validity[i] = nullptr;
    }
    if (!device_ptrs) {
        size_t total = 0;
        bool small = true;
        auto plan = [&](const void* p, size_t bytes) {
            if (!p) return;
            slice_off.push_back(total);
            total += (bytes + 8 + BUFFER_SLACK + 255) & ~(size_t)255;
            if (bytes > (1u << 20)) small = false;
        };
        for (int i = 0; i < n_cols && small; ++i) {
            const bhip_column_desc& d = cols[i];
            if (d.dtype < DT_INT32 || d.dtype > DT_LAST) { small = false; break; }
            const size_t db = d.dtype == DT_UTF8 ? (size_t)d.data_bytes : d.dtype == DT_BOOLEAN ? bitmap_bytes(n_rows) : (size_t)n_rows * dtype_width(d.dtype);
            plan(d.data ? d.data : (const void*)"", db);
            if (d.offsets) plan(d.offsets, (size_t)(n_rows + 1) * 4);
            if (validity[i]) plan(validity[i], bitmap_bytes(n_rows));
        }
        if (small && total > 0 && total <= (4u << 20)) {
            staging.assign(total, 0);
            block = make_buffer(ex, total);
        } else {
            slice_off.clear();
        }
    }
    size_t next_slice = 0;
    auto slice = [&](const void* host, size_t copy_bytes, size_t logical_bytes) -> BufferPtr {
        const size_t off = slice_off[next_slice++];
        if (copy_bytes && host) memcpy(staging.data() + off, host, copy_bytes);
        return std::make_shared<Buffer>(ctx, block, static_cast<uint8_t*>(block->ptr()) + off, logical_bytes);
    };
    for (int i = 0; i < n_cols; ++i) {
        const bhip_column_desc& d = cols[i];
        if (!d.name) fail(BHIP_EINVAL, "column without a name");
        if (d.dtype < DT_INT32 || d.dtype > DT_LAST) fail(BHIP_ENOTIMPL, std::string("unsupported data type for column ") + d.name);
        schema->fields.push_back(Field{d.name, d.dtype, d.nullable != 0 || d.validity != nullptr});
        Column c;
        c.dtype = d.dtype;
        c.length = n_rows;
        size_t data_bytes;
        if (d.dtype == DT_UTF8) {
            if (!d.offsets) fail(BHIP_EINVAL, std::string("Utf8 column without offsets: ") + d.name);
            data_bytes = (size_t)d.data_bytes;
            c.data_bytes = d.data_bytes;
        } else if (d.dtype == DT_BOOLEAN) {
            data_bytes = (size_t)((n_rows + 7) / 8);
        } else {
            data_bytes = (size_t)n_rows * dtype_width(d.dtype);
        }
        if (!d.data && data_bytes > 0) fail(BHIP_EINVAL, std::string("column without data: ") + d.name);
        if (device_ptrs) {
            c.data = std::make_shared<Buffer>(ctx, const_cast<void*>(d.data), data_bytes);
            if (d.offsets) c.offsets = std::make_shared<Buffer>(ctx, const_cast<int32_t*>(d.offsets), (size_t)(n_rows + 1) * 4);
            if (d.validity) c.validity = std::make_shared<Buffer>(ctx, const_cast<uint8_t*>(d.validity), bitmap_bytes(n_rows));
        } else if (block) {
            const size_t padded = d.dtype == DT_BOOLEAN ? bitmap_bytes(n_rows) : data_bytes;
            c.data = slice(d.data, data_bytes, padded + 8);
            if (d.offsets) c.offsets = slice(d.offsets, (size_t)(n_rows + 1) * 4, (size_t)(n_rows + 1) * 4);
            if (validity[i]) c.validity = slice(validity[i], (size_t)((n_rows + 7) / 8), bitmap_bytes(n_rows) + 8);
        } else {
            // device copies are padded to whole 64-bit words (bitmaps are read as u64)
            const size_t padded = d.dtype == DT_BOOLEAN ? bitmap_bytes(n_rows) : data_bytes;
            c.data = make_buffer(ex, padded + 8);
            if (d.dtype == DT_BOOLEAN) HIP_CHECK(hipMemset(c.data->ptr(), 0, padded + 8));
            if (data_bytes) HIP_CHECK(hipMemcpy(c.data->ptr(), d.data, data_bytes, hipMemcpyHostToDevice));
            if (d.offsets) {
                c.offsets = make_buffer(ex, (size_t)(n_rows + 1) * 4);
                HIP_CHECK(hipMemcpy(c.offsets->ptr(), d.offsets, (size_t)(n_rows + 1) * 4, hipMemcpyHostToDevice));
            }
            if (validity[i]) {
                c.validity = make_buffer(ex, bitmap_bytes(n_rows) + 8);
                HIP_CHECK(hipMemset(c.validity->ptr(), 0, bitmap_bytes(n_rows) + 8));
                HIP_CHECK(hipMemcpy(c.validity->ptr(), validity[i], (size_t)((n_rows + 7) / 8), hipMemcpyHostToDevice));
            }
        }
        batch->cols.push_back(std::move(c));
    }
    if (block) HIP_CHECK(hipMemcpy(block->ptr(), staging.data(), staging.size(), hipMemcpyHostToDevice));
    batch->schema = schema;
    return batch;
}

void column_to_host(const Batch& b, int i, void* data, int32_t* offsets, uint8_t* validity) {
    if (i < 0 || i >= (int)b.cols.size()) fail(BHIP_EINVAL, "column index out of range");
    b.ctx->set_device();
    HIP_CHECK(hipDeviceSynchronize());
    const Column& c = b.cols[i];
    if (data) {
        size_t bytes;
        if (c.dtype == DT_UTF8) bytes = (size_t)c.data_bytes;
        else if (c.dtype == DT_BOOLEAN) bytes = (size_t)((c.length + 7) / 8);
        else bytes = (size_t)c.length * dtype_width(c.dtype);
        if (bytes) HIP_CHECK(hipMemcpy(data, c.data->ptr(), bytes, hipMemcpyDeviceToHost));
    }
    if (offsets && c.offsets) HIP_CHECK(hipMemcpy(offsets, c.offsets->ptr(), (size_t)(c.length + 1) * 4, hipMemcpyDeviceToHost));
    if (validity) {
        const size_t bytes = (size_t)((c.length + 7) / 8);
        if (c.validity) { if (bytes) HIP_CHECK(hipMemcpy(validity, c.validity->ptr(), bytes, hipMemcpyDeviceToHost)); }
        else memset(validity, 0xFF, bytes);
    }
}

}  // namespace bhip
