// core.hpp — host-side runtime objects: errors, the per-GPU context (caching allocator, stream
// pool, pinned staging), device buffers, columns, schemas, record batches.
//
// Reference conventions being mirrored: RecordBatch columns are Arc'd immutable buffers
// (rust/core/src/memory_stream.rs:29-92); errors are Result<_, DataFusionError> values, never
// panics (rust/core/src/execution_plans/unresolved_shuffle.rs:83-90; rust/core/src/error.rs).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ballista_hip.h"
#include "../kernels.h"

namespace bhip {

// ---- errors ---------------------------------------------------------------------------------
struct Error : std::runtime_error {
    bhip_status code;
    Error(bhip_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void fail(bhip_status c, const std::string& m) { throw Error(c, m); }
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        if (e == hipErrorOutOfMemory) fail(BHIP_EOOM, std::string(what) + ": " + hipGetErrorString(e));
        fail(BHIP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    }
}
#define HIP_CHECK(expr) ::bhip::hip_check((expr), #expr)

void set_last_error(const std::string& m);
const char* get_last_error();

const char* dtype_name(int dt);
int dtype_width(int dt);   // bytes of a fixed-width value; 0 for Utf8 / Boolean

// ---- context ----------------------------------------------------------------------------------
class Context;

struct Block {               // one allocation of the caching allocator
    void* ptr = nullptr;
    size_t bytes = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t ready = nullptr;   // recorded on last_stream only when ANOTHER stream takes the block over (Context::alloc)
};

class Context : public std::enable_shared_from_this<Context> {
public:
    explicit Context(int device);
    ~Context();
    int device() const { return device_; }
    int cus() const { return cus_; }
    void set_device() const { HIP_CHECK(hipSetDevice(device_)); }

    // stream-aware caching allocator: a released block is reused by the same stream at once and
    // by another stream only after the release event
    void* alloc(size_t bytes, hipStream_t stream);
    void free(void* ptr, hipStream_t stream);
    void memory(uint64_t* in_use, uint64_t* peak);
    void trim();

    hipStream_t acquire_stream();
    void release_stream(hipStream_t s);

    // "everything enqueued on `stream` so far has finished", optionally with `bytes` (<= 56, a multiple of 4) of device
    // memory brought along: a one-thread kernel writes them and then a sequence number into a pinned host slot the caller
    // spins on.  A runtime wait (hipStreamSynchronize after a device-to-host copy) costs 30-40 us of wake-up latency on this
    // stack; an operator tree makes dozens of them per query.  Falls back to the runtime wait after ~2 ms of spinning.
    void wait_stream(hipStream_t stream, const void* dev_src = nullptr, void* host_dst = nullptr, size_t bytes = 0);

    // kernel timing hook (bench roofline): accumulated by the aggregate operator
    void add_kernel_time(double ms, uint64_t launches, const char* kernel = nullptr);
    std::string kernel_name();
    void kernel_time(bool reset, double* ms, uint64_t* launches);
    bool timing_enabled() const { return timing_ > 0; }
    int timing_level() const { return timing_; }
    // per-kernel-name totals (BHIP_KERNEL_TIMING=1): event pairs queued by KernelTimer are resolved when read
    struct KernelStat { double ms = 0; uint64_t launches = 0; uint64_t bytes = 0; };    // bytes: algorithmic bytes of the timed launches
    void push_timed(hipEvent_t a, hipEvent_t b, const char* name, uint64_t bytes = 0);
    hipEvent_t timing_event();
    std::map<std::string, KernelStat> kernel_stats(bool reset);

private:
    int device_;
    int cus_ = 256;
    int timing_ = 0;
    std::mutex mu_;
    std::multimap<size_t, Block> free_blocks_;
    std::map<void*, Block> live_;
    uint64_t in_use_ = 0, peak_ = 0, cached_ = 0;
    std::vector<hipStream_t> stream_pool_;
    static constexpr size_t N_SLOTS = 1024;
    struct HostSlot { volatile uint64_t w[8]; };      // one cache line of pinned host memory: w[0] = sequence, w[1..7] = payload
    HostSlot* slots_ = nullptr;
    std::atomic<uint64_t> slot_seq_{0};
    bool spin_wait_ = true;
    double k_ms_ = 0;
    uint64_t k_launches_ = 0;
    std::string k_name_;
    struct PendingTimed { hipEvent_t a, b; const char* name; uint64_t bytes; };
    std::vector<PendingTimed> pending_timed_;
    std::vector<hipEvent_t> event_pool_;
    std::map<std::string, KernelStat> k_stats_;
};
using ContextPtr = std::shared_ptr<Context>;

// execution context of one task: the context + the HIP stream every kernel of the task runs on
struct Exec {
    ContextPtr ctx;
    hipStream_t stream;
    LaunchCfg cfg() const { return LaunchCfg{ctx->cus(), stream}; }
};

// times the launches enqueued on ex.stream during its lifetime under `name` (no-op unless BHIP_KERNEL_TIMING=1);
// nothing waits here: the event pair is read when the statistics are (bhip_ctx_kernel_stats)
struct KernelTimer {
    const Exec& ex;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    // rows: what the launch covers.  BHIP_KERNEL_TIMING=1 times launches over >= 2^18 rows only (an event pair per tiny
    // launch would stretch the small-launch tail it is there to measure); =2 times every launch
    uint64_t bytes = 0;
    KernelTimer(const Exec& e, const char* n, int64_t rows = -1, uint64_t algorithmic_bytes = 0) : ex(e), name(n), bytes(algorithmic_bytes) {
        const int level = ex.ctx->timing_level();
        if (level <= 0 || (level == 1 && rows < (1 << 18))) return;
        a = ex.ctx->timing_event();
        b = ex.ctx->timing_event();
        hipEventRecord(a, ex.stream);
    }
    void stop() {
        if (!a) return;
        hipEventRecord(b, ex.stream);
        ex.ctx->push_timed(a, b, name, bytes);
        a = b = nullptr;
    }
    ~KernelTimer() { stop(); }
};

#define TIMED_LAUNCH(ex, name, call) do { ::bhip::KernelTimer _kt((ex), (name)); HIP_CHECK(call); } while (0)
#define TIMED_LAUNCH_N(ex, name, rows, call) do { ::bhip::KernelTimer _kt((ex), (name), (int64_t)(rows)); HIP_CHECK(call); } while (0)
// ... with the launch's algorithmic bytes, for the roofline line of bench.py
#define TIMED_LAUNCH_B(ex, name, rows, bytes, call) do { ::bhip::KernelTimer _kt((ex), (name), (int64_t)(rows), (uint64_t)(bytes)); HIP_CHECK(call); } while (0)

// ---- device buffers -----------------------------------------------------------------------------
constexpr size_t BUFFER_SLACK = 16;

class Buffer {
public:
    Buffer(ContextPtr ctx, size_t bytes, hipStream_t stream);          // owned
    Buffer(ContextPtr ctx, void* borrowed, size_t bytes);              // borrowed device pointer
    // slice of `parent` (kept alive); the slicer leaves BUFFER_SLACK readable bytes behind every slice
    Buffer(ContextPtr ctx, std::shared_ptr<Buffer> parent, void* ptr, size_t bytes);
    ~Buffer();
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    void* ptr() const { return ptr_; }
    size_t bytes() const { return bytes_; }
    // owned buffers carry BUFFER_SLACK readable bytes past bytes(): kernels may over-read short strings
    bool owned() const { return owned_ || parent_ != nullptr; }
    template <class T> T* as() const { return reinterpret_cast<T*>(ptr_); }
    void set_stream(hipStream_t s) { stream_ = s; }
private:
    ContextPtr ctx_;
    void* ptr_;
    size_t bytes_;
    bool owned_;
    hipStream_t stream_;
    std::shared_ptr<Buffer> parent_;
};
using BufferPtr = std::shared_ptr<Buffer>;
BufferPtr make_buffer(const Exec& ex, size_t bytes);
inline size_t bitmap_bytes(int64_t n_bits) { return (size_t)((n_bits + 63) / 64) * 8; }

// scratch that lives for one operator call
struct Temp {
    const Exec& ex;
    std::vector<BufferPtr> held;
    explicit Temp(const Exec& e) : ex(e) {}
    template <class T> T* get(size_t count) {
        held.push_back(make_buffer(ex, count * sizeof(T) + 16));
        return held.back()->as<T>();
    }
};

// ---- columns / schema / batches -------------------------------------------------------------------
struct Column {
    int dtype = 0;
    int64_t length = 0;
    BufferPtr data;        // values | Utf8 bytes | Boolean bitmap
    BufferPtr offsets;     // Utf8
    BufferPtr validity;    // may be null
    int64_t data_bytes = 0;   // Utf8 value bytes
    // A VIEW (only between a HashJoinExec and the HashJoinExec that asked for it, host/ops_join.cpp): row i is row view_idx[i] of
    // *view_base, NULL where the index is 0xFFFFFFFF; data / offsets / validity are unset.  take_columns composes the indices, so a
    // payload column that passes through several joins is gathered ONCE, for the rows that survive them all.
    std::shared_ptr<const Column> view_base;
    BufferPtr view_idx;
    bool view_may_null = false;
    bool is_view() const { return (bool)view_base; }
    ColumnRef ref() const {
        ColumnRef r;
        r.data = data ? data->ptr() : nullptr;
        r.offsets = offsets ? offsets->as<int32_t>() : nullptr;
        r.validity = validity ? validity->as<uint64_t>() : nullptr;
        r.dtype = dtype;
        r.data_bytes = (int32_t)data_bytes;
        return r;
    }
    int64_t memory_size() const;
};

struct Field {
    std::string name;
    int dtype;
    bool nullable;
    // LargeUtf8 at the boundary (Arrow C data, IPC files, wire schemas): 64-bit offsets there; the device column is an ordinary Utf8
    // column (int32 offsets: < 2 GiB of value bytes per batch, refused beyond).  The flag follows the column through the operators.
    bool large = false;
    // Binary at the boundary (sha224 .. sha512 produce it): the same buffers as Utf8, nothing reads them as text
    bool binary = false;
};

struct Schema {
    std::vector<Field> fields;
    int index_of(const std::string& name) const {
        for (size_t i = 0; i < fields.size(); ++i)
            if (fields[i].name == name) return (int)i;
        return -1;
    }
};
using SchemaPtr = std::shared_ptr<const Schema>;

struct Batch {
    SchemaPtr schema;
    std::vector<Column> cols;
    int64_t n_rows = 0;
    ContextPtr ctx;
    int64_t memory_size() const;
};
using BatchPtr = std::shared_ptr<const Batch>;

// host <-> device movement
BatchPtr batch_from_host(const ContextPtr& ctx, int n_cols, const bhip_column_desc* cols, int64_t n_rows, bool device_ptrs);
void column_to_host(const Batch& b, int i, void* data, int32_t* offsets, uint8_t* validity);
// '|'-separated TPC-H text (host memory) -> device batch of the projected fields (tbl.cpp, kernels_tbl.hip)
BatchPtr batch_from_tbl(const ContextPtr& ctx, const void* text_host, int64_t n_bytes, int n_fields, const bhip_column_desc* fields,
                        int n_proj, const int32_t* projection);

// whole-batch operations (ops_basic.cpp)
// permutation: `indices` holds every input row exactly once (Utf8 value bytes are then known without a read-back)
BatchPtr take_batch(const Exec& ex, const Batch& in, const uint32_t* indices, int64_t n_out, SchemaPtr schema = nullptr,
                    bool permutation = false);
Column take_column(const Exec& ex, const Column& c, const uint32_t* indices, int64_t n_out, int64_t known_bytes = -1);
BatchPtr concat_batches(const Exec& ex, const SchemaPtr& schema, const std::vector<BatchPtr>& parts);
BatchPtr slice_head(const Exec& ex, const Batch& in, int64_t n);

// BHIP_TRACE_HOST=1: one stderr line per call, "[bhip-host] <us since the first point> <what>" — where the host's time goes between
// the launches of a task (the device side is rocprofv3's job)
void trace_point(const char* what);

// read a small device value once the task's stream has caught up (Context::wait_stream)
template <class T>
T read_device(const Exec& ex, const T* dev) {
    T host;
    if (sizeof(T) <= 56 && sizeof(T) % 4 == 0) {
        ex.ctx->wait_stream(ex.stream, dev, &host, sizeof(T));
        return host;
    }
    HIP_CHECK(hipMemcpyAsync(&host, dev, sizeof(T), hipMemcpyDeviceToHost, ex.stream));
    HIP_CHECK(hipStreamSynchronize(ex.stream));
    return host;
}
inline void stream_wait(const Exec& ex) { ex.ctx->wait_stream(ex.stream); }

}  // namespace bhip

// opaque C handles
struct bhip_ctx { bhip::ContextPtr p; std::atomic<int> rc{1}; };
struct bhip_batch { bhip::BatchPtr p; std::atomic<int> rc{1}; std::vector<std::string> names; };
