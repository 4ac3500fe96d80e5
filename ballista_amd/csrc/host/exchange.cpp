// exchange.cpp — batches between the GPUs of one node: the shuffle of a repartitioned stage and the gather of partial
// aggregate states, over RCCL (xGMI) on device memory.
//
// What it replaces: a Ballista stage boundary writes each output partition as an Arrow IPC file
// (rust/core/src/utils.rs:49-84) and the next stage's ShuffleReaderExec pulls it over Flight / TCP
// (rust/core/src/execution_plans/shuffle_reader.rs:77-99, rust/core/src/client.rs:123-208).  With one process per GPU on
// one node a partition never leaves device memory:
//
//   pack     every buffer of a batch (values, Utf8 offsets, validity bitmaps) is laid out in ONE device block, 64-byte
//            aligned, described by a fixed-size header of int64s (rows, then per column: data bytes, offsets flag, validity flag);
//   headers  one ncclAllGather of the header matrix: every rank learns the size of every block of the exchange;
//   blocks   all_gather of small blocks (partial aggregate states: < 16 KiB) rides along in the same ncclAllGather; everything
//            else is one grouped ncclSend / ncclRecv per peer pair — xGMI is point to point, 7 peers = 7 links busy at once;
//   unpack   received blocks are NOT copied again: the columns of the resulting batch are slices of the block.
//
// RCCL is loaded with dlopen at the first communicator (the library itself does not link it), through the one HIP runtime of
// the process; nothing here needs PyTorch.  The 128-byte unique id travels by whatever channel the ranks already share
// (bench.py: a gloo broadcast on the CPU; an executor: its scheduler RPC).
//
// TRANSPORTS.  Everything above the byte movers — header matrices, block layout, who sends what to whom in which order, the
// streaming shuffle — is ONE piece of code over a two-call interface (Transport: all_gather of equal-sized device regions, one
// grouped exchange of point-to-point device regions).  Three implementations:
//   rccl      ncclAllGather / grouped ncclSend + ncclRecv: the product path, one process per GPU;
//   loopback  N communicators inside ONE process on ONE device, rendezvous on the host, hipMemcpyAsync device to device: the
//             N-rank code paths (N = 2, 3, 8) run on the single-GPU test box (tests/test_exchange_gpu.py);
//   host      the caller moves host bytes (two callbacks): bench.py's gloo rehearsal of the N-rank flow with ranks sharing a GPU,
//             or an executor that only has its Flight / TCP channel between processes.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>

#include "../sort_kernels.h"
#include "../util_kernels.h"
#include "plan.hpp"

namespace bhip {

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (auto n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { err = std::string("cannot load librccl: ") + dlerror(); return; }
#define BHIP_SYM(field, name)                                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                   \
    if (!r.field) { err = std::string("librccl lacks ") + name; return; }
        BHIP_SYM(GetUniqueId, "ncclGetUniqueId")
        BHIP_SYM(CommInitRank, "ncclCommInitRank")
        BHIP_SYM(CommDestroy, "ncclCommDestroy")
        BHIP_SYM(AllGather, "ncclAllGather")
        BHIP_SYM(Send, "ncclSend")
        BHIP_SYM(Recv, "ncclRecv")
        BHIP_SYM(GroupStart, "ncclGroupStart")
        BHIP_SYM(GroupEnd, "ncclGroupEnd")
        BHIP_SYM(GetErrorString, "ncclGetErrorString")
#undef BHIP_SYM
    });
    if (!err.empty()) fail(BHIP_EHIP, err);
    return r;
}

void nccl_check(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) fail(BHIP_EHIP, std::string(what) + ": " + rccl().GetErrorString(e));
}
#define NCCL_CHECK(expr) nccl_check((expr), #expr)

constexpr size_t ALIGN = 64;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

}  // namespace

// ---- pack / unpack -----------------------------------------------------------------------------------------------------
// header: [0] = rows, [1] = block bytes, then per column [data bytes, has offsets, has validity]
size_t pack_header_words(const Schema& s) { return 2 + 3 * s.fields.size(); }

static size_t column_data_bytes(const Column& c, int64_t n_rows) {
    if (c.dtype == DT_UTF8) return (size_t)c.data_bytes;
    if (c.dtype == DT_BOOLEAN) return bitmap_bytes(n_rows);
    return (size_t)n_rows * dtype_width(c.dtype);
}

void pack_header(const Batch& b, int64_t* h) {
    h[0] = b.n_rows;
    size_t total = 0;
    for (size_t i = 0; i < b.cols.size(); ++i) {
        const Column& c = b.cols[i];
        const size_t db = column_data_bytes(c, b.n_rows);
        h[2 + 3 * i] = (int64_t)db;
        h[3 + 3 * i] = c.dtype == DT_UTF8 ? 1 : 0;
        h[4 + 3 * i] = c.validity ? 1 : 0;
        total += align_up(db + BUFFER_SLACK);                       // slack: kernels may over-read short strings
        if (c.dtype == DT_UTF8) total += align_up((size_t)(b.n_rows + 1) * 4);
        if (c.validity) total += align_up(bitmap_bytes(b.n_rows) + 8);
    }
    h[1] = (int64_t)total;
}

// copies every buffer of `b` to `dst` (device) in header order
void pack_batch(const Exec& ex, const Batch& b, uint8_t* dst) {
    struct Piece { const void* src; size_t bytes; size_t at; };
    std::vector<Piece> pieces;
    size_t at = 0;
    for (auto& c : b.cols) {
        const size_t db = column_data_bytes(c, b.n_rows);
        pieces.push_back({c.data ? c.data->ptr() : nullptr, db, at});
        at += align_up(db + BUFFER_SLACK);
        if (c.dtype == DT_UTF8) {
            // (a 0-row Utf8 column may come without an offsets buffer — empty partitions are common after hash partitioning;
            // the block is zero-filled where nothing is copied, and offsets[0] = 0 is the whole offsets array of an empty column)
            pieces.push_back({c.offsets ? c.offsets->ptr() : nullptr, (size_t)(b.n_rows + 1) * 4, at});
            at += align_up((size_t)(b.n_rows + 1) * 4);
        }
        if (c.validity) {
            pieces.push_back({c.validity->ptr(), bitmap_bytes(b.n_rows), at});
            at += align_up(bitmap_bytes(b.n_rows) + 8);
        }
    }
    bool small = pieces.size() <= PACK_MAX;
    for (auto& p : pieces) small = small && p.bytes <= (1u << 20);
    if (small) {
        PackDesc pd;
        pd.n = 0;
        for (auto& p : pieces) {
            if (!p.bytes || !p.src) continue;
            pd.src[pd.n] = p.src;
            pd.bytes[pd.n] = (uint32_t)p.bytes;
            pd.dst[pd.n] = (uint32_t)p.at;
            ++pd.n;
        }
        if (pd.n) HIP_CHECK(launch_pack_buffers(ex.cfg(), pd, dst));
        for (auto& p : pieces)                       // a buffer the batch does not have (offsets of a 0-row Utf8 column): zeros
            if (p.bytes && !p.src) HIP_CHECK(hipMemsetAsync(dst + p.at, 0, p.bytes, ex.stream));
        return;
    }
    for (auto& p : pieces)
        if (p.bytes && p.src) HIP_CHECK(hipMemcpyAsync(dst + p.at, p.src, p.bytes, hipMemcpyDeviceToDevice, ex.stream));
        else if (p.bytes) HIP_CHECK(hipMemsetAsync(dst + p.at, 0, p.bytes, ex.stream));
}

// the batch whose buffers are slices of `block` (kept alive by the columns)
BatchPtr unpack_batch(const ContextPtr& ctx, const SchemaPtr& schema, const BufferPtr& block, size_t block_offset, const int64_t* h) {
    auto b = std::make_shared<Batch>();
    b->schema = schema;
    b->ctx = ctx;
    b->n_rows = h[0];
    uint8_t* base = block->as<uint8_t>() + block_offset;
    size_t at = 0;
    for (size_t i = 0; i < schema->fields.size(); ++i) {
        Column c;
        c.dtype = schema->fields[i].dtype;
        c.length = b->n_rows;
        const size_t db = (size_t)h[2 + 3 * i];
        c.data = std::make_shared<Buffer>(ctx, block, base + at, db);
        at += align_up(db + BUFFER_SLACK);
        if (h[3 + 3 * i]) {
            c.offsets = std::make_shared<Buffer>(ctx, block, base + at, (size_t)(b->n_rows + 1) * 4);
            at += align_up((size_t)(b->n_rows + 1) * 4);
            c.data_bytes = (int64_t)db;
        }
        if (h[4 + 3 * i]) {
            c.validity = std::make_shared<Buffer>(ctx, block, base + at, bitmap_bytes(b->n_rows));
            at += align_up(bitmap_bytes(b->n_rows) + 8);
        }
        b->cols.push_back(std::move(c));
    }
    if ((int64_t)at != h[1]) fail(BHIP_EEXEC, "exchange: block layout does not match its header");
    return b;
}

// ---- transports ------------------------------------------------------------------------------------------------------------
struct Xfer { void* ptr; size_t bytes; int peer; };

class Transport {
public:
    virtual ~Transport() = default;
    virtual const char* name() const = 0;
    // recv[r * bytes, (r + 1) * bytes) = rank r's `send` region (device memory); ordered on `stream`
    virtual void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) = 0;
    // ONE grouped exchange: every send reaches the matching receive of its peer; several regions per peer pair are matched in
    // order.  Both sides derive their lists from common knowledge (header / count matrices), so sizes agree by construction.
    virtual void exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t stream) = 0;
};

class RcclTransport : public Transport {
public:
    RcclTransport(const uint8_t* id, int world, int rank) {
        ncclUniqueId uid;
        static_assert(sizeof(uid) == BHIP_COMM_ID_BYTES, "unique id size");
        memcpy(&uid, id, sizeof(uid));
        NCCL_CHECK(rccl().CommInitRank(&comm_, world, uid, rank));
    }
    ~RcclTransport() override { if (comm_) rccl().CommDestroy(comm_); }
    const char* name() const override { return "rccl"; }
    void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        NCCL_CHECK(rccl().AllGather(send, recv, bytes, ncclUint8, comm_, stream));
    }
    void exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t stream) override {
        if (sends.empty() && recvs.empty()) return;
        // xGMI is point to point: one grouped launch keeps every peer's link busy at once
        NCCL_CHECK(rccl().GroupStart());
        for (auto& x : sends) NCCL_CHECK(rccl().Send(x.ptr, x.bytes, ncclUint8, x.peer, comm_, stream));
        for (auto& x : recvs) NCCL_CHECK(rccl().Recv(x.ptr, x.bytes, ncclUint8, x.peer, comm_, stream));
        NCCL_CHECK(rccl().GroupEnd());
    }
private:
    ncclComm_t comm_ = nullptr;
};

// N ranks of ONE process on one device: what each rank posts is visible to the others after a host rendezvous, the bytes move
// with hipMemcpyAsync.  A rank that never arrives (its thread failed) breaks the hub: the others get an error after the
// timeout instead of hanging.
struct LoopbackHub {
    explicit LoopbackHub(int w) : world(w), posts((size_t)w) {}
    const int world;
    struct Post { const void* send = nullptr; size_t bytes = 0; std::vector<Xfer> sends; hipEvent_t ready = nullptr; };
    std::vector<Post> posts;
    int joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    void barrier() {
        std::unique_lock<std::mutex> g(mu);
        if (broken) fail(BHIP_EEXEC, "loopback exchange: a peer rank failed");
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        static const int timeout_s = [] { const char* v = getenv("BHIP_LOOPBACK_TIMEOUT_S"); return v ? atoi(v) : 120; }();
        if (!cv.wait_for(g, std::chrono::seconds(timeout_s), [&] { return generation != gen || broken; })) {
            broken = true;
            cv.notify_all();
            fail(BHIP_EEXEC, "loopback exchange: a peer rank did not arrive (collective calls must be made by every rank)");
        }
        if (broken && generation == gen) fail(BHIP_EEXEC, "loopback exchange: a peer rank failed");
    }
};

class LoopbackTransport : public Transport {
public:
    LoopbackTransport(const uint8_t* id, int world, int rank) : rank_(rank), key_(reinterpret_cast<const char*>(id), BHIP_COMM_ID_BYTES) {
        std::lock_guard<std::mutex> g(registry_mu());
        auto& slot = registry()[key_];
        hub_ = slot.lock();
        if (!hub_) { hub_ = std::make_shared<LoopbackHub>(world); slot = hub_; }
        if (hub_->world != world) fail(BHIP_EINVAL, "loopback communicator: ranks disagree about the world size");
        if (hub_->joined >= world) fail(BHIP_EINVAL, "loopback communicator: more ranks than the world holds (ids are single-use)");
        ++hub_->joined;
        HIP_CHECK(hipEventCreateWithFlags(&ready_, hipEventDisableTiming));
    }
    ~LoopbackTransport() override {
        if (ready_) hipEventDestroy(ready_);
        std::lock_guard<std::mutex> g(registry_mu());
        hub_.reset();
        auto it = registry().find(key_);
        if (it != registry().end() && it->second.expired()) registry().erase(it);
    }
    const char* name() const override { return "loopback"; }
    void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        HIP_CHECK(hipEventRecord(ready_, stream));
        auto& mine = hub_->posts[(size_t)rank_];
        mine.send = send; mine.bytes = bytes; mine.ready = ready_;
        hub_->barrier();
        for (int r = 0; r < hub_->world; ++r) {
            const auto& p = hub_->posts[(size_t)r];
            if (p.bytes != bytes) { finish(stream); fail(BHIP_EEXEC, "loopback all_gather: ranks posted regions of different sizes"); }
            HIP_CHECK(hipStreamWaitEvent(stream, p.ready, 0));
            if (bytes) HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t*>(recv) + (size_t)r * bytes, p.send, bytes, hipMemcpyDeviceToDevice, stream));
        }
        finish(stream);
    }
    void exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t stream) override {
        HIP_CHECK(hipEventRecord(ready_, stream));
        auto& mine = hub_->posts[(size_t)rank_];
        mine.sends = sends; mine.ready = ready_;
        hub_->barrier();
        std::vector<size_t> taken((size_t)hub_->world, 0);          // per peer: how many of its sends to me are matched already
        std::string err;
        for (auto& rv : recvs) {
            const auto& p = hub_->posts[(size_t)rv.peer];
            size_t& k = taken[(size_t)rv.peer];
            while (k < p.sends.size() && p.sends[k].peer != rank_) ++k;
            if (k == p.sends.size()) { err = "a receive without a matching send"; break; }
            if (p.sends[k].bytes != rv.bytes) { err = "send and receive sizes differ (" + std::to_string(p.sends[k].bytes) + " vs " + std::to_string(rv.bytes) + ")"; break; }
            HIP_CHECK(hipStreamWaitEvent(stream, p.ready, 0));
            HIP_CHECK(hipMemcpyAsync(rv.ptr, p.sends[k].ptr, rv.bytes, hipMemcpyDeviceToDevice, stream));
            ++k;
        }
        if (err.empty())
            for (int r = 0; r < hub_->world && err.empty(); ++r) {
                const auto& p = hub_->posts[(size_t)r];
                size_t k = taken[(size_t)r];
                while (k < p.sends.size() && p.sends[k].peer != rank_) ++k;
                if (k != p.sends.size()) err = "a send without a matching receive";
            }
        finish(stream);
        if (!err.empty()) fail(BHIP_EEXEC, "loopback exchange: " + err);
    }
private:
    // the copies read the peers' buffers: nobody may release or overwrite its own until every rank's copies have completed
    void finish(hipStream_t stream) {
        const hipError_t e = hipStreamSynchronize(stream);
        hub_->barrier();
        HIP_CHECK(e);
    }
    static std::mutex& registry_mu() { static std::mutex m; return m; }
    static std::map<std::string, std::weak_ptr<LoopbackHub>>& registry() { static std::map<std::string, std::weak_ptr<LoopbackHub>> r; return r; }
    int rank_;
    std::string key_;
    std::shared_ptr<LoopbackHub> hub_;
    hipEvent_t ready_ = nullptr;
};

// the caller moves host bytes: regions are staged device -> host, handed to the callbacks, staged host -> device
class HostTransport : public Transport {
public:
    HostTransport(const bhip_comm_host_transport& cb, int world) : cb_(cb), world_(world) {
        if (!cb.all_gather || !cb.exchange) fail(BHIP_EINVAL, "host transport: both callbacks are required");
    }
    const char* name() const override { return "host"; }
    void all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        std::vector<uint8_t> out(bytes ? bytes : 1), in((bytes ? bytes : 1) * (size_t)world_);
        if (bytes) HIP_CHECK(hipMemcpyAsync(out.data(), send, bytes, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (cb_.all_gather(cb_.user, out.data(), in.data(), (uint64_t)bytes) != 0) fail(BHIP_EEXEC, "host transport: all_gather callback failed");
        if (bytes) HIP_CHECK(hipMemcpyAsync(recv, in.data(), bytes * (size_t)world_, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    void exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t stream) override {
        std::vector<std::vector<uint8_t>> sb(sends.size()), rb(recvs.size());
        std::vector<bhip_comm_region> s(sends.size()), r(recvs.size());
        for (size_t i = 0; i < sends.size(); ++i) {
            sb[i].resize(sends[i].bytes ? sends[i].bytes : 1);
            if (sends[i].bytes) HIP_CHECK(hipMemcpyAsync(sb[i].data(), sends[i].ptr, sends[i].bytes, hipMemcpyDeviceToHost, stream));
            s[i] = bhip_comm_region{sb[i].data(), (uint64_t)sends[i].bytes, sends[i].peer};
        }
        for (size_t i = 0; i < recvs.size(); ++i) {
            rb[i].resize(recvs[i].bytes ? recvs[i].bytes : 1);
            r[i] = bhip_comm_region{rb[i].data(), (uint64_t)recvs[i].bytes, recvs[i].peer};
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        if (cb_.exchange(cb_.user, (int32_t)s.size(), s.data(), (int32_t)r.size(), r.data()) != 0) fail(BHIP_EEXEC, "host transport: exchange callback failed");
        for (size_t i = 0; i < recvs.size(); ++i)
            if (recvs[i].bytes) HIP_CHECK(hipMemcpyAsync(recvs[i].ptr, rb[i].data(), recvs[i].bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
private:
    bhip_comm_host_transport cb_;
    int world_;
};

static BatchPtr empty_batch(const Exec& ex, const SchemaPtr& schema) {
    auto e = std::make_shared<Batch>();
    e->schema = schema;
    e->ctx = ex.ctx;
    for (auto& f : schema->fields) {
        Column c;
        c.dtype = f.dtype;
        c.data = make_buffer(ex, 8);
        if (f.dtype == DT_UTF8) { c.offsets = make_buffer(ex, 8); HIP_CHECK(hipMemsetAsync(c.offsets->ptr(), 0, 8, ex.stream)); }
        e->cols.push_back(c);
    }
    return e;
}

static void same_schema(const Schema& a, const Schema& b, const char* what) {
    bool ok = a.fields.size() == b.fields.size();
    for (size_t i = 0; ok && i < a.fields.size(); ++i) ok = a.fields[i].dtype == b.fields[i].dtype;
    if (!ok) fail(BHIP_EINVAL, std::string(what) + ": the batches of one collective call must share ONE schema");
}

// ---- communicator --------------------------------------------------------------------------------------------------------
class Communicator {
public:
    Communicator(ContextPtr ctx, std::unique_ptr<Transport> t, int world, int rank) : ctx_(std::move(ctx)), t_(std::move(t)), world_(world), rank_(rank) {
        ctx_->set_device();
        stream_ = ctx_->acquire_stream();
        aux_ = ctx_->acquire_stream();
        for (auto& e : ev_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ~Communicator() {
        for (auto e : ev_) if (e) hipEventDestroy(e);
        t_.reset();
        if (stream_) ctx_->release_stream(stream_);
        if (aux_) ctx_->release_stream(aux_);
    }
    int world() const { return world_; }
    int rank() const { return rank_; }
    const ContextPtr& ctx() const { return ctx_; }
    const char* transport() const { return t_->name(); }

    struct Stats { double seconds = 0; uint64_t bytes_out = 0, calls = 0; };
    Stats stats(bool reset) {
        std::lock_guard<std::mutex> g(mu_);
        Stats s = stats_;
        if (reset) stats_ = Stats{};
        return s;
    }

    // every rank's batch, in rank order (the order MergeExec concatenates partitions in)
    std::vector<BatchPtr> all_gather(const BatchPtr& mine) {
        std::vector<BatchPtr> out(world_);
        if (world_ == 1) { out[0] = mine; return out; }
        std::lock_guard<std::mutex> g(mu_);
        const auto t0 = std::chrono::steady_clock::now();
        ctx_->set_device();
        Exec ex{ctx_, stream_};
        const SchemaPtr schema = mine->schema;
        const size_t H = pack_header_words(*schema);
        // round 1: header + block when it fits the slot, in one collective
        constexpr size_t SLOT = 16384;
        std::vector<int64_t> h(H);
        pack_header(*mine, h.data());
        const size_t head_bytes = align_up(H * 8);
        if (head_bytes > SLOT) fail(BHIP_ENOTIMPL, "all_gather: more columns than a header slot holds");
        const bool fits = head_bytes + (size_t)h[1] <= SLOT;
        auto send = make_buffer(ex, SLOT);
        auto recv = make_buffer(ex, SLOT * (size_t)world_);
        HIP_CHECK(hipMemsetAsync(send->ptr(), 0, fits ? SLOT : head_bytes, stream_));
        HIP_CHECK(hipMemcpyAsync(send->ptr(), h.data(), H * 8, hipMemcpyHostToDevice, stream_));
        if (fits) pack_batch(ex, *mine, send->as<uint8_t>() + head_bytes);
        t_->all_gather(send->ptr(), recv->ptr(), SLOT, stream_);
        std::vector<int64_t> heads((size_t)world_ * H);
        // every rank's header in ONE strided copy (a copy call per rank is ~10 us each: eight of them per all_gather of an 8-GPU step)
        HIP_CHECK(hipMemcpy2DAsync(heads.data(), H * 8, recv->ptr(), SLOT, H * 8, (size_t)world_, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int r = 0; r < world_; ++r) check_header(&heads[(size_t)r * H], *schema, "all_gather");
        bool all_fit = true;
        for (int r = 0; r < world_; ++r) all_fit = all_fit && head_bytes + (size_t)heads[(size_t)r * H + 1] <= SLOT;
        if (all_fit) {
            for (int r = 0; r < world_; ++r)
                out[r] = r == rank_ ? mine : unpack_batch(ctx_, schema, recv, (size_t)r * SLOT + head_bytes, &heads[(size_t)r * H]);
            recv->set_stream(nullptr);
            account(t0, (uint64_t)h[1] * (uint64_t)(world_ - 1));
            return out;
        }
        // round 2: blocks of any size, point to point (every rank took the same branch: the headers are common knowledge)
        auto mine_block = make_buffer(ex, (size_t)h[1] + ALIGN);
        pack_batch(ex, *mine, mine_block->as<uint8_t>());
        std::vector<BufferPtr> blocks(world_);
        std::vector<Xfer> sends, recvs;
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            const size_t nb = (size_t)heads[(size_t)p * H + 1];
            blocks[p] = make_buffer(ex, nb + ALIGN);
            if (h[1]) sends.push_back(Xfer{mine_block->ptr(), (size_t)h[1], p});
            if (nb) recvs.push_back(Xfer{blocks[p]->ptr(), nb, p});
        }
        t_->exchange(sends, recvs, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int r = 0; r < world_; ++r) {
            if (r != rank_) blocks[r]->set_stream(nullptr);
            out[r] = r == rank_ ? mine : unpack_batch(ctx_, schema, blocks[r], 0, &heads[(size_t)r * H]);
        }
        account(t0, (uint64_t)h[1] * (uint64_t)(world_ - 1));
        return out;
    }

    // parts[d] goes to rank d; out[s] = the batch rank s held for this rank
    std::vector<BatchPtr> all_to_all(const std::vector<BatchPtr>& parts) {
        if ((int)parts.size() != world_) fail(BHIP_EINVAL, "all_to_all needs one batch per rank");
        for (auto& p : parts) {
            if (!p) fail(BHIP_EINVAL, "all_to_all: null part");
            same_schema(*parts[0]->schema, *p->schema, "all_to_all");
        }
        std::vector<BatchPtr> out(world_);
        if (world_ == 1) { out[0] = parts[0]; return out; }
        std::lock_guard<std::mutex> g(mu_);
        const auto t0 = std::chrono::steady_clock::now();
        ctx_->set_device();
        Exec ex{ctx_, stream_};
        const SchemaPtr schema = parts[0]->schema;
        const size_t H = pack_header_words(*schema);
        // headers of my world outgoing blocks -> everyone (world x world x H int64)
        std::vector<int64_t> mine((size_t)world_ * H);
        for (int d = 0; d < world_; ++d) pack_header(*parts[d], &mine[(size_t)d * H]);
        const std::vector<int64_t> all = gather_words(mine);
        // pack while nothing else is pending
        std::vector<BufferPtr> sendb(world_), recvb(world_);
        uint64_t out_bytes = 0;
        for (int d = 0; d < world_; ++d) {
            if (d == rank_) continue;
            const size_t nb = (size_t)mine[(size_t)d * H + 1];
            sendb[d] = make_buffer(ex, nb + ALIGN);
            pack_batch(ex, *parts[d], sendb[d]->as<uint8_t>());
            out_bytes += nb;
        }
        auto head_of = [&](int src, int dst) { return &all[((size_t)src * world_ + dst) * H]; };
        for (int s = 0; s < world_; ++s) check_header(head_of(s, rank_), *schema, "all_to_all");
        std::vector<Xfer> sends, recvs;
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            const size_t out_b = (size_t)mine[(size_t)p * H + 1], in_b = (size_t)head_of(p, rank_)[1];
            recvb[p] = make_buffer(ex, in_b + ALIGN);
            if (out_b) sends.push_back(Xfer{sendb[p]->ptr(), out_b, p});
            if (in_b) recvs.push_back(Xfer{recvb[p]->ptr(), in_b, p});
        }
        t_->exchange(sends, recvs, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int s = 0; s < world_; ++s) {
            if (s != rank_) recvb[s]->set_stream(nullptr);
            out[s] = s == rank_ ? parts[s] : unpack_batch(ctx_, schema, recvb[s], 0, head_of(s, rank_));
        }
        account(t0, out_bytes);
        return out;
    }

    // RepartitionExec(Hash([key], world)) + the shuffle read in one call: the rows of every rank's `in` whose key hashes to this
    // rank, in source-rank order and input order within a source (what all_to_all of bhip_batch_hash_partition + concat returns).
    //
    // Fixed-width NULL-free columns and one NULL-free integer key (every exchange of the TPC-H joins) STREAM: a count pass over
    // the key column sizes everything (a [chunk][destination] matrix, gathered from every rank), the result columns are
    // allocated once at their exact size, and the input goes through in chunks of `chunk_rows` — partition_scatter into one of
    // two staging buffers on the auxiliary stream while the previous chunk's grouped send / receive is in flight on the
    // communicator's stream; a peer's rows land directly at their final position.  Device memory beyond input and result:
    // two chunks (not the N partitions + N packed blocks + N received blocks + the concatenation the general path holds).
    BatchPtr shuffle(const BatchPtr& in, const std::string& key, int64_t chunk_rows, bhip_shuffle_stats* st) {
        if (st) memset(st, 0, sizeof(*st));
        const int ki = in->schema->index_of(key);
        if (ki < 0) fail(BHIP_EINVAL, "shuffle: no column named '" + key + "'");
        if (world_ > 256) fail(BHIP_ENOTIMPL, "shuffle over more than 256 ranks");
        const Column& kc = in->cols[(size_t)ki];
        const int kw = kc.validity ? 0 : (kc.dtype == DT_INT32 || kc.dtype == DT_DATE32) ? 4 : (kc.dtype == DT_INT64 || kc.dtype == DT_UINT64) ? 8 : 0;
        bool fixed = kw != 0 && (int)in->cols.size() <= TAKE_MANY_MAX;
        for (auto& c : in->cols) fixed = fixed && !c.validity && !c.is_view() && c.dtype != DT_UTF8 && c.dtype != DT_BOOLEAN;
        static const bool no_stream = [] { const char* v = getenv("BHIP_NO_STREAMING_SHUFFLE"); return v && atoi(v) != 0; }();
        const auto t0 = std::chrono::steady_clock::now();
        // ONE gather tells every rank everything: each rank chunks its OWN rows (at most MAXC chunks: the matrix has a common size),
        // counts them per destination on the device and publishes [rows, "streams", chunks, counts[MAXC][world]].  Every rank must
        // take the same path: if any rank cannot stream (a NULL-able column there), all take the general one.
        constexpr int64_t MAXC = 64;
        const int64_t quantum = partition_chunk_quantum();
        const int64_t n = in->n_rows;
        const bool stream_me = fixed && !no_stream;
        if (chunk_rows <= 0) chunk_rows = 64ll << 20;                                  // 64 Mi rows: 1.8 GB of Q5's 28-byte lineitem rows per staging buffer
        chunk_rows = std::max(chunk_rows, (n + MAXC - 1) / MAXC);
        chunk_rows = (chunk_rows + quantum - 1) / quantum * quantum;
        if (chunk_rows > n) chunk_rows = std::max<int64_t>(quantum, (n + quantum - 1) / quantum * quantum);
        const int64_t C_me = (n + chunk_rows - 1) / chunk_rows;
        const size_t W = 3 + (size_t)MAXC * world_;
        std::vector<int64_t> me(W, 0);
        me[0] = n; me[1] = stream_me ? 1 : 0; me[2] = C_me;
        std::vector<int64_t> all;
        double ms_count = 0;
        {
            std::lock_guard<std::mutex> g(mu_);
            ctx_->set_device();
            Exec ex{ctx_, stream_};
            // ---- count pass: counts[c][d] = rows of my chunk c that go to rank d ----------------------------------------------
            if (stream_me && n > 0) {
                Temp tmp(ex);
                uint64_t* dev = tmp.get<uint64_t>((size_t)MAXC * world_);
                HIP_CHECK(hipMemsetAsync(dev, 0, (size_t)MAXC * world_ * 8, stream_));
                TIMED_LAUNCH_B(ex, "partition_count", n, (uint64_t)n * kw, partition_count(ex.cfg(), kc.data->ptr(), kw, n, (uint32_t)world_, chunk_rows, dev));
                HIP_CHECK(hipMemcpyAsync(&me[3], dev, (size_t)MAXC * world_ * 8, hipMemcpyDeviceToHost, stream_));
                HIP_CHECK(hipStreamSynchronize(stream_));
            }
            ms_count = ms_since(t0);
            all = world_ == 1 ? me : gather_words(me);                                   // [src][3 + chunk * world + dst]
        }
        bool all_stream = true;
        int64_t C = 0;                                                                   // chunk rounds every rank walks (its own chunks may run out earlier)
        for (int r = 0; r < world_; ++r) {
            all_stream = all_stream && all[(size_t)r * W + 1] != 0;
            C = std::max(C, all[(size_t)r * W + 2]);
        }
        if (!all_stream) {
            // general path: any column type, NULLs
            ctx_->set_device();
            Exec ex{ctx_, stream_};
            std::vector<ExprPtr> exprs = {make_column(key)};
            std::vector<BatchPtr> parts;
            {
                std::lock_guard<std::mutex> g(mu_);
                parts = hash_partition_batch(ex, in, exprs, world_);
                HIP_CHECK(hipStreamSynchronize(stream_));
            }
            if (st) for (int d = 0; d < world_ && d < BHIP_SHUFFLE_MAX_PEERS; ++d) st->rows_to[d] = (uint64_t)parts[(size_t)d]->n_rows;
            auto got = all_to_all(parts);
            std::lock_guard<std::mutex> g(mu_);
            std::vector<BatchPtr> live;
            for (auto& b : got) if (b->n_rows) live.push_back(b);
            BatchPtr out = live.empty() ? got[0] : live.size() == 1 ? live[0] : concat_batches(ex, in->schema, live);
            HIP_CHECK(hipStreamSynchronize(stream_));
            if (st) {
                st->rows_in = (uint64_t)in->n_rows; st->rows_out = (uint64_t)out->n_rows; st->streamed = 0;
                st->ms_total = ms_since(t0);
            }
            return out;
        }
        std::lock_guard<std::mutex> g(mu_);
        ctx_->set_device();
        Exec ex{ctx_, stream_}, ax{ctx_, aux_};
        auto cnt = [&](int src, int64_t c, int dst) { return all[(size_t)src * W + 3 + (size_t)c * world_ + dst]; };
        // rows I receive from each source, where each source's rows start in the result
        std::vector<int64_t> from((size_t)world_, 0), start((size_t)world_ + 1, 0);
        for (int s = 0; s < world_; ++s) {
            for (int64_t c = 0; c < C; ++c) from[(size_t)s] += cnt(s, c, rank_);
            start[(size_t)s + 1] = start[(size_t)s] + from[(size_t)s];
        }
        const int64_t n_out = start[(size_t)world_];
        if (n_out > 0xFFFFFFF0ll) fail(BHIP_ENOTIMPL, "shuffle: this rank would receive more than 2^32-16 rows in one batch");
        // ---- result columns, two staging sets -----------------------------------------------------------------------------------
        const size_t n_cols = in->cols.size();
        auto out = std::make_shared<Batch>();
        out->schema = in->schema;
        out->ctx = ctx_;
        out->n_rows = n_out;
        size_t row_bytes = 0;
        for (auto& c : in->cols) {
            Column oc;
            oc.dtype = c.dtype;
            oc.length = n_out;
            oc.data = make_buffer(ex, (size_t)n_out * dtype_width(c.dtype) + 8);
            out->cols.push_back(oc);
            row_bytes += (size_t)dtype_width(c.dtype);
        }
        const int64_t stage_rows = std::min<int64_t>(chunk_rows, std::max<int64_t>(n, 1));
        std::vector<BufferPtr> stage[2];
        BufferPtr scatter_tmp[2];
        const int n_stage = C > 1 ? 2 : 1;
        for (int b = 0; b < n_stage; ++b) {
            for (auto& c : in->cols) stage[b].push_back(make_buffer(ax, (size_t)stage_rows * dtype_width(c.dtype) + 8));
            scatter_tmp[b] = make_buffer(ax, partition_scatter_temp_bytes(stage_rows) + 64);
        }
        std::vector<int64_t> got((size_t)world_, 0);                    // rows of each source already placed
        uint64_t bytes_remote = 0, bytes_local = 0;
        // the producer's stream finished before this call (the plan node waits); the result buffers were allocated on stream_
        for (int64_t c = 0; c < C; ++c) {
            const int b = (int)(c % n_stage);
            const int64_t lo = c * chunk_rows, rows = c < C_me ? std::min(chunk_rows, n - lo) : 0;
            if (c >= n_stage) HIP_CHECK(hipStreamWaitEvent(aux_, ev_[2 + b], 0));       // the exchange that read this staging set is done
            if (rows > 0) {
                TakeMany tm;
                tm.n = 0;
                for (size_t ci = 0; ci < n_cols; ++ci) {
                    const int w = dtype_width(in->cols[ci].dtype);
                    tm.src[tm.n] = in->cols[ci].data->as<uint8_t>() + (size_t)lo * w;
                    tm.dst[tm.n] = stage[b][ci]->ptr();
                    tm.width[tm.n] = w;
                    ++tm.n;
                }
                KernelTimer kt(ax, "partition_scatter", rows, (uint64_t)rows * (2 * row_bytes + 2 * (size_t)kw));
                HIP_CHECK(partition_scatter(ax.cfg(), kc.data->as<uint8_t>() + (size_t)lo * kw, kw, rows, (uint32_t)world_, tm, scatter_tmp[b]->ptr(), nullptr));
            }
            HIP_CHECK(hipEventRecord(ev_[b], aux_));
            HIP_CHECK(hipStreamWaitEvent(stream_, ev_[b], 0));
            // this chunk's sends (from the staging set, partition-contiguous) and receives (straight into the result)
            std::vector<Xfer> sends, recvs;
            int64_t first = 0;
            for (int d = 0; d < world_; ++d) {
                const int64_t k = c < C_me ? cnt(rank_, c, d) : 0;
                if (k > 0) {
                    for (size_t ci = 0; ci < n_cols; ++ci) {
                        const size_t w = (size_t)dtype_width(in->cols[ci].dtype);
                        uint8_t* src = stage[b][ci]->as<uint8_t>() + (size_t)first * w;
                        if (d == rank_) {
                            uint8_t* dst = out->cols[ci].data->as<uint8_t>() + (size_t)(start[(size_t)rank_] + got[(size_t)rank_]) * w;
                            HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)k * w, hipMemcpyDeviceToDevice, stream_));
                            bytes_local += (uint64_t)k * w;
                        } else {
                            sends.push_back(Xfer{src, (size_t)k * w, d});
                            bytes_remote += (uint64_t)k * w;
                        }
                    }
                    if (st && d < BHIP_SHUFFLE_MAX_PEERS) st->rows_to[d] += (uint64_t)k;
                }
                first += k;
            }
            got[(size_t)rank_] += c < C_me ? cnt(rank_, c, rank_) : 0;
            for (int s = 0; s < world_; ++s) {
                if (s == rank_) continue;
                const int64_t k = cnt(s, c, rank_);
                if (k <= 0) continue;
                for (size_t ci = 0; ci < n_cols; ++ci) {
                    const size_t w = (size_t)dtype_width(in->cols[ci].dtype);
                    recvs.push_back(Xfer{out->cols[ci].data->as<uint8_t>() + (size_t)(start[(size_t)s] + got[(size_t)s]) * w, (size_t)k * w, s});
                }
                got[(size_t)s] += k;
            }
            t_->exchange(sends, recvs, stream_);         // every rank, every chunk (a rendezvous transport counts calls), even with nothing to move
            HIP_CHECK(hipEventRecord(ev_[2 + b], stream_));
        }
        HIP_CHECK(hipStreamSynchronize(stream_));
        HIP_CHECK(hipStreamSynchronize(aux_));
        for (auto& c : out->cols) c.data->set_stream(nullptr);
        for (int s = 0; s < world_; ++s)
            if (got[(size_t)s] != from[(size_t)s]) fail(BHIP_EEXEC, "shuffle: received rows do not add up to the count matrix");
        if (st) {
            st->rows_in = (uint64_t)n; st->rows_out = (uint64_t)n_out; st->streamed = 1; st->chunks = (uint64_t)C;
            st->bytes_sent_remote = bytes_remote; st->bytes_kept_local = bytes_local;
            st->staging_bytes = (uint64_t)n_stage * ((uint64_t)stage_rows * row_bytes + partition_scatter_temp_bytes(stage_rows));
            st->ms_count = ms_count; st->ms_total = ms_since(t0);
        }
        stats_.seconds += ms_since(t0) * 1e-3;
        stats_.bytes_out += bytes_remote;
        stats_.calls += 1;
        return out;
    }

private:
    static double ms_since(std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void account(std::chrono::steady_clock::time_point t0, uint64_t bytes_out) {
        stats_.seconds += ms_since(t0) * 1e-3;
        stats_.bytes_out += bytes_out;
        stats_.calls += 1;
    }
    // a header another rank sent: plausible for `schema`?  (a rank that called with a different schema would otherwise make this
    // one slice a block by a foreign layout)
    static void check_header(const int64_t* h, const Schema& schema, const char* what) {
        bool ok = h[0] >= 0 && h[1] >= 0;
        for (size_t i = 0; ok && i < schema.fields.size(); ++i) {
            const bool utf8 = schema.fields[i].dtype == DT_UTF8;
            ok = h[2 + 3 * i] >= 0 && (h[3 + 3 * i] != 0) == utf8;
            const int w = dtype_width(schema.fields[i].dtype);
            if (ok && w > 0) ok = h[2 + 3 * i] == h[0] * w;
        }
        if (!ok) fail(BHIP_EEXEC, std::string(what) + ": a peer's block header does not fit this rank's schema (collective calls need ONE schema on every rank)");
    }
    // all_gather of a few int64 words per rank (the same count on every rank) -> [rank][words]
    std::vector<int64_t> gather_words(const std::vector<int64_t>& mine) {
        Exec ex{ctx_, stream_};
        const size_t nb = mine.size() * 8;
        auto hs = make_buffer(ex, nb);
        auto hr = make_buffer(ex, nb * (size_t)world_);
        HIP_CHECK(hipMemcpyAsync(hs->ptr(), mine.data(), nb, hipMemcpyHostToDevice, stream_));
        t_->all_gather(hs->ptr(), hr->ptr(), nb, stream_);
        std::vector<int64_t> all(mine.size() * (size_t)world_);
        HIP_CHECK(hipMemcpyAsync(all.data(), hr->ptr(), all.size() * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        return all;
    }

    ContextPtr ctx_;
    std::unique_ptr<Transport> t_;
    int world_, rank_;
    hipStream_t stream_ = nullptr, aux_ = nullptr;
    hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};          // [0,1] staging set scattered, [2,3] staging set sent
    std::mutex mu_;                                                    // one collective at a time per communicator
    Stats stats_;
};

// ---- the exchange as plan nodes ----------------------------------------------------------------------------------------------
// A stage boundary inside ONE plan per rank: where the reference writes a stage's partitions to IPC files and the next stage's
// ShuffleReaderExec pulls them (rust/scheduler/src/planner.rs:136-171, rust/core/src/execution_plans/shuffle_reader.rs:77-99),
// these nodes move the batches between the ranks' GPUs when they are executed — so a rank's whole distributed query is one
// operator tree and one bhip_plan_collect per step.
class ExchangeNode : public UnaryExec {
protected:
    std::shared_ptr<Communicator> comm_;
    struct Cache { std::mutex mu; bool done = false; std::vector<BatchPtr> parts; };
    std::shared_ptr<Cache> cache_ = std::make_shared<Cache>();
    // every partition of the input as ONE batch, complete in memory (the communicator works on its own streams)
    BatchPtr drain_input(const Exec& ex) const {
        std::vector<BatchPtr> all;
        const int n_in = input_->output_partitioning().count;
        for (int p = 0; p < n_in; ++p) {
            auto s = input_->execute(p, ex);
            while (BatchPtr b = s->next())
                if (b->n_rows) all.push_back(materialize_batch(ex, b));
        }
        BatchPtr one = all.empty() ? empty_batch(ex, input_->schema()) : all.size() == 1 ? all[0] : concat_batches(ex, input_->schema(), all);
        stream_wait(ex);
        return one;
    }
};

class AllGatherExec : public ExchangeNode {
public:
    AllGatherExec(std::shared_ptr<Communicator> comm, PlanPtr input) {
        comm_ = std::move(comm);
        input_ = std::move(input);
        ctx_ = input_->context() ? input_->context() : comm_->ctx();
    }
    const char* name() const override { return "AllGatherExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, comm_->world(), {}}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override {
        if (c.size() != 1) fail(BHIP_EINVAL, "AllGatherExec wrong number of children");
        return std::make_shared<AllGatherExec>(comm_, c[0]);
    }
    std::string describe() const override { return std::string("AllGatherExec: ") + comm_->transport() + ", world=" + std::to_string(comm_->world()); }
    StreamPtr execute(int partition, const Exec& ex) const override {
        check_partition(*this, partition);
        auto self = std::static_pointer_cast<const AllGatherExec>(shared_from_this());
        return StreamPtr(new LazyStream(schema(), [self, partition, ex]() {
            std::lock_guard<std::mutex> g(self->cache_->mu);
            if (!self->cache_->done) {
                self->cache_->parts = self->comm_->all_gather(self->drain_input(ex));
                self->cache_->done = true;
            }
            return std::vector<BatchPtr>{self->cache_->parts[(size_t)partition]};
        }));
    }
};

class ShuffleExchangeExec : public ExchangeNode {
public:
    ShuffleExchangeExec(std::shared_ptr<Communicator> comm, PlanPtr input, std::string key, int64_t chunk_rows)
        : key_(std::move(key)), chunk_rows_(chunk_rows) {
        comm_ = std::move(comm);
        input_ = std::move(input);
        ctx_ = input_->context() ? input_->context() : comm_->ctx();
        if (input_->schema()->index_of(key_) < 0) fail(BHIP_EINVAL, "ShuffleExchangeExec: No field named '" + key_ + "'");
    }
    const char* name() const override { return "ShuffleExchangeExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    // this rank's partition of Hash([key], world)
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, 1, {}}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override {
        if (c.size() != 1) fail(BHIP_EINVAL, "ShuffleExchangeExec wrong number of children");
        return std::make_shared<ShuffleExchangeExec>(comm_, c[0], key_, chunk_rows_);
    }
    std::string describe() const override {
        return "ShuffleExchangeExec: partitioning=Hash([" + key_ + "], " + std::to_string(comm_->world()) + "), " + comm_->transport() + ", rank " +
               std::to_string(comm_->rank());
    }
    StreamPtr execute(int partition, const Exec& ex) const override {
        check_partition(*this, partition);
        auto self = std::static_pointer_cast<const ShuffleExchangeExec>(shared_from_this());
        return StreamPtr(new LazyStream(schema(), [self, ex]() {
            std::lock_guard<std::mutex> g(self->cache_->mu);
            if (!self->cache_->done) {
                self->cache_->parts = {self->comm_->shuffle(self->drain_input(ex), self->key_, self->chunk_rows_, nullptr)};
                self->cache_->done = true;
            }
            return self->cache_->parts;
        }));
    }
private:
    std::string key_;
    int64_t chunk_rows_;
};

}  // namespace bhip

struct bhip_comm { std::shared_ptr<bhip::Communicator> c; };

using namespace bhip;

#define BHIP_X_BEGIN try {
#define BHIP_X_END                                                                                     \
    return BHIP_OK;                                                                                    \
    }                                                                                                  \
    catch (const bhip::Error& e) { bhip::set_last_error(e.what()); return e.code; }                    \
    catch (const std::exception& e) { bhip::set_last_error(std::string("internal error: ") + e.what()); return BHIP_EINVAL; }

extern "C" {

bhip_status bhip_comm_unique_id(uint8_t* id) {
    BHIP_X_BEGIN
    if (!id) fail(BHIP_EINVAL, "null argument: id");
    ncclUniqueId uid;
    NCCL_CHECK(rccl().GetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    BHIP_X_END
}

static void check_world(int32_t world, int32_t rank) {
    if (world < 1 || rank < 0 || rank >= world) fail(BHIP_EINVAL, "communicator: rank outside the world");
}

bhip_status bhip_comm_create(bhip_ctx* ctx, const uint8_t* id, int32_t world, int32_t rank, bhip_comm** out) {
    BHIP_X_BEGIN
    if (!ctx || !id || !out) fail(BHIP_EINVAL, "null argument");
    check_world(world, rank);
    ctx->p->set_device();
    auto h = std::make_unique<bhip_comm>();
    h->c = std::make_shared<Communicator>(ctx->p, std::make_unique<RcclTransport>(id, world, rank), world, rank);
    *out = h.release();
    BHIP_X_END
}

bhip_status bhip_comm_create_loopback(bhip_ctx* ctx, const uint8_t* id, int32_t world, int32_t rank, bhip_comm** out) {
    BHIP_X_BEGIN
    if (!ctx || !id || !out) fail(BHIP_EINVAL, "null argument");
    check_world(world, rank);
    ctx->p->set_device();
    auto h = std::make_unique<bhip_comm>();
    h->c = std::make_shared<Communicator>(ctx->p, std::make_unique<LoopbackTransport>(id, world, rank), world, rank);
    *out = h.release();
    BHIP_X_END
}

bhip_status bhip_comm_create_host(bhip_ctx* ctx, const bhip_comm_host_transport* transport, int32_t world, int32_t rank, bhip_comm** out) {
    BHIP_X_BEGIN
    if (!ctx || !transport || !out) fail(BHIP_EINVAL, "null argument");
    check_world(world, rank);
    ctx->p->set_device();
    auto h = std::make_unique<bhip_comm>();
    h->c = std::make_shared<Communicator>(ctx->p, std::make_unique<HostTransport>(*transport, world), world, rank);
    *out = h.release();
    BHIP_X_END
}

void bhip_comm_release(bhip_comm* comm) { delete comm; }

bhip_status bhip_comm_info(bhip_comm* comm, int32_t* world, int32_t* rank, const char** transport) {
    BHIP_X_BEGIN
    if (!comm) fail(BHIP_EINVAL, "null argument");
    if (world) *world = comm->c->world();
    if (rank) *rank = comm->c->rank();
    if (transport) *transport = comm->c->transport();
    BHIP_X_END
}

bhip_status bhip_comm_stats(bhip_comm* comm, int32_t reset, double* seconds, uint64_t* bytes_out, uint64_t* calls) {
    BHIP_X_BEGIN
    if (!comm) fail(BHIP_EINVAL, "null argument");
    const auto s = comm->c->stats(reset != 0);
    if (seconds) *seconds = s.seconds;
    if (bytes_out) *bytes_out = s.bytes_out;
    if (calls) *calls = s.calls;
    BHIP_X_END
}

static bhip_batch* wrap(BatchPtr b) {
    auto h = new bhip_batch();
    h->p = std::move(b);
    return h;
}

bhip_status bhip_comm_all_gather(bhip_comm* comm, bhip_batch* mine, bhip_batch** out) {
    BHIP_X_BEGIN
    if (!comm || !mine || !out) fail(BHIP_EINVAL, "null argument");
    auto got = comm->c->all_gather(mine->p);
    for (size_t i = 0; i < got.size(); ++i) out[i] = wrap(got[i]);
    BHIP_X_END
}

bhip_status bhip_comm_all_to_all(bhip_comm* comm, bhip_batch* const* parts, bhip_batch** out) {
    BHIP_X_BEGIN
    if (!comm || !parts || !out) fail(BHIP_EINVAL, "null argument");
    std::vector<BatchPtr> in;
    for (int i = 0; i < comm->c->world(); ++i) {
        if (!parts[i]) fail(BHIP_EINVAL, "null argument: part");
        in.push_back(parts[i]->p);
    }
    auto got = comm->c->all_to_all(in);
    for (size_t i = 0; i < got.size(); ++i) out[i] = wrap(got[i]);
    BHIP_X_END
}

bhip_status bhip_comm_shuffle(bhip_comm* comm, bhip_batch* batch, const char* key_column, int64_t chunk_rows, bhip_batch** out,
                              bhip_shuffle_stats* stats) {
    BHIP_X_BEGIN
    if (!comm || !batch || !key_column || !out) fail(BHIP_EINVAL, "null argument");
    *out = wrap(comm->c->shuffle(batch->p, key_column, chunk_rows, stats));
    BHIP_X_END
}

bhip_status bhip_plan_all_gather(bhip_comm* comm, bhip_plan* input, bhip_plan** out) {
    BHIP_X_BEGIN
    if (!comm || !input || !out) fail(BHIP_EINVAL, "null argument");
    auto h = new bhip_plan();
    try { h->p = std::make_shared<AllGatherExec>(comm->c, input->p); } catch (...) { delete h; throw; }
    *out = h;
    BHIP_X_END
}

bhip_status bhip_plan_shuffle(bhip_comm* comm, bhip_plan* input, const char* key_column, int64_t chunk_rows, bhip_plan** out) {
    BHIP_X_BEGIN
    if (!comm || !input || !key_column || !out) fail(BHIP_EINVAL, "null argument");
    auto h = new bhip_plan();
    try { h->p = std::make_shared<ShuffleExchangeExec>(comm->c, input->p, key_column, chunk_rows); } catch (...) { delete h; throw; }
    *out = h;
    BHIP_X_END
}

// pack / unpack on their own: what a transport other than RCCL (the gloo rehearsal of bench.py, a test) moves
bhip_status bhip_batch_pack(bhip_batch* batch, int64_t* header, int32_t header_cap, void* host_block, int64_t block_cap, int64_t* block_bytes) {
    BHIP_X_BEGIN
    if (!batch || !header || !block_bytes) fail(BHIP_EINVAL, "null argument");
    const Batch& b = *batch->p;
    const size_t H = pack_header_words(*b.schema);
    if ((size_t)header_cap < H) fail(BHIP_EINVAL, "bhip_batch_pack: header needs " + std::to_string(H) + " words");
    pack_header(b, header);
    *block_bytes = header[1];
    if (!host_block) return BHIP_OK;                                   // size query
    if (block_cap < header[1]) fail(BHIP_EINVAL, "bhip_batch_pack: block buffer too small");
    b.ctx->set_device();
    Exec ex{b.ctx, b.ctx->acquire_stream()};
    try {
        auto dev = make_buffer(ex, (size_t)header[1] + 64);
        HIP_CHECK(hipMemsetAsync(dev->ptr(), 0, (size_t)header[1], ex.stream));
        pack_batch(ex, b, dev->as<uint8_t>());
        HIP_CHECK(hipMemcpyAsync(host_block, dev->ptr(), (size_t)header[1], hipMemcpyDeviceToHost, ex.stream));
        HIP_CHECK(hipStreamSynchronize(ex.stream));
    } catch (...) { b.ctx->release_stream(ex.stream); throw; }
    b.ctx->release_stream(ex.stream);
    BHIP_X_END
}

bhip_status bhip_batch_unpack(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* schema, const int64_t* header, const void* host_block,
                              bhip_batch** out) {
    BHIP_X_BEGIN
    if (!ctx || !header || !out || (n_cols > 0 && !schema)) fail(BHIP_EINVAL, "null argument");
    auto s = std::make_shared<Schema>();
    for (int i = 0; i < n_cols; ++i) s->fields.push_back(Field{schema[i].name ? schema[i].name : "", schema[i].dtype, schema[i].nullable != 0});
    ctx->p->set_device();
    Exec ex{ctx->p, ctx->p->acquire_stream()};
    BatchPtr b;
    try {
        auto dev = make_buffer(ex, (size_t)header[1] + 64);
        if (header[1]) {
            if (!host_block) fail(BHIP_EINVAL, "null argument: host_block");
            HIP_CHECK(hipMemcpyAsync(dev->ptr(), host_block, (size_t)header[1], hipMemcpyHostToDevice, ex.stream));
        }
        HIP_CHECK(hipStreamSynchronize(ex.stream));
        dev->set_stream(nullptr);
        b = unpack_batch(ctx->p, s, dev, 0, header);
    } catch (...) { ctx->p->release_stream(ex.stream); throw; }
    ctx->p->release_stream(ex.stream);
    *out = wrap(b);
    BHIP_X_END
}

}  // extern "C"
