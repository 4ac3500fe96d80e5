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
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

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
            pieces.push_back({c.offsets->ptr(), (size_t)(b.n_rows + 1) * 4, at});
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
        return;
    }
    for (auto& p : pieces)
        if (p.bytes && p.src) HIP_CHECK(hipMemcpyAsync(dst + p.at, p.src, p.bytes, hipMemcpyDeviceToDevice, ex.stream));
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

// ---- communicator --------------------------------------------------------------------------------------------------------
class Communicator {
public:
    Communicator(ContextPtr ctx, const uint8_t* id, int world, int rank) : ctx_(std::move(ctx)), world_(world), rank_(rank) {
        if (world < 1 || rank < 0 || rank >= world) fail(BHIP_EINVAL, "communicator: rank outside the world");
        ctx_->set_device();
        ncclUniqueId uid;
        static_assert(sizeof(uid) == BHIP_COMM_ID_BYTES, "unique id size");
        memcpy(&uid, id, sizeof(uid));
        NCCL_CHECK(rccl().CommInitRank(&comm_, world, uid, rank));
        stream_ = ctx_->acquire_stream();
    }
    ~Communicator() {
        if (comm_) rccl().CommDestroy(comm_);
        if (stream_) ctx_->release_stream(stream_);
    }
    int world() const { return world_; }
    int rank() const { return rank_; }
    const ContextPtr& ctx() const { return ctx_; }

    // every rank's batch, in rank order (the order MergeExec concatenates partitions in)
    std::vector<BatchPtr> all_gather(const BatchPtr& mine) {
        std::vector<BatchPtr> out(world_);
        if (world_ == 1) { out[0] = mine; return out; }
        ctx_->set_device();
        Exec ex{ctx_, stream_};
        const SchemaPtr schema = mine->schema;
        const size_t H = pack_header_words(*schema);
        // round 1: header + block when it fits the slot, in one collective
        constexpr size_t SLOT = 16384;
        std::vector<int64_t> h(H);
        pack_header(*mine, h.data());
        const size_t head_bytes = align_up(H * 8);
        const bool fits = head_bytes + (size_t)h[1] <= SLOT;
        auto send = make_buffer(ex, SLOT);
        auto recv = make_buffer(ex, SLOT * (size_t)world_);
        HIP_CHECK(hipMemsetAsync(send->ptr(), 0, head_bytes, stream_));
        HIP_CHECK(hipMemcpyAsync(send->ptr(), h.data(), H * 8, hipMemcpyHostToDevice, stream_));
        if (fits) pack_batch(ex, *mine, send->as<uint8_t>() + head_bytes);
        NCCL_CHECK(rccl().AllGather(send->ptr(), recv->ptr(), SLOT, ncclUint8, comm_, stream_));
        std::vector<int64_t> heads((size_t)world_ * H);
        for (int r = 0; r < world_; ++r)
            HIP_CHECK(hipMemcpyAsync(&heads[(size_t)r * H], recv->as<uint8_t>() + (size_t)r * SLOT, H * 8, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        bool all_fit = true;
        for (int r = 0; r < world_; ++r) all_fit = all_fit && head_bytes + (size_t)heads[(size_t)r * H + 1] <= SLOT;
        if (all_fit) {
            for (int r = 0; r < world_; ++r)
                out[r] = r == rank_ ? mine : unpack_batch(ctx_, schema, recv, (size_t)r * SLOT + head_bytes, &heads[(size_t)r * H]);
            return out;
        }
        // round 2: blocks of any size, point to point (every rank took the same branch: the headers are common knowledge)
        auto mine_block = make_buffer(ex, (size_t)h[1] + ALIGN);
        pack_batch(ex, *mine, mine_block->as<uint8_t>());
        std::vector<BufferPtr> blocks(world_);
        NCCL_CHECK(rccl().GroupStart());
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            const size_t nb = (size_t)heads[(size_t)p * H + 1];
            blocks[p] = make_buffer(ex, nb + ALIGN);
            if (h[1]) NCCL_CHECK(rccl().Send(mine_block->ptr(), (size_t)h[1], ncclUint8, p, comm_, stream_));
            if (nb) NCCL_CHECK(rccl().Recv(blocks[p]->ptr(), nb, ncclUint8, p, comm_, stream_));
        }
        NCCL_CHECK(rccl().GroupEnd());
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int r = 0; r < world_; ++r) out[r] = r == rank_ ? mine : unpack_batch(ctx_, schema, blocks[r], 0, &heads[(size_t)r * H]);
        return out;
    }

    // parts[d] goes to rank d; out[s] = the batch rank s held for this rank
    std::vector<BatchPtr> all_to_all(const std::vector<BatchPtr>& parts) {
        if ((int)parts.size() != world_) fail(BHIP_EINVAL, "all_to_all needs one batch per rank");
        std::vector<BatchPtr> out(world_);
        if (world_ == 1) { out[0] = parts[0]; return out; }
        ctx_->set_device();
        Exec ex{ctx_, stream_};
        const SchemaPtr schema = parts[0]->schema;
        const size_t H = pack_header_words(*schema);
        // headers of my world outgoing blocks -> everyone (world x world x H int64)
        std::vector<int64_t> mine((size_t)world_ * H);
        for (int d = 0; d < world_; ++d) pack_header(*parts[d], &mine[(size_t)d * H]);
        auto hs = make_buffer(ex, mine.size() * 8);
        auto hr = make_buffer(ex, mine.size() * 8 * (size_t)world_);
        HIP_CHECK(hipMemcpyAsync(hs->ptr(), mine.data(), mine.size() * 8, hipMemcpyHostToDevice, stream_));
        NCCL_CHECK(rccl().AllGather(hs->ptr(), hr->ptr(), mine.size() * 8, ncclUint8, comm_, stream_));
        std::vector<int64_t> all((size_t)world_ * world_ * H);
        HIP_CHECK(hipMemcpyAsync(all.data(), hr->ptr(), all.size() * 8, hipMemcpyDeviceToHost, stream_));
        // pack while the headers travel back
        std::vector<BufferPtr> sendb(world_), recvb(world_);
        for (int d = 0; d < world_; ++d) {
            if (d == rank_) continue;
            const size_t nb = (size_t)mine[(size_t)d * H + 1];
            sendb[d] = make_buffer(ex, nb + ALIGN);
            pack_batch(ex, *parts[d], sendb[d]->as<uint8_t>());
        }
        HIP_CHECK(hipStreamSynchronize(stream_));
        auto head_of = [&](int src, int dst) { return &all[((size_t)src * world_ + dst) * H]; };
        NCCL_CHECK(rccl().GroupStart());
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            const size_t out_b = (size_t)mine[(size_t)p * H + 1], in_b = (size_t)head_of(p, rank_)[1];
            recvb[p] = make_buffer(ex, in_b + ALIGN);
            if (out_b) NCCL_CHECK(rccl().Send(sendb[p]->ptr(), out_b, ncclUint8, p, comm_, stream_));
            if (in_b) NCCL_CHECK(rccl().Recv(recvb[p]->ptr(), in_b, ncclUint8, p, comm_, stream_));
        }
        NCCL_CHECK(rccl().GroupEnd());
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int s = 0; s < world_; ++s) out[s] = s == rank_ ? parts[s] : unpack_batch(ctx_, schema, recvb[s], 0, head_of(s, rank_));
        return out;
    }

private:
    ContextPtr ctx_;
    int world_, rank_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
};

}  // namespace bhip

struct bhip_comm { std::unique_ptr<bhip::Communicator> c; };

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

bhip_status bhip_comm_create(bhip_ctx* ctx, const uint8_t* id, int32_t world, int32_t rank, bhip_comm** out) {
    BHIP_X_BEGIN
    if (!ctx || !id || !out) fail(BHIP_EINVAL, "null argument");
    auto h = std::make_unique<bhip_comm>();
    h->c = std::make_unique<Communicator>(ctx->p, id, world, rank);
    *out = h.release();
    BHIP_X_END
}

void bhip_comm_release(bhip_comm* comm) { delete comm; }

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
