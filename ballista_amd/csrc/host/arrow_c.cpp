// arrow_c.cpp — Arrow C Data / C Stream interface at the host edge of the library.
//
// The Arrow C Stream Interface (get_schema / get_next / get_last_error / release) is the C
// analogue of the reference's `RecordBatchStream` (rust/core/src/memory_stream.rs:57-92); the
// C Data Interface carries one RecordBatch as a struct array.  The Rust executor side would
// produce/consume these through arrow-rs' `ffi` module.
#include <cstdlib>
#include <cstring>

#include "../util_kernels.h"
#include <memory>
#include <mutex>
#include "plan.hpp"

using namespace bhip;

// Arrow C Data Interface format strings <-> bhip_dtype (shared with ipc.cpp)
namespace bhip {
int dtype_from_format(const char* f) {
    if (!f) return 0;
    static const struct { const char* fmt; int dt; } table[] = {
        {"i", DT_INT32}, {"l", DT_INT64}, {"C", DT_UINT8}, {"L", DT_UINT64}, {"g", DT_FLOAT64}, {"tdD", DT_DATE32}, {"b", DT_BOOLEAN},
        {"u", DT_UTF8}, {"c", DT_INT8}, {"s", DT_INT16}, {"S", DT_UINT16}, {"I", DT_UINT32}, {"f", DT_FLOAT32}, {"tdm", DT_DATE64},
        {"tss:", DT_TIMESTAMP_S}, {"tsm:", DT_TIMESTAMP_MS}, {"tsu:", DT_TIMESTAMP_US}, {"tsn:", DT_TIMESTAMP_NS}, {"U", DT_LARGE_UTF8}, {"z", DT_BINARY}};
    for (auto& e : table)
        if (!strcmp(f, e.fmt)) return e.dt;
    return 0;
}

const char* format_of_dtype(int dt) {
    switch (dt) {
        case DT_INT32: return "i";
        case DT_INT64: return "l";
        case DT_UINT8: return "C";
        case DT_UINT64: return "L";
        case DT_FLOAT64: return "g";
        case DT_DATE32: return "tdD";
        case DT_BOOLEAN: return "b";
        case DT_INT8: return "c";
        case DT_INT16: return "s";
        case DT_UINT16: return "S";
        case DT_UINT32: return "I";
        case DT_FLOAT32: return "f";
        case DT_DATE64: return "tdm";
        case DT_TIMESTAMP_S: return "tss:";
        case DT_TIMESTAMP_MS: return "tsm:";
        case DT_TIMESTAMP_US: return "tsu:";
        case DT_TIMESTAMP_NS: return "tsn:";
        case DT_LARGE_UTF8: return "U";
        case DT_BINARY: return "z";
        default: return "u";
    }
}
}  // namespace bhip

namespace {

const char* format_of(int dt) { return format_of_dtype(dt); }

// what the device operators carry: every primitive type the serde ships (rust/core/proto/ballista.proto:755-790) except
// LargeUtf8, Binary, Float16, Time32/64, Interval, Duration, Decimal and the nested types — a batch of those is refused here
// (BHIP_ENOTIMPL: keep the CPU operator); timestamps with a time zone likewise
int device_dtype_from_format(const char* f) {
    const int dt = dtype_from_format(f);
    return (dt >= DT_INT32 && dt <= DT_LAST) || dt == DT_LARGE_UTF8 || dt == DT_BINARY ? dt : 0;
}

// copy n bits starting at bit `off` of src into a fresh, zero-padded bitmap
std::vector<uint8_t> realign_bits(const uint8_t* src, int64_t off, int64_t n) {
    std::vector<uint8_t> out((size_t)((n + 63) / 64) * 8 + 8, 0);
    if ((off & 7) == 0) { memcpy(out.data(), src + off / 8, (size_t)((n + 7) / 8)); }
    else
        for (int64_t i = 0; i < n; ++i) {
            const int64_t s = off + i;
            if ((src[s >> 3] >> (s & 7)) & 1) out[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        }
    // clear bits past n in the last byte
    if (n & 7) out[(size_t)(n >> 3)] &= (uint8_t)((1u << (n & 7)) - 1u);
    return out;
}

// ---- export ------------------------------------------------------------------------------------------
struct ExportedArray {
    std::vector<void*> owned;                 // malloc'd buffers
    std::vector<const void*> buffers;
    std::vector<ArrowArray> child_storage;
    std::vector<ArrowArray*> child_ptrs;
    std::vector<ExportedArray*> child_priv;
    ~ExportedArray() { for (void* p : owned) free(p); }
};

void release_array(ArrowArray* a) {
    if (!a || !a->release) return;
    auto* priv = static_cast<ExportedArray*>(a->private_data);
    for (auto& c : priv->child_storage)
        if (c.release) c.release(&c);
    delete priv;
    a->release = nullptr;
}

struct ExportedSchema {
    std::string format, name;
    std::vector<ArrowSchema> child_storage;
    std::vector<ArrowSchema*> child_ptrs;
};

void release_schema(ArrowSchema* s) {
    if (!s || !s->release) return;
    auto* priv = static_cast<ExportedSchema*>(s->private_data);
    for (auto& c : priv->child_storage)
        if (c.release) c.release(&c);
    delete priv;
    s->release = nullptr;
}

void export_schema(const Schema& schema, ArrowSchema* out) {
    auto* top = new ExportedSchema();
    top->format = "+s";
    top->name = "";
    top->child_storage.resize(schema.fields.size());
    for (size_t i = 0; i < schema.fields.size(); ++i) {
        auto* cp = new ExportedSchema();
        cp->format = format_of(schema.fields[i].binary ? (int)DT_BINARY : schema.fields[i].large ? (int)DT_LARGE_UTF8 : schema.fields[i].dtype);
        cp->name = schema.fields[i].name;
        ArrowSchema& c = top->child_storage[i];
        memset(&c, 0, sizeof(c));
        c.format = cp->format.c_str();
        c.name = cp->name.c_str();
        c.flags = schema.fields[i].nullable ? ARROW_FLAG_NULLABLE : 0;
        c.release = release_schema;
        c.private_data = cp;
        top->child_ptrs.push_back(&c);
    }
    memset(out, 0, sizeof(*out));
    out->format = top->format.c_str();
    out->name = top->name.c_str();
    out->n_children = (int64_t)schema.fields.size();
    out->children = top->child_ptrs.data();
    out->release = release_schema;
    out->private_data = top;
}

// small batches (stage outputs of aggregates: a few rows x a dozen columns): every buffer is packed into one
// device block by ONE kernel and comes back in ONE copy, instead of one blocking copy per buffer
static bool plan_small_export(const Batch& b, PackDesc& d, size_t& total) {
    const int64_t n = b.n_rows;
    d.n = 0;
    total = 0;
    auto add = [&](const void* dev, size_t bytes) -> bool {
        if (d.n >= PACK_MAX || bytes > (1u << 20)) return false;
        d.src[d.n] = dev; d.bytes[d.n] = (uint32_t)bytes; d.dst[d.n] = (uint32_t)total;
        ++d.n;
        total += (bytes + 63) & ~(size_t)63;
        return true;
    };
    for (const Column& c : b.cols) {
        if (c.validity && !add(c.validity->ptr(), (size_t)((n + 7) / 8))) return false;
        if (c.dtype == DT_UTF8) {
            if (!add(c.offsets->ptr(), (size_t)(n + 1) * 4) || !add(c.data->ptr(), (size_t)c.data_bytes)) return false;
        } else if (c.dtype == DT_BOOLEAN) {
            if (!add(c.data->ptr(), (size_t)((n + 7) / 8))) return false;
        } else if (!add(c.data->ptr(), (size_t)n * dtype_width(c.dtype))) return false;
    }
    return total <= (4u << 20) && d.n > 0;
}

void export_batch(const Batch& b, ArrowArray* out) {
    b.ctx->set_device();
    HIP_CHECK(hipDeviceSynchronize());
    auto* top = new ExportedArray();
    std::unique_ptr<ExportedArray> guard(top);
    top->child_storage.resize(b.cols.size());
    for (auto& c : top->child_storage) memset(&c, 0, sizeof(c));
    PackDesc pd;
    size_t packed_total = 0;
    uint8_t* packed = nullptr;
    int packed_next = 0;
    if (plan_small_export(b, pd, packed_total)) {
        packed = static_cast<uint8_t*>(malloc(packed_total + 64));
        if (!packed) fail(BHIP_EOOM, "host allocation failed");
        top->owned.push_back(packed);
        void* dev_block = b.ctx->alloc(packed_total + 64, nullptr);
        hipError_t e = launch_pack_buffers(LaunchCfg{b.ctx->cus(), nullptr}, pd, static_cast<uint8_t*>(dev_block));
        if (e == hipSuccess) e = hipMemcpy(packed, dev_block, packed_total, hipMemcpyDeviceToHost);
        b.ctx->free(dev_block, nullptr);
        HIP_CHECK(e);
    }
    for (size_t i = 0; i < b.cols.size(); ++i) {
        const Column& c = b.cols[i];
        auto* cp = new ExportedArray();
        ArrowArray& a = top->child_storage[i];
        a.private_data = cp;
        a.release = release_array;
        auto host_copy = [&](const void* dev, size_t bytes) -> void* {
            if (packed) return packed + pd.dst[packed_next++];       // same order as plan_small_export
            void* h = malloc(bytes ? bytes : 8);
            if (!h) fail(BHIP_EOOM, "host allocation failed");
            cp->owned.push_back(h);
            if (bytes) HIP_CHECK(hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost));
            return h;
        };
        const int64_t n = b.n_rows;
        const void* validity = c.validity ? host_copy(c.validity->ptr(), (size_t)((n + 7) / 8)) : nullptr;
        cp->buffers.push_back(validity);
        if (c.dtype == DT_UTF8) {
            const void* off32 = host_copy(c.offsets->ptr(), (size_t)(n + 1) * 4);
            if (i < b.schema->fields.size() && b.schema->fields[i].large) {              // LargeUtf8: the same offsets as int64
                auto* wide = static_cast<int64_t*>(malloc(((size_t)n + 1) * 8));
                if (!wide) fail(BHIP_EOOM, "host allocation failed");
                cp->owned.push_back(wide);
                for (int64_t r = 0; r <= n; ++r) wide[r] = static_cast<const int32_t*>(off32)[r];
                cp->buffers.push_back(wide);
            } else {
                cp->buffers.push_back(off32);
            }
            cp->buffers.push_back(host_copy(c.data->ptr(), (size_t)c.data_bytes));
        } else if (c.dtype == DT_BOOLEAN) {
            cp->buffers.push_back(host_copy(c.data->ptr(), (size_t)((n + 7) / 8)));
        } else {
            cp->buffers.push_back(host_copy(c.data->ptr(), (size_t)n * dtype_width(c.dtype)));
        }
        a.length = n;
        a.null_count = c.validity ? -1 : 0;
        a.offset = 0;
        a.n_buffers = (int64_t)cp->buffers.size();
        a.buffers = cp->buffers.data();
        top->child_ptrs.push_back(&a);
    }
    top->buffers.push_back(nullptr);
    memset(out, 0, sizeof(*out));
    out->length = b.n_rows;
    out->null_count = 0;
    out->n_buffers = 1;
    out->buffers = top->buffers.data();
    out->n_children = (int64_t)b.cols.size();
    out->children = top->child_ptrs.data();
    out->release = release_array;
    out->private_data = guard.release();
}

// ---- C stream ------------------------------------------------------------------------------------------
struct StreamPriv {
    bhip_stream* s;
    std::string last_error;
};

int stream_get_schema(ArrowArrayStream* st, ArrowSchema* out) {
    auto* p = static_cast<StreamPriv*>(st->private_data);
    try {
        export_schema(*p->s->s->schema(), out);
        return 0;
    } catch (const std::exception& e) { p->last_error = e.what(); return 5 /* EIO */; }
}

int stream_get_next(ArrowArrayStream* st, ArrowArray* out) {
    auto* p = static_cast<StreamPriv*>(st->private_data);
    try {
        p->s->ex.ctx->set_device();
        BatchPtr b = p->s->s->next();
        if (!b) { memset(out, 0, sizeof(*out)); return 0; }   // released array = end of stream
        HIP_CHECK(hipStreamSynchronize(p->s->ex.stream));
        export_batch(*b, out);
        return 0;
    } catch (const std::exception& e) { p->last_error = e.what(); return 5; }
}

const char* stream_last_error(ArrowArrayStream* st) { return static_cast<StreamPriv*>(st->private_data)->last_error.c_str(); }

void stream_release(ArrowArrayStream* st) {
    if (!st || !st->release) return;
    auto* p = static_cast<StreamPriv*>(st->private_data);
    delete p->s;
    delete p;
    st->release = nullptr;
}

}  // namespace

extern "C" {

bhip_status bhip_batch_import_arrow(bhip_ctx* ctx, struct ArrowArray* array, struct ArrowSchema* schema, bhip_batch** out) {
    try {
        if (!ctx || !array || !schema || !out) fail(BHIP_EINVAL, "null argument");
        if (!schema->format || strcmp(schema->format, "+s") != 0) fail(BHIP_EINVAL, "expected a struct array (RecordBatch)");
        if (array->n_children != schema->n_children) fail(BHIP_EINVAL, "array / schema children mismatch");
        const int n_cols = (int)array->n_children;
        const int64_t n_rows = array->length;
        std::vector<bhip_column_desc> descs(n_cols);
        std::vector<std::vector<uint8_t>> bit_storage;
        std::vector<std::vector<int32_t>> off_storage;
        std::vector<bool> large_cols, binary_cols;
        for (int i = 0; i < n_cols; ++i) {
            const ArrowSchema* cs = schema->children[i];
            const ArrowArray* ca = array->children[i];
            const int dt = device_dtype_from_format(cs->format);
            if (!dt) fail(BHIP_ENOTIMPL, std::string("unsupported Arrow type '") + cs->format + "' for column " + (cs->name ? cs->name : ""));
            if (ca->dictionary) fail(BHIP_ENOTIMPL, "dictionary arrays are not supported");
            // the array must have the buffers its declared format implies (a producer whose batches do not match
            // the stream's schema would otherwise be read out of bounds)
            const bool large = dt == DT_LARGE_UTF8, binary = dt == DT_BINARY;
            const int64_t need_buffers = (dt == DT_UTF8 || large || binary) ? 3 : 2;
            if (ca->n_buffers < need_buffers || ca->length != n_rows)
                fail(BHIP_EINVAL, std::string("Arrow array of column ") + (cs->name ? cs->name : "") + " does not match its schema");
            const int64_t off = ca->offset + array->offset;
            bhip_column_desc& d = descs[i];
            memset(&d, 0, sizeof(d));
            d.name = cs->name ? cs->name : "";
            d.dtype = (large || binary) ? (int)DT_UTF8 : dt;
            d.nullable = (cs->flags & ARROW_FLAG_NULLABLE) ? 1 : 0;
            large_cols.push_back(large);
            binary_cols.push_back(binary);
            const uint8_t* validity = ca->n_buffers > 0 ? static_cast<const uint8_t*>(ca->buffers[0]) : nullptr;
            if (validity && ca->null_count != 0) {
                bit_storage.push_back(realign_bits(validity, off, n_rows));
                d.validity = bit_storage.back().data();
            }
            if (large) {
                // 64-bit offsets, rebased to 0 and narrowed: a batch's strings must fit the device column's int32 offsets
                const int64_t* offsets = static_cast<const int64_t*>(ca->buffers[1]) + off;
                off_storage.emplace_back((size_t)n_rows + 1);
                auto& o = off_storage.back();
                if (!ca->buffers[1] && n_rows > 0)
                    fail(BHIP_EINVAL, std::string("Arrow array of column ") + (cs->name ? cs->name : "") + " has no offsets buffer");
                const int64_t first = ca->buffers[1] ? offsets[0] : 0;
                if (ca->buffers[1] && offsets[n_rows] - first > 0x7FFFFFFFll)
                    fail(BHIP_ENOTIMPL, std::string("LargeUtf8 column ") + (cs->name ? cs->name : "") + " holds more than 2 GiB of value bytes in one batch");
                for (int64_t r = 0; r <= n_rows; ++r) o[(size_t)r] = ca->buffers[1] ? (int32_t)(offsets[r] - first) : 0;
                d.offsets = o.data();
                d.data = static_cast<const uint8_t*>(ca->buffers[2]) + first;
                d.data_bytes = o[(size_t)n_rows];
                if (!ca->buffers[2]) d.data = "";
            } else if (dt == DT_UTF8 || binary) {
                const int32_t* offsets = static_cast<const int32_t*>(ca->buffers[1]) + off;
                // rebase to 0 so only the referenced bytes are shipped
                off_storage.emplace_back((size_t)n_rows + 1);
                auto& o = off_storage.back();
                // the C Data spec lets a length-0 array come without an offsets buffer; any other array must have one
                if (!ca->buffers[1] && n_rows > 0)
                    fail(BHIP_EINVAL, std::string("Arrow array of column ") + (cs->name ? cs->name : "") + " has no offsets buffer");
                const int32_t first = ca->buffers[1] ? offsets[0] : 0;
                for (int64_t r = 0; r <= n_rows; ++r) o[(size_t)r] = ca->buffers[1] ? offsets[r] - first : 0;
                d.offsets = o.data();
                d.data = static_cast<const uint8_t*>(ca->buffers[2]) + first;
                d.data_bytes = o[(size_t)n_rows];
                if (!ca->buffers[2]) d.data = "";
            } else if (dt == DT_BOOLEAN) {
                bit_storage.push_back(realign_bits(static_cast<const uint8_t*>(ca->buffers[1]), off, n_rows));
                d.data = bit_storage.back().data();
            } else {
                d.data = static_cast<const uint8_t*>(ca->buffers[1]) + off * dtype_width(dt);
            }
        }
        BatchPtr b = batch_from_host(ctx->p, n_cols, descs.data(), n_rows, false);
        // schema nullability follows the Arrow field flag
        auto s = std::make_shared<Schema>(*b->schema);
        for (int i = 0; i < n_cols; ++i) {
            s->fields[i].nullable = descs[i].nullable || descs[i].validity;
            s->fields[i].large = large_cols[i];
            s->fields[i].binary = binary_cols[i];
        }
        auto nb = std::make_shared<Batch>(*b);
        nb->schema = s;
        auto h = new bhip_batch();
        h->p = nb;
        *out = h;
        if (array->release) array->release(array);
        return BHIP_OK;
    } catch (const bhip::Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return BHIP_EINVAL; }
}

}  // extern "C"

namespace {

// Leaf over host-side Arrow C streams (the C image of a DataFusion child operator's RecordBatchStream, one per output
// partition): a partition's stream is drained on its first execute, each batch imported to the device; later executes replay
// the device batches.
class ArrowStreamExec : public bhip::ExecutionPlan {
public:
    ArrowStreamExec(bhip::ContextPtr ctx, int n, ArrowArrayStream* const* streams, bhip::SchemaPtr schema) : schema_(std::move(schema)) {
        ctx_ = std::move(ctx);
        own_.p = ctx_;                  // a context handle of its own for the batch import
        for (int i = 0; i < n; ++i) {
            auto p = std::make_unique<Part>();
            p->stream = *streams[i];    // the streams are moved into the plan (C stream interface ownership)
            streams[i]->release = nullptr;
            parts_.push_back(std::move(p));
        }
    }
    ~ArrowStreamExec() override {
        for (auto& p : parts_)
            if (p->stream.release) p->stream.release(&p->stream);
    }
    const char* name() const override { return "ArrowStreamExec"; }
    bhip::SchemaPtr schema() const override { return schema_; }
    bhip::Partitioning output_partitioning() const override { return bhip::Partitioning{BHIP_PART_UNKNOWN, (int)parts_.size(), {}}; }
    std::vector<bhip::PlanPtr> children() const override { return {}; }
    bhip::PlanPtr with_new_children(const std::vector<bhip::PlanPtr>& c) const override {
        if (!c.empty()) fail(BHIP_EINVAL, "ArrowStreamExec has no children");
        return shared_from_this();
    }
    std::string describe() const override { return "ArrowStreamExec: partitions=" + std::to_string(parts_.size()); }
    bhip::StreamPtr execute(int partition, const bhip::Exec&) const override {
        if (partition < 0 || partition >= (int)parts_.size()) fail(BHIP_EINVAL, "ArrowStreamExec invalid partition " + std::to_string(partition));
        Part& P = *parts_[partition];
        std::lock_guard<std::mutex> g(P.mu);
        if (!P.drained) {
            for (;;) {
                ArrowArray arr;
                memset(&arr, 0, sizeof(arr));
                if (P.stream.get_next(&P.stream, &arr) != 0) {
                    const char* m = P.stream.get_last_error ? P.stream.get_last_error(&P.stream) : nullptr;
                    fail(BHIP_EEXEC, std::string("Arrow stream: ") + (m ? m : "get_next failed"));
                }
                if (!arr.release) break;                       // end of stream
                ArrowSchema sch;
                memset(&sch, 0, sizeof(sch));
                if (P.stream.get_schema(&P.stream, &sch) != 0) { arr.release(&arr); fail(BHIP_EEXEC, "Arrow stream: get_schema failed"); }
                bhip_batch* b = nullptr;
                const bhip_status st = bhip_batch_import_arrow(&own_, &arr, &sch, &b);
                if (sch.release) sch.release(&sch);
                if (st != BHIP_OK) { if (arr.release) arr.release(&arr); fail(st, bhip_last_error()); }
                P.batches.push_back(b->p);
                bhip_batch_release(b);
            }
            P.drained = true;
            P.stream.release(&P.stream);                       // the producer is done with: let it go now
            P.stream.release = nullptr;
        }
        return bhip::StreamPtr(new bhip::VecStream(schema_, P.batches));
    }
private:
    struct Part {
        ArrowArrayStream stream;
        std::mutex mu;
        bool drained = false;
        std::vector<bhip::BatchPtr> batches;
    };
    mutable bhip_ctx own_;
    bhip::SchemaPtr schema_;
    std::vector<std::unique_ptr<Part>> parts_;
};

}  // namespace

extern "C" {

bhip_status bhip_plan_arrow_streams(bhip_ctx* ctx, int32_t n_partitions, struct ArrowArrayStream* const* streams, bhip_plan** out) {
    try {
        if (!ctx || !streams || !out || n_partitions < 1) fail(BHIP_EINVAL, "null argument");
        for (int i = 0; i < n_partitions; ++i)
            if (!streams[i] || !streams[i]->get_schema || !streams[i]->release) fail(BHIP_EINVAL, "released or null Arrow stream");
        ArrowArrayStream* stream = streams[0];
        ArrowSchema sch;
        memset(&sch, 0, sizeof(sch));
        if (stream->get_schema(stream, &sch) != 0) fail(BHIP_EEXEC, "Arrow stream: get_schema failed");
        auto schema = std::make_shared<bhip::Schema>();
        std::string err;
        if (!sch.format || strcmp(sch.format, "+s") != 0) err = "expected a struct schema (RecordBatch stream)";
        for (int64_t i = 0; err.empty() && i < sch.n_children; ++i) {
            const ArrowSchema* cs = sch.children[i];
            const int dt = device_dtype_from_format(cs->format);
            if (!dt) err = std::string("unsupported Arrow type '") + cs->format + "' for column " + (cs->name ? cs->name : "");
            else schema->fields.push_back(bhip::Field{cs->name ? cs->name : "", (dt == DT_LARGE_UTF8 || dt == DT_BINARY) ? (int)DT_UTF8 : dt, (cs->flags & ARROW_FLAG_NULLABLE) != 0, dt == DT_LARGE_UTF8, dt == DT_BINARY});
        }
        if (sch.release) sch.release(&sch);
        if (!err.empty()) fail(strncmp(err.c_str(), "unsupported", 11) == 0 ? BHIP_ENOTIMPL : BHIP_EINVAL, err);
        auto* h = new bhip_plan();
        h->p = std::make_shared<ArrowStreamExec>(ctx->p, (int)n_partitions, streams, schema);
        *out = h;
        return BHIP_OK;
    } catch (const bhip::Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return BHIP_EINVAL; }
}

bhip_status bhip_plan_arrow_stream(bhip_ctx* ctx, struct ArrowArrayStream* stream, bhip_plan** out) {
    return bhip_plan_arrow_streams(ctx, 1, &stream, out);
}

bhip_status bhip_batch_export_arrow(bhip_batch* batch, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    try {
        if (!batch || !out_array || !out_schema) fail(BHIP_EINVAL, "null argument");
        export_schema(*batch->p->schema, out_schema);
        try { export_batch(*batch->p, out_array); } catch (...) { release_schema(out_schema); throw; }
        return BHIP_OK;
    } catch (const bhip::Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return BHIP_EINVAL; }
}

bhip_status bhip_stream_export_arrow(bhip_stream* stream, struct ArrowArrayStream* out) {
    if (!stream || !out) { set_last_error("null argument"); return BHIP_EINVAL; }
    auto* p = new StreamPriv{stream, ""};
    out->get_schema = stream_get_schema;
    out->get_next = stream_get_next;
    out->get_last_error = stream_last_error;
    out->release = stream_release;
    out->private_data = p;
    return BHIP_OK;
}

}  // extern "C"
