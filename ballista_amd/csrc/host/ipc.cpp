// ipc.cpp — Arrow IPC FILE format, written and read by the library itself: the two sides of a Ballista stage boundary.
//
//   write   `utils::write_stream_to_disk` (rust/core/src/utils.rs:49-84): drain a RecordBatchStream into
//           `<work_dir>/<job>/<stage>/<partition>/data.arrow` with arrow's `FileWriter`, count rows / batches / bytes
//           (PartitionStats, rust/core/src/serde/scheduler/mod.rs:96-190) — `bhip_stream_write_ipc`;
//   read    the file `ShuffleReaderExec` / `do_get(FetchPartition)` serve (rust/executor/src/flight_service.rs:193-228) —
//           `bhip_plan_ipc_files`, a leaf with one partition per file.
//
// The format work (Flatbuffers metadata, 8-byte aligned bodies, footer) is host-only and operates on Arrow C Data Interface
// structs (`bhip_ipc_write_file` / `bhip_ipc_open_file`), so it is testable without a GPU against pyarrow in both directions;
// the device entry points are the existing export / import around it.  Metadata version V5, no compression, no dictionaries
// (what arrow-rs 4.0's FileWriter produces for the types of this path).  Flatbuffers schema: arrow/format/{Schema,Message,File}.fbs.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>

#include "plan.hpp"

namespace bhip {

int dtype_from_format(const char* fmt);       // arrow_c.cpp
const char* format_of_dtype(int dt);

namespace {

// ---- Flatbuffers: a minimal back-to-front builder and a reader --------------------------------------------------------
class Fb {
public:
    Fb() : buf_(1024), head_(1024) {}
    size_t size() const { return buf_.size() - head_; }
    const uint8_t* data() const { return buf_.data() + head_; }

    void align(size_t a) { while ((size() % a) != 0) push_byte(0); }
    // make room so that after writing `len` more bytes the cursor is `a`-aligned
    void prep(size_t a, size_t len) {
        if (a > max_align_) max_align_ = a;
        while (((size() + len) % a) != 0) push_byte(0);
    }
    template <class T> void push(T v) {
        grow(sizeof(T));
        head_ -= sizeof(T);
        memcpy(&buf_[head_], &v, sizeof(T));
    }
    uint32_t string(const std::string& s) {
        prep(4, s.size() + 1);
        push_byte(0);
        grow(s.size());
        head_ -= s.size();
        memcpy(&buf_[head_], s.data(), s.size());
        push<uint32_t>((uint32_t)s.size());
        return (uint32_t)size();
    }
    uint32_t offsets_vector(const std::vector<uint32_t>& offs) {
        prep(4, offs.size() * 4);
        for (size_t i = offs.size(); i-- > 0;) push<uint32_t>((uint32_t)(size() - offs[i] + 4));
        push<uint32_t>((uint32_t)offs.size());
        return (uint32_t)size();
    }
    // vector of 16-byte structs of two int64 (FieldNode, Buffer)
    uint32_t pairs_vector(const std::vector<std::pair<int64_t, int64_t>>& v) {
        prep(8, v.size() * 16);
        prep(4, v.size() * 16);
        for (size_t i = v.size(); i-- > 0;) { push<int64_t>(v[i].second); push<int64_t>(v[i].first); }
        push<uint32_t>((uint32_t)v.size());
        return (uint32_t)size();
    }
    // vector of Block{offset: long, metaDataLength: int, (pad), bodyLength: long}
    struct Block { int64_t offset; int32_t meta; int64_t body; };
    uint32_t blocks_vector(const std::vector<Block>& v) {
        prep(8, v.size() * 24);
        prep(4, v.size() * 24);
        for (size_t i = v.size(); i-- > 0;) { push<int64_t>(v[i].body); push<int32_t>(0); push<int32_t>(v[i].meta); push<int64_t>(v[i].offset); }
        push<uint32_t>((uint32_t)v.size());
        return (uint32_t)size();
    }

    void start_table(int n_fields) { slots_.assign(n_fields, 0); object_start_ = size(); }
    template <class T> void add(int slot, T v, T def) {
        if (v == def) return;
        prep(sizeof(T), 0);
        push<T>(v);
        slots_[slot] = (uint32_t)size();
    }
    template <class T> void add_always(int slot, T v) {
        prep(sizeof(T), 0);
        push<T>(v);
        slots_[slot] = (uint32_t)size();
    }
    void add_offset(int slot, uint32_t target) {
        if (!target) return;
        prep(4, 0);
        push<uint32_t>((uint32_t)(size() - target + 4));
        slots_[slot] = (uint32_t)size();
    }
    uint32_t end_table() {
        prep(4, 0);
        push<int32_t>(0);                                   // soffset to the vtable, patched below
        const uint32_t table = (uint32_t)size();
        int n = (int)slots_.size();
        while (n > 0 && slots_[n - 1] == 0) --n;            // trailing absent fields are trimmed
        for (int i = n; i-- > 0;) push<uint16_t>(slots_[i] ? (uint16_t)(table - slots_[i]) : 0);
        push<uint16_t>((uint16_t)(table - object_start_));
        push<uint16_t>((uint16_t)((n + 2) * 2));
        const uint32_t vt = (uint32_t)size();
        const int32_t soff = (int32_t)(vt - table);
        memcpy(&buf_[buf_.size() - table], &soff, 4);
        return table;
    }
    void finish(uint32_t root) {
        prep(max_align_ > 8 ? max_align_ : 8, 4);
        push<uint32_t>((uint32_t)(size() - root + 4));
    }

private:
    void push_byte(uint8_t b) { grow(1); buf_[--head_] = b; }
    void grow(size_t n) {
        if (head_ >= n) return;
        const size_t old = buf_.size(), used = size();
        size_t cap = old * 2;
        while (cap - used < n) cap *= 2;
        std::vector<uint8_t> nb(cap);
        memcpy(&nb[cap - used], &buf_[head_], used);
        buf_.swap(nb);
        head_ = cap - used;
    }
    std::vector<uint8_t> buf_;
    size_t head_;
    size_t max_align_ = 4;
    std::vector<uint32_t> slots_;
    size_t object_start_ = 0;
};

// reader: a table is a position in a byte range
struct FbTable {
    const uint8_t* base = nullptr;      // start of the flatbuffer
    size_t len = 0;
    size_t pos = 0;                     // position of the table
    bool ok() const { return base != nullptr; }
    template <class T> T rd(size_t at) const {
        if (at > len || sizeof(T) > len - at) fail(BHIP_EEXEC, "Arrow IPC: metadata runs past its buffer");    // (no `at + size`: it could wrap)
        T v;
        memcpy(&v, base + at, sizeof(T));
        return v;
    }
    size_t field(int slot) const {      // absolute position of the field, 0 = absent
        const int32_t soff = rd<int32_t>(pos);
        const int64_t vt_s = (int64_t)pos - (int64_t)soff;
        if (vt_s < 0 || (uint64_t)vt_s >= len) fail(BHIP_EEXEC, "Arrow IPC: vtable outside the metadata");
        const size_t vt = (size_t)vt_s;
        const uint16_t vsize = rd<uint16_t>(vt);
        const size_t entry = 4 + 2 * (size_t)slot;
        if (entry + 2 > vsize) return 0;
        const uint16_t off = rd<uint16_t>(vt + entry);
        return off ? pos + off : 0;
    }
    template <class T> T scalar(int slot, T def) const { const size_t f = field(slot); return f ? rd<T>(f) : def; }
    FbTable table(int slot) const {
        const size_t f = field(slot);
        if (!f) return FbTable{};
        return FbTable{base, len, f + rd<uint32_t>(f)};
    }
    std::string str(int slot) const {
        const size_t f = field(slot);
        if (!f) return std::string();
        const size_t s = f + rd<uint32_t>(f);
        const uint32_t n = rd<uint32_t>(s);
        if (s > len || len - s < 4 || n > len - s - 4) fail(BHIP_EEXEC, "Arrow IPC: string runs past the metadata");
        return std::string((const char*)base + s + 4, n);
    }
    // vector: position of element 0 and the count
    bool vec(int slot, size_t& first, uint32_t& n) const {
        const size_t f = field(slot);
        if (!f) { first = 0; n = 0; return false; }
        const size_t v = f + rd<uint32_t>(f);
        n = rd<uint32_t>(v);
        first = v + 4;
        if (n > len) fail(BHIP_EEXEC, "Arrow IPC: vector longer than the metadata that holds it");
        return true;
    }
    FbTable vec_table(size_t first, uint32_t i) const {
        const size_t e = first + 4 * (size_t)i;
        return FbTable{base, len, e + rd<uint32_t>(e)};
    }
};

FbTable fb_root(const uint8_t* p, size_t len) {
    FbTable t{p, len, 0};
    t.pos = t.rd<uint32_t>(0);
    if (t.pos > len || len - t.pos < 4) fail(BHIP_EEXEC, "Arrow IPC: root table outside the metadata");
    return t;
}

// ---- Arrow type <-> Flatbuffers Type union ---------------------------------------------------------------------------------
enum { T_Int = 2, T_FloatingPoint = 3, T_Binary = 4, T_Utf8 = 5, T_Bool = 6, T_Date = 8, T_Timestamp = 10, T_LargeUtf8 = 20 };

uint32_t write_type(Fb& fb, int dt, uint8_t& type_tag) {
    auto int_type = [&](int bits, bool is_signed) {
        fb.start_table(2);
        fb.add<int32_t>(0, bits, 0);
        fb.add<uint8_t>(1, is_signed ? 1 : 0, 0);
        type_tag = T_Int;
        return fb.end_table();
    };
    switch (dt) {
        case DT_INT8: return int_type(8, true);
        case DT_INT16: return int_type(16, true);
        case DT_INT32: return int_type(32, true);
        case DT_INT64: return int_type(64, true);
        case DT_UINT8: return int_type(8, false);
        case DT_UINT16: return int_type(16, false);
        case DT_UINT32: return int_type(32, false);
        case DT_UINT64: return int_type(64, false);
        case DT_FLOAT32:
        case DT_FLOAT64:
            fb.start_table(1);
            fb.add<int16_t>(0, dt == DT_FLOAT64 ? 2 : 1, 0);      // Precision: HALF, SINGLE, DOUBLE
            type_tag = T_FloatingPoint;
            return fb.end_table();
        case DT_UTF8: fb.start_table(0); type_tag = T_Utf8; return fb.end_table();
        case DT_LARGE_UTF8: fb.start_table(0); type_tag = T_LargeUtf8; return fb.end_table();
        case DT_BINARY: fb.start_table(0); type_tag = T_Binary; return fb.end_table();
        case DT_BOOLEAN: fb.start_table(0); type_tag = T_Bool; return fb.end_table();
        case DT_DATE32:
        case DT_DATE64:
            fb.start_table(1);
            fb.add<int16_t>(0, dt == DT_DATE32 ? 0 : 1, 1);      // DateUnit: DAY = 0, MILLISECOND = 1 (default)
            type_tag = T_Date;
            return fb.end_table();
        case DT_TIMESTAMP_S:
        case DT_TIMESTAMP_MS:
        case DT_TIMESTAMP_US:
        case DT_TIMESTAMP_NS:
            fb.start_table(2);
            fb.add<int16_t>(0, (int16_t)(dt - DT_TIMESTAMP_S), 0);
            type_tag = T_Timestamp;
            return fb.end_table();
        default: fail(BHIP_ENOTIMPL, std::string("Arrow IPC: cannot write type ") + dtype_name(dt));
    }
}

int read_type(uint8_t tag, const FbTable& t, const std::string& field_name) {
    switch (tag) {
        case T_Int: {
            const int bits = t.scalar<int32_t>(0, 0);
            const bool sg = t.scalar<uint8_t>(1, 0) != 0;
            switch (bits) {
                case 8: return sg ? DT_INT8 : DT_UINT8;
                case 16: return sg ? DT_INT16 : DT_UINT16;
                case 32: return sg ? DT_INT32 : DT_UINT32;
                case 64: return sg ? DT_INT64 : DT_UINT64;
            }
            break;
        }
        case T_FloatingPoint: {
            const int p = t.scalar<int16_t>(0, 0);
            if (p == 2) return DT_FLOAT64;
            if (p == 1) return DT_FLOAT32;
            break;
        }
        case T_Utf8: return DT_UTF8;
        case T_LargeUtf8: return DT_LARGE_UTF8;
        case T_Binary: return DT_BINARY;
        case T_Bool: return DT_BOOLEAN;
        case T_Date: return t.scalar<int16_t>(0, 1) == 0 ? DT_DATE32 : DT_DATE64;
        case T_Timestamp:
            if (t.str(1).empty()) return DT_TIMESTAMP_S + t.scalar<int16_t>(0, 0);
            break;
        default: break;
    }
    fail(BHIP_ENOTIMPL, "Arrow IPC: column '" + field_name + "' has a type outside the GPU path (type id " + std::to_string(tag) + ")");
}

uint32_t write_schema(Fb& fb, const Schema& s) {
    std::vector<uint32_t> fields;
    for (auto& f : s.fields) {
        const uint32_t name = fb.string(f.name);
        uint8_t tag = 0;
        const uint32_t type = write_type(fb, f.binary ? (int)DT_BINARY : f.large ? (int)DT_LARGE_UTF8 : f.dtype, tag);
        const uint32_t children = fb.offsets_vector({});
        fb.start_table(7);                                     // Field: name, nullable, type_type, type, dictionary, children, custom_metadata
        fb.add_offset(0, name);
        fb.add<uint8_t>(1, f.nullable ? 1 : 0, 0);
        fb.add<uint8_t>(2, tag, 0);
        fb.add_offset(3, type);
        fb.add_offset(5, children);
        fields.push_back(fb.end_table());
    }
    const uint32_t fv = fb.offsets_vector(fields);
    fb.start_table(4);                                         // Schema: endianness, fields, custom_metadata, features
    fb.add_offset(1, fv);
    return fb.end_table();
}

SchemaPtr read_schema(const FbTable& st) {
    auto s = std::make_shared<Schema>();
    if (st.scalar<int16_t>(0, 0) != 0) fail(BHIP_ENOTIMPL, "Arrow IPC: big-endian file");
    size_t first;
    uint32_t n;
    st.vec(1, first, n);
    for (uint32_t i = 0; i < n; ++i) {
        const FbTable f = st.vec_table(first, i);
        Field fld;
        fld.name = f.str(0);
        fld.nullable = f.scalar<uint8_t>(1, 0) != 0;
        if (f.field(4)) fail(BHIP_ENOTIMPL, "Arrow IPC: dictionary-encoded column '" + fld.name + "'");
        fld.dtype = read_type(f.scalar<uint8_t>(2, 0), f.table(3), fld.name);
        if (fld.dtype == DT_LARGE_UTF8) { fld.dtype = DT_UTF8; fld.large = true; }
        if (fld.dtype == DT_BINARY) { fld.dtype = DT_UTF8; fld.binary = true; }
        s->fields.push_back(fld);
    }
    return s;
}

enum { MSG_SCHEMA = 1, MSG_DICTIONARY = 2, MSG_RECORD_BATCH = 3 };
constexpr int16_t METADATA_V5 = 4;

std::vector<uint8_t> message_bytes(Fb& fb, uint8_t header_type, uint32_t header, int64_t body_len) {
    fb.start_table(5);                                         // Message: version, header_type, header, bodyLength, custom_metadata
    fb.add<int16_t>(0, METADATA_V5, 0);
    fb.add<uint8_t>(1, header_type, 0);
    fb.add_offset(2, header);
    fb.add<int64_t>(3, body_len, 0);
    fb.finish(fb.end_table());
    // encapsulated message: continuation marker, metadata length (padded so the body starts 8-byte aligned), metadata
    const size_t meta = fb.size();
    const size_t padded = (meta + 8 + 7) / 8 * 8 - 8;
    std::vector<uint8_t> out(8 + padded, 0);
    const uint32_t cont = 0xFFFFFFFFu;
    const int32_t len = (int32_t)padded;
    memcpy(&out[0], &cont, 4);
    memcpy(&out[4], &len, 4);
    memcpy(&out[8], fb.data(), meta);
    return out;
}

inline int64_t pad8(int64_t n) { return (n + 7) / 8 * 8; }

// ---- host columns (Arrow C Data Interface) ----------------------------------------------------------------------------------
bool bit_get(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

// bits [off, off + n) of `src` as a bitmap starting at bit 0
std::vector<uint8_t> slice_bits(const uint8_t* src, int64_t off, int64_t n) {
    std::vector<uint8_t> out((size_t)pad8((n + 7) / 8), 0);
    if ((off & 7) == 0) {
        memcpy(out.data(), src + (off >> 3), (size_t)((n + 7) / 8));
        if (n & 7) out[(size_t)(n >> 3)] &= (uint8_t)((1u << (n & 7)) - 1);
    } else {
        for (int64_t i = 0; i < n; ++i)
            if (bit_get(src, off + i)) out[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    }
    return out;
}

struct BodyPiece { const void* ptr; int64_t bytes; std::vector<uint8_t> owned; };

// the body buffers of one column in IPC order (validity, [offsets,] data), re-based to array offset 0
void column_pieces(const ArrowArray& a, int dt, std::vector<BodyPiece>& out, int64_t& null_count) {
    const int64_t n = a.length, off = a.offset;
    const uint8_t* validity = a.n_buffers > 0 ? static_cast<const uint8_t*>(a.buffers[0]) : nullptr;
    null_count = 0;
    if (validity && a.null_count != 0) {
        BodyPiece p{nullptr, 0, slice_bits(validity, off, n)};
        for (int64_t i = 0; i < n; ++i) null_count += !bit_get(p.owned.data(), i);
        if (null_count) { p.bytes = (n + 7) / 8; out.push_back(std::move(p)); }
        else out.push_back(BodyPiece{nullptr, 0, {}});
    } else {
        out.push_back(BodyPiece{nullptr, 0, {}});                  // V5: an absent validity buffer has length 0
    }
    if (dt == DT_LARGE_UTF8) {
        const int64_t* o = static_cast<const int64_t*>(a.buffers[1]) + off;
        const char* d = static_cast<const char*>(a.buffers[2]);
        BodyPiece po{nullptr, (n + 1) * 8, {}};
        const int64_t base = a.buffers[1] ? o[0] : 0;
        if (!a.buffers[1]) po.owned.assign(16, 0);
        else if (base == 0) po.ptr = o;
        else {
            po.owned.resize((size_t)(n + 1) * 8);
            int64_t* r = reinterpret_cast<int64_t*>(po.owned.data());
            for (int64_t i = 0; i <= n; ++i) r[i] = o[i] - base;
        }
        out.push_back(std::move(po));
        out.push_back(BodyPiece{d ? d + base : nullptr, (n && a.buffers[1]) ? (int64_t)(o[n] - base) : 0, {}});
    } else if (dt == DT_UTF8) {
        const int32_t* o = static_cast<const int32_t*>(a.buffers[1]) + off;
        const char* d = static_cast<const char*>(a.buffers[2]);
        BodyPiece po{nullptr, (n + 1) * 4, {}};
        const int32_t base = a.buffers[1] ? o[0] : 0;
        if (!a.buffers[1]) po.owned.assign(8, 0);                       // an empty array may come without an offsets buffer
        else if (base == 0) po.ptr = o;
        else {
            po.owned.resize((size_t)(n + 1) * 4);
            int32_t* r = reinterpret_cast<int32_t*>(po.owned.data());
            for (int64_t i = 0; i <= n; ++i) r[i] = o[i] - base;
        }
        out.push_back(std::move(po));
        out.push_back(BodyPiece{d ? d + base : nullptr, (n && a.buffers[1]) ? (int64_t)(o[n] - base) : 0, {}});
    } else if (dt == DT_BOOLEAN) {
        BodyPiece p{nullptr, (n + 7) / 8, slice_bits(static_cast<const uint8_t*>(a.buffers[1]), off, n)};
        out.push_back(std::move(p));
    } else {
        const int w = dtype_width(dt);
        out.push_back(BodyPiece{static_cast<const uint8_t*>(a.buffers[1]) + off * w, n * w, {}});
    }
}

int64_t write_all(FILE* f, const void* p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) fail(BHIP_EEXEC, "Ballista Error: write to the IPC file failed");
    return (int64_t)n;
}

SchemaPtr schema_of_c(const ArrowSchema& sch) {
    auto s = std::make_shared<Schema>();
    if (!sch.format || strcmp(sch.format, "+s") != 0) fail(BHIP_EINVAL, "Arrow IPC: expected a struct schema (RecordBatch stream)");
    for (int64_t i = 0; i < sch.n_children; ++i) {
        const ArrowSchema* c = sch.children[i];
        const int dt = dtype_from_format(c->format);
        if (!dt) fail(BHIP_ENOTIMPL, std::string("Arrow IPC: unsupported Arrow type '") + c->format + "' for column " + (c->name ? c->name : ""));
        s->fields.push_back(Field{c->name ? c->name : "", (dt == DT_LARGE_UTF8 || dt == DT_BINARY) ? (int)DT_UTF8 : dt, (c->flags & ARROW_FLAG_NULLABLE) != 0, dt == DT_LARGE_UTF8, dt == DT_BINARY});
    }
    return s;
}

}  // namespace

// FileWriter: schema message, one message per batch, end-of-stream marker, footer.  Returns PartitionStats.
void ipc_write_file(ArrowArrayStream* stream, const std::string& path, uint64_t* num_rows, uint64_t* num_batches, uint64_t* num_bytes) {
    ArrowSchema csch;
    memset(&csch, 0, sizeof(csch));
    if (stream->get_schema(stream, &csch) != 0) fail(BHIP_EEXEC, "Arrow stream: get_schema failed");
    SchemaPtr schema;
    try { schema = schema_of_c(csch); } catch (...) { if (csch.release) csch.release(&csch); throw; }
    if (csch.release) csch.release(&csch);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) fail(BHIP_EEXEC, "Failed to create partition file at " + path + ": " + strerror(errno));       // utils.rs:53-58
    struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
    int64_t pos = 0;
    pos += write_all(f, "ARROW1\0\0", 8);
    {
        Fb fb;
        auto msg = message_bytes(fb, MSG_SCHEMA, write_schema(fb, *schema), 0);
        pos += write_all(f, msg.data(), msg.size());
    }
    std::vector<Fb::Block> blocks;
    uint64_t rows = 0, batches = 0, bytes = 0;
    static const uint8_t zeros[8] = {0};
    for (;;) {
        ArrowArray arr;
        memset(&arr, 0, sizeof(arr));
        if (stream->get_next(stream, &arr) != 0) {
            const char* m = stream->get_last_error ? stream->get_last_error(stream) : nullptr;
            fail(BHIP_EEXEC, std::string("Arrow stream: ") + (m ? m : "get_next failed"));
        }
        if (!arr.release) break;
        struct Rel { ArrowArray* a; ~Rel() { if (a->release) a->release(a); } } rel{&arr};
        if (arr.n_children != (int64_t)schema->fields.size()) fail(BHIP_EINVAL, "Arrow IPC: a batch does not match the stream's schema");
        std::vector<BodyPiece> pieces;
        std::vector<std::pair<int64_t, int64_t>> nodes, buffers;
        int64_t body = 0;
        for (int64_t c = 0; c < arr.n_children; ++c) {
            const ArrowArray& a = *arr.children[c];
            if (a.length != arr.length) fail(BHIP_EINVAL, "Arrow IPC: column length differs from the batch length");
            const size_t first = pieces.size();
            int64_t nulls = 0;
            column_pieces(a, schema->fields[c].large ? (int)DT_LARGE_UTF8 : schema->fields[c].dtype, pieces, nulls);
            nodes.push_back({a.length, nulls});
            for (size_t k = first; k < pieces.size(); ++k) {
                buffers.push_back({body, pieces[k].bytes});
                body += pad8(pieces[k].bytes);
                bytes += (uint64_t)pieces[k].bytes;
            }
        }
        Fb fb;
        const uint32_t bv = fb.pairs_vector(buffers);
        const uint32_t nv = fb.pairs_vector(nodes);
        fb.start_table(4);                                     // RecordBatch: length, nodes, buffers, compression
        fb.add<int64_t>(0, arr.length, 0);
        fb.add_offset(1, nv);
        fb.add_offset(2, bv);
        auto msg = message_bytes(fb, MSG_RECORD_BATCH, fb.end_table(), body);
        blocks.push_back(Fb::Block{pos, (int32_t)msg.size(), body});
        pos += write_all(f, msg.data(), msg.size());
        for (auto& p : pieces) {
            const void* src = p.owned.empty() ? p.ptr : p.owned.data();
            pos += write_all(f, src, (size_t)p.bytes);
            pos += write_all(f, zeros, (size_t)(pad8(p.bytes) - p.bytes));
        }
        rows += (uint64_t)arr.length;
        batches += 1;
    }
    const uint32_t eos[2] = {0xFFFFFFFFu, 0};
    pos += write_all(f, eos, 8);
    {
        Fb fb;
        const uint32_t rb = fb.blocks_vector(blocks);
        const uint32_t dict = fb.blocks_vector({});
        const uint32_t sch = write_schema(fb, *schema);
        fb.start_table(5);                                     // Footer: version, schema, dictionaries, recordBatches, custom_metadata
        fb.add<int16_t>(0, METADATA_V5, 0);
        fb.add_offset(1, sch);
        fb.add_offset(2, dict);
        fb.add_offset(3, rb);
        fb.finish(fb.end_table());
        write_all(f, fb.data(), fb.size());
        const int32_t flen = (int32_t)fb.size();
        write_all(f, &flen, 4);
        write_all(f, "ARROW1", 6);
    }
    if (fflush(f) != 0) fail(BHIP_EEXEC, "Ballista Error: write to the IPC file failed");
    if (num_rows) *num_rows = rows;
    if (num_batches) *num_batches = batches;
    if (num_bytes) *num_bytes = bytes;
}

// ---- FileReader as an Arrow C stream ---------------------------------------------------------------------------------------------
namespace {

struct IpcFile {
    std::vector<uint8_t> bytes;          // the whole file (stage outputs of this path are read once, front to back)
    SchemaPtr schema;
    std::vector<Fb::Block> blocks;
    size_t next = 0;
    std::string error;
    std::string path;
};

struct ExportedBatch {                   // owns what an exported ArrowArray points to
    std::shared_ptr<IpcFile> file;
    std::vector<ArrowArray> children;
    std::vector<ArrowArray*> child_ptrs;
    std::vector<std::vector<const void*>> buffers;
    std::vector<const void*> top_buffers;
};

void release_exported(ArrowArray* a) {
    delete static_cast<ExportedBatch*>(a->private_data);
    a->release = nullptr;
}
void release_child(ArrowArray* a) { a->release = nullptr; }

void load_file(IpcFile& F) {
    std::ifstream in(F.path, std::ios::binary | std::ios::ate);
    if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + F.path);
    const std::streamsize n = in.tellg();
    F.bytes.resize((size_t)n);
    in.seekg(0);
    if (n && !in.read(reinterpret_cast<char*>(F.bytes.data()), n)) fail(BHIP_EEXEC, "Ballista Error: cannot read " + F.path);
    const uint8_t* p = F.bytes.data();
    const size_t len = F.bytes.size();
    if (len < 18 || memcmp(p, "ARROW1", 6) != 0 || memcmp(p + len - 6, "ARROW1", 6) != 0)
        fail(BHIP_EEXEC, "Arrow IPC: " + F.path + " is not an Arrow file (magic missing)");
    int32_t flen;
    memcpy(&flen, p + len - 10, 4);
    if (flen <= 0 || (size_t)flen > len - 18) fail(BHIP_EEXEC, "Arrow IPC: corrupt footer length in " + F.path);
    const FbTable footer = fb_root(p + len - 10 - flen, (size_t)flen);
    const FbTable st = footer.table(1);
    if (!st.ok()) fail(BHIP_EEXEC, "Arrow IPC: footer without a schema");
    F.schema = read_schema(st);
    size_t first;
    uint32_t n_dict = 0, n_rb = 0;
    footer.vec(2, first, n_dict);
    if (n_dict) fail(BHIP_ENOTIMPL, "Arrow IPC: dictionary batches");
    if (footer.vec(3, first, n_rb))
        for (uint32_t i = 0; i < n_rb; ++i) {
            const size_t e = first + 24 * (size_t)i;
            F.blocks.push_back(Fb::Block{footer.rd<int64_t>(e), footer.rd<int32_t>(e + 8), footer.rd<int64_t>(e + 16)});
        }
}

void export_next(const std::shared_ptr<IpcFile>& Fp, ArrowArray* out) {
    IpcFile& F = *Fp;
    memset(out, 0, sizeof(*out));
    if (F.next >= F.blocks.size()) return;                      // end of stream: released array
    const Fb::Block blk = F.blocks[F.next++];
    const uint8_t* p = F.bytes.data();
    const size_t len = F.bytes.size();
    // the footer's Block fields come from the file: every comparison by subtraction, nothing that could wrap
    if (blk.offset < 8 || (uint64_t)blk.offset > len || blk.meta < 12 || (uint64_t)blk.meta > len - (size_t)blk.offset || blk.body < 0 ||
        (uint64_t)blk.body > len - (size_t)blk.offset - (size_t)blk.meta)
        fail(BHIP_EEXEC, "Arrow IPC: record batch block outside the file");
    if ((blk.offset & 7) || (blk.meta & 7)) fail(BHIP_EEXEC, "Arrow IPC: record batch block not 8-byte aligned");
    size_t m = (size_t)blk.offset;
    uint32_t first_word;
    memcpy(&first_word, p + m, 4);
    const size_t prefix = first_word == 0xFFFFFFFFu ? 8 : 4;    // V4 files have no continuation marker
    const FbTable msg = fb_root(p + m + prefix, (size_t)blk.meta - prefix);
    if (msg.scalar<uint8_t>(1, 0) != MSG_RECORD_BATCH) fail(BHIP_EEXEC, "Arrow IPC: block is not a record batch");
    const FbTable rb = msg.table(2);
    if (rb.field(3)) fail(BHIP_ENOTIMPL, "Arrow IPC: compressed record batch bodies");
    const int64_t n_rows = rb.scalar<int64_t>(0, 0);
    if (n_rows < 0 || n_rows > 0xFFFFFFF0ll) fail(BHIP_EEXEC, "Arrow IPC: record batch row count out of range");
    size_t nodes, bufs;
    uint32_t n_nodes, n_bufs;
    rb.vec(1, nodes, n_nodes);
    rb.vec(2, bufs, n_bufs);
    const uint8_t* body = p + m + (size_t)blk.meta;
    auto X = std::make_unique<ExportedBatch>();
    X->file = Fp;
    const size_t nc = F.schema->fields.size();
    if (n_nodes != nc) fail(BHIP_EEXEC, "Arrow IPC: record batch does not match the file's schema");
    X->children.resize(nc);
    X->buffers.resize(nc);
    uint32_t bi = 0;
    auto next_buf = [&](int64_t& blen) -> const void* {
        if (bi >= n_bufs) fail(BHIP_EEXEC, "Arrow IPC: record batch has too few buffers");
        const int64_t off = rb.rd<int64_t>(bufs + 16 * (size_t)bi), bl = rb.rd<int64_t>(bufs + 16 * (size_t)bi + 8);
        ++bi;
        if (off < 0 || bl < 0 || off > blk.body || bl > blk.body - off) fail(BHIP_EEXEC, "Arrow IPC: buffer outside the record batch body");
        if (off & 7) fail(BHIP_EEXEC, "Arrow IPC: buffer not 8-byte aligned in the record batch body");       // (the format requires it; the consumers read words)
        blen = bl;
        return bl ? body + off : nullptr;
    };
    for (size_t c = 0; c < nc; ++c) {
        ArrowArray& a = X->children[c];
        memset(&a, 0, sizeof(a));
        a.length = rb.rd<int64_t>(nodes + 16 * c);
        a.null_count = rb.rd<int64_t>(nodes + 16 * c + 8);
        if (a.length != n_rows) fail(BHIP_EEXEC, "Arrow IPC: field node length differs from the batch length");
        if (a.null_count < 0 || a.null_count > n_rows) fail(BHIP_EEXEC, "Arrow IPC: field node null count out of range");
        const int dt = F.schema->fields[c].large ? (int)DT_LARGE_UTF8 : F.schema->fields[c].dtype;
        int64_t bl;
        const void* validity = next_buf(bl);
        if (a.null_count > 0 && (!validity || bl < (n_rows + 7) / 8)) fail(BHIP_EEXEC, "Arrow IPC: validity buffer too short");
        X->buffers[c].push_back(a.null_count > 0 ? validity : nullptr);
        static const int32_t zero_offset[4] = {0, 0, 0, 0};
        if (dt == DT_LARGE_UTF8) {
            const void* o = next_buf(bl);
            if (n_rows > 0 && bl < (n_rows + 1) * 8) fail(BHIP_EEXEC, "Arrow IPC: offsets buffer too short");
            const int64_t* oi = o ? static_cast<const int64_t*>(o) : reinterpret_cast<const int64_t*>(zero_offset);
            int64_t dl;
            const void* d = next_buf(dl);
            if (n_rows > 0 && (oi[0] < 0 || oi[n_rows] < oi[0] || oi[n_rows] > dl)) fail(BHIP_EEXEC, "Arrow IPC: string offsets outside the data buffer");
            for (int64_t i = 0; i < n_rows; ++i)                  // interior offsets reach a device gather: they must ascend
                if (oi[i + 1] < oi[i]) fail(BHIP_EEXEC, "Arrow IPC: string offsets do not ascend");
            X->buffers[c].push_back(oi);
            X->buffers[c].push_back(d ? d : (const void*)zero_offset);
        } else if (dt == DT_UTF8) {
            const void* o = next_buf(bl);
            if (n_rows > 0 && bl < (n_rows + 1) * 4) fail(BHIP_EEXEC, "Arrow IPC: offsets buffer too short");
            const int32_t* oi = o ? static_cast<const int32_t*>(o) : zero_offset;
            int64_t dl;
            const void* d = next_buf(dl);
            if (n_rows > 0 && (oi[0] < 0 || oi[n_rows] < oi[0] || oi[n_rows] > dl)) fail(BHIP_EEXEC, "Arrow IPC: string offsets outside the data buffer");
            for (int64_t i = 0; i < n_rows; ++i)
                if (oi[i + 1] < oi[i]) fail(BHIP_EEXEC, "Arrow IPC: string offsets do not ascend");
            X->buffers[c].push_back(oi);
            X->buffers[c].push_back(d ? d : (const void*)zero_offset);
        } else {
            const void* d = next_buf(bl);
            const int64_t need = dt == DT_BOOLEAN ? (n_rows + 7) / 8 : n_rows * dtype_width(dt);
            if (bl < need) fail(BHIP_EEXEC, "Arrow IPC: data buffer too short");
            X->buffers[c].push_back(d ? d : (const void*)zero_offset);
        }
        a.n_buffers = (int64_t)X->buffers[c].size();
        a.buffers = X->buffers[c].data();
        a.release = release_child;
    }
    for (auto& c : X->children) X->child_ptrs.push_back(&c);
    X->top_buffers.push_back(nullptr);
    out->length = n_rows;
    out->null_count = 0;
    out->n_buffers = 1;
    out->buffers = X->top_buffers.data();
    out->n_children = (int64_t)nc;
    out->children = X->child_ptrs.data();
    out->release = release_exported;
    out->private_data = X.release();
}

// schema export (struct of the columns)
struct SchemaHold {
    std::vector<ArrowSchema> children;
    std::vector<ArrowSchema*> ptrs;
    std::vector<std::string> names;
};
void release_schema_child(ArrowSchema* s) { s->release = nullptr; }
void release_schema_top(ArrowSchema* s) {
    delete static_cast<SchemaHold*>(s->private_data);
    s->release = nullptr;
}
void export_schema_c(const Schema& s, ArrowSchema* out) {
    auto H = std::make_unique<SchemaHold>();
    H->children.resize(s.fields.size());
    for (auto& f : s.fields) H->names.push_back(f.name);
    for (size_t i = 0; i < s.fields.size(); ++i) {
        ArrowSchema& c = H->children[i];
        memset(&c, 0, sizeof(c));
        c.format = format_of_dtype(s.fields[i].binary ? (int)DT_BINARY : s.fields[i].large ? (int)DT_LARGE_UTF8 : s.fields[i].dtype);
        c.name = H->names[i].c_str();
        c.flags = s.fields[i].nullable ? ARROW_FLAG_NULLABLE : 0;
        c.release = release_schema_child;
        H->ptrs.push_back(&c);
    }
    memset(out, 0, sizeof(*out));
    out->format = "+s";
    out->name = "";
    out->n_children = (int64_t)s.fields.size();
    out->children = H->ptrs.data();
    out->release = release_schema_top;
    out->private_data = H.release();
}

struct StreamPriv { std::shared_ptr<IpcFile> file; };

int s_get_schema(ArrowArrayStream* s, ArrowSchema* out) {
    auto* P = static_cast<StreamPriv*>(s->private_data);
    try { export_schema_c(*P->file->schema, out); return 0; }
    catch (const std::exception& e) { P->file->error = e.what(); return 5; }
}
int s_get_next(ArrowArrayStream* s, ArrowArray* out) {
    auto* P = static_cast<StreamPriv*>(s->private_data);
    try { export_next(P->file, out); return 0; }
    catch (const std::exception& e) { P->file->error = e.what(); return 5; }
}
const char* s_last_error(ArrowArrayStream* s) { return static_cast<StreamPriv*>(s->private_data)->file->error.c_str(); }
void s_release(ArrowArrayStream* s) {
    delete static_cast<StreamPriv*>(s->private_data);
    s->release = nullptr;
}

}  // namespace

void ipc_open_file(const std::string& path, ArrowArrayStream* out) {
    auto F = std::make_shared<IpcFile>();
    F->path = path;
    load_file(*F);
    auto* P = new StreamPriv{F};
    out->get_schema = s_get_schema;
    out->get_next = s_get_next;
    out->get_last_error = s_last_error;
    out->release = s_release;
    out->private_data = P;
}

}  // namespace bhip

using namespace bhip;

#define BHIP_I_BEGIN try {
#define BHIP_I_END                                                                                     \
    return BHIP_OK;                                                                                    \
    }                                                                                                  \
    catch (const bhip::Error& e) { bhip::set_last_error(e.what()); return e.code; }                    \
    catch (const std::exception& e) { bhip::set_last_error(std::string("internal error: ") + e.what()); return BHIP_EINVAL; }

extern "C" {

bhip_status bhip_ipc_write_file(struct ArrowArrayStream* stream, const char* path, uint64_t* num_rows, uint64_t* num_batches, uint64_t* num_bytes) {
    BHIP_I_BEGIN
    if (!stream || !path || !stream->get_next) fail(BHIP_EINVAL, "null argument");
    ipc_write_file(stream, path, num_rows, num_batches, num_bytes);
    BHIP_I_END
}

bhip_status bhip_ipc_open_file(const char* path, struct ArrowArrayStream* out) {
    BHIP_I_BEGIN
    if (!path || !out) fail(BHIP_EINVAL, "null argument");
    ipc_open_file(path, out);
    BHIP_I_END
}

bhip_status bhip_stream_write_ipc(bhip_stream* stream, const char* path, uint64_t* num_rows, uint64_t* num_batches, uint64_t* num_bytes) {
    BHIP_I_BEGIN
    if (!stream) fail(BHIP_EINVAL, "null argument");
    if (!path) { bhip_stream_release(stream); fail(BHIP_EINVAL, "null argument: path"); }    // consumed on every path
    ArrowArrayStream cs;
    memset(&cs, 0, sizeof(cs));
    const bhip_status st = bhip_stream_export_arrow(stream, &cs);      // consumes the stream
    if (st != BHIP_OK) return st;
    struct Rel { ArrowArrayStream* s; ~Rel() { if (s->release) s->release(s); } } rel{&cs};
    ipc_write_file(&cs, path, num_rows, num_batches, num_bytes);
    BHIP_I_END
}

bhip_status bhip_plan_ipc_files(bhip_ctx* ctx, int32_t n_files, const char* const* paths, bhip_plan** out) {
    BHIP_I_BEGIN
    if (!ctx || !paths || !out || n_files < 1) fail(BHIP_EINVAL, "null argument");
    std::vector<ArrowArrayStream> streams((size_t)n_files);
    std::vector<ArrowArrayStream*> ptrs;
    for (auto& s : streams) { memset(&s, 0, sizeof(s)); ptrs.push_back(&s); }
    struct Rel { std::vector<ArrowArrayStream>& v; ~Rel() { for (auto& s : v) if (s.release) s.release(&s); } } rel{streams};
    for (int i = 0; i < n_files; ++i) {
        if (!paths[i]) fail(BHIP_EINVAL, "null argument: path");
        ipc_open_file(paths[i], &streams[i]);
    }
    const bhip_status st = bhip_plan_arrow_streams(ctx, n_files, ptrs.data(), out);
    if (st != BHIP_OK) return st;
    BHIP_I_END
}

}  // extern "C"
