// parquet_host.cpp — see parquet_host.hpp.  Format: the Parquet file layout the reference's ParquetExec reads
// (rust/core/src/serde/physical_plan/from_proto.rs:111-121; `--format parquet` of rust/benchmarks/tpch/src/main.rs:147-150).
#include "parquet_host.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>

namespace bhip {
namespace pq {

static HostAllocFn g_alloc = nullptr;
static HostFreeFn g_free = nullptr;
void set_host_allocator(HostAllocFn alloc, HostFreeFn free_fn) { g_alloc = alloc; g_free = free_fn; }
void* host_alloc(size_t bytes) {
    void* p = g_alloc ? g_alloc(bytes) : malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void host_free(void* p, size_t bytes) {
    if (g_free) g_free(p, bytes);
    else free(p);
}

namespace {

// ---- Thrift compact protocol ---------------------------------------------------------------------------------------------
struct Thrift {
    const uint8_t* p;
    const uint8_t* end;
    int depth = 0;
    static constexpr int MAX_DEPTH = 64;
    void need(size_t n) const {
        if ((size_t)(end - p) < n) fail(BHIP_EEXEC, "Parquet: truncated metadata");
    }
    uint64_t varint() {
        uint64_t v = 0;
        for (int s = 0; s < 70; s += 7) {
            need(1);
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << (s < 64 ? s : 63);
            if (!(b & 0x80)) return v;
        }
        fail(BHIP_EEXEC, "Parquet: bad varint");
    }
    int64_t zigzag() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
    std::string binary() {
        const uint64_t n = varint();
        if (n > (uint64_t)(end - p)) fail(BHIP_EEXEC, "Parquet: string past the end of the metadata");
        std::string s((const char*)p, (size_t)n);
        p += n;
        return s;
    }
    // field header: false at STOP.  `type` = compact type id, `id` = field id
    bool field(int16_t& id, int& type) {
        need(1);
        const uint8_t b = *p++;
        if (b == 0) return false;
        type = b & 0x0F;
        const int delta = b >> 4;
        if (delta) id = (int16_t)(id + delta);
        else id = (int16_t)zigzag();
        return true;
    }
    void list_header(int& elem_type, uint32_t& n) {
        need(1);
        const uint8_t b = *p++;
        elem_type = b & 0x0F;
        n = b >> 4;
        if (n == 15) {
            const uint64_t v = varint();
            // every element takes at least one byte: a count beyond the remaining bytes is corrupt (and would be a loop bound)
            if (v > (uint64_t)(end - p)) fail(BHIP_EEXEC, "Parquet: list longer than the metadata that holds it");
            n = (uint32_t)v;
        }
    }
    struct Nest {
        Thrift& t;
        explicit Nest(Thrift& th) : t(th) { if (++t.depth > MAX_DEPTH) fail(BHIP_EEXEC, "Parquet: metadata nested deeper than 64 levels"); }
        ~Nest() { --t.depth; }
    };
    void skip(int type) {
        switch (type) {
            case 1: case 2: break;                       // bool in the header
            case 3: need(1); ++p; break;
            case 4: case 5: case 6: varint(); break;
            case 7: need(8); p += 8; break;
            case 8: binary(); break;
            case 9: case 10: {
                Nest nest(*this);
                int et; uint32_t n;
                list_header(et, n);
                for (uint32_t i = 0; i < n; ++i) skip_elem(et);
            } break;
            case 11: {
                Nest nest(*this);
                const uint64_t n = varint();
                if (n > (uint64_t)(end - p)) fail(BHIP_EEXEC, "Parquet: map longer than the metadata that holds it");
                if (n) {
                    need(1);
                    const uint8_t kv = *p++;
                    for (uint64_t i = 0; i < n; ++i) { skip_elem(kv >> 4); skip_elem(kv & 15); }
                }
            } break;
            case 12: {
                Nest nest(*this);
                int16_t id = 0; int t;
                while (field(id, t)) skip(t);
            } break;
            default: fail(BHIP_EEXEC, "Parquet: unknown thrift type " + std::to_string(type));
        }
    }
    void skip_elem(int type) {                           // list elements carry bools as a byte
        if (type == 1 || type == 2) { need(1); ++p; }
        else skip(type);
    }
};

PqColumn read_schema_element(Thrift& t, int& num_children) {
    PqColumn c;
    num_children = 0;
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        switch (id) {
            case 1: c.phys = (int)t.zigzag(); break;
            case 3: c.repetition = (int)t.zigzag(); break;
            case 4: c.name = t.binary(); break;
            case 5: num_children = (int)t.zigzag(); break;
            case 6: c.converted = (int)t.zigzag(); break;
            case 10: {                                   // LogicalType union: 1 STRING, 6 DATE
                if (ty != 12) { t.skip(ty); break; }
                Thrift::Nest nest(t);
                int16_t lid = 0; int lt;
                while (t.field(lid, lt)) {
                    if (lid == 1) c.logical_string = true;
                    if (lid == 6) c.logical_date = true;
                    t.skip(lt);
                }
            } break;
            default: t.skip(ty);
        }
    }
    return c;
}

PqChunk read_column_chunk(Thrift& t) {
    PqChunk ch;
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        if (id == 3 && ty == 12) {                       // ColumnMetaData
            Thrift::Nest nest(t);
            int16_t mid = 0; int mt;
            while (t.field(mid, mt)) {
                switch (mid) {
                    case 4: ch.codec = (int)t.zigzag(); break;
                    case 5: ch.num_values = t.zigzag(); break;
                    case 7: ch.compressed = t.zigzag(); break;
                    case 9: ch.data_off = t.zigzag(); break;
                    case 11: ch.dict_off = t.zigzag(); break;
                    default: t.skip(mt);
                }
            }
        } else
            t.skip(ty);
    }
    return ch;
}

}  // namespace

PqFile read_footer(const std::string& path) {
    PqFile F;
    F.path = path;
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + path);
    const int64_t size = in.tellg();
    F.size = size;
    if (size < 12) fail(BHIP_EEXEC, "Parquet: " + path + " is too short");
    char tail[8];
    in.seekg(size - 8);
    in.read(tail, 8);
    if (!in || memcmp(tail + 4, "PAR1", 4) != 0) fail(BHIP_EEXEC, "Parquet: " + path + " is not a Parquet file (magic missing)");
    uint32_t flen;
    memcpy(&flen, tail, 4);
    if ((int64_t)flen > size - 12) fail(BHIP_EEXEC, "Parquet: corrupt footer length in " + path);
    std::vector<uint8_t> meta(flen ? flen : 1);
    in.seekg(size - 8 - (int64_t)flen);
    in.read(reinterpret_cast<char*>(meta.data()), flen);
    if (!in) fail(BHIP_EEXEC, "Parquet: cannot read the footer of " + path);
    Thrift t{meta.data(), meta.data() + flen};
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        if (id == 2 && ty == 9) {                        // schema
            int et; uint32_t n;
            t.list_header(et, n);
            if (et != 12) fail(BHIP_EEXEC, "Parquet: the schema is not a list of structs");
            for (uint32_t i = 0; i < n; ++i) {
                int children = 0;
                PqColumn c = read_schema_element(t, children);
                if (i == 0) continue;                    // the root
                if (children) fail(BHIP_ENOTIMPL, "Parquet: nested column '" + c.name + "'");
                F.cols.push_back(c);
            }
        } else if (id == 3 && (ty == 4 || ty == 5 || ty == 6)) {
            F.num_rows = t.zigzag();
        } else if (id == 4 && ty == 9) {                 // row groups
            int et; uint32_t n;
            t.list_header(et, n);
            if (et != 12) fail(BHIP_EEXEC, "Parquet: the row groups are not a list of structs");
            for (uint32_t i = 0; i < n; ++i) {
                PqRowGroup g;
                int16_t gid = 0; int gt;
                while (t.field(gid, gt)) {
                    if (gid == 1 && gt == 9) {
                        int cet; uint32_t cn;
                        t.list_header(cet, cn);
                        if (cet != 12) fail(BHIP_EEXEC, "Parquet: the column chunks are not a list of structs");
                        for (uint32_t k = 0; k < cn; ++k) g.cols.push_back(read_column_chunk(t));
                    } else if (gid == 3 && (gt == 4 || gt == 5 || gt == 6)) {
                        g.num_rows = t.zigzag();
                    } else
                        t.skip(gt);
                }
                if (g.num_rows < 0) fail(BHIP_EEXEC, "Parquet: negative row count in a row group");
                F.groups.push_back(g);
            }
        } else
            t.skip(ty);
    }
    for (auto& c : F.cols) {
        if (c.repetition == 2) fail(BHIP_ENOTIMPL, "Parquet: repeated column '" + c.name + "'");
        switch (c.phys) {
            case PQ_BOOLEAN: c.dtype = DT_BOOLEAN; break;
            case PQ_INT32:
                if (c.logical_date || c.converted == 6) c.dtype = DT_DATE32;
                else if (c.converted == -1 || c.converted == 17) c.dtype = DT_INT32;       // none / INT_32
                else if (c.converted == 13) c.dtype = DT_UINT32;                           // UINT_32: the same four bytes
                else if (c.converted == 15 || c.converted == 16 || c.converted == 11 || c.converted == 12) {
                    c.dtype = DT_INT32;                                                    // INT_8 / INT_16 / UINT_8 / UINT_16: four-byte pages, narrowed after the decode
                    c.out_dtype = c.converted == 15 ? DT_INT8 : c.converted == 16 ? DT_INT16 : c.converted == 11 ? DT_UINT8 : DT_UINT16;
                }
                break;
            case PQ_INT64:
                if (c.converted == -1 || c.converted == 18) c.dtype = DT_INT64;           // none / INT_64
                else if (c.converted == 14) c.dtype = DT_UINT64;                          // UINT_64
                else if (c.converted == 9) c.dtype = DT_TIMESTAMP_MS;                     // TIMESTAMP_MILLIS
                else if (c.converted == 10) c.dtype = DT_TIMESTAMP_US;                    // TIMESTAMP_MICROS
                break;
            case PQ_FLOAT: c.dtype = DT_FLOAT32; break;
            case PQ_DOUBLE: c.dtype = DT_FLOAT64; break;
            case PQ_BYTE_ARRAY: c.dtype = DT_UTF8; break;
            default: break;
        }
        if (!c.out_dtype) c.out_dtype = c.dtype;
    }
    return F;
}

std::vector<uint8_t> read_chunk_bytes(const PqFile& F, const PqChunk& ch, const std::string& column_name) {
    const int64_t start = (ch.dict_off > 0 && ch.dict_off < ch.data_off) ? ch.dict_off : ch.data_off;
    // offsets and sizes come from the footer: inside the file, compared by subtraction
    if (start < 4 || start > F.size || ch.compressed < 0 || ch.compressed > F.size - start)
        fail(BHIP_EEXEC, "Parquet: column chunk of '" + column_name + "' lies outside the file");
    std::vector<uint8_t> raw((size_t)ch.compressed);
    std::ifstream in(F.path, std::ios::binary);
    if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + F.path);
    in.seekg(start);
    if (ch.compressed && !in.read(reinterpret_cast<char*>(raw.data()), ch.compressed))
        fail(BHIP_EEXEC, "Parquet: column chunk of '" + column_name + "' runs past the end of the file");
    return raw;
}

namespace {

// ---- Snappy (raw format) ---------------------------------------------------------------------------------------------------
void snappy_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    const uint8_t* p = src;
    const uint8_t* end = src + n;
    uint64_t ulen = 0;
    for (int s = 0;; s += 7) {
        if (p >= end || s > 35) fail(BHIP_EEXEC, "Parquet: bad Snappy preamble");
        const uint8_t b = *p++;
        ulen |= (uint64_t)(b & 0x7F) << s;
        if (!(b & 0x80)) break;
    }
    if (ulen != expect) fail(BHIP_EEXEC, "Parquet: Snappy length does not match the page header");
    // a copy element of 3 bytes yields at most 64: a claimed length beyond that ratio cannot be what the stream decodes to
    if (ulen > 64 * (uint64_t)n + 64) fail(BHIP_EEXEC, "Parquet: Snappy length impossible for the compressed size");
    out.resize((size_t)ulen);
    size_t o = 0;
    while (p < end) {
        const uint8_t tag = *p++;
        if ((tag & 3) == 0) {
            size_t len = (tag >> 2) + 1;
            if (len > 60) {
                const int nb = (int)len - 60;
                if (end - p < nb) fail(BHIP_EEXEC, "Parquet: truncated Snappy literal");
                len = 0;
                for (int i = 0; i < nb; ++i) len |= (size_t)p[i] << (8 * i);
                len += 1;
                p += nb;
            }
            if ((size_t)(end - p) < len || len > out.size() - o) fail(BHIP_EEXEC, "Parquet: corrupt Snappy literal");
            memcpy(&out[o], p, len);
            p += len;
            o += len;
        } else {
            size_t len, off;
            if ((tag & 3) == 1) {
                if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated Snappy copy");
                len = ((tag >> 2) & 7) + 4;
                off = ((size_t)(tag >> 5) << 8) | *p++;
            } else if ((tag & 3) == 2) {
                if (end - p < 2) fail(BHIP_EEXEC, "Parquet: truncated Snappy copy");
                len = (tag >> 2) + 1;
                off = (size_t)p[0] | ((size_t)p[1] << 8);
                p += 2;
            } else {
                if (end - p < 4) fail(BHIP_EEXEC, "Parquet: truncated Snappy copy");
                len = (tag >> 2) + 1;
                off = (size_t)p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24);
                p += 4;
            }
            if (off == 0 || off > o || len > out.size() - o) fail(BHIP_EEXEC, "Parquet: corrupt Snappy copy");
            for (size_t i = 0; i < len; ++i) out[o + i] = out[o + i - off];       // may overlap
            o += len;
        }
    }
    if (o != out.size()) fail(BHIP_EEXEC, "Parquet: Snappy stream ended early");
}

// ---- RLE / bit-packed hybrid -------------------------------------------------------------------------------------------------
// run table of `n_values` values starting at p: the device kernel expands it (launch_pq_expand_runs); a packed run's bits are
// checked to lie inside [base, end) for the values the page actually takes from it
void parse_runs(const uint8_t* base, const uint8_t* p, const uint8_t* end, int bit_width, int64_t n_values, std::vector<PqRun>& runs) {
    int64_t out = 0;
    const int vbytes = (bit_width + 7) / 8;
    while (out < n_values) {
        uint64_t h = 0;
        for (int s = 0;; s += 7) {
            if (p >= end || s > 35) fail(BHIP_EEXEC, "Parquet: truncated RLE / bit-packed data");
            const uint8_t b = *p++;
            h |= (uint64_t)(b & 0x7F) << s;
            if (!(b & 0x80)) break;
        }
        PqRun r;
        r.out_start = (uint32_t)out;
        if (h & 1) {                                     // bit-packed: (h >> 1) groups of 8 values
            const int64_t cnt = (int64_t)(h >> 1) * 8;
            const int64_t bytes = (int64_t)(h >> 1) * bit_width;
            r.count = (uint32_t)std::min<int64_t>(cnt, n_values - out);
            // the last group of a page may be cut short by the writer, but never inside the values the page still needs
            const int64_t needed = ((int64_t)r.count * bit_width + 7) / 8;
            if ((int64_t)(end - p) < needed) fail(BHIP_EEXEC, "Parquet: truncated bit-packed run");
            r.packed = (uint32_t)bit_width;              // (width 0 — a one-entry dictionary — is an RLE run of index 0: below)
            r.value = (uint32_t)(p - base);              // byte offset of the run's bits
            if (bit_width == 0) r.value = 0;
            p += std::min<int64_t>(bytes, end - p);
        } else {
            const int64_t cnt = (int64_t)(h >> 1);
            if (end - p < vbytes) fail(BHIP_EEXEC, "Parquet: truncated RLE run");
            uint32_t v = 0;
            for (int i = 0; i < vbytes; ++i) v |= (uint32_t)p[i] << (8 * i);
            p += vbytes;
            r.count = (uint32_t)std::min<int64_t>(cnt, n_values - out);
            r.packed = 0;
            r.value = v;
        }
        if (r.count == 0) fail(BHIP_EEXEC, "Parquet: empty run");
        runs.push_back(r);
        out += r.count;
    }
}

// definition levels of a flat optional column (bit width 1) -> validity bits; returns the number of valid values
template <class Vec>
int64_t decode_def_levels(const uint8_t* p, const uint8_t* end, int64_t n, Vec& validity) {
    validity.assign((size_t)((n + 63) / 64) * 8 + 8, 0);
    if (n > 0 && (!p || p >= end)) fail(BHIP_EEXEC, "Parquet: an optional column's page without definition levels");
    int64_t out = 0, valid = 0;
    while (out < n) {
        uint64_t h = 0;
        for (int s = 0;; s += 7) {
            if (p >= end || s > 35) fail(BHIP_EEXEC, "Parquet: truncated definition levels");
            const uint8_t b = *p++;
            h |= (uint64_t)(b & 0x7F) << s;
            if (!(b & 0x80)) break;
        }
        if (h & 1) {
            const int64_t groups = (int64_t)(h >> 1);
            if (groups == 0) fail(BHIP_EEXEC, "Parquet: empty run of definition levels");
            for (int64_t g = 0; g < groups && out < n; ++g) {
                if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated definition levels");
                const uint8_t byte = *p++;
                for (int b = 0; b < 8 && out < n; ++b, ++out)
                    if ((byte >> b) & 1) { validity[(size_t)(out >> 3)] |= (uint8_t)(1u << (out & 7)); ++valid; }
            }
        } else {
            int64_t cnt = (int64_t)(h >> 1);
            if (cnt == 0) fail(BHIP_EEXEC, "Parquet: empty run of definition levels");
            if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated definition levels");
            const uint8_t v = *p++;
            cnt = std::min<int64_t>(cnt, n - out);
            if (v & 1) {
                for (int64_t i = 0; i < cnt; ++i) validity[(size_t)((out + i) >> 3)] |= (uint8_t)(1u << ((out + i) & 7));
                valid += cnt;
            }
            out += cnt;
        }
    }
    return valid;
}

struct PageHeader { int type = -1; int64_t usize = 0, csize = 0; int64_t n_values = 0; int encoding = 0; int64_t n_nulls = -1; int64_t def_bytes = 0, rep_bytes = 0; bool v2_compressed = true; };

PageHeader read_page_header(Thrift& t) {
    PageHeader h;
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        if (id == 1 && ty >= 4 && ty <= 6) h.type = (int)t.zigzag();
        else if (id == 2 && ty >= 4 && ty <= 6) h.usize = t.zigzag();
        else if (id == 3 && ty >= 4 && ty <= 6) h.csize = t.zigzag();
        else if ((id == 5 || id == 7 || id == 8) && ty == 12) {
            Thrift::Nest nest(t);
            int16_t sid = 0; int st;
            while (t.field(sid, st)) {
                const bool is_int = st >= 4 && st <= 6;
                if (id == 5) {                           // DataPageHeader
                    if (sid == 1 && is_int) h.n_values = t.zigzag();
                    else if (sid == 2 && is_int) h.encoding = (int)t.zigzag();
                    else t.skip(st);
                } else if (id == 7) {                    // DictionaryPageHeader
                    if (sid == 1 && is_int) h.n_values = t.zigzag();
                    else if (sid == 2 && is_int) h.encoding = (int)t.zigzag();
                    else t.skip(st);
                } else {                                 // DataPageHeaderV2
                    if (sid == 1 && is_int) h.n_values = t.zigzag();
                    else if (sid == 2 && is_int) h.n_nulls = t.zigzag();
                    else if (sid == 4 && is_int) h.encoding = (int)t.zigzag();
                    else if (sid == 5 && is_int) h.def_bytes = t.zigzag();
                    else if (sid == 6 && is_int) h.rep_bytes = t.zigzag();
                    else if (sid == 7 && (st == 1 || st == 2)) h.v2_compressed = st == 1;
                    else t.skip(st);
                }
            }
        } else
            t.skip(ty);
    }
    return h;
}

size_t phys_width(int phys) { return (phys == PQ_INT32 || phys == PQ_FLOAT) ? 4 : (phys == PQ_INT64 || phys == PQ_DOUBLE) ? 8 : 0; }

void parse_dictionary(const PqColumn& pc, const uint8_t* vals, size_t nbytes, int64_t n, HostDict& d) {
    if (n < 0) fail(BHIP_EEXEC, "Parquet: negative dictionary size");
    d.present = true;
    d.n = n;
    d.offsets.clear();
    d.bytes.clear();
    if (pc.phys == PQ_BYTE_ARRAY) {
        if ((uint64_t)n > nbytes / 4) fail(BHIP_EEXEC, "Parquet: truncated dictionary page");        // every value has a 4-byte length
        d.offsets.assign((size_t)n + 1, 0);
        d.bytes.reserve(nbytes);
        const uint8_t* p = vals;
        const uint8_t* end = vals + nbytes;
        for (int64_t i = 0; i < n; ++i) {
            if (end - p < 4) fail(BHIP_EEXEC, "Parquet: truncated dictionary page");
            uint32_t len;
            memcpy(&len, p, 4);
            p += 4;
            if ((size_t)(end - p) < len) fail(BHIP_EEXEC, "Parquet: truncated dictionary string");
            d.bytes.insert(d.bytes.end(), p, p + len);
            p += len;
            if (d.bytes.size() > 0x7FFFFFF0u) fail(BHIP_ENOTIMPL, "Parquet: more than 2 GiB of dictionary strings");
            d.offsets[(size_t)i + 1] = (int32_t)d.bytes.size();
        }
    } else {
        const size_t w = phys_width(pc.phys);
        if (pc.phys == PQ_BOOLEAN || !w) fail(BHIP_ENOTIMPL, "Parquet: dictionary-encoded column of physical type " + std::to_string(pc.phys));
        if ((uint64_t)n > nbytes / w) fail(BHIP_EEXEC, "Parquet: truncated dictionary page");
        d.bytes.assign(vals, vals + w * (size_t)n);
    }
}

}  // namespace

// pages -> one set of chunk buffers where that is possible (parquet_host.hpp: ChunkStaging); BHIP_PARQUET_PER_PAGE=1 keeps the pages
static void stage_chunk(HostChunk& hc, size_t width) {
    static const bool per_page = [] { const char* v = getenv("BHIP_PARQUET_PER_PAGE"); return v && atoi(v) != 0; }();
    if (per_page || hc.pages.size() < 2) return;
    bool any_nulls = false, all_dict = true, all_fixed = true;
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
    if (any_nulls) return;
    if (all_fixed && width) {
        hc.staged_bytes.reserve(bytes_total);
        for (HostPage& pg : hc.pages) {
            hc.staged_bytes.insert(hc.staged_bytes.end(), pg.bytes.begin(), pg.bytes.begin() + width * (size_t)pg.n);
            HostVec<uint8_t>().swap(pg.bytes);
        }
        hc.staged = STAGED_FIXED;
    } else if (all_dict && bytes_total < 0xFFFFFFF0u && rows_total < 0xFFFFFFF0ll) {
        hc.staged_bytes.reserve(bytes_total);
        hc.staged_runs.reserve(runs_total);
        int64_t row = 0;
        for (HostPage& pg : hc.pages) {
            const size_t byte_base = hc.staged_bytes.size();
            hc.staged_bytes.insert(hc.staged_bytes.end(), pg.bytes.begin(), pg.bytes.end());
            for (PqRun r : pg.runs) {
                r.out_start += (uint32_t)row;
                if (r.packed) r.value += (uint32_t)byte_base;
                hc.staged_runs.push_back(r);
            }
            row += pg.n;
            HostVec<uint8_t>().swap(pg.bytes);
            std::vector<PqRun>().swap(pg.runs);
        }
        hc.staged = STAGED_DICT;
    }
}

HostChunk parse_chunk(const uint8_t* raw, size_t len, const PqColumn& pc, const PqChunk& ch, int64_t n_rows) {
    if (ch.codec != 0 && ch.codec != 1) fail(BHIP_ENOTIMPL, "Parquet: column '" + pc.name + "' uses compression codec " + std::to_string(ch.codec) + " (UNCOMPRESSED and SNAPPY are read)");
    HostChunk out;
    const bool optional = pc.repetition == 1;
    const size_t width = phys_width(pc.phys);
    const uint8_t* p = raw;
    const uint8_t* end = raw + len;
    std::vector<uint8_t> page;
    while (out.rows < n_rows) {
        if (p >= end) fail(BHIP_EEXEC, "Parquet: column chunk of '" + pc.name + "' ends before its last row");
        Thrift t{p, end};
        const PageHeader h = read_page_header(t);
        p = t.p;
        if (h.csize < 0 || h.usize < 0 || (int64_t)(end - p) < h.csize) fail(BHIP_EEXEC, "Parquet: page of '" + pc.name + "' runs past its chunk");
        const uint8_t* body = p;
        p += h.csize;
        if (h.type == 1) continue;                       // index page
        // ---- page payload, decompressed ------------------------------------------------------------------------------------
        const uint8_t* def_ptr = nullptr;
        int64_t def_len = 0;
        const uint8_t* vals;
        size_t vals_len;
        if (h.type == 3) {                               // V2: levels first, uncompressed; the rest compressed on its own
            if (h.rep_bytes < 0 || h.def_bytes < 0 || h.def_bytes > h.csize || h.def_bytes > h.usize || h.rep_bytes > h.csize - h.def_bytes)
                fail(BHIP_EEXEC, "Parquet: level lengths of a V2 page of '" + pc.name + "' do not fit the page");
            if (h.rep_bytes) fail(BHIP_ENOTIMPL, "Parquet: repetition levels");
            def_ptr = body;
            def_len = h.def_bytes;
            const uint8_t* rest = body + h.def_bytes;
            const size_t rest_c = (size_t)(h.csize - h.def_bytes), rest_u = (size_t)(h.usize - h.def_bytes);
            if (ch.codec == 1 && h.v2_compressed) { snappy_decode(rest, rest_c, page, rest_u); vals = page.data(); vals_len = page.size(); }
            else { vals = rest; vals_len = rest_c; }
        } else {
            if (ch.codec == 1) { snappy_decode(body, (size_t)h.csize, page, (size_t)h.usize); vals = page.data(); vals_len = page.size(); }
            else { vals = body; vals_len = (size_t)h.csize; }
            if (h.type == 0 && optional) {               // V1: [4-byte length][RLE definition levels]
                if (vals_len < 4) fail(BHIP_EEXEC, "Parquet: truncated data page");
                uint32_t dl;
                memcpy(&dl, vals, 4);
                if ((size_t)dl > vals_len - 4) fail(BHIP_EEXEC, "Parquet: definition levels run past the page");
                def_ptr = vals + 4;
                def_len = dl;
                vals += 4 + (size_t)dl;
                vals_len -= 4 + (size_t)dl;
            }
        }
        if (h.type == 2) {                               // dictionary page
            if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICT) fail(BHIP_ENOTIMPL, "Parquet: dictionary page encoding " + std::to_string(h.encoding));
            parse_dictionary(pc, vals, vals_len, h.n_values, out.dict);
            continue;
        }
        if (h.type != 0 && h.type != 3) fail(BHIP_ENOTIMPL, "Parquet: page type " + std::to_string(h.type));
        const int64_t n = h.n_values;
        if (n <= 0 || n > n_rows - out.rows) fail(BHIP_EEXEC, "Parquet: page row count does not fit its row group");
        out.pages.emplace_back();
        HostPage& pg = out.pages.back();
        pg.n = n;
        pg.n_valid = n;
        if (optional) {
            pg.n_valid = decode_def_levels(def_ptr, def_ptr ? def_ptr + def_len : nullptr, n, pg.validity);
            pg.has_nulls = pg.n_valid < n;
            if (pg.has_nulls) {
                pg.prefix.assign((size_t)((n + 63) / 64) + 1, 0);
                for (size_t i = 0; i + 1 < pg.prefix.size(); ++i) {
                    uint64_t w;
                    memcpy(&w, pg.validity.data() + 8 * i, 8);
                    pg.prefix[i + 1] = pg.prefix[i] + (uint32_t)__builtin_popcountll(w);
                }
            }
        }
        const int64_t n_valid = pg.n_valid;
        if (h.encoding == ENC_PLAIN_DICT || h.encoding == ENC_RLE_DICT) {
            // ---- dictionary indices: the run table here, expansion + NULL re-insertion + gather on the device ----------------
            if (!out.dict.present) fail(BHIP_EEXEC, "Parquet: dictionary-encoded page without a dictionary page");
            if (vals_len < 1) fail(BHIP_EEXEC, "Parquet: truncated dictionary-index page");
            const int bw = vals[0];
            if (bw > 32) fail(BHIP_EEXEC, "Parquet: dictionary index width " + std::to_string(bw));
            pg.kind = PG_DICT;
            pg.bit_width = bw;
            if (n_valid) parse_runs(vals, vals + 1, vals + vals_len, bw, n_valid, pg.runs);
            pg.bytes.assign(vals, vals + vals_len);
        } else if (h.encoding == ENC_PLAIN || (h.encoding == ENC_RLE && pc.phys == PQ_BOOLEAN)) {
            std::vector<uint8_t> rle_bits;
            if (h.encoding == ENC_RLE) {
                // BOOLEAN of data page V2: [4-byte length] + RLE / bit-packed hybrid of width 1 -> the dense bit vector PLAIN would carry
                if (vals_len < 4) fail(BHIP_EEXEC, "Parquet: truncated RLE BOOLEAN page");
                uint32_t bl;
                memcpy(&bl, vals, 4);
                if ((size_t)bl > vals_len - 4) fail(BHIP_EEXEC, "Parquet: RLE BOOLEAN data runs past the page");
                decode_def_levels(vals + 4, vals + 4 + bl, n_valid, rle_bits);
                vals = rle_bits.data();
                vals_len = rle_bits.size();
            }
            if (pc.phys == PQ_BYTE_ARRAY) {
                // length-prefixed strings: a sequential walk, NULL rows repeat the running offset
                pg.kind = PG_STRINGS;
                pg.offsets.assign((size_t)n + 1, 0);
                pg.bytes.reserve(vals_len);
                const uint8_t* q = vals;
                const uint8_t* qe = vals + vals_len;
                for (int64_t i = 0; i < n; ++i) {
                    const bool valid = !pg.has_nulls || ((pg.validity[(size_t)(i >> 3)] >> (i & 7)) & 1);
                    if (valid) {
                        if (qe - q < 4) fail(BHIP_EEXEC, "Parquet: truncated string page");
                        uint32_t slen;
                        memcpy(&slen, q, 4);
                        q += 4;
                        if ((size_t)(qe - q) < slen) fail(BHIP_EEXEC, "Parquet: truncated string value");
                        pg.bytes.insert(pg.bytes.end(), q, q + slen);
                        q += slen;
                        if (pg.bytes.size() > 0x7FFFFFF0u) fail(BHIP_ENOTIMPL, "Parquet: more than 2 GiB of strings in one page");
                    }
                    pg.offsets[(size_t)i + 1] = (int32_t)pg.bytes.size();
                }
            } else if (pc.phys == PQ_BOOLEAN) {
                pg.kind = PG_BOOL;
                if (vals_len < (size_t)((n_valid + 7) / 8)) fail(BHIP_EEXEC, "Parquet: truncated BOOLEAN page");
                pg.bytes.assign((size_t)((n + 63) / 64) * 8 + 8, 0);
                if (pg.has_nulls) {
                    int64_t k = 0;
                    for (int64_t i = 0; i < n; ++i)
                        if ((pg.validity[(size_t)(i >> 3)] >> (i & 7)) & 1) {
                            if ((vals[(size_t)(k >> 3)] >> (k & 7)) & 1) pg.bytes[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
                            ++k;
                        }
                } else {
                    memcpy(pg.bytes.data(), vals, (size_t)((n + 7) / 8));
                    if (n & 7) pg.bytes[(size_t)(n >> 3)] &= (uint8_t)((1u << (n & 7)) - 1);
                }
            } else {
                if (!width) fail(BHIP_ENOTIMPL, "Parquet: PLAIN pages of physical type " + std::to_string(pc.phys));
                if (vals_len / width < (size_t)n_valid) fail(BHIP_EEXEC, "Parquet: truncated PLAIN page of '" + pc.name + "'");
                pg.kind = PG_FIXED;
                pg.bytes.assign(vals, vals + width * (size_t)n_valid);
            }
        } else {
            fail(BHIP_ENOTIMPL, "Parquet: column '" + pc.name + "' uses encoding " + std::to_string(h.encoding) + " (PLAIN and RLE_DICTIONARY are read)");
        }
        out.rows += n;
    }
    stage_chunk(out, width);
    return out;
}

int64_t host_walk(const std::string& path) {
    const PqFile F = read_footer(path);
    int64_t rows = 0;
    for (auto& g : F.groups) {
        if (g.num_rows == 0) continue;
        if (g.num_rows > 0xFFFFFFF0ll) fail(BHIP_ENOTIMPL, "Parquet: row group of more than 2^32 rows");
        for (size_t ci = 0; ci < F.cols.size(); ++ci) {
            if (!F.cols[ci].dtype) continue;                         // a type outside the GPU path: the plan refuses it, nothing is parsed
            if (ci >= g.cols.size()) fail(BHIP_EEXEC, "Parquet: row group without column " + std::to_string(ci));
            const std::vector<uint8_t> raw = read_chunk_bytes(F, g.cols[ci], F.cols[ci].name);
            const HostChunk hc = parse_chunk(raw.data(), raw.size(), F.cols[ci], g.cols[ci], g.num_rows);
            // touch what the device half would upload (the sanitizer sees every byte the walk produced)
            uint64_t sum = 0;
            for (auto& pg : hc.pages) {
                for (auto b : pg.bytes) sum += b;
                for (auto& r : pg.runs) sum += r.count;
            }
            (void)sum;
        }
        rows += g.num_rows;
    }
    return rows;
}

}  // namespace pq
}  // namespace bhip
