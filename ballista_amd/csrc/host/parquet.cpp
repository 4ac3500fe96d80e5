// parquet.cpp — ParquetExec: the scan leaf the serde builds at rust/core/src/serde/physical_plan/from_proto.rs:111-121
// (`ParquetExec::try_from_files(filenames, projection, None, batch_size, num_partitions)`), what `--format parquet` of the
// reference's TPC-H benchmark reads (rust/benchmarks/tpch/src/main.rs:147-150; its `convert` writes Snappy by default, :84-86).
//
// Split of the work.  HOST: the footer (Thrift compact protocol), page headers, Snappy, definition levels, the run headers of
// the RLE / bit-packed hybrid, length-prefixed strings — everything that is a sequential byte walk.  DEVICE: every per-value
// step: plain values are one copy, dictionary indices are expanded from the run table by one kernel (one thread per value:
// binary search of its run, bit extraction), NULLs are re-inserted by rank (prefix popcounts of the validity words), dictionary
// values — strings included — are gathered with the take kernels FilterExec uses.  One output batch per row group.
//
// Supported: flat schemas; INT32 (Int32 / Date32), INT64, DOUBLE, BOOLEAN, BYTE_ARRAY (Utf8); required and optional fields;
// PLAIN, PLAIN_DICTIONARY / RLE_DICTIONARY; data pages V1 and V2; UNCOMPRESSED and SNAPPY.  Anything else is BHIP_ENOTIMPL at
// plan time (types, nesting) or at the page that needs it (codec, encoding): the caller keeps its CPU ParquetExec.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "../util_kernels.h"
#include "plan.hpp"

namespace bhip {

namespace {

// ---- Thrift compact protocol ---------------------------------------------------------------------------------------------
struct Thrift {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t varint() {
        uint64_t v = 0;
        for (int s = 0; s < 70; s += 7) {
            if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated metadata");
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
        if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated struct");
        const uint8_t b = *p++;
        if (b == 0) return false;
        type = b & 0x0F;
        const int delta = b >> 4;
        if (delta) id = (int16_t)(id + delta);
        else id = (int16_t)zigzag();
        return true;
    }
    void list_header(int& elem_type, uint32_t& n) {
        if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated list");
        const uint8_t b = *p++;
        elem_type = b & 0x0F;
        n = b >> 4;
        if (n == 15) n = (uint32_t)varint();
    }
    void skip(int type) {
        switch (type) {
            case 1: case 2: break;                       // bool in the header
            case 3: ++p; break;
            case 4: case 5: case 6: varint(); break;
            case 7: p += 8; break;
            case 8: binary(); break;
            case 9: case 10: {
                int et; uint32_t n;
                list_header(et, n);
                for (uint32_t i = 0; i < n; ++i) skip_elem(et);
            } break;
            case 11: {
                const uint32_t n = (uint32_t)varint();
                if (n) {
                    const uint8_t kv = *p++;
                    for (uint32_t i = 0; i < n; ++i) { skip_elem(kv >> 4); skip_elem(kv & 15); }
                }
            } break;
            case 12: {
                int16_t id = 0; int t;
                while (field(id, t)) skip(t);
            } break;
            default: fail(BHIP_EEXEC, "Parquet: unknown thrift type " + std::to_string(type));
        }
        if (p > end) fail(BHIP_EEXEC, "Parquet: truncated metadata");
    }
    void skip_elem(int type) {                           // list elements carry bools as a byte
        if (type == 1 || type == 2) ++p;
        else skip(type);
    }
};

enum { PQ_BOOLEAN = 0, PQ_INT32 = 1, PQ_INT64 = 2, PQ_INT96 = 3, PQ_FLOAT = 4, PQ_DOUBLE = 5, PQ_BYTE_ARRAY = 6, PQ_FIXED = 7 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICT = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_RLE_DICT = 8 };

struct PqColumn {
    std::string name;
    int phys = -1, converted = -1, repetition = 0;
    int dtype = 0;          // the type the pages decode to (INT32 pages annotated INT_8 .. UINT_16 decode as Int32 ...)
    int out_dtype = 0;      // ... and are narrowed to this column type afterwards (launch_narrow_i32); else == dtype
    bool logical_date = false, logical_string = false;
};
struct PqChunk { int codec = 0; int64_t num_values = 0, data_off = 0, dict_off = -1, compressed = 0; };
struct PqRowGroup { int64_t num_rows = 0; std::vector<PqChunk> cols; };
struct PqFile { std::string path; std::vector<PqColumn> cols; std::vector<PqRowGroup> groups; int64_t num_rows = 0; };

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

PqFile read_footer(const std::string& path) {
    PqFile F;
    F.path = path;
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + path);
    const int64_t size = in.tellg();
    if (size < 12) fail(BHIP_EEXEC, "Parquet: " + path + " is too short");
    char tail[8];
    in.seekg(size - 8);
    in.read(tail, 8);
    if (memcmp(tail + 4, "PAR1", 4) != 0) fail(BHIP_EEXEC, "Parquet: " + path + " is not a Parquet file (magic missing)");
    uint32_t flen;
    memcpy(&flen, tail, 4);
    if ((int64_t)flen + 12 > size) fail(BHIP_EEXEC, "Parquet: corrupt footer length in " + path);
    std::vector<uint8_t> meta(flen);
    in.seekg(size - 8 - (int64_t)flen);
    in.read(reinterpret_cast<char*>(meta.data()), flen);
    Thrift t{meta.data(), meta.data() + meta.size()};
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        if (id == 2 && ty == 9) {                        // schema
            int et; uint32_t n;
            t.list_header(et, n);
            for (uint32_t i = 0; i < n; ++i) {
                int children = 0;
                PqColumn c = read_schema_element(t, children);
                if (i == 0) continue;                    // the root
                if (children) fail(BHIP_ENOTIMPL, "Parquet: nested column '" + c.name + "'");
                F.cols.push_back(c);
            }
        } else if (id == 3) {
            F.num_rows = t.zigzag();
        } else if (id == 4 && ty == 9) {                 // row groups
            int et; uint32_t n;
            t.list_header(et, n);
            for (uint32_t i = 0; i < n; ++i) {
                PqRowGroup g;
                int16_t gid = 0; int gt;
                while (t.field(gid, gt)) {
                    if (gid == 1 && gt == 9) {
                        int cet; uint32_t cn;
                        t.list_header(cet, cn);
                        for (uint32_t k = 0; k < cn; ++k) g.cols.push_back(read_column_chunk(t));
                    } else if (gid == 3) {
                        g.num_rows = t.zigzag();
                    } else
                        t.skip(gt);
                }
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
            if ((size_t)(end - p) < len || o + len > out.size()) fail(BHIP_EEXEC, "Parquet: corrupt Snappy literal");
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
            if (off == 0 || off > o || o + len > out.size()) fail(BHIP_EEXEC, "Parquet: corrupt Snappy copy");
            for (size_t i = 0; i < len; ++i) out[o + i] = out[o + i - off];       // may overlap
            o += len;
        }
    }
    if (o != out.size()) fail(BHIP_EEXEC, "Parquet: Snappy stream ended early");
}

// ---- RLE / bit-packed hybrid -------------------------------------------------------------------------------------------------
// run table of `n_values` values starting at p: the device kernel expands it (launch_pq_expand_runs)
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
            if (end - p < bytes && out + cnt <= n_values) fail(BHIP_EEXEC, "Parquet: truncated bit-packed run");
            r.count = (uint32_t)std::min<int64_t>(cnt, n_values - out);
            r.packed = 1;
            r.value = (uint32_t)(p - base);              // byte offset of the run's bits
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
        if (r.count == 0 && (h >> 1) == 0) fail(BHIP_EEXEC, "Parquet: empty run");
        runs.push_back(r);
        out += r.count;
    }
}

// definition levels of a flat optional column (bit width 1) -> validity bits; returns the number of valid values
int64_t decode_def_levels(const uint8_t* p, const uint8_t* end, int64_t n, std::vector<uint8_t>& validity) {
    validity.assign((size_t)((n + 63) / 64) * 8 + 8, 0);
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
            for (int64_t g = 0; g < groups && out < n; ++g) {
                if (p >= end) fail(BHIP_EEXEC, "Parquet: truncated definition levels");
                const uint8_t byte = *p++;
                for (int b = 0; b < 8 && out < n; ++b, ++out)
                    if ((byte >> b) & 1) { validity[(size_t)(out >> 3)] |= (uint8_t)(1u << (out & 7)); ++valid; }
            }
        } else {
            int64_t cnt = (int64_t)(h >> 1);
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

// ---- one column chunk -> one device column ----------------------------------------------------------------------------------------
struct PageHeader { int type = -1; int64_t usize = 0, csize = 0; int64_t n_values = 0; int encoding = 0; int64_t n_nulls = -1; int64_t def_bytes = 0, rep_bytes = 0; bool v2_compressed = true; };

PageHeader read_page_header(Thrift& t) {
    PageHeader h;
    int16_t id = 0; int ty;
    while (t.field(id, ty)) {
        if (id == 1) h.type = (int)t.zigzag();
        else if (id == 2) h.usize = t.zigzag();
        else if (id == 3) h.csize = t.zigzag();
        else if ((id == 5 || id == 7 || id == 8) && ty == 12) {
            int16_t sid = 0; int st;
            while (t.field(sid, st)) {
                if (id == 5) {                           // DataPageHeader
                    if (sid == 1) h.n_values = t.zigzag();
                    else if (sid == 2) h.encoding = (int)t.zigzag();
                    else t.skip(st);
                } else if (id == 7) {                    // DictionaryPageHeader
                    if (sid == 1) h.n_values = t.zigzag();
                    else if (sid == 2) h.encoding = (int)t.zigzag();
                    else t.skip(st);
                } else {                                 // DataPageHeaderV2
                    if (sid == 1) h.n_values = t.zigzag();
                    else if (sid == 2) h.n_nulls = t.zigzag();
                    else if (sid == 4) h.encoding = (int)t.zigzag();
                    else if (sid == 5) h.def_bytes = t.zigzag();
                    else if (sid == 6) h.rep_bytes = t.zigzag();
                    else if (sid == 7) h.v2_compressed = st == 1;
                    else t.skip(st);
                }
            }
        } else
            t.skip(ty);
    }
    return h;
}

BufferPtr upload(const Exec& ex, const void* host, size_t bytes) {
    BufferPtr b = make_buffer(ex, bytes + 16);
    if (bytes) HIP_CHECK(hipMemcpyAsync(b->ptr(), host, bytes, hipMemcpyHostToDevice, ex.stream));
    return b;
}

// dictionary of a chunk as a device column
Column dictionary_column(const Exec& ex, const PqColumn& pc, const uint8_t* vals, size_t nbytes, int64_t n) {
    Column d;
    d.dtype = pc.dtype;
    d.length = n;
    if (pc.phys == PQ_BYTE_ARRAY) {
        std::vector<int32_t> off((size_t)n + 1, 0);
        std::vector<uint8_t> bytes;
        bytes.reserve(nbytes);
        const uint8_t* p = vals;
        const uint8_t* end = vals + nbytes;
        for (int64_t i = 0; i < n; ++i) {
            if (end - p < 4) fail(BHIP_EEXEC, "Parquet: truncated dictionary page");
            uint32_t len;
            memcpy(&len, p, 4);
            p += 4;
            if ((size_t)(end - p) < len) fail(BHIP_EEXEC, "Parquet: truncated dictionary string");
            bytes.insert(bytes.end(), p, p + len);
            p += len;
            off[(size_t)i + 1] = (int32_t)bytes.size();
        }
        d.offsets = upload(ex, off.data(), off.size() * 4);
        d.data = upload(ex, bytes.data(), bytes.size());
        d.data_bytes = (int64_t)bytes.size();
    } else {
        const size_t w = (pc.phys == PQ_INT32 || pc.phys == PQ_FLOAT) ? 4 : 8;
        if (pc.phys == PQ_BOOLEAN) fail(BHIP_ENOTIMPL, "Parquet: dictionary-encoded BOOLEAN");
        if (nbytes < w * (size_t)n) fail(BHIP_EEXEC, "Parquet: truncated dictionary page");
        d.data = upload(ex, vals, w * (size_t)n);
    }
    return d;
}

Column decode_chunk(const Exec& ex, std::ifstream& in, const PqColumn& pc, const PqChunk& ch, int64_t n_rows) {
    if (ch.codec != 0 && ch.codec != 1) fail(BHIP_ENOTIMPL, "Parquet: column '" + pc.name + "' uses compression codec " + std::to_string(ch.codec) + " (UNCOMPRESSED and SNAPPY are read)");
    const int64_t start = (ch.dict_off > 0 && ch.dict_off < ch.data_off) ? ch.dict_off : ch.data_off;
    std::vector<uint8_t> raw((size_t)ch.compressed);
    in.clear();
    in.seekg(start);
    if (!in.read(reinterpret_cast<char*>(raw.data()), ch.compressed)) fail(BHIP_EEXEC, "Parquet: column chunk of '" + pc.name + "' runs past the end of the file");
    const bool optional = pc.repetition == 1;
    const size_t width = (pc.phys == PQ_INT32 || pc.phys == PQ_FLOAT) ? 4 : (pc.phys == PQ_INT64 || pc.phys == PQ_DOUBLE) ? 8 : 0;
    Column dict;
    bool have_dict = false;
    std::vector<Column> pieces;
    std::vector<BufferPtr> keep;                         // device scratch that must outlive the kernels of this chunk
    int64_t rows_done = 0;
    const uint8_t* p = raw.data();
    const uint8_t* end = raw.data() + raw.size();
    const LaunchCfg cfg = ex.cfg();
    std::vector<uint8_t> page, validity;
    while (rows_done < n_rows) {
        if (p >= end) fail(BHIP_EEXEC, "Parquet: column chunk of '" + pc.name + "' ends before its last row");
        Thrift t{p, end};
        const PageHeader h = read_page_header(t);
        p = t.p;
        if (h.csize < 0 || (int64_t)(end - p) < h.csize) fail(BHIP_EEXEC, "Parquet: page of '" + pc.name + "' runs past its chunk");
        const uint8_t* body = p;
        p += h.csize;
        if (h.type == 1) continue;                       // index page
        // ---- page payload, decompressed ------------------------------------------------------------------------------------
        const uint8_t* def_ptr = nullptr;
        int64_t def_len = 0;
        const uint8_t* vals;
        size_t vals_len;
        if (h.type == 3) {                               // V2: levels first, uncompressed; the rest compressed on its own
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
                if ((size_t)dl + 4 > vals_len) fail(BHIP_EEXEC, "Parquet: definition levels run past the page");
                def_ptr = vals + 4;
                def_len = dl;
                vals += 4 + dl;
                vals_len -= 4 + dl;
            }
        }
        if (h.type == 2) {                               // dictionary page
            if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICT) fail(BHIP_ENOTIMPL, "Parquet: dictionary page encoding " + std::to_string(h.encoding));
            dict = dictionary_column(ex, pc, vals, vals_len, h.n_values);
            have_dict = true;
            continue;
        }
        if (h.type != 0 && h.type != 3) fail(BHIP_ENOTIMPL, "Parquet: page type " + std::to_string(h.type));
        const int64_t n = h.n_values;
        if (n <= 0 || rows_done + n > n_rows) fail(BHIP_EEXEC, "Parquet: page row count does not fit its row group");
        int64_t n_valid = n;
        BufferPtr dvalid, dprefix;
        bool has_nulls = false;
        if (optional) {
            n_valid = decode_def_levels(def_ptr, def_ptr + def_len, n, validity);
            has_nulls = n_valid < n;
            if (has_nulls) {
                std::vector<uint32_t> prefix((size_t)((n + 63) / 64) + 1, 0);
                const uint64_t* w = reinterpret_cast<const uint64_t*>(validity.data());
                for (size_t i = 0; i + 1 < prefix.size(); ++i) prefix[i + 1] = prefix[i] + (uint32_t)__builtin_popcountll(w[i]);
                dvalid = upload(ex, validity.data(), validity.size());
                dprefix = upload(ex, prefix.data(), prefix.size() * 4);
                keep.push_back(dprefix);
            }
        }
        Column c;
        c.dtype = pc.dtype;
        c.length = n;
        if (has_nulls) c.validity = dvalid;
        if (h.encoding == ENC_PLAIN_DICT || h.encoding == ENC_RLE_DICT) {
            // ---- dictionary indices: run table on the host, expansion + NULL re-insertion + gather on the device --------------
            if (!have_dict) fail(BHIP_EEXEC, "Parquet: dictionary-encoded page without a dictionary page");
            if (vals_len < 1) fail(BHIP_EEXEC, "Parquet: truncated dictionary-index page");
            const int bw = vals[0];
            if (bw > 32) fail(BHIP_EEXEC, "Parquet: dictionary index width " + std::to_string(bw));
            std::vector<PqRun> runs;
            if (n_valid) parse_runs(vals, vals + 1, vals + vals_len, bw, n_valid, runs);
            BufferPtr dbytes = upload(ex, vals, vals_len), druns = upload(ex, runs.data(), runs.size() * sizeof(PqRun));
            BufferPtr dense = make_buffer(ex, (size_t)n_valid * 4 + 16);
            keep.push_back(dbytes); keep.push_back(druns); keep.push_back(dense);
            if (n_valid) TIMED_LAUNCH_N(ex, "pq_expand_runs", n_valid, launch_pq_expand_runs(cfg, druns->as<PqRun>(), (uint32_t)runs.size(), dbytes->as<uint8_t>(), bw,
                                                                                            (uint32_t)n_valid, (uint32_t)dict.length, dense->as<uint32_t>()));
            const uint32_t* idx = dense->as<uint32_t>();
            if (has_nulls) {
                BufferPtr full = make_buffer(ex, (size_t)n * 4 + 16);
                keep.push_back(full);
                TIMED_LAUNCH_N(ex, "pq_scatter_valid", n, launch_pq_scatter_valid(cfg, dvalid->as<uint64_t>(), dprefix->as<uint32_t>(), dense->ptr(), 4, n, full->ptr(), 1));
                idx = full->as<uint32_t>();
            }
            Column g = has_nulls ? take_column_nullable(ex, dict, idx, n) : take_column(ex, dict, idx, n);
            g.dtype = pc.dtype;
            if (has_nulls) g.validity = dvalid;          // the definition levels ARE the validity
            else g.validity = nullptr;
            c = g;
            c.length = n;
        } else if (h.encoding == ENC_PLAIN || (h.encoding == ENC_RLE && pc.phys == PQ_BOOLEAN)) {
            std::vector<uint8_t> rle_bits;
            if (h.encoding == ENC_RLE) {
                // BOOLEAN of data page V2: [4-byte length] + RLE / bit-packed hybrid of width 1 -> the dense bit vector PLAIN would carry
                if (vals_len < 4) fail(BHIP_EEXEC, "Parquet: truncated RLE BOOLEAN page");
                uint32_t bl;
                memcpy(&bl, vals, 4);
                if ((size_t)bl + 4 > vals_len) fail(BHIP_EEXEC, "Parquet: RLE BOOLEAN data runs past the page");
                decode_def_levels(vals + 4, vals + 4 + bl, n_valid, rle_bits);
                vals = rle_bits.data();
                vals_len = rle_bits.size();
            }
            if (pc.phys == PQ_BYTE_ARRAY) {
                // length-prefixed strings: a sequential walk (host), NULL rows repeat the running offset
                std::vector<int32_t> off((size_t)n + 1, 0);
                std::vector<uint8_t> bytes;
                bytes.reserve(vals_len);
                const uint8_t* q = vals;
                const uint8_t* qe = vals + vals_len;
                for (int64_t i = 0; i < n; ++i) {
                    const bool valid = !has_nulls || ((validity[(size_t)(i >> 3)] >> (i & 7)) & 1);
                    if (valid) {
                        if (qe - q < 4) fail(BHIP_EEXEC, "Parquet: truncated string page");
                        uint32_t len;
                        memcpy(&len, q, 4);
                        q += 4;
                        if ((size_t)(qe - q) < len) fail(BHIP_EEXEC, "Parquet: truncated string value");
                        bytes.insert(bytes.end(), q, q + len);
                        q += len;
                    }
                    off[(size_t)i + 1] = (int32_t)bytes.size();
                }
                c.offsets = upload(ex, off.data(), off.size() * 4);
                c.data = upload(ex, bytes.data(), bytes.size());
                c.data_bytes = (int64_t)bytes.size();
            } else if (pc.phys == PQ_BOOLEAN) {
                if (has_nulls) {
                    std::vector<uint8_t> full((size_t)((n + 63) / 64) * 8 + 8, 0);
                    int64_t k = 0;
                    for (int64_t i = 0; i < n; ++i)
                        if ((validity[(size_t)(i >> 3)] >> (i & 7)) & 1) {
                            if ((vals[(size_t)(k >> 3)] >> (k & 7)) & 1) full[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
                            ++k;
                        }
                    c.data = upload(ex, full.data(), full.size());
                } else {
                    if (vals_len < (size_t)((n + 7) / 8)) fail(BHIP_EEXEC, "Parquet: truncated BOOLEAN page");
                    std::vector<uint8_t> full((size_t)((n + 63) / 64) * 8 + 8, 0);
                    memcpy(full.data(), vals, (size_t)((n + 7) / 8));
                    if (n & 7) full[(size_t)(n >> 3)] &= (uint8_t)((1u << (n & 7)) - 1);
                    c.data = upload(ex, full.data(), full.size());
                }
            } else {
                if (vals_len < width * (size_t)n_valid) fail(BHIP_EEXEC, "Parquet: truncated PLAIN page of '" + pc.name + "'");
                BufferPtr dense = upload(ex, vals, width * (size_t)n_valid);
                if (has_nulls) {
                    c.data = make_buffer(ex, width * (size_t)n + 16);
                    keep.push_back(dense);
                    TIMED_LAUNCH_N(ex, "pq_scatter_valid", n, launch_pq_scatter_valid(cfg, dvalid->as<uint64_t>(), dprefix->as<uint32_t>(), dense->ptr(), (int)width, n,
                                                                                     c.data->ptr(), 0));
                } else
                    c.data = dense;
            }
        } else {
            fail(BHIP_ENOTIMPL, "Parquet: column '" + pc.name + "' uses encoding " + std::to_string(h.encoding) + " (PLAIN and RLE_DICTIONARY are read)");
        }
        pieces.push_back(c);
        rows_done += n;
        // the page buffers (`page`, `validity`, run tables) are reused by the next page: the copies above must have left the host
        stream_wait(ex);
    }
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
    return concat_batches(ex, s, parts)->cols[0];
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
            std::vector<BatchPtr> out;
            for (int fi = self->part_files_[partition].first; fi < self->part_files_[partition].second; ++fi) {
                const PqFile& F = self->files_[fi];
                std::ifstream in(F.path, std::ios::binary);
                if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + F.path);
                for (auto& g : F.groups) {
                    if (g.num_rows == 0) continue;
                    if (g.num_rows > 0xFFFFFFF0ll) fail(BHIP_ENOTIMPL, "Parquet: row group of more than 2^32 rows");
                    auto b = std::make_shared<Batch>();
                    b->schema = self->schema_;
                    b->ctx = ex.ctx;
                    b->n_rows = g.num_rows;
                    for (uint32_t ci : self->proj_) {
                        if (ci >= g.cols.size()) fail(BHIP_EEXEC, "Parquet: row group without column " + std::to_string(ci));
                        Column col = decode_chunk(ex, in, F.cols[ci], g.cols[ci], g.num_rows);
                        if (F.cols[ci].out_dtype != F.cols[ci].dtype) {                   // Int32 values of an INT_8 .. UINT_16 column
                            Column narrow = col;
                            narrow.dtype = F.cols[ci].out_dtype;
                            narrow.data = make_buffer(ex, (size_t)col.length * dtype_width(narrow.dtype) + 8);
                            TIMED_LAUNCH_N(ex, "narrow_i32", col.length, launch_narrow_i32(ex.cfg(), col.data->as<int32_t>(), col.length, dtype_width(narrow.dtype), narrow.data->ptr()));
                            col = narrow;
                        }
                        b->cols.push_back(col);
                    }
                    out.push_back(b);
                }
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
