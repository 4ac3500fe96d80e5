// parquet_host.hpp — the HOST half of ParquetExec (parquet_host.cpp): everything that is a sequential byte walk over bytes that
// come from a file — the footer (Thrift compact protocol), page headers, Snappy, definition levels, the run headers of the
// RLE / bit-packed hybrid, length-prefixed strings.  No device call in here: the same code runs under the CPU sanitizer build
// (tests/c/host_fuzz.cpp) and on the decode threads of ParquetExec (parquet.cpp), whose device half consumes HostChunk.
//
// Every length, count and offset read from the file is checked against the bytes that hold it before it is used as a size, an
// index or a loop bound; recursion over nested Thrift values is capped.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../util_kernels.h"
#include "core.hpp"

namespace bhip {
namespace pq {

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
struct PqFile { std::string path; int64_t size = 0; std::vector<PqColumn> cols; std::vector<PqRowGroup> groups; int64_t num_rows = 0; };

PqFile read_footer(const std::string& path);
// the bytes of one column chunk (dictionary page first when there is one); bounds checked against the file size
std::vector<uint8_t> read_chunk_bytes(const PqFile& F, const PqChunk& ch, const std::string& column_name);

// Storage of what the walk produces.  The big buffers (value bytes, offsets, validity) come from an allocator the device half
// installs: a pool of PINNED host blocks, so that the copies to the device are truly asynchronous and a block's pages are
// faulted in once per process, not once per row group (with plain std::vector the scan spent more time in first-touch page
// faults of its ~5 GB of decoded pages than in its kernels).  Default: malloc / free — what the sanitizer harness runs on.
typedef void* (*HostAllocFn)(size_t bytes);
typedef void (*HostFreeFn)(void* p, size_t bytes);
void set_host_allocator(HostAllocFn alloc, HostFreeFn free_fn);
void* host_alloc(size_t bytes);
void host_free(void* p, size_t bytes);
template <class T>
struct HostAlloc {
    using value_type = T;
    HostAlloc() = default;
    template <class U> HostAlloc(const HostAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(host_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t n) { host_free(p, n * sizeof(T)); }
    template <class U> bool operator==(const HostAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HostAlloc<U>&) const { return false; }
};
template <class T> using HostVec = std::vector<T, HostAlloc<T>>;

// what the host walk leaves of one data page for the device
enum PageKind { PG_DICT = 0, PG_FIXED = 1, PG_STRINGS = 2, PG_BOOL = 3 };
struct HostPage {
    int64_t n = 0, n_valid = 0;
    bool has_nulls = false;
    HostVec<uint8_t> validity;          // LSB-first, padded to 64-bit words (+ 8), when the column is optional
    std::vector<uint32_t> prefix;       // set bits before each validity word, when has_nulls
    int kind = PG_FIXED;
    int bit_width = 0;                  // PG_DICT
    std::vector<PqRun> runs;            // PG_DICT: the run table of the n_valid indices; `value` of a packed run = byte offset in `bytes`
    HostVec<uint8_t> bytes;             // PG_DICT: the index bytes; PG_FIXED: n_valid dense values; PG_STRINGS: value bytes;
                                        // PG_BOOL: n bits (NULLs re-inserted as 0), padded to 64-bit words
    HostVec<int32_t> offsets;           // PG_STRINGS: n + 1 (a NULL row repeats the running offset)
};
struct HostDict {
    bool present = false;
    int64_t n = 0;
    HostVec<int32_t> offsets;           // BYTE_ARRAY dictionaries: n + 1
    HostVec<uint8_t> bytes;             // value bytes (strings) or n fixed-width values
};
// a NULL-free chunk whose pages are all of one kind leaves the walk as ONE set of buffers (the page buffers are released): the device
// half then issues one copy per chunk instead of one per page buffer (~1,600 hipMemcpyAsync calls per GiB of lineitem, 110 of the
// calling thread's 330 ms: profiles/r03_parquet_scan.txt)
enum ChunkStaging { STAGED_NONE = 0, STAGED_FIXED = 1, STAGED_DICT = 2 };
struct HostChunk {
    HostDict dict;
    std::vector<HostPage> pages;      // staged chunks keep n / n_valid of their pages, not the bytes or run tables
    int64_t rows = 0;
    int staged = STAGED_NONE;
    HostVec<uint8_t> staged_bytes;    // STAGED_FIXED: the column's values; STAGED_DICT: the index bytes of every page behind each other
    HostVec<PqRun> staged_runs;       // STAGED_DICT: ONE run table (out_start and byte offsets re-based to the chunk)
};
// pure host: headers, decompression, levels, run tables, string walks of every page of the chunk
HostChunk parse_chunk(const uint8_t* raw, size_t len, const PqColumn& pc, const PqChunk& ch, int64_t n_rows);

// footer + every page of every chunk of every supported column, no device: rows walked (the sanitizer harness's entry point)
int64_t host_walk(const std::string& path);

}  // namespace pq
}  // namespace bhip
