// sort_kernels.h — launchers of kernels_sort.hip (internal C++ interface).
#pragma once
#include "util_kernels.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

size_t radix_sort_temp_bytes(int64_t n);
// bits that differ between any two keys (device u64)
// n <= small_sort_max(): stable sort of (key, value) pairs in place, one launch, no host round trip
int small_sort_max();
hipError_t small_sort_pairs(const LaunchCfg& cfg, uint64_t* keys, uint32_t* vals, int64_t n);
hipError_t radix_key_diff(const LaunchCfg& cfg, const uint64_t* keys, int64_t n, uint64_t* diff_out);
// one stable 8-bit pass on byte `byte` (0 = least significant)
hipError_t radix_pass(const LaunchCfg& cfg, const uint64_t* keys, const uint32_t* vals, int64_t n, int byte,
                      uint64_t* keys_out, uint32_t* vals_out, void* temp);

// a whole multi-key sort + gather of every column for n_rows <= ROWSORT_MAX_ROWS, one launch (kernels_sort.hip)
constexpr int ROWSORT_MAX_ROWS = 256, ROWSORT_MAX_KEYS = 8, ROWSORT_MAX_COLS = 32;
struct RowSortArgs {
    int32_t n_rows, n_keys, n_cols;
    ColumnRef key[ROWSORT_MAX_KEYS];               // most significant first
    uint8_t desc[ROWSORT_MAX_KEYS], nulls_first[ROWSORT_MAX_KEYS];
    ColumnRef col[ROWSORT_MAX_COLS];               // the columns to reorder
    uint8_t width[ROWSORT_MAX_COLS];               // bytes per value of a fixed-width column
    void* out_data[ROWSORT_MAX_COLS];
    int32_t* out_offsets[ROWSORT_MAX_COLS];        // Utf8
    uint64_t* out_validity[ROWSORT_MAX_COLS];      // nullptr: the column has no validity bitmap
};
hipError_t launch_rowsort(const LaunchCfg& cfg, const RowSortArgs& A);

// a whole multi-key sort of fixed-width keys for mid-sized inputs (2 .. bucket_sort_max_rows()): one split on the most significant
// differing bits of the composite key (about log2(n) of them) + ranks by comparison inside the bins — 8 launches instead of 5 per 8-bit pass (kernels_sort.hip)
constexpr int BSORT_MAX_WORDS = 4;
struct BucketSortKeys {
    int32_t n_words;                               // most significant first
    ColumnRef col[BSORT_MAX_WORDS];
    uint8_t null_rank[BSORT_MAX_WORDS];            // 1: the word is the NULL rank of `col` (nulls_first), 0: its value image (desc)
    uint8_t desc[BSORT_MAX_WORDS], nulls_first[BSORT_MAX_WORDS];
};
int64_t bucket_sort_max_rows();
size_t bucket_sort_temp_bytes(int64_t n, int n_words);
hipError_t bucket_sort(const LaunchCfg& cfg, const BucketSortKeys& K, int64_t n, void* temp, uint32_t* perm, uint32_t** status_dev);

hipError_t launch_sort_key_fixed(const LaunchCfg& cfg, const ColumnRef& c, const uint32_t* perm, int64_t n, bool desc, uint64_t* out);
hipError_t launch_sort_key_utf8(const LaunchCfg& cfg, const ColumnRef& c, const uint32_t* perm, int64_t n, int chunk, bool desc,
                                uint64_t* out);
hipError_t launch_sort_key_null(const LaunchCfg& cfg, const uint64_t* validity, const uint32_t* perm, int64_t n, bool nulls_first,
                                uint64_t* out);
hipError_t launch_utf8_max_len(const LaunchCfg& cfg, const int32_t* offsets, int64_t n, uint32_t* out);
// hash partitioning of fixed-width columns by ONE NULL-free Int32 / Date32 (key_width 4) or Int64 / UInt64 (8) key:
// histogram + scan + one scatter of every payload column (util_kernels.h: TakeMany, widths 1 / 4 / 8);
// first_host[0..n_parts] = partition boundaries in the scattered columns.  n_parts <= 256.
size_t partition_scatter_temp_bytes(int64_t n);
hipError_t partition_scatter(const LaunchCfg& cfg, const void* keys, int key_width, int64_t n, uint32_t n_parts, const TakeMany& cols,
                             void* temp, uint32_t* first_host);
// first_host == nullptr: only enqueues (no read-back, no wait) — for callers that already hold the partition sizes
// partition_count: counts[chunk * n_parts + p] += rows of chunk `chunk` (rows_per_chunk each, a multiple of partition_chunk_quantum())
// that go to partition p; counts zeroed by the caller
hipError_t partition_count(const LaunchCfg& cfg, const void* keys, int key_width, int64_t n, uint32_t n_parts, int64_t rows_per_chunk,
                           uint64_t* counts);
int64_t partition_chunk_quantum();
hipError_t launch_hash_to_pid(const LaunchCfg& cfg, const uint64_t* hashes, int64_t n, uint32_t n_parts, uint64_t* out);
hipError_t launch_partition_bounds(const LaunchCfg& cfg, const uint64_t* sorted_keys, int64_t n, uint32_t n_parts, uint32_t* first);

}  // namespace bhip
