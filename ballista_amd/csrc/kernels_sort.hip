// kernels_sort.hip — stable LSD radix sort of (u64 key, u32 row id) pairs and the key
// normalisation that turns SortExec's lexicographic multi-key order into unsigned integer order.
//
// SortExec (rust/core/src/serde/physical_plan/from_proto.rs:291-331) needs: all rows of the single
// input partition, PhysicalSortExpr{expr, descending, nulls_first} per key.  The same stable pass
// also implements RepartitionExec(Hash)'s split (key = partition id) and keeps input order inside
// every partition.
//
// One 8-bit pass = histogram -> exclusive scan -> scatter.  Ranks inside a workgroup come from
// wave64 ballots (8 ballots give the lanes holding the same digit), so the pass is stable without
// any sorting network.  HBM traffic per pass: read keys+ids twice, write once (12 B x 3 per row);
// passes whose byte is the same in every key are skipped (one OR-reduction per key tells).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "sort_kernels.h"
#include "util_kernels.h"
#include "vm_device.h"

namespace bhip {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_ITEMS = 16;                       // sub-tiles per chunk
constexpr int SORT_CHUNK = SORT_BLOCK * SORT_ITEMS;  // 4096 rows per workgroup

// ITEMS sub-tiles of 256 rows per workgroup: 16 for large inputs; 4 for up to 2 Mi rows, where a pass is latency-bound (three
// barriers and a dependent load per sub-tile) and 4x the workgroups hide it (tools/exp_sort.py, two keys: 200 K rows 0.45 -> 0.33 ms,
// 1.13 M rows 0.60 -> 0.52 ms, 4 M rows 1.08 -> 1.17 ms)
constexpr int SORT_ITEMS_SMALL = 4;
static int sort_items_for(int64_t n) {
    static const int forced = [] { const char* v = getenv("BHIP_SORT_ITEMS"); return v ? atoi(v) : 0; }();       // 4 | 16: A/B
    if (forced == SORT_ITEMS_SMALL || forced == SORT_ITEMS) return forced;
    return n <= (1ll << 21) ? SORT_ITEMS_SMALL : SORT_ITEMS;
}

template <int ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
radix_hist_kernel(const uint64_t* keys, int64_t n, int shift, uint32_t* hist, int n_blocks) {
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (SORT_BLOCK * ITEMS);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t j = base + i * SORT_BLOCK + threadIdx.x;
        if (j < n) atomicAdd(&s_hist[(keys[j] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];   // digit-major
}

template <int ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
radix_scatter_kernel(const uint64_t* keys, const uint32_t* vals, int64_t n, int shift, const uint32_t* offsets,
                     int n_blocks, uint64_t* keys_out, uint32_t* vals_out) {
    __shared__ uint32_t s_base[256];            // next output position of each digit for this workgroup
    __shared__ uint32_t s_wave[4][256];         // per-wave digit counts of the current sub-tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_base[tid] = offsets[(size_t)tid * n_blocks + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * (SORT_BLOCK * ITEMS);
    for (int i = 0; i < ITEMS; ++i) {
        for (int w = 0; w < 4; ++w) s_wave[w][tid] = 0;
        __syncthreads();
        const int64_t j = base + i * SORT_BLOCK + tid;
        const bool in = j < n;
        uint64_t key = 0;
        uint32_t val = 0;
        uint32_t digit = 0;
        if (in) { key = keys[j]; val = vals[j]; digit = (uint32_t)(key >> shift) & 0xFF; }
        // lanes of this wave holding the same digit
        uint64_t same = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((digit >> b) & 1);
            same &= ((digit >> b) & 1) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (in && rank == 0) s_wave[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (in) {
            uint32_t pos = s_base[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += s_wave[w][digit];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        s_base[tid] += s_wave[0][tid] + s_wave[1][tid] + s_wave[2][tid] + s_wave[3][tid];
        __syncthreads();
    }
}

// ---- hash partitioning of fixed-width columns in two passes over the key column + one over the payload --------
// RepartitionExec(Hash(key), n) for ONE NULL-free integer key (every exchange of the TPC-H joins): the partition id
// (row hash % n) is recomputed from the key column in both passes instead of being materialised, and
// the scatter pass writes every payload column straight to its partition-contiguous position — each byte is read
// once and written once, where sort-by-partition-id + one gather per partition re-reads every cache line of every
// column once per partition.  Rows keep their input order inside a partition (same stable ranks as the radix pass).
template <int KEYW>
__device__ inline uint32_t partition_of(const void* keys, int64_t j, uint32_t n_parts) {
    uint64_t bits;
    if (KEYW == 4) bits = (uint64_t)(int64_t) reinterpret_cast<const int32_t*>(keys)[j];      // sign-extended
    else bits = reinterpret_cast<const uint64_t*>(keys)[j];
    return (uint32_t)(mix64(bits) % n_parts);                                                  // h = mix64(0 ^ bits)
}

template <int KEYW>
__global__ void __launch_bounds__(SORT_BLOCK)
partition_hist_kernel(const void* keys, int64_t n, uint32_t n_parts, uint32_t* hist, int n_blocks) {
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t j = base + i * SORT_BLOCK + threadIdx.x;
        if (j < n) atomicAdd(&s_hist[partition_of<KEYW>(keys, j, n_parts)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];   // partition-major
}

template <int KEYW>
__global__ void __launch_bounds__(SORT_BLOCK)
partition_scatter_kernel(const void* keys, int64_t n, uint32_t n_parts, const uint32_t* offsets, int n_blocks, TakeMany cols) {
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_wave[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_base[tid] = offsets[(size_t)tid * n_blocks + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
    for (int i = 0; i < SORT_ITEMS; ++i) {
        for (int w = 0; w < 4; ++w) s_wave[w][tid] = 0;
        __syncthreads();
        const int64_t j = base + i * SORT_BLOCK + tid;
        const bool in = j < n;
        const uint32_t digit = in ? partition_of<KEYW>(keys, j, n_parts) : 0u;
        uint64_t same = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((digit >> b) & 1);
            same &= ((digit >> b) & 1) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (in && rank == 0) s_wave[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (in) {
            uint32_t pos = s_base[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += s_wave[w][digit];
            for (int c = 0; c < cols.n; ++c) {
                const int w = cols.width[c];
                if (w == 8) reinterpret_cast<uint64_t*>(cols.dst[c])[pos] = reinterpret_cast<const uint64_t*>(cols.src[c])[j];
                else if (w == 4) reinterpret_cast<uint32_t*>(cols.dst[c])[pos] = reinterpret_cast<const uint32_t*>(cols.src[c])[j];
                else if (w == 2) reinterpret_cast<uint16_t*>(cols.dst[c])[pos] = reinterpret_cast<const uint16_t*>(cols.src[c])[j];
                else reinterpret_cast<uint8_t*>(cols.dst[c])[pos] = reinterpret_cast<const uint8_t*>(cols.src[c])[j];
            }
        }
        __syncthreads();
        s_base[tid] += s_wave[0][tid] + s_wave[1][tid] + s_wave[2][tid] + s_wave[3][tid];
        __syncthreads();
    }
}

// counts[(row / rows_per_chunk) * n_parts + partition] += 1 for every row: the streaming shuffle (host/exchange.cpp) sizes every
// chunk's sends and every receive buffer from this matrix before a single payload byte moves.  rows_per_chunk is a multiple of
// SORT_CHUNK, so a workgroup's rows belong to one chunk: an LDS histogram, then n_parts global atomics per workgroup.
template <int KEYW>
__global__ void __launch_bounds__(SORT_BLOCK)
partition_count_kernel(const void* keys, int64_t n, uint32_t n_parts, int64_t rows_per_chunk, unsigned long long* counts) {
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t j = base + i * SORT_BLOCK + threadIdx.x;
        if (j < n) atomicAdd(&s_hist[partition_of<KEYW>(keys, j, n_parts)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_parts && s_hist[threadIdx.x])
        atomicAdd(&counts[(size_t)(base / rows_per_chunk) * n_parts + threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

hipError_t partition_count(const LaunchCfg& cfg, const void* keys, int key_width, int64_t n, uint32_t n_parts, int64_t rows_per_chunk,
                           uint64_t* counts) {
    if (n_parts == 0 || n_parts > 256 || (key_width != 4 && key_width != 8) || rows_per_chunk <= 0 || rows_per_chunk % SORT_CHUNK) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const int n_blocks = (int)((n + SORT_CHUNK - 1) / SORT_CHUNK);
    if (key_width == 4) hipLaunchKernelGGL(partition_count_kernel<4>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, rows_per_chunk, (unsigned long long*)counts);
    else hipLaunchKernelGGL(partition_count_kernel<8>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, rows_per_chunk, (unsigned long long*)counts);
    return hipGetLastError();
}
int64_t partition_chunk_quantum() { return SORT_CHUNK; }

size_t partition_scatter_temp_bytes(int64_t n) {
    const int64_t n_blocks = (n + SORT_CHUNK - 1) / SORT_CHUNK;
    const size_t hist = (size_t)256 * (n_blocks > 0 ? n_blocks : 1) * 4;
    return 2 * hist + exclusive_scan_temp_bytes(256 * n_blocks) + 64;
}

// first[p] (host, n_parts + 1 entries) = first output row of partition p
hipError_t partition_scatter(const LaunchCfg& cfg, const void* keys, int key_width, int64_t n, uint32_t n_parts, const TakeMany& cols,
                             void* temp, uint32_t* first_host) {
    if (n_parts == 0 || n_parts > 256 || (key_width != 4 && key_width != 8)) return hipErrorInvalidValue;
    if (first_host) for (uint32_t p = 0; p <= n_parts; ++p) first_host[p] = 0;
    if (n == 0) return hipSuccess;
    const int n_blocks = (int)((n + SORT_CHUNK - 1) / SORT_CHUNK);
    const size_t hist_elems = (size_t)256 * n_blocks;
    uint32_t* hist = reinterpret_cast<uint32_t*>(temp);
    uint32_t* offsets = hist + hist_elems;
    void* scan_tmp = offsets + hist_elems + 4;
    if (key_width == 4) hipLaunchKernelGGL(partition_hist_kernel<4>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, hist, n_blocks);
    else hipLaunchKernelGGL(partition_hist_kernel<8>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, hist, n_blocks);
    hipError_t e = exclusive_scan_u32_u32(cfg.stream, hist, (int64_t)hist_elems, offsets, false, nullptr, scan_tmp);
    if (e != hipSuccess) return e;
    if (key_width == 4) hipLaunchKernelGGL(partition_scatter_kernel<4>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, offsets, n_blocks, cols);
    else hipLaunchKernelGGL(partition_scatter_kernel<8>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, n_parts, offsets, n_blocks, cols);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (!first_host) return hipSuccess;              // the caller knows the partition sizes (partition_count): nothing to read back, no wait
    // offsets[p * n_blocks] = rows of partitions < p
    e = hipMemcpy2DAsync(first_host, 4, offsets, (size_t)n_blocks * 4, 4, n_parts, hipMemcpyDeviceToHost, cfg.stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(cfg.stream);
    first_host[n_parts] = (uint32_t)n;
    return e;
}

// OR over all keys of (key XOR first key): bits that differ somewhere
__global__ void __launch_bounds__(SORT_BLOCK)
key_diff_kernel(const uint64_t* keys, int64_t n, uint64_t* out) {
    const uint64_t first = keys[0];
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK)
        acc |= keys[i] ^ first;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_down((uint32_t)acc, d, 64), hi = __shfl_down((uint32_t)(acc >> 32), d, 64);
        acc |= ((uint64_t)hi << 32) | lo;
    }
    // one atomic per workgroup: thousands of same-address atomics serialise (~12 ns each: 96 us for 1.1 M keys before)
    __shared__ uint64_t s_acc[SORT_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t all = 0;
        for (int w = 0; w < SORT_BLOCK / 64; ++w) all |= s_acc[w];
        if (all) atomicOr((unsigned long long*)out, (unsigned long long)all);
    }
}

size_t radix_sort_temp_bytes(int64_t n) {
    const int64_t chunk = (int64_t)SORT_BLOCK * sort_items_for(n);
    const int64_t n_blocks = (n + chunk - 1) / chunk;
    const size_t hist = (size_t)256 * (n_blocks > 0 ? n_blocks : 1) * 4;
    return 2 * hist + exclusive_scan_temp_bytes(256 * n_blocks) + 64;
}

hipError_t radix_key_diff(const LaunchCfg& cfg, const uint64_t* keys, int64_t n, uint64_t* diff_out) {
    hipError_t e = hipMemsetAsync(diff_out, 0, 8, cfg.stream);
    if (e != hipSuccess || n == 0) return e;
    int64_t g = (n + SORT_BLOCK - 1) / SORT_BLOCK;
    if (g > (int64_t)cfg.device_cus * 4) g = (int64_t)cfg.device_cus * 4;
    hipLaunchKernelGGL(key_diff_kernel, dim3((unsigned)g), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, diff_out);
    return hipGetLastError();
}

// ---- small inputs: one workgroup, stable rank sort in LDS, in place ----------------------------------
// (a Q1 / Q5 result is 4-5 rows: eight radix passes of three launches each plus a host round trip for
// the byte-difference mask would cost ~100x the work; rank = #smaller + #equal-and-earlier is stable)
constexpr int SMALL_SORT_MAX = 1024;
__global__ void __launch_bounds__(SORT_BLOCK)
small_sort_pairs_kernel(uint64_t* keys, uint32_t* vals, int n) {
    __shared__ uint64_t k[SMALL_SORT_MAX];
    __shared__ uint32_t v[SMALL_SORT_MAX];
    for (int i = threadIdx.x; i < n; i += SORT_BLOCK) { k[i] = keys[i]; v[i] = vals[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += SORT_BLOCK) {
        const uint64_t ki = k[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const uint64_t kj = k[j];                       // same address for the whole wave: LDS broadcast
            rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
        keys[rank] = ki;
        vals[rank] = v[i];
    }
}

int small_sort_max() { return SMALL_SORT_MAX; }

hipError_t small_sort_pairs(const LaunchCfg& cfg, uint64_t* keys, uint32_t* vals, int64_t n) {
    if (n <= 1) return hipSuccess;
    if (n > SMALL_SORT_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(small_sort_pairs_kernel, dim3(1), dim3(SORT_BLOCK), 0, cfg.stream, keys, vals, (int)n);
    return hipGetLastError();
}

// one stable pass on byte `byte` of the keys: (keys, vals) -> (keys_out, vals_out)
hipError_t radix_pass(const LaunchCfg& cfg, const uint64_t* keys, const uint32_t* vals, int64_t n, int byte,
                      uint64_t* keys_out, uint32_t* vals_out, void* temp) {
    if (n == 0) return hipSuccess;
    const bool small = sort_items_for(n) == SORT_ITEMS_SMALL;
    const int64_t chunk = (int64_t)SORT_BLOCK * sort_items_for(n);
    const int n_blocks = (int)((n + chunk - 1) / chunk);
    const size_t hist_elems = (size_t)256 * n_blocks;
    uint32_t* hist = reinterpret_cast<uint32_t*>(temp);
    uint32_t* offsets = hist + hist_elems;
    void* scan_tmp = offsets + hist_elems + 4;
    if (small) hipLaunchKernelGGL(radix_hist_kernel<SORT_ITEMS_SMALL>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, byte * 8, hist, n_blocks);
    else hipLaunchKernelGGL(radix_hist_kernel<SORT_ITEMS>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, n, byte * 8, hist, n_blocks);
    hipError_t e = exclusive_scan_u32_u32(cfg.stream, hist, (int64_t)hist_elems, offsets, false, nullptr, scan_tmp);
    if (e != hipSuccess) return e;
    if (small)
        hipLaunchKernelGGL(radix_scatter_kernel<SORT_ITEMS_SMALL>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, vals, n, byte * 8,
                           offsets, n_blocks, keys_out, vals_out);
    else
        hipLaunchKernelGGL(radix_scatter_kernel<SORT_ITEMS>, dim3(n_blocks), dim3(SORT_BLOCK), 0, cfg.stream, keys, vals, n, byte * 8,
                           offsets, n_blocks, keys_out, vals_out);
    return hipGetLastError();
}

// ---- key normalisation ---------------------------------------------------------------------------
// order-preserving u64 image of a fixed-width value: unsigned order of the images == the type's order
// (floats: total order by sign-magnitude flip, -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN)
__device__ inline uint64_t fixed_key_image(const ColumnRef& c, uint32_t row) {
    constexpr uint64_t SIGN = 0x8000000000000000ull;
    switch (c.dtype) {
        case DT_INT8: return (uint64_t)(int64_t) reinterpret_cast<const int8_t*>(c.data)[row] ^ SIGN;
        case DT_INT16: return (uint64_t)(int64_t) reinterpret_cast<const int16_t*>(c.data)[row] ^ SIGN;
        case DT_INT32:
        case DT_DATE32: return (uint64_t)(int64_t) reinterpret_cast<const int32_t*>(c.data)[row] ^ SIGN;
        case DT_INT64:
        case DT_DATE64:
        case DT_TIMESTAMP_S:
        case DT_TIMESTAMP_MS:
        case DT_TIMESTAMP_US:
        case DT_TIMESTAMP_NS: return reinterpret_cast<const uint64_t*>(c.data)[row] ^ SIGN;
        case DT_UINT8: return reinterpret_cast<const uint8_t*>(c.data)[row];
        case DT_UINT16: return reinterpret_cast<const uint16_t*>(c.data)[row];
        case DT_UINT32: return reinterpret_cast<const uint32_t*>(c.data)[row];
        case DT_UINT64: return reinterpret_cast<const uint64_t*>(c.data)[row];
        case DT_BOOLEAN: return (reinterpret_cast<const uint8_t*>(c.data)[row >> 3] >> (row & 7)) & 1u;
        case DT_FLOAT32: {
            const uint32_t b = reinterpret_cast<const uint32_t*>(c.data)[row];
            return (b >> 31) ? (uint32_t)~b : (b | 0x80000000u);
        }
        default: {   // Float64
            const uint64_t b = reinterpret_cast<const uint64_t*>(c.data)[row];
            return (b >> 63) ? ~b : (b | SIGN);
        }
    }
}
__device__ inline bool row_valid(const uint64_t* validity, uint32_t row) {
    return validity == nullptr || ((validity[row >> 6] >> (row & 63)) & 1ull);
}

// out[i] = order-preserving u64 image of column value at row perm[i]
__global__ void __launch_bounds__(SORT_BLOCK)
sort_key_fixed_kernel(ColumnRef c, const uint32_t* perm, int64_t n, int descending, uint64_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint32_t row = perm[i];
        uint64_t k = fixed_key_image(c, row);
        if (!row_valid(c.validity, row)) k = 0;   // ties among NULLs
        out[i] = descending ? ~k : k;
    }
}

static int sgrid(const LaunchCfg& cfg, int64_t n);

// ---- mid-sized inputs: ONE split on the most significant DIFFERING bits of the composite key, then ranks inside the (tiny) bins -----
// (the result of a high-cardinality aggregate — TPC-H Q3 sorts 1.13 M groups by revenue DESC, o_orderdate: the LSD path is 15 passes
// x 5 launches for them, ~40 us each however few rows there are.)  The composite key of a row = its WORDS, most significant first:
// per sort expression an optional NULL-rank word (the column has a validity bitmap) and the value's order-preserving image
// (sort_key_fixed_kernel's, 0 for NULL rows) — unsigned lexicographic order of the words, ties by row number, IS the order the
// stable LSD passes produce.  Bits that are the same in every row carry no order, so the split digit is made of the highest
// bits that differ somewhere (log2(n) of them, 12 .. 20: about one bin per row, since images of floats leave most exponent patterns
// unused — Q3's revenues spend 5 of the bits on exponents, a third of the rows sit in one octave; a low-cardinality first key gives
// its few bits and the digit continues at the top of the next word).  Rows are dropped into their digit's bin through an atomic
// cursor (any order: the row number in the comparison makes the result unique); a row's place inside its bin = the number of the
// bin's rows that sort before it, counted by comparing with each (bins hold a handful of rows; a first version sorted segments of
// bins with a bitonic network in LDS: 0.48 ms for Q3's groups, all of it LDS traffic of 28-byte records through 66 exchange steps).
// A bin larger than BSORT_BIN_MAX rows (heavily repeated leading bits: NULLs, a few distinct values) raises `status` and the caller
// falls back to the LSD passes, which are short exactly then.
constexpr int BSORT_MIN_BITS = 12, BSORT_MAX_BITS = 20;
constexpr int BSORT_BIN_MAX = 512;

__device__ inline uint64_t bsort_word(const BucketSortKeys& K, int w, uint32_t row) {
    const ColumnRef& c = K.col[w];
    const bool valid = row_valid(c.validity, row);
    if (K.null_rank[w]) return K.nulls_first[w] ? (valid ? 1u : 0u) : (valid ? 0u : 1u);
    uint64_t k = valid ? fixed_key_image(c, row) : 0ull;     // ties among NULLs
    return K.desc[w] ? ~k : k;
}

// words[w * n + i] = word w of row i; diff[w] |= bits of word w that differ from row 0's
template <int W>
__global__ void __launch_bounds__(SORT_BLOCK)
bsort_words_kernel(BucketSortKeys K, int64_t n, uint64_t* words, unsigned long long* diff) {
    uint64_t first[W], acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { first[w] = bsort_word(K, w, 0u); acc[w] = 0; }
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t x = bsort_word(K, w, (uint32_t)i);
            words[(size_t)w * n + i] = x;
            acc[w] |= x ^ first[w];
        }
    }
    __shared__ uint64_t s_acc[SORT_BLOCK / 64][W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint64_t a = acc[w];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = __shfl_down((uint32_t)a, d, 64), hi = __shfl_down((uint32_t)(a >> 32), d, 64);
            a |= ((uint64_t)hi << 32) | lo;
        }
        if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6][w] = a;
    }
    __syncthreads();
    if (threadIdx.x < W) {
        uint64_t all = 0;
        for (int v = 0; v < SORT_BLOCK / 64; ++v) all |= s_acc[v][threadIdx.x];
        // thousands of same-address atomics serialise (55 us for Q3's 1.13 M rows before): only a workgroup that adds a bit sends one
        // (a stale read costs a redundant atomic, never a missing bit)
        if (all & ~__hip_atomic_load(&diff[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&diff[threadIdx.x], (unsigned long long)all);
    }
}

// mask[w] = the bits of word w that belong to the split digit: the `bits` most significant set bits of diff[0..W)
template <int W>
__device__ inline void bsort_digit_masks(const unsigned long long* diff, uint64_t* s_mask, int bits) {
    if (threadIdx.x == 0) {
        int left = bits;
        for (int w = 0; w < W; ++w) {
            uint64_t d = diff[w], m = 0;
            while (d && left > 0) {
                const int b = 63 - __clzll((long long)d);
                m |= 1ull << b;
                d &= ~(1ull << b);
                --left;
            }
            s_mask[w] = m;
        }
    }
    __syncthreads();
}
template <int W>
__device__ inline uint32_t bsort_digit(const uint64_t* x, const uint64_t* s_mask) {
    uint32_t digit = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint64_t m = s_mask[w];                       // the same in every lane: a scalar loop
        while (m) {
            const int b = 63 - __clzll((long long)m);
            digit = (digit << 1) | (uint32_t)((x[w] >> b) & 1ull);
            m &= ~(1ull << b);
        }
    }
    return digit;
}

template <int W>
__global__ void __launch_bounds__(SORT_BLOCK)
bsort_hist_kernel(const uint64_t* words, int64_t n, const unsigned long long* diff, int bits, uint32_t* bins, uint32_t* digits) {
    __shared__ uint64_t s_mask[W];
    bsort_digit_masks<W>(diff, s_mask, bits);
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        uint64_t x[W];
#pragma unroll
        for (int w = 0; w < W; ++w) x[w] = words[(size_t)w * n + i];
        const uint32_t d = bsort_digit<W>(x, s_mask);
        digits[i] = d;
        atomicAdd(&bins[d], 1u);
    }
}

// row i -> position first[digit] + (rows of the bin that arrived before it): ONE record of W + 1 words {words, row | digit << 32} per row
// (five separate scattered 4- and 8-byte stores per row, each a partial line, were 98 us for Q3's 1.13 M rows)
template <int W>
__global__ void __launch_bounds__(SORT_BLOCK)
bsort_scatter_kernel(const uint64_t* words, const uint32_t* digits, int64_t n, const uint32_t* first, uint32_t* fill, uint64_t* records) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint32_t d = digits[i];
        const uint32_t pos = first[d] + atomicAdd(&fill[d], 1u);
        uint64_t* r = records + (size_t)pos * (W + 1);
#pragma unroll
        for (int w = 0; w < W; ++w) r[w] = words[(size_t)w * n + i];
        r[W] = (uint64_t)(uint32_t)i | ((uint64_t)d << 32);
    }
}

// the row at position p of the binned order goes to first[bin] + (rows of its bin that sort before it)
template <int W>
__global__ void __launch_bounds__(SORT_BLOCK)
bsort_rank_kernel(const uint64_t* records, int64_t n, const uint32_t* first, const uint32_t* bins, uint32_t* perm, uint32_t* status) {
    for (int64_t p = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint64_t* mine = records + (size_t)p * (W + 1);
        const uint64_t tag = mine[W];
        const uint32_t d = (uint32_t)(tag >> 32), me = (uint32_t)tag, s = first[d], c = bins[d];
        if (c == 1) { perm[s] = me; continue; }
        if (c > (uint32_t)BSORT_BIN_MAX) { status[0] = 1u; continue; }
        uint64_t x[W];
#pragma unroll
        for (int w = 0; w < W; ++w) x[w] = mine[w];
        uint32_t before = 0;
        for (uint32_t q = s; q < s + c; ++q) {
            const uint64_t* other = records + (size_t)q * (W + 1);
            bool lt = (uint32_t)other[W] < me;             // ties keep input order
#pragma unroll
            for (int w = W - 1; w >= 0; --w) {
                const uint64_t y = other[w];
                lt = y != x[w] ? y < x[w] : lt;
            }
            before += lt ? 1u : 0u;
        }
        perm[s + before] = me;
    }
}

int64_t bucket_sort_max_rows() { return 1ll << 22; }
static int bsort_bits_for(int64_t n) {
    int bits = BSORT_MIN_BITS;
    while (bits < BSORT_MAX_BITS && (1ll << bits) < n) ++bits;
    return bits;
}
size_t bucket_sort_temp_bytes(int64_t n, int n_words) {
    const size_t words = ((size_t)n_words * n * 8 + 63) & ~(size_t)63, records = ((size_t)(n_words + 1) * n * 8 + 63) & ~(size_t)63;
    const size_t n_bins = (size_t)1 << bsort_bits_for(n);
    return words + records + (((size_t)n * 4 + 63) & ~(size_t)63)                             // words, records, digits
           + n_bins * 4 * 3 + exclusive_scan_temp_bytes((int64_t)n_bins) + 512;               // first, bins, fill, scan, diff + status
}

// perm[i] = the row at position i of the sorted order.  *status_dev (device, 4 bytes inside temp) != 0 afterwards: a bin overflowed, `perm`
// is incomplete and the caller must sort some other way.  Only enqueues.
hipError_t bucket_sort(const LaunchCfg& cfg, const BucketSortKeys& K, int64_t n, void* temp, uint32_t* perm, uint32_t** status_dev) {
    if (K.n_words < 1 || K.n_words > BSORT_MAX_WORDS || n < 2 || n > bucket_sort_max_rows()) return hipErrorInvalidValue;
    const int W = K.n_words, bits = bsort_bits_for(n);
    const uint32_t n_bins = 1u << bits;
    uint8_t* p = reinterpret_cast<uint8_t*>(temp);
    auto carve = [&](size_t bytes) { uint8_t* q = p; p += (bytes + 63) & ~(size_t)63; return q; };
    uint64_t* words = reinterpret_cast<uint64_t*>(carve((size_t)W * n * 8));
    uint64_t* records = reinterpret_cast<uint64_t*>(carve((size_t)(W + 1) * n * 8));
    uint32_t* digits = reinterpret_cast<uint32_t*>(carve((size_t)n * 4));
    uint32_t* first = reinterpret_cast<uint32_t*>(carve((size_t)n_bins * 4));
    void* scan_tmp = carve(exclusive_scan_temp_bytes((int64_t)n_bins));
    // zeroed together: bins, fill, diff, status
    uint8_t* zero0 = p;
    uint32_t* bins = reinterpret_cast<uint32_t*>(carve((size_t)n_bins * 4));
    uint32_t* fill = reinterpret_cast<uint32_t*>(carve((size_t)n_bins * 4));
    unsigned long long* diff = reinterpret_cast<unsigned long long*>(carve(BSORT_MAX_WORDS * 8));
    uint32_t* status = reinterpret_cast<uint32_t*>(carve(4));
    hipError_t e = hipMemsetAsync(zero0, 0, (size_t)(p - zero0), cfg.stream);
    if (e != hipSuccess) return e;
    *status_dev = status;
    const int g = sgrid(cfg, n);
#define BSORT_HEAD(W_)                                                                                                                 \
    {                                                                                                                                  \
        hipLaunchKernelGGL(bsort_words_kernel<W_>, dim3(g), dim3(SORT_BLOCK), 0, cfg.stream, K, n, words, diff);                       \
        hipLaunchKernelGGL(bsort_hist_kernel<W_>, dim3(g), dim3(SORT_BLOCK), 0, cfg.stream, words, n, diff, bits, bins, digits);       \
    }
#define BSORT_TAIL(W_)                                                                                                                 \
    {                                                                                                                                  \
        hipLaunchKernelGGL(bsort_scatter_kernel<W_>, dim3(g), dim3(SORT_BLOCK), 0, cfg.stream, words, digits, n, first, fill, records);  \
        hipLaunchKernelGGL(bsort_rank_kernel<W_>, dim3(g), dim3(SORT_BLOCK), 0, cfg.stream, records, n, first, bins, perm, status);      \
    }
    switch (W) {
        case 1: BSORT_HEAD(1) break;
        case 2: BSORT_HEAD(2) break;
        case 3: BSORT_HEAD(3) break;
        default: BSORT_HEAD(4) break;
    }
    if ((e = exclusive_scan_u32_u32(cfg.stream, bins, (int64_t)n_bins, first, false, nullptr, scan_tmp)) != hipSuccess) return e;
    switch (W) {
        case 1: BSORT_TAIL(1) break;
        case 2: BSORT_TAIL(2) break;
        case 3: BSORT_TAIL(3) break;
        default: BSORT_TAIL(4) break;
    }
#undef BSORT_HEAD
#undef BSORT_TAIL
    return hipGetLastError();
}

// ---- a whole SortExec over at most ROWSORT_MAX_ROWS rows in ONE launch of one workgroup -------------------------------
// (the result of a low-cardinality aggregate: TPC-H Q1 sorts 4 rows, Q5 5 — the general path is ~20 launches for them).
// Thread i owns input row i: its output position is the number of rows that sort before it under the lexicographic
// (key, descending, nulls_first) order, ties by input position — the same stable order the LSD passes produce.  Then thread r
// writes output row r of every column (fixed width, Boolean and validity bits by ballot, Utf8 through a block prefix sum).
__device__ inline int rowsort_cmp_key(const ColumnRef& c, uint32_t a, uint32_t b) {   // ascending, both rows valid
    if (c.dtype == DT_UTF8) {
        const int32_t oa = c.offsets[a], ob = c.offsets[b];
        const int la = c.offsets[a + 1] - oa, lb = c.offsets[b + 1] - ob;
        const uint8_t* sa = reinterpret_cast<const uint8_t*>(c.data) + oa;
        const uint8_t* sb = reinterpret_cast<const uint8_t*>(c.data) + ob;
        const int m = la < lb ? la : lb;
        for (int p = 0; p < m; ++p)
            if (sa[p] != sb[p]) return sa[p] < sb[p] ? -1 : 1;
        // one is a prefix of the other: zero-padded chunks first, then the length (sort_key_utf8_kernel)
        for (int p = m; p < la; ++p) if (sa[p] != 0) return 1;
        for (int p = m; p < lb; ++p) if (sb[p] != 0) return -1;
        return la < lb ? -1 : la > lb ? 1 : 0;
    }
    const uint64_t ka = fixed_key_image(c, a), kb = fixed_key_image(c, b);
    return ka < kb ? -1 : ka > kb ? 1 : 0;
}

__global__ void __launch_bounds__(ROWSORT_MAX_ROWS)
rowsort_kernel(RowSortArgs A) {
    __shared__ uint32_t s_src[ROWSORT_MAX_ROWS];
    __shared__ uint32_t s_wave[ROWSORT_MAX_ROWS / 64];
    const int n = A.n_rows, i = threadIdx.x, lane = i & 63, wave = i >> 6;
    if (i < n) {
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            int c = 0;
            for (int k = 0; k < A.n_keys && c == 0; ++k) {
                const ColumnRef& kc = A.key[k];
                const bool vj = row_valid(kc.validity, (uint32_t)j), vi = row_valid(kc.validity, (uint32_t)i);
                if (vj && vi) {
                    c = rowsort_cmp_key(kc, (uint32_t)j, (uint32_t)i);
                    if (A.desc[k]) c = -c;
                } else if (vj != vi) {
                    c = (!vj) == (A.nulls_first[k] != 0) ? -1 : 1;      // the NULL row goes first iff nulls_first
                }
            }
            rank += (c < 0 || (c == 0 && j < i)) ? 1 : 0;
        }
        s_src[rank] = (uint32_t)i;
    }
    __syncthreads();
    const bool in = i < n;
    const uint32_t src = in ? s_src[i] : 0u;
    const int n_words = (n + 63) >> 6;
    for (int ci = 0; ci < A.n_cols; ++ci) {
        const ColumnRef& c = A.col[ci];
        const bool valid = in && row_valid(c.validity, src);
        if (A.out_validity[ci] != nullptr) {
            const uint64_t w = __ballot(valid);
            if (lane == 0 && wave < n_words) A.out_validity[ci][wave] = w;
        }
        if (c.dtype == DT_UTF8) {
            const int32_t o0 = in ? c.offsets[src] : 0;
            const uint32_t len = in ? (uint32_t)(c.offsets[src + 1] - o0) : 0u;
            uint32_t x = len;                      // inclusive scan inside the wave, then the wave totals
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            if (lane == 63) s_wave[wave] = x;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < ROWSORT_MAX_ROWS / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
            __syncthreads();
            const uint32_t d0 = before + x - len;
            if (in) {
                A.out_offsets[ci][i] = (int32_t)d0;
                const uint8_t* s = reinterpret_cast<const uint8_t*>(c.data) + o0;
                uint8_t* d = reinterpret_cast<uint8_t*>(A.out_data[ci]) + d0;
                for (uint32_t b = 0; b < len; ++b) d[b] = s[b];
            }
            if (i == 0) A.out_offsets[ci][n] = (int32_t)total;
        } else if (c.dtype == DT_BOOLEAN) {
            const bool bit = in && ((reinterpret_cast<const uint8_t*>(c.data)[src >> 3] >> (src & 7)) & 1u);
            const uint64_t w = __ballot(bit);
            if (lane == 0 && wave < n_words) reinterpret_cast<uint64_t*>(A.out_data[ci])[wave] = w;
        } else if (in) {
            switch (A.width[ci]) {
                case 1: reinterpret_cast<uint8_t*>(A.out_data[ci])[i] = reinterpret_cast<const uint8_t*>(c.data)[src]; break;
                case 2: reinterpret_cast<uint16_t*>(A.out_data[ci])[i] = reinterpret_cast<const uint16_t*>(c.data)[src]; break;
                case 4: reinterpret_cast<uint32_t*>(A.out_data[ci])[i] = reinterpret_cast<const uint32_t*>(c.data)[src]; break;
                default: reinterpret_cast<uint64_t*>(A.out_data[ci])[i] = reinterpret_cast<const uint64_t*>(c.data)[src]; break;
            }
        }
    }
}

hipError_t launch_rowsort(const LaunchCfg& cfg, const RowSortArgs& A) {
    if (A.n_rows < 1 || A.n_rows > ROWSORT_MAX_ROWS || A.n_keys > ROWSORT_MAX_KEYS || A.n_cols > ROWSORT_MAX_COLS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rowsort_kernel, dim3(1), dim3(ROWSORT_MAX_ROWS), 0, cfg.stream, A);
    return hipGetLastError();
}

// Utf8: chunk `chunk` = bytes [8*chunk, 8*chunk+8) big-endian, zero padded; chunk == -1: the length
__global__ void __launch_bounds__(SORT_BLOCK)
sort_key_utf8_kernel(ColumnRef c, const uint32_t* perm, int64_t n, int chunk, int descending, uint64_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint32_t row = perm[i];
        const int32_t o0 = c.offsets[row], len = c.offsets[row + 1] - o0;
        uint64_t k = 0;
        if (chunk < 0) k = (uint64_t)(uint32_t)len;
        else {
            const uint8_t* s = reinterpret_cast<const uint8_t*>(c.data) + o0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int p = chunk * 8 + b;
                k = (k << 8) | (p < len ? s[p] : 0u);
            }
        }
        if (c.validity != nullptr && !((c.validity[row >> 6] >> (row & 63)) & 1ull)) k = 0;
        out[i] = descending ? ~k : k;
    }
}

// null rank: nulls_first -> NULL = 0, valid = 1 ; nulls last -> NULL = 1, valid = 0
__global__ void __launch_bounds__(SORT_BLOCK)
sort_key_null_kernel(const uint64_t* validity, const uint32_t* perm, int64_t n, int nulls_first, uint64_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint32_t row = perm[i];
        const bool valid = (validity[row >> 6] >> (row & 63)) & 1ull;
        out[i] = nulls_first ? (valid ? 1 : 0) : (valid ? 0 : 1);
    }
}

__global__ void __launch_bounds__(SORT_BLOCK)
utf8_max_len_kernel(const int32_t* offsets, int64_t n, uint32_t* out) {
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
        m = len > m ? len : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// partition id = hash % n  as a sort key
__global__ void __launch_bounds__(SORT_BLOCK)
hash_to_pid_kernel(const uint64_t* hashes, int64_t n, uint32_t n_parts, uint64_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SORT_BLOCK)
        out[i] = hashes[i] % n_parts;
}

// first[p] = first position whose (sorted) key is >= p, for p in [0, n_parts]
__global__ void __launch_bounds__(SORT_BLOCK)
partition_bounds_kernel(const uint64_t* sorted_keys, int64_t n, uint32_t n_parts, uint32_t* first) {
    for (int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; i <= n; i += (int64_t)gridDim.x * SORT_BLOCK) {
        const uint64_t cur = i < n ? sorted_keys[i] : n_parts;
        const uint64_t prev = i > 0 ? sorted_keys[i - 1] + 1 : 0;
        for (uint64_t p = prev; p <= cur && p <= n_parts; ++p) first[p] = (uint32_t)i;
    }
}

static int sgrid(const LaunchCfg& cfg, int64_t n) {
    int64_t g = (n + SORT_BLOCK - 1) / SORT_BLOCK;
    const int64_t cap = (int64_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_sort_key_fixed(const LaunchCfg& cfg, const ColumnRef& c, const uint32_t* perm, int64_t n, bool desc, uint64_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sort_key_fixed_kernel, dim3(sgrid(cfg, n)), dim3(SORT_BLOCK), 0, cfg.stream, c, perm, n, desc ? 1 : 0, out);
    return hipGetLastError();
}
hipError_t launch_sort_key_utf8(const LaunchCfg& cfg, const ColumnRef& c, const uint32_t* perm, int64_t n, int chunk, bool desc,
                                uint64_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sort_key_utf8_kernel, dim3(sgrid(cfg, n)), dim3(SORT_BLOCK), 0, cfg.stream, c, perm, n, chunk, desc ? 1 : 0, out);
    return hipGetLastError();
}
hipError_t launch_sort_key_null(const LaunchCfg& cfg, const uint64_t* validity, const uint32_t* perm, int64_t n, bool nulls_first,
                                uint64_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sort_key_null_kernel, dim3(sgrid(cfg, n)), dim3(SORT_BLOCK), 0, cfg.stream, validity, perm, n,
                       nulls_first ? 1 : 0, out);
    return hipGetLastError();
}
hipError_t launch_utf8_max_len(const LaunchCfg& cfg, const int32_t* offsets, int64_t n, uint32_t* out) {
    hipError_t e = hipMemsetAsync(out, 0, 4, cfg.stream);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(utf8_max_len_kernel, dim3(sgrid(cfg, n)), dim3(SORT_BLOCK), 0, cfg.stream, offsets, n, out);
    return hipGetLastError();
}
hipError_t launch_hash_to_pid(const LaunchCfg& cfg, const uint64_t* hashes, int64_t n, uint32_t n_parts, uint64_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hash_to_pid_kernel, dim3(sgrid(cfg, n)), dim3(SORT_BLOCK), 0, cfg.stream, hashes, n, n_parts, out);
    return hipGetLastError();
}
hipError_t launch_partition_bounds(const LaunchCfg& cfg, const uint64_t* sorted_keys, int64_t n, uint32_t n_parts, uint32_t* first) {
    hipLaunchKernelGGL(partition_bounds_kernel, dim3(sgrid(cfg, n + 1)), dim3(SORT_BLOCK), 0, cfg.stream, sorted_keys, n, n_parts, first);
    return hipGetLastError();
}

}  // namespace bhip
