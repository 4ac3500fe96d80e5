// kernels_util.hip — data-movement kernels around the scan kernels: prefix sums, selection
// bitmap -> row indices, gathers ("take") for fixed-width / bitmap / Utf8 columns, group-table
// emission, concatenation helpers.
//
// These are the MI355X bodies of arrow-rs `filter` / `take` / `concat`, which the reference
// reaches through FilterExec, HashJoinExec, SortExec, RepartitionExec and CoalesceBatchesExec
// (rust/core/src/serde/physical_plan/from_proto.rs:81-92,122-147,253-331).  All are HBM-bound:
// sequential reads/writes are one element per lane (coalesced); gathers read at row
// granularity and write coalesced.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "util_kernels.h"
#include "vm_device.h"

namespace bhip {

// =============================================================================================
// exclusive prefix sum  out[i] = sum_{j<i} in[j]   (uint32 in, uint64/uint32/int32 out)
// three phases: per-chunk sums -> one-workgroup scan of the sums -> per-chunk local scan
// =============================================================================================
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SCAN_BLOCK * SCAN_ITEMS;   // 4096

__device__ inline uint64_t block_exclusive_scan(uint64_t v, uint64_t* s_wave, uint64_t* total) {
    // wave inclusive scan by shuffles, then 4 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)x, d, 64), hi = __shfl_up((uint32_t)(x >> 32), d, 64);
        const uint64_t y = ((uint64_t)hi << 32) | lo;
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint64_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const uint64_t s = s_wave[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    if (total) *total = tot;
    return wave_off + x - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
scan_chunk_sums_kernel(const uint32_t* in, int64_t n, uint64_t* chunk_sums) {
    __shared__ uint64_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + i * SCAN_BLOCK + threadIdx.x;
        if (j < n) v += in[j];
    }
    uint64_t tot;
    block_exclusive_scan(v, s_wave, &tot);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
scan_sums_kernel(uint64_t* chunk_sums, int64_t n_chunks, uint64_t* total_out) {
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    for (int64_t b = 0; b < n_chunks; b += SCAN_BLOCK) {
        const int64_t j = b + threadIdx.x;
        const uint64_t v = j < n_chunks ? chunk_sums[j] : 0;
        uint64_t tot;
        const uint64_t ex = block_exclusive_scan(v, s_wave, &tot);
        if (j < n_chunks) chunk_sums[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <class OutT>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_apply_kernel(const uint32_t* in, int64_t n, const uint64_t* chunk_offsets, OutT* out, int write_total) {
    __shared__ uint64_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    // thread owns SCAN_ITEMS consecutive elements so the scan order is the element order
    uint32_t x[SCAN_ITEMS];
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        x[i] = j < n ? in[j] : 0;
        sum += x[i];
    }
    uint64_t run = chunk_offsets[blockIdx.x] + block_exclusive_scan(sum, s_wave, nullptr);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        if (j < n) out[j] = (OutT)run;
        run += x[i];
        if (write_total && j == n - 1) out[n] = (OutT)run;
    }
}

// up to SCAN_TWO_LAUNCH_CHUNKS chunks: no scan_sums launch — every workgroup adds up the RAW sums of the chunks before its own (a few KB
// from L2) and the last one writes the grand total: two launches instead of three (a Q3 step runs eight scans of this size: the tile
// counts of a probe, the rank map's granules, the run flags of the aggregate, the bins of the bucket sort)
constexpr int64_t SCAN_TWO_LAUNCH_CHUNKS = 2048;
template <class OutT>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_apply_raw_kernel(const uint32_t* in, int64_t n, const uint64_t* chunk_sums, OutT* out, int write_total, uint64_t* total_out) {
    __shared__ uint64_t s_wave[4];
    uint64_t before = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += SCAN_BLOCK) before += chunk_sums[j];
    uint64_t chunk_off;
    block_exclusive_scan(before, s_wave, &chunk_off);
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    uint32_t x[SCAN_ITEMS];
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        x[i] = j < n ? in[j] : 0;
        sum += x[i];
    }
    uint64_t tot;
    uint64_t run = chunk_off + block_exclusive_scan(sum, s_wave, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        if (j < n) out[j] = (OutT)run;
        run += x[i];
        if (write_total && j == n - 1) out[n] = (OutT)run;
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = chunk_off + tot;
}

// n <= SCAN_ONE_LAUNCH_MAX: the whole scan by ONE workgroup in one launch, chunk after chunk with a running carry (group
// tables, small gathers: three launches cost more than the serial walk)
constexpr int64_t SCAN_ONE_LAUNCH_MAX = 8 * (int64_t)SCAN_CHUNK;    // ~3 us per serial chunk against ~10 us per extra launch
template <class OutT>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_one_chunk_kernel(const uint32_t* in, int64_t n, OutT* out, int write_total, uint64_t* total_out) {
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    for (int64_t base = 0; base < n; base += SCAN_CHUNK) {
        uint32_t x[SCAN_ITEMS];
        uint64_t sum = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
            x[i] = j < n ? in[j] : 0;
            sum += x[i];
        }
        uint64_t tot;
        uint64_t run = carry + block_exclusive_scan(sum, s_wave, &tot);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
            if (j < n) out[j] = (OutT)run;
            run += x[i];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        if (write_total) out[n] = (OutT)carry;
        if (total_out) *total_out = carry;
    }
}

size_t exclusive_scan_temp_bytes(int64_t n) {
    const int64_t n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    return (size_t)(n_chunks > 0 ? n_chunks : 1) * sizeof(uint64_t);
}

template <class OutT>
static hipError_t exclusive_scan_t(hipStream_t st, const uint32_t* in, int64_t n, OutT* out, bool write_total,
                                   uint64_t* total_out, void* temp) {
    uint64_t* sums = reinterpret_cast<uint64_t*>(temp);
    const int64_t n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (n_chunks == 0) {
        hipError_t e = hipSuccess;
        if (total_out) e = hipMemsetAsync(total_out, 0, sizeof(uint64_t), st);
        if (e == hipSuccess && write_total) e = hipMemsetAsync(out, 0, sizeof(OutT), st);
        return e;
    }
    if (n <= SCAN_ONE_LAUNCH_MAX) {
        hipLaunchKernelGGL((scan_one_chunk_kernel<OutT>), dim3(1), dim3(SCAN_BLOCK), 0, st, in, n, out, write_total ? 1 : 0, total_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, in, n, sums);
    if (n_chunks <= SCAN_TWO_LAUNCH_CHUNKS) {
        hipLaunchKernelGGL((scan_apply_raw_kernel<OutT>), dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, in, n, sums, out, write_total ? 1 : 0, total_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, n_chunks, total_out);
    hipLaunchKernelGGL((scan_apply_kernel<OutT>), dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, in, n, sums, out,
                       write_total ? 1 : 0);
    return hipGetLastError();
}

hipError_t exclusive_scan_u32_u64(hipStream_t st, const uint32_t* in, int64_t n, uint64_t* out, bool write_total,
                                  uint64_t* total_out, void* temp) {
    return exclusive_scan_t<uint64_t>(st, in, n, out, write_total, total_out, temp);
}
hipError_t exclusive_scan_u32_i32(hipStream_t st, const uint32_t* in, int64_t n, int32_t* out, bool write_total,
                                  uint64_t* total_out, void* temp) {
    return exclusive_scan_t<int32_t>(st, in, n, out, write_total, total_out, temp);
}
hipError_t exclusive_scan_u32_u32(hipStream_t st, const uint32_t* in, int64_t n, uint32_t* out, bool write_total,
                                  uint64_t* total_out, void* temp) {
    return exclusive_scan_t<uint32_t>(st, in, n, out, write_total, total_out, temp);
}

// =============================================================================================
// rank map of the narrow join (kernels_join.hip): bits32[g] = the build keys of granule g (32 consecutive key values, one bit
// each)  ->  pack[g] = bits32[g] | (number of build keys before granule g) << 32.  The same three phases as the scan above with
// the popcount taken on the fly and the packed word written by the last one: a probe reads ONE 8-byte word per row and has both
// the membership bit and the rank.
// =============================================================================================
// Large maps: a workgroup takes RP_CHUNK granules in rounds of 1024 — every lane one 16-byte load of four consecutive
// granules, so a wave's load covers 1 KiB contiguous (and its two 16-byte stores of four packed words 2 KiB) — with a running
// carry; the ranks inside a round come from a 32-bit block scan (a chunk holds < 2^20 keys).  The first version gave every
// thread 16 consecutive granules: each of its load and store instructions then touched 64 different cache lines, and the
// 1.5 GB map of an SF1000 order-key build (187 M granules) took 1.44 ms — a third of the rank's whole join leg.
constexpr int RP_ROUND = SCAN_BLOCK * 4;                 // 1024 granules per round
constexpr int RP_ROUNDS = 16;
constexpr int RP_CHUNK = RP_ROUND * RP_ROUNDS;           // 16384 granules = 64 KiB of key-set bits per workgroup

__device__ inline uint32_t block_exclusive_scan_u32(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const uint32_t sw = s_wave[w];
        if (w < wave) wave_off += sw;
        tot += sw;
    }
    __syncthreads();
    *total = tot;
    return wave_off + x - v;
}

__device__ inline uint4 rp_load4(const uint32_t* __restrict__ bits32, int64_t j, int64_t n) {
    // (the bits buffer is allocated in 64-bit words + 1: a 16-byte load that starts below n may read up to 12 bytes of padding
    // only when n is not a multiple of 4 — guarded element-wise there)
    if (j + 4 <= n) return *reinterpret_cast<const uint4*>(bits32 + j);
    uint4 x = make_uint4(0, 0, 0, 0);
    if (j < n) x.x = bits32[j];
    if (j + 1 < n) x.y = bits32[j + 1];
    if (j + 2 < n) x.z = bits32[j + 2];
    return x;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
rank_pack_sums_kernel(const uint32_t* __restrict__ bits32, int64_t n, uint64_t* __restrict__ chunk_sums) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * RP_CHUNK;
    uint32_t v = 0;
#pragma unroll 4
    for (int r = 0; r < RP_ROUNDS; ++r) {
        const int64_t j = base + (int64_t)r * RP_ROUND + (int64_t)threadIdx.x * 4;
        const uint4 x = rp_load4(bits32, j, n);
        v += (uint32_t)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w));
    }
    uint32_t tot;
    block_exclusive_scan_u32(v, s_wave, &tot);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

// RAW: chunk_offsets holds the chunks' own sums (no scan_sums launch in between): the workgroup adds up those before its own, the last
// one writes the grand total
template <bool RAW>
__global__ void __launch_bounds__(SCAN_BLOCK)
rank_pack_apply_kernel(const uint32_t* __restrict__ bits32, int64_t n, const uint64_t* __restrict__ chunk_offsets, uint64_t* __restrict__ pack,
                       uint64_t* __restrict__ total_out) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * RP_CHUNK;
    uint64_t carry;
    if (RAW) {
        __shared__ uint64_t s_wave64[4];
        uint64_t before = 0;
        for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += SCAN_BLOCK) before += chunk_offsets[j];
        block_exclusive_scan(before, s_wave64, &carry);
        if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = carry + chunk_offsets[blockIdx.x];
    } else {
        carry = chunk_offsets[blockIdx.x];
    }
    for (int r = 0; r < RP_ROUNDS; ++r) {
        const int64_t j = base + (int64_t)r * RP_ROUND + (int64_t)threadIdx.x * 4;
        if (base + (int64_t)r * RP_ROUND >= n) break;                                   // workgroup-uniform
        const uint4 x = rp_load4(bits32, j, n);
        const uint32_t c0 = (uint32_t)__popc(x.x), c1 = (uint32_t)__popc(x.y), c2 = (uint32_t)__popc(x.z), c3 = (uint32_t)__popc(x.w);
        uint32_t tot;
        const uint64_t run = carry + block_exclusive_scan_u32(c0 + c1 + c2 + c3, s_wave, &tot);
        const uint64_t p0 = (uint64_t)x.x | (run << 32), p1 = (uint64_t)x.y | ((run + c0) << 32), p2 = (uint64_t)x.z | ((run + c0 + c1) << 32),
                       p3 = (uint64_t)x.w | ((run + c0 + c1 + c2) << 32);
        if (j + 4 <= n) {
            ulonglong2* o = reinterpret_cast<ulonglong2*>(pack + j);
            o[0] = make_ulonglong2(p0, p1);
            o[1] = make_ulonglong2(p2, p3);
        } else {
            if (j < n) pack[j] = p0;
            if (j + 1 < n) pack[j + 1] = p1;
            if (j + 2 < n) pack[j + 2] = p2;
        }
        carry += tot;
    }
}

// small maps (<= SCAN_ONE_LAUNCH_MAX granules: a dimension table's keys) in one launch of one workgroup
__global__ void __launch_bounds__(SCAN_BLOCK)
rank_pack_one_kernel(const uint32_t* __restrict__ bits32, int64_t n, uint64_t* __restrict__ pack, uint64_t* total_out) {
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    for (int64_t base = 0; base < n; base += SCAN_CHUNK) {
        uint32_t x[SCAN_ITEMS];
        uint64_t sum = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
            x[i] = j < n ? bits32[j] : 0;
            sum += (uint32_t)__popc(x[i]);
        }
        uint64_t tot;
        uint64_t run = carry + block_exclusive_scan(sum, s_wave, &tot);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
            if (j < n) pack[j] = (uint64_t)x[i] | (run << 32);
            run += (uint32_t)__popc(x[i]);
        }
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

hipError_t launch_rank_pack(hipStream_t st, const uint32_t* bits32, int64_t n, uint64_t* pack, uint64_t* total_out, void* temp) {
    if (n <= 0) return total_out ? hipMemsetAsync(total_out, 0, sizeof(uint64_t), st) : hipSuccess;
    if (n <= SCAN_ONE_LAUNCH_MAX) {
        hipLaunchKernelGGL(rank_pack_one_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, bits32, n, pack, total_out);
        return hipGetLastError();
    }
    uint64_t* sums = reinterpret_cast<uint64_t*>(temp);                   // (exclusive_scan_temp_bytes(n) words: sized for the smaller SCAN_CHUNK)
    const int64_t n_chunks = (n + RP_CHUNK - 1) / RP_CHUNK;
    hipLaunchKernelGGL(rank_pack_sums_kernel, dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, bits32, n, sums);
    if (n_chunks <= SCAN_TWO_LAUNCH_CHUNKS) {
        hipLaunchKernelGGL(rank_pack_apply_kernel<true>, dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, bits32, n, (const uint64_t*)sums, pack, total_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, n_chunks, total_out);
    hipLaunchKernelGGL(rank_pack_apply_kernel<false>, dim3((unsigned)n_chunks), dim3(SCAN_BLOCK), 0, st, bits32, n, (const uint64_t*)sums, pack, (uint64_t*)nullptr);
    return hipGetLastError();
}

// =============================================================================================
// selection bitmap -> ascending row indices (FilterExec keeps row order)
// =============================================================================================
// One wave per 1024-row tile, no LDS, no barrier: every lane takes 16 rows (one 16-bit piece of the bitmap, loaded as such), a
// shuffle prefix sum of the popcounts gives the lane's first output position, then the lanes write the indices of their set bits
// one bit per step — as many steps as the fullest piece holds bits.  (Word by word with every lane testing its own bit was 16
// steps of ~15 instructions per tile whatever the density; a wave64 vector instruction is four cycles, and the kernel was
// bound by exactly that: 0.20 ms for the 600 M-row bitmap of the stand-alone Filter at 1.3 % density.)
__global__ void __launch_bounds__(BLOCK)
select_indices_kernel(const uint64_t* bitmap, const uint64_t* tile_offsets, int64_t n_rows, uint32_t* indices) {
    static_assert(SEL_TILE == 64 * 16, "64 lanes x 16 rows");
    const int lane = threadIdx.x & 63;
    const int64_t n_tiles = (n_rows + SEL_TILE - 1) / SEL_TILE;
    const int64_t n_pieces = (n_rows + 15) / 16;
    const uint16_t* __restrict__ pieces = reinterpret_cast<const uint16_t*>(bitmap);      // little-endian: piece q of a word = its bits 16q..16q+15
    const int64_t wave_global = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (BLOCK / 64);
    auto load_piece = [&](int64_t t) -> uint32_t {
        const int64_t pi = t * (SEL_TILE / 16) + lane;
        uint32_t bits = pi < n_pieces ? pieces[pi] : 0u;
        // rows beyond n_rows are clear by construction of the producers; mask them anyway
        if (pi == n_pieces - 1 && (n_rows & 15)) bits &= (1u << (n_rows & 15)) - 1u;
        return bits;
    };
    uint32_t bits_next = wave_global < n_tiles ? load_piece(wave_global) : 0u;
    uint64_t out_next = wave_global < n_tiles ? tile_offsets[wave_global] : 0ull;
    for (int64_t t = wave_global; t < n_tiles; t += n_waves) {
        uint32_t bits = bits_next;
        const uint64_t out_base = out_next;
        if (t + n_waves < n_tiles) {                                 // wave-uniform
            bits_next = load_piece(t + n_waves);
            out_next = tile_offsets[t + n_waves];
        }
        const uint32_t mine = (uint32_t)__popc(bits);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        uint32_t* __restrict__ out = indices + out_base + (incl - mine);
        const uint32_t first_row = (uint32_t)(t * SEL_TILE) + (uint32_t)lane * 16u;
        uint32_t k = 0;
        while (__any(bits != 0)) {
            if (bits) {
                out[k++] = first_row + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
            }
        }
    }
}

hipError_t launch_select_indices(const LaunchCfg& cfg, const uint64_t* bitmap, const uint64_t* tile_offsets,
                                 int64_t n_rows, uint32_t* indices) {
    const int64_t n_tiles = (n_rows + SEL_TILE - 1) / SEL_TILE;
    if (n_tiles == 0) return hipSuccess;
    int64_t grid = (int64_t)cfg.device_cus * 8;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(select_indices_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, bitmap, tile_offsets,
                       n_rows, indices);
    return hipGetLastError();
}

// =============================================================================================
// take (gather by row index); index 0xFFFFFFFF = NULL row (outer joins)
// =============================================================================================
constexpr uint32_t NULL_INDEX = 0xFFFFFFFFu;

template <class T>
__global__ void __launch_bounds__(BLOCK)
take_fixed_kernel(const T* src, const uint32_t* idx, int64_t n, T* dst) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint32_t j = idx[i];
        dst[i] = j == NULL_INDEX ? T(0) : src[j];
    }
}

// validity / Boolean bitmaps: out bit i = (idx != NULL) && src bit idx (src == nullptr: all set)
__global__ void __launch_bounds__(BLOCK)
take_bitmap_kernel(const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst) {
    const int64_t n_round = (n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        bool bit = false;
        if (i < n) {
            const uint32_t j = idx[i];
            if (j != NULL_INDEX) bit = src == nullptr ? true : ((src[j >> 6] >> (j & 63)) & 1ull);
        }
        const uint64_t word = __ballot(bit);
        if ((threadIdx.x & 63) == 0) dst[i >> 6] = word;
    }
}

// every fixed-width column and every bitmap of a batch in ONE launch: blockIdx.y selects the buffer
__global__ void __launch_bounds__(BLOCK)
take_many_kernel(TakeMany d, const uint32_t* idx, int64_t n) {
    const int c = blockIdx.y;
    const void* src = d.src[c];
    void* dst = d.dst[c];
    const int width = d.width[c];
    if (width == 0) {                                      // bitmap (src == nullptr: all set)
        const uint64_t* bs = reinterpret_cast<const uint64_t*>(src);
        uint64_t* bd = reinterpret_cast<uint64_t*>(dst);
        const int64_t n_round = (n + 63) & ~(int64_t)63;
        for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
            bool bit = false;
            if (i < n) {
                const uint32_t j = idx[i];
                if (j != NULL_INDEX) bit = bs == nullptr ? true : ((bs[j >> 6] >> (j & 63)) & 1ull);
            }
            const uint64_t word = __ballot(bit);
            if ((threadIdx.x & 63) == 0) bd[i >> 6] = word;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint32_t j = idx[i];
        if (width == 8) reinterpret_cast<uint64_t*>(dst)[i] = j == NULL_INDEX ? 0ull : reinterpret_cast<const uint64_t*>(src)[j];
        else if (width == 4) reinterpret_cast<uint32_t*>(dst)[i] = j == NULL_INDEX ? 0u : reinterpret_cast<const uint32_t*>(src)[j];
        else if (width == 2) reinterpret_cast<uint16_t*>(dst)[i] = j == NULL_INDEX ? (uint16_t)0 : reinterpret_cast<const uint16_t*>(src)[j];
        else reinterpret_cast<uint8_t*>(dst)[i] = j == NULL_INDEX ? (uint8_t)0 : reinterpret_cast<const uint8_t*>(src)[j];
    }
}

__global__ void __launch_bounds__(BLOCK)
take_utf8_lengths_kernel(const int32_t* offsets, const uint32_t* idx, int64_t n, uint32_t* lengths) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint32_t j = idx[i];
        lengths[i] = j == NULL_INDEX ? 0u : (uint32_t)(offsets[j + 1] - offsets[j]);
    }
}

// The 64 strings a wave gathers land back to back in the output.  They are assembled in LDS (one 8-byte load per short string,
// byte stores into LDS) and written out with coalesced 4-byte stores: ~7x fewer global-memory instructions than a byte loop per
// row for TPC-H's short names.  A wave whose strings exceed the staging area copies row by row.
constexpr int TAKE_STAGE = 2048;
struct __attribute__((packed)) Unaligned64 { uint64_t v; };
struct __attribute__((packed)) Unaligned32 { uint32_t v; };

__global__ void __launch_bounds__(BLOCK)
take_utf8_copy_kernel(const int32_t* src_off, const uint8_t* src, int64_t src_bytes, const uint32_t* idx, int64_t n,
                      const int32_t* dst_off, uint8_t* dst) {
    __shared__ uint8_t s_stage[BLOCK / 64][TAKE_STAGE];
    uint8_t* stage = s_stage[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int64_t n_round = (n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t i0 = i - lane;                                          // first row of this wave's 64
        const int64_t i1 = i0 + 64 < n ? i0 + 64 : n;
        const int32_t wave_d0 = dst_off[i0], total = dst_off[i1] - wave_d0;     // wave-uniform
        int32_t s0 = 0, len = 0, d0 = 0;
        if (i < n) {
            const uint32_t j = idx[i];
            d0 = dst_off[i];
            if (j != NULL_INDEX) { s0 = src_off[j]; len = src_off[j + 1] - s0; }
        }
        if (total <= TAKE_STAGE) {
            const int off = d0 - wave_d0;
            if (len > 0 && len <= 8 && (int64_t)s0 + 8 <= src_bytes) {
                uint64_t v = reinterpret_cast<const Unaligned64*>(src + s0)->v;
                for (int b = 0; b < len; ++b) { stage[off + b] = (uint8_t)v; v >>= 8; }
            } else {
                for (int b = 0; b < len; ++b) stage[off + b] = src[s0 + b];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint8_t* out = dst + wave_d0;
            for (int p = lane * 4; p < total; p += 256) {
                if (p + 4 <= total) {
                    uint32_t w;
                    __builtin_memcpy(&w, stage + p, 4);
                    reinterpret_cast<Unaligned32*>(out + p)->v = w;
                } else {
                    for (int b = p; b < total; ++b) out[b] = stage[b];
                }
            }
            __builtin_amdgcn_wave_barrier();                                  // the next round overwrites the staging area
        } else {
            for (int32_t b = 0; b < len; ++b) dst[d0 + b] = src[s0 + b];
        }
    }
}

// out[i] = inner[idx[i]], NULL_INDEX where idx[i] is NULL_INDEX (indices of a view column seen through another gather)
__global__ void __launch_bounds__(BLOCK)
compose_indices_kernel(const uint32_t* inner, const uint32_t* idx, int64_t n, uint32_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint32_t j = idx[i];
        out[i] = j == NULL_INDEX ? NULL_INDEX : inner[j];
    }
}

static int grid_for(const LaunchCfg& cfg, int64_t n, int per_thread = 1) {
    int64_t g = (n + (int64_t)BLOCK * per_thread - 1) / ((int64_t)BLOCK * per_thread);
    const int64_t cap = (int64_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_take_fixed(const LaunchCfg& cfg, const void* src, int width, const uint32_t* idx, int64_t n, void* dst) {
    if (n == 0) return hipSuccess;
    const int grid = grid_for(cfg, n);
    switch (width) {
        case 1: hipLaunchKernelGGL(take_fixed_kernel<uint8_t>, dim3(grid), dim3(BLOCK), 0, cfg.stream,
                                   (const uint8_t*)src, idx, n, (uint8_t*)dst); break;
        case 2: hipLaunchKernelGGL(take_fixed_kernel<uint16_t>, dim3(grid), dim3(BLOCK), 0, cfg.stream,
                                   (const uint16_t*)src, idx, n, (uint16_t*)dst); break;
        case 4: hipLaunchKernelGGL(take_fixed_kernel<uint32_t>, dim3(grid), dim3(BLOCK), 0, cfg.stream,
                                   (const uint32_t*)src, idx, n, (uint32_t*)dst); break;
        case 8: hipLaunchKernelGGL(take_fixed_kernel<uint64_t>, dim3(grid), dim3(BLOCK), 0, cfg.stream,
                                   (const uint64_t*)src, idx, n, (uint64_t*)dst); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// four-byte integers -> their low 1 or 2 bytes (Parquet INT32 pages of INT_8 / INT_16 / UINT_8 / UINT_16 columns)
__global__ void __launch_bounds__(BLOCK)
narrow_i32_kernel(const int32_t* src, int64_t n, int width, void* dst) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        if (width == 1) static_cast<uint8_t*>(dst)[i] = (uint8_t)src[i];
        else static_cast<uint16_t*>(dst)[i] = (uint16_t)src[i];
    }
}
hipError_t launch_narrow_i32(const LaunchCfg& cfg, const int32_t* src, int64_t n, int width, void* dst) {
    if (n == 0) return hipSuccess;
    if (width != 1 && width != 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(narrow_i32_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, src, n, width, dst);
    return hipGetLastError();
}
hipError_t launch_compose_indices(const LaunchCfg& cfg, const uint32_t* inner, const uint32_t* idx, int64_t n, uint32_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(compose_indices_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, inner, idx, n, out);
    return hipGetLastError();
}
hipError_t launch_take_many(const LaunchCfg& cfg, const TakeMany& d, const uint32_t* idx, int64_t n) {
    if (n == 0 || d.n == 0) return hipSuccess;
    hipLaunchKernelGGL(take_many_kernel, dim3(grid_for(cfg, n), d.n), dim3(BLOCK), 0, cfg.stream, d, idx, n);
    return hipGetLastError();
}

hipError_t launch_take_bitmap(const LaunchCfg& cfg, const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(take_bitmap_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, src, idx, n, dst);
    return hipGetLastError();
}

hipError_t launch_take_utf8_lengths(const LaunchCfg& cfg, const int32_t* offsets, const uint32_t* idx, int64_t n,
                                    uint32_t* lengths) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(take_utf8_lengths_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, offsets, idx, n, lengths);
    return hipGetLastError();
}

hipError_t launch_take_utf8_copy(const LaunchCfg& cfg, const int32_t* src_off, const uint8_t* src, int64_t src_bytes, const uint32_t* idx,
                                 int64_t n, const int32_t* dst_off, uint8_t* dst) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(take_utf8_copy_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, src_off, src, src_bytes, idx, n,
                       dst_off, dst);
    return hipGetLastError();
}

// =============================================================================================
// small helpers
// =============================================================================================
// several regions set to a 32-bit pattern each in ONE launch (a join build or a hash aggregate clears 4-6 small buffers in a row: a
// hipMemsetAsync is a launch of its own, ~4 us on the stream plus the gap before the next)
__global__ void __launch_bounds__(BLOCK)
fill_many_kernel(FillMany F) {
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x, gsize = (int64_t)gridDim.x * BLOCK;
    for (int r = 0; r < F.n; ++r) {
        uint32_t* p = static_cast<uint32_t*>(F.ptr[r]);
        const int64_t words = (int64_t)(F.bytes[r] >> 2);
        const uint32_t v = F.value[r];
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const uint4 v4 = make_uint4(v, v, v, v);
            const int64_t quads = words >> 2;
            for (int64_t i = gtid; i < quads; i += gsize) reinterpret_cast<uint4*>(p)[i] = v4;
            for (int64_t i = (quads << 2) + gtid; i < words; i += gsize) p[i] = v;
        } else {
            for (int64_t i = gtid; i < words; i += gsize) p[i] = v;
        }
    }
}
hipError_t launch_fill_many(const LaunchCfg& cfg, const FillMany& F) {
    if (F.n < 0 || F.n > FILL_MANY_MAX) return hipErrorInvalidValue;
    uint64_t most = 0;
    for (int r = 0; r < F.n; ++r) {
        if ((F.bytes[r] & 3) || (reinterpret_cast<uintptr_t>(F.ptr[r]) & 3)) return hipErrorInvalidValue;
        most = F.bytes[r] > most ? F.bytes[r] : most;
    }
    if (most == 0) return hipSuccess;
    int64_t g = (int64_t)((most / 16 + BLOCK - 1) / BLOCK);
    const int64_t cap = (int64_t)cfg.device_cus * 8;
    g = g < 1 ? 1 : (g > cap ? cap : g);
    hipLaunchKernelGGL(fill_many_kernel, dim3((unsigned)g), dim3(BLOCK), 0, cfg.stream, F);
    return hipGetLastError();
}

__global__ void __launch_bounds__(BLOCK)
iota_u32_kernel(uint32_t* out, int64_t n, uint32_t start) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        out[i] = start + (uint32_t)i;
}
hipError_t launch_iota_u32(const LaunchCfg& cfg, uint32_t* out, int64_t n, uint32_t start) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, out, n, start);
    return hipGetLastError();
}

// dst_off[i] = src_off[i] - src_off[0] + add   (i in [0, n]) : Utf8 offsets of a concatenation part
__global__ void __launch_bounds__(BLOCK)
rebase_offsets_kernel(const int32_t* src_off, int64_t n_plus_1, int32_t add, int32_t* dst_off) {
    const int32_t first = src_off[0];
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_plus_1; i += (int64_t)gridDim.x * BLOCK)
        dst_off[i] = src_off[i] - first + add;
}
hipError_t launch_rebase_offsets(const LaunchCfg& cfg, const int32_t* src_off, int64_t n_plus_1, int32_t add, int32_t* dst_off) {
    hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid_for(cfg, n_plus_1)), dim3(BLOCK), 0, cfg.stream, src_off,
                       n_plus_1, add, dst_off);
    return hipGetLastError();
}

// copy `n_bits` bits from src (starting at bit src_bit0) to dst starting at bit dst_bit0.
// One wave builds one 64-bit destination word; partial edge words are merged with atomics.
__global__ void __launch_bounds__(BLOCK)
copy_bits_kernel(const uint64_t* src, int64_t src_bit0, uint64_t* dst, int64_t dst_bit0, int64_t n_bits) {
    const int64_t first_word = dst_bit0 >> 6, last_word = (dst_bit0 + n_bits - 1) >> 6;
    const int64_t n_words = last_word - first_word + 1;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t w = wave; w < n_words; w += n_waves) {
        const int64_t dbit = ((first_word + w) << 6) + lane;
        const int64_t rel = dbit - dst_bit0;
        bool in = rel >= 0 && rel < n_bits;
        bool bit = false;
        if (in) {
            if (src == nullptr) bit = true;
            else { const int64_t sbit = src_bit0 + rel; bit = (src[sbit >> 6] >> (sbit & 63)) & 1ull; }
        }
        const uint64_t val = __ballot(bit), mask = __ballot(in);
        if (lane == 0) {
            if (mask == ~0ull) dst[first_word + w] = val;
            else {
                atomicAnd((unsigned long long*)&dst[first_word + w], (unsigned long long)~mask);
                atomicOr((unsigned long long*)&dst[first_word + w], (unsigned long long)val);
            }
        }
    }
}
hipError_t launch_copy_bits(const LaunchCfg& cfg, const uint64_t* src, int64_t src_bit0, uint64_t* dst, int64_t dst_bit0,
                            int64_t n_bits) {
    if (n_bits == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_bits_kernel, dim3(grid_for(cfg, (n_bits + 63) / 64 * 64)), dim3(BLOCK), 0, cfg.stream, src,
                       src_bit0, dst, dst_bit0, n_bits);
    return hipGetLastError();
}

// =============================================================================================
// group table -> Arrow columns
// =============================================================================================
__device__ inline uint64_t key_get(const GroupRec& g, int pos, int width) {
    uint64_t v;
    if (pos >= 8) v = g.k1 >> (8 * (pos - 8));
    else {
        v = g.k0 >> (8 * pos);
        if (pos > 0 && pos + width > 8) v |= g.k1 << (8 * (8 - pos));
    }
    if (width < 8) v &= (1ull << (8 * width)) - 1ull;
    return v;
}

// one group column; `pos` = byte position of the part inside the packed key
__global__ void __launch_bounds__(BLOCK)
emit_group_key_kernel(const GroupRec* table, int64_t n_groups, EmitKeySpec spec, void* data, uint64_t* validity,
                      uint32_t* utf8_lengths, const ScanStatus* dev_n) {
    if (dev_n) n_groups = dev_n->n_groups;                  // the host has not read the count yet (small tables, ops_agg.cpp)
    const int64_t n_round = (n_groups + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        bool valid = false;
        if (i < n_groups) {
            const GroupRec& g = table[i];
            int pos = spec.pos, width = spec.width;
            valid = true;
            if (spec.nullable) { valid = key_get(g, pos, 1) != 0; pos += 1; width -= 1; }
            if (spec.dtype == DT_UTF8) {
                if (utf8_lengths) utf8_lengths[i] = valid ? (uint32_t)key_get(g, pos, 1) : 0u;
            } else {
                uint64_t v = valid ? key_get(g, pos, width) : 0;
                if (spec.dtype != DT_BOOLEAN) dt_store(spec.dtype, data, i, v);   // (Boolean: written below as a ballot word; Float32 keys hold the double's bits)
                if (spec.dtype == DT_BOOLEAN) {
                    const uint64_t word = __ballot(v & 1);
                    if ((threadIdx.x & 63) == 0) reinterpret_cast<uint64_t*>(data)[i >> 6] = word;
                }
            }
        } else if (spec.dtype == DT_BOOLEAN) {
            const uint64_t word = __ballot(false);
            if ((threadIdx.x & 63) == 0) reinterpret_cast<uint64_t*>(data)[i >> 6] = word;
        }
        if (validity != nullptr) {
            const uint64_t vw = __ballot(valid);
            if ((threadIdx.x & 63) == 0) validity[i >> 6] = vw;
        }
    }
}

// a Utf8 key column of at most SCAN_CHUNK groups in ONE launch of one workgroup: lengths, their prefix sum, offsets, the
// bytes, the validity words and the byte total (a thread owns SCAN_ITEMS consecutive groups, so four lanes make a word)
__global__ void __launch_bounds__(SCAN_BLOCK)
emit_group_utf8_small_kernel(const GroupRec* table, int64_t n_groups, EmitKeySpec spec, uint64_t* validity, int32_t* offsets,
                             uint8_t* bytes, uint64_t* total_out, const ScanStatus* dev_n) {
    if (dev_n) n_groups = dev_n->n_groups;
    static_assert(SCAN_ITEMS == 16, "four lanes x 16 groups = one validity word");
    __shared__ uint64_t s_wave[4];
    uint32_t len[SCAN_ITEMS];
    uint64_t sum = 0;
    uint32_t vmask = 0;
    const int pos = spec.pos + (spec.nullable ? 1 : 0);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = (int64_t)threadIdx.x * SCAN_ITEMS + i;
        len[i] = 0;
        if (j < n_groups) {
            const GroupRec& g = table[j];
            const bool valid = !spec.nullable || key_get(g, spec.pos, 1) != 0;
            if (valid) { len[i] = (uint32_t)key_get(g, pos, 1); vmask |= 1u << i; }
        }
        sum += len[i];
    }
    if (validity != nullptr) {
        const uint64_t m0 = vmask;
        const uint64_t m1 = __shfl_down(vmask, 1, 64), m2 = __shfl_down(vmask, 2, 64), m3 = __shfl_down(vmask, 3, 64);
        const int64_t word = threadIdx.x >> 2;
        if ((threadIdx.x & 3) == 0 && word < (n_groups + 63) / 64) validity[word] = m0 | (m1 << 16) | (m2 << 32) | (m3 << 48);
    }
    uint64_t tot;
    uint64_t run = block_exclusive_scan(sum, s_wave, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = (int64_t)threadIdx.x * SCAN_ITEMS + i;
        if (j < n_groups) {
            offsets[j] = (int32_t)run;
            const GroupRec& g = table[j];
            for (uint32_t b = 0; b < len[i]; ++b) bytes[run + b] = (uint8_t)key_get(g, pos + 1 + (int)b, 1);
        }
        run += len[i];
    }
    if (threadIdx.x == 0) {
        offsets[n_groups] = (int32_t)tot;
        if (total_out) *total_out = tot;
    }
}

__global__ void __launch_bounds__(BLOCK)
emit_group_utf8_kernel(const GroupRec* table, int64_t n_groups, EmitKeySpec spec, const int32_t* offsets, uint8_t* bytes) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_groups; i += (int64_t)gridDim.x * BLOCK) {
        const GroupRec& g = table[i];
        int pos = spec.pos;
        if (spec.nullable) { if (key_get(g, pos, 1) == 0) continue; pos += 1; }
        const int len = (int)key_get(g, pos, 1);
        const int32_t d0 = offsets[i];
        for (int b = 0; b < len; ++b) bytes[d0 + b] = (uint8_t)key_get(g, pos + 1 + b, 1);
    }
}

// value of one output column of a group (shared by the per-column and the all-in-one emit kernels)
__device__ inline uint64_t emit_value_of(const GroupRec& g, const EmitValueSpec& spec, bool& valid) {
    const uint64_t cnt_a = spec.count_is_rows ? g.rows : g.nvalid[spec.acc_a];
    uint64_t v = 0;
    valid = true;
    switch (spec.kind) {
        case EMIT_VALUE: v = g.acc[spec.acc_a]; valid = cnt_a > 0; break;
        case EMIT_COUNT: v = cnt_a; break;
        case EMIT_ROWS: v = g.rows; break;
        case EMIT_RAW: v = g.acc[spec.acc_a]; break;
        case EMIT_AVG: valid = cnt_a > 0; v = valid ? d2u(u2d(g.acc[spec.acc_a]) / (double)cnt_a) : 0; break;
        case EMIT_AVG_ACC: {
            const uint64_t c = g.acc[spec.acc_b];
            valid = c > 0 && g.nvalid[spec.acc_a] > 0;
            v = valid ? d2u(u2d(g.acc[spec.acc_a]) / (double)c) : 0;
        } break;
        default: break;
    }
    return valid ? v : 0;
}

// all value columns of an aggregate's output in ONE launch: blockIdx.y selects the column
__global__ void __launch_bounds__(BLOCK)
emit_group_values_kernel(const GroupRec* table, int64_t n_groups, EmitValueBatch batch, const ScanStatus* dev_n) {
    if (dev_n) n_groups = dev_n->n_groups;
    const EmitValueSpec spec = batch.spec[blockIdx.y];
    void* data = batch.data[blockIdx.y];
    uint64_t* validity = batch.validity[blockIdx.y];
    const int64_t n_round = (n_groups + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        bool valid = false;
        if (i < n_groups) {
            const uint64_t v = emit_value_of(table[i], spec, valid);
            dt_store(spec.dtype, data, i, v);
        }
        if (validity != nullptr) {
            const uint64_t vw = __ballot(valid);
            if ((threadIdx.x & 63) == 0) validity[i >> 6] = vw;
        }
    }
}

__global__ void __launch_bounds__(EMIT_ALL_MAX_GROUPS)
emit_all_kernel(EmitAllArgs A) {
    __shared__ uint32_t s_wave[EMIT_ALL_MAX_GROUPS / 64];
    const int n = (int)A.status->n_groups;
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const bool in = i < n;
    const int n_words = (n + 63) >> 6;
    const GroupRec& g = A.table[in ? i : 0];          // (threads past the count read record 0 and write nothing)
    for (int c = 0; c < A.n_keys; ++c) {
        const EmitKeySpec spec = A.key[c];
        int pos = spec.pos, width = spec.width;
        bool valid = in;
        if (spec.nullable) { valid = in && key_get(g, pos, 1) != 0; pos += 1; width -= 1; }
        if (A.key_validity[c] != nullptr) {
            const uint64_t w = __ballot(valid);
            if (lane == 0 && wave < n_words) A.key_validity[c][wave] = w;
        }
        if (spec.dtype == DT_UTF8) {
            const uint32_t len = valid ? (uint32_t)key_get(g, pos, 1) : 0u;
            uint32_t x = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            if (lane == 63) s_wave[wave] = x;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < EMIT_ALL_MAX_GROUPS / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
            __syncthreads();
            const uint32_t d0 = before + x - len;
            if (in) {
                A.key_offsets[c][i] = (int32_t)d0;
                uint8_t* out = reinterpret_cast<uint8_t*>(A.key_data[c]) + d0;
                for (uint32_t b = 0; b < len; ++b) out[b] = (uint8_t)key_get(g, pos + 1 + (int)b, 1);
            }
            if (i == 0) { A.key_offsets[c][n] = (int32_t)total; *A.key_total[c] = total; }
        } else if (spec.dtype == DT_BOOLEAN) {
            const uint64_t w = __ballot(valid && (key_get(g, pos, width) & 1));
            if (lane == 0 && wave < n_words) reinterpret_cast<uint64_t*>(A.key_data[c])[wave] = w;
        } else if (in) {
            dt_store(spec.dtype, A.key_data[c], i, valid ? key_get(g, pos, width) : 0);
        }
    }
    for (int c = 0; c < A.n_values; ++c) {
        bool valid = false;
        uint64_t v = 0;
        if (in) v = emit_value_of(g, A.value[c], valid);
        if (in) dt_store(A.value[c].dtype, A.value_data[c], i, v);
        if (A.value_validity[c] != nullptr) {
            const uint64_t w = __ballot(valid);
            if (lane == 0 && wave < n_words) A.value_validity[c][wave] = w;
        }
    }
}
__global__ void __launch_bounds__(BLOCK)
emit_slots_kernel(SlotSource S, int64_t n_groups, EmitAllArgs A) {
    const int64_t n_round = (n_groups + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        const bool in = i < n_groups;
        const int64_t j = in ? i : 0;
        GroupRec g;                                      // (only its key words are ever read: key_get)
        const uint32_t o = S.head[j];
        g.k0 = S.keys128[2ull * o];
        g.k1 = S.keys128[2ull * o + 1];
        const uint64_t rows = S.rows[j];
        const uint64_t* acc = S.acc + (size_t)j * S.n_acc;
        const uint64_t* nvalid = S.nvalid ? S.nvalid + (size_t)j * S.n_acc : nullptr;
        for (int c = 0; c < A.n_keys; ++c) {
            const EmitKeySpec spec = A.key[c];
            int pos = spec.pos, width = spec.width;
            bool valid = in;
            if (spec.nullable) { valid = in && key_get(g, pos, 1) != 0; pos += 1; width -= 1; }
            if (A.key_validity[c] != nullptr) {
                const uint64_t w = __ballot(valid);
                if ((threadIdx.x & 63) == 0) A.key_validity[c][i >> 6] = w;
            }
            if (in) dt_store(spec.dtype, A.key_data[c], i, valid ? key_get(g, pos, width) : 0);
        }
        for (int c = 0; c < A.n_values; ++c) {
            const EmitValueSpec spec = A.value[c];
            // (emit_value_of, reading the slot arrays instead of a GroupRec)
            const uint64_t cnt_a = spec.count_is_rows || !nvalid ? rows : nvalid[spec.acc_a];
            bool valid = true;
            uint64_t v = 0;
            switch (spec.kind) {
                case EMIT_VALUE: v = acc[spec.acc_a]; valid = cnt_a > 0; break;
                case EMIT_COUNT: v = cnt_a; break;
                case EMIT_ROWS: v = rows; break;
                case EMIT_RAW: v = acc[spec.acc_a]; break;
                case EMIT_AVG: valid = cnt_a > 0; v = valid ? d2u(u2d(acc[spec.acc_a]) / (double)cnt_a) : 0; break;
                case EMIT_AVG_ACC: {
                    const uint64_t cn = acc[spec.acc_b];
                    valid = cn > 0 && (nvalid ? nvalid[spec.acc_a] : rows) > 0;
                    v = valid ? d2u(u2d(acc[spec.acc_a]) / (double)cn) : 0;
                } break;
                default: break;
            }
            valid = valid && in;
            if (in) dt_store(spec.dtype, A.value_data[c], i, valid ? v : 0);
            if (A.value_validity[c] != nullptr) {
                const uint64_t w = __ballot(valid);
                if ((threadIdx.x & 63) == 0) A.value_validity[c][i >> 6] = w;
            }
        }
    }
}
hipError_t launch_emit_slots(const LaunchCfg& cfg, const SlotSource& S, int64_t n_groups, const EmitAllArgs& A) {
    if (A.n_keys > EMIT_ALL_MAX_KEYS || A.n_values > EMIT_ALL_MAX_VALUES) return hipErrorInvalidValue;
    if (n_groups <= 0) return hipSuccess;
    int64_t g = (n_groups + BLOCK - 1) / BLOCK;
    const int64_t cap = (int64_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(emit_slots_kernel, dim3((unsigned)g), dim3(BLOCK), 0, cfg.stream, S, n_groups, A);
    return hipGetLastError();
}

hipError_t launch_emit_all(const LaunchCfg& cfg, const EmitAllArgs& A) {
    if (A.n_keys > EMIT_ALL_MAX_KEYS || A.n_values > EMIT_ALL_MAX_VALUES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(emit_all_kernel, dim3(1), dim3(EMIT_ALL_MAX_GROUPS), 0, cfg.stream, A);
    return hipGetLastError();
}

// ---- join / repartition key of ONE NULL-free integer column: the packed-key image the expression VM builds
// (low `width` bytes of the value in word 0, word 1 = 0), as a streaming pass
template <class T>
__global__ void __launch_bounds__(BLOCK)
widen_key_kernel(const T* src, int64_t n, uint64_t* keys128) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        ulonglong2 k;
        k.x = sizeof(T) == 4 ? (uint64_t)(uint32_t)src[i] : (uint64_t)src[i];
        k.y = 0;
        reinterpret_cast<ulonglong2*>(keys128)[i] = k;
    }
}
hipError_t launch_widen_key(const LaunchCfg& cfg, const void* src, int width, int64_t n, uint64_t* keys128) {
    if (n == 0) return hipSuccess;
    if (width == 4)
        hipLaunchKernelGGL(widen_key_kernel<uint32_t>, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, (const uint32_t*)src, n, keys128);
    else if (width == 8)
        hipLaunchKernelGGL(widen_key_kernel<uint64_t>, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, (const uint64_t*)src, n, keys128);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ void __launch_bounds__(BLOCK)
pack_fixed_keys_kernel(FixedKeyParts K, int64_t n, uint64_t* keys128) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        uint64_t k0 = 0, k1 = 0;
        for (int p = 0; p < K.n; ++p) {
            const int w = K.width[p], pos = K.pos[p];
            uint64_t v;
            if (w == 8) v = reinterpret_cast<const uint64_t*>(K.src[p])[i];
            else if (w == 4) v = reinterpret_cast<const uint32_t*>(K.src[p])[i];
            else if (w == 2) v = reinterpret_cast<const uint16_t*>(K.src[p])[i];
            else v = reinterpret_cast<const uint8_t*>(K.src[p])[i];
            if (pos < 8) {
                k0 |= v << (8 * pos);
                if (pos + w > 8) k1 |= v >> (8 * (8 - pos));
            } else {
                k1 |= v << (8 * (pos - 8));
            }
        }
        ulonglong2 k;
        k.x = k0; k.y = k1;
        reinterpret_cast<ulonglong2*>(keys128)[i] = k;
    }
}
hipError_t launch_pack_fixed_keys(const LaunchCfg& cfg, const FixedKeyParts& K, int64_t n, uint64_t* keys128) {
    if (K.n < 1 || K.n > FIXED_KEY_PARTS_MAX) return hipErrorInvalidValue;
    for (int p = 0; p < K.n; ++p)
        if ((K.width[p] != 1 && K.width[p] != 2 && K.width[p] != 4 && K.width[p] != 8) || K.pos[p] + K.width[p] > 16) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_fixed_keys_kernel, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, K, n, keys128);
    return hipGetLastError();
}

// ---- pack many small device buffers into one block (one launch + one D2H copy on export) ---------------
__global__ void __launch_bounds__(BLOCK)
pack_buffers_kernel(PackDesc d, uint8_t* out) {
    const int b = blockIdx.x;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(d.src[b]);
    uint8_t* dst = out + d.dst[b];
    for (uint32_t i = threadIdx.x; i < d.bytes[b]; i += BLOCK) dst[i] = src[i];
}
// ---- Parquet value decode -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK)
pq_expand_runs_kernel(const PqRun* __restrict__ runs, uint32_t n_runs, const uint8_t* __restrict__ page, int /*bit width: per run*/, uint32_t n_values,
                      uint32_t limit, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_values; i += gridDim.x * BLOCK) {
        uint32_t lo = 0, hi = n_runs;                   // last run with out_start <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (runs[mid].out_start <= i) lo = mid; else hi = mid;
        }
        const PqRun r = runs[lo];
        uint32_t v;
        if (!r.packed) v = r.value;
        else {
            const int bw = (int)r.packed;                 // the run's own width (pages of one chunk differ as the dictionary grows)
            const uint64_t bit = (uint64_t)(i - r.out_start) * (uint64_t)bw;
            const uint8_t* p = page + r.value + (bit >> 3);
            uint64_t w = 0;                              // up to 32 + 7 bits
            for (int b = 0; b < 5; ++b) w |= (uint64_t)p[b] << (8 * b);
            v = (uint32_t)((w >> (bit & 7)) & ((bw >= 32) ? 0xFFFFFFFFull : ((1ull << bw) - 1ull)));
        }
        out[i] = v < limit ? v : 0xFFFFFFFFu;
    }
}
hipError_t launch_pq_expand_runs(const LaunchCfg& cfg, const PqRun* runs, uint32_t n_runs, const uint8_t* page, int bit_width, uint32_t n_values,
                                 uint32_t limit, uint32_t* out) {
    if (n_values == 0 || n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(pq_expand_runs_kernel, dim3(grid_for(cfg, n_values)), dim3(BLOCK), 0, cfg.stream, runs, n_runs, page, bit_width, n_values,
                       limit, out);
    return hipGetLastError();
}

template <class T>
__global__ void __launch_bounds__(BLOCK)
pq_scatter_valid_kernel(const uint64_t* __restrict__ validity, const uint32_t* __restrict__ prefix, const T* __restrict__ src, int64_t n,
                        T* __restrict__ dst, T null_value) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t w = validity[i >> 6];
        const int b = (int)(i & 63);
        dst[i] = ((w >> b) & 1ull) ? src[prefix[i >> 6] + (uint32_t)__popcll(w & ((1ull << b) - 1ull))] : null_value;
    }
}
hipError_t launch_pq_scatter_valid(const LaunchCfg& cfg, const uint64_t* validity, const uint32_t* word_prefix, const void* src, int width,
                                   int64_t n, void* dst, int null_index) {
    if (n == 0) return hipSuccess;
    if (width == 4)
        hipLaunchKernelGGL(pq_scatter_valid_kernel<uint32_t>, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, validity, word_prefix,
                           static_cast<const uint32_t*>(src), n, static_cast<uint32_t*>(dst), null_index ? 0xFFFFFFFFu : 0u);
    else if (width == 8)
        hipLaunchKernelGGL(pq_scatter_valid_kernel<uint64_t>, dim3(grid_for(cfg, n)), dim3(BLOCK), 0, cfg.stream, validity, word_prefix,
                           static_cast<const uint64_t*>(src), n, static_cast<uint64_t*>(dst), (uint64_t)0);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// host notification without a runtime wait: copy up to 56 bytes to a pinned host slot, then publish a sequence number
// (system-scope release) the host spins on — what stands behind read_device() / stream_wait() in host/core.hpp
__global__ void publish_kernel(const uint32_t* __restrict__ src, int n_words, volatile uint32_t* payload, volatile uint64_t* seq_word, uint64_t seq) {
    for (int i = 0; i < n_words; ++i) payload[i] = src[i];
    __threadfence_system();
    *seq_word = seq;
}
hipError_t launch_publish(hipStream_t stream, const void* src, int n_words, void* slot, uint64_t seq) {
    auto w = static_cast<volatile uint64_t*>(slot);
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, stream, static_cast<const uint32_t*>(src), n_words,
                       reinterpret_cast<volatile uint32_t*>(w + 1), w, seq);
    return hipGetLastError();
}

hipError_t launch_pack_buffers(const LaunchCfg& cfg, const PackDesc& d, uint8_t* out) {
    if (d.n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_buffers_kernel, dim3(d.n), dim3(BLOCK), 0, cfg.stream, d, out);
    return hipGetLastError();
}

hipError_t launch_emit_group_key(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                 void* data, uint64_t* validity, uint32_t* utf8_lengths, const ScanStatus* dev_n) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_group_key_kernel, dim3(grid_for(cfg, n_groups)), dim3(BLOCK), 0, cfg.stream, table, n_groups,
                       spec, data, validity, utf8_lengths, dev_n);
    return hipGetLastError();
}
hipError_t launch_emit_group_utf8_small(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                        uint64_t* validity, int32_t* offsets, uint8_t* bytes, uint64_t* total_out, const ScanStatus* dev_n) {
    if (n_groups <= 0 || n_groups > EMIT_UTF8_SMALL_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(emit_group_utf8_small_kernel, dim3(1), dim3(SCAN_BLOCK), 0, cfg.stream, table, n_groups, spec, validity,
                       offsets, bytes, total_out, dev_n);
    return hipGetLastError();
}
hipError_t launch_emit_group_utf8(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                  const int32_t* offsets, uint8_t* bytes) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_group_utf8_kernel, dim3(grid_for(cfg, n_groups)), dim3(BLOCK), 0, cfg.stream, table, n_groups,
                       spec, offsets, bytes);
    return hipGetLastError();
}
hipError_t launch_emit_group_value(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups,
                                   const EmitValueSpec& spec, void* data, uint64_t* validity) {
    if (n_groups == 0) return hipSuccess;
    EmitValueBatch b;
    b.n = 1; b.spec[0] = spec; b.data[0] = data; b.validity[0] = validity;
    return launch_emit_group_values(cfg, table, n_groups, b, nullptr);
}
hipError_t launch_emit_group_values(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitValueBatch& batch,
                                    const ScanStatus* dev_n) {
    if (n_groups == 0 || batch.n == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_group_values_kernel, dim3(grid_for(cfg, n_groups), batch.n), dim3(BLOCK), 0, cfg.stream, table,
                       n_groups, batch, dev_n);
    return hipGetLastError();
}

}  // namespace bhip
