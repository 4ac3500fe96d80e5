// kernels_range.hip — FilterExec's predicate pass for predicates that are an AND of `column <op> literal`
// comparisons over NULL-free Int32 / Date32 / Float64 columns (reference operator: FilterExec,
// rust/core/src/serde/physical_plan/from_proto.rs:81-92; TPC-H: every filter of Q1 / Q3 / Q6 and the date
// filters of Q5).  Same plan table as the aggregate fast path (sop.h: one [lo, hi] range per column, built by
// host/sop.cpp), same wide-load row mapping as lean_kernel.h: a thread owns two consecutive rows of each
// 512-row sub-tile, so a Float64 predicate column is one global_load_dwordx4 per row pair.
//
// Output = the selection bitmap (bit i = row i kept) + the number of kept rows of each 1024-row tile — the
// interface of the expression-VM kernel it stands in for (kernels_scan.hip::scan_pred_bitmap_kernel), so
// the index pass and the gathers after it are unchanged.  Algorithmic bytes: the predicate columns once
// (Q6: 28 B/row) + 1 bit/row written.
#include "lean_kernel.h"

namespace bhip {

// bit i of x -> bit 2i
__device__ inline uint64_t spread32(uint64_t x) {
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// own tile constants: the selection interface fixes 1024-row tiles whatever lean_kernel.h is tuned to
constexpr int RB_U = 2;
constexpr int RB_SUB = BLOCK * 2;
constexpr int RB_TILE = RB_SUB * RB_U;
constexpr int RB_ROWS = 2 * RB_U;

template <int NRANGE>
__global__ void __launch_bounds__(BLOCK)
range_bitmap_kernel(const SopProgram* __restrict__ Sp, uint64_t* bitmap, uint32_t* tile_counts) {
    static_assert(RB_TILE == SEL_TILE, "selection tiles are 1024 rows");
    const SopProgram& S = *Sp;
    constexpr int U = RB_U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ranges = S.n_ranges;
    const int64_t n_rows = S.n_rows;
    const int64_t n_tiles = n_rows / RB_TILE;                 // full tiles
    const int64_t grid = gridDim.x;

    const BHIP_GLOBAL char* rp[NRANGE];
    const BHIP_GLOBAL uint64_t* rvp[NRANGE];   // validity bitmap or null: a NULL fails every comparison
    bool r32[NRANGE];
    double rlo[NRANGE], rhi[NRANGE];
#pragma unroll
    for (int p = 0; p < NRANGE; ++p) {
        rp[p] = nullptr; rvp[p] = nullptr; r32[p] = false; rlo[p] = -__builtin_huge_val(); rhi[p] = __builtin_huge_val();
        if (p < n_ranges) {
            r32[p] = S.ranges[p].is32 != 0;
            rp[p] = (const BHIP_GLOBAL char*)S.cols[S.ranges[p].col].data;
            rvp[p] = (const BHIP_GLOBAL uint64_t*)S.cols[S.ranges[p].col].validity;
            rlo[p] = S.ranges[p].lo; rhi[p] = S.ranges[p].hi;
        }
    }
    const uint32_t t8 = (uint32_t)tid * 8u, t16 = (uint32_t)tid * 16u;

    for (int64_t t = blockIdx.x; t < n_tiles; t += grid) {
        const int64_t row0 = t * RB_TILE;
        LeanU4 rv[NRANGE][U];
#pragma unroll
        for (int p = 0; p < NRANGE; ++p)
            if (p < n_ranges) {
                if (r32[p]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const LeanU2 v = lean_ld2(rp[p] + row0 * 4 + u * (RB_SUB * 4) + t8);
                        rv[p][u].x = v.x; rv[p][u].y = v.y; rv[p][u].z = 0; rv[p][u].w = 0;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) rv[p][u] = lean_ld4(rp[p] + row0 * 8 + u * (RB_SUB * 8) + t16);
                }
            }
        bool live[RB_ROWS];
#pragma unroll
        for (int r = 0; r < RB_ROWS; ++r) live[r] = true;
#pragma unroll
        for (int p = 0; p < NRANGE; ++p)
            if (p < n_ranges) {
                if (r32[p]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double a = (double)(int32_t)rv[p][u].x, b = (double)(int32_t)rv[p][u].y;
                        live[2 * u] = live[2 * u] && a >= rlo[p] && a <= rhi[p];
                        live[2 * u + 1] = live[2 * u + 1] && b >= rlo[p] && b <= rhi[p];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double a = u2d(((uint64_t)rv[p][u].y << 32) | rv[p][u].x), b = u2d(((uint64_t)rv[p][u].w << 32) | rv[p][u].z);
                        live[2 * u] = live[2 * u] && a >= rlo[p] && a <= rhi[p];
                        live[2 * u + 1] = live[2 * u + 1] && b >= rlo[p] && b <= rhi[p];
                    }
                }
                if (rvp[p]) {           // the word of rows [64k, 64k + 64) holding this lane's pair (32 lanes share it)
                    const uint32_t bit = (2u * (uint32_t)tid) & 63u;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint64_t w = rvp[p][(row0 >> 6) + u * (RB_SUB / 64) + (tid >> 5)];
                        live[2 * u] = live[2 * u] && ((w >> bit) & 1ull);
                        live[2 * u + 1] = live[2 * u + 1] && ((w >> (bit + 1u)) & 1ull);
                    }
                }
            }
        // lane l holds rows 2l, 2l+1 of this wave's 128-row span: interleave the two ballots into row order
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t b0 = __ballot(live[2 * u]), b1 = __ballot(live[2 * u + 1]);
            const uint64_t w_lo = spread32(b0) | (spread32(b1) << 1);
            const uint64_t w_hi = spread32(b0 >> 32) | (spread32(b1 >> 32) << 1);
            cnt += (uint32_t)(__popcll(b0) + __popcll(b1));
            if (lane == 0) {
                uint64_t* out = bitmap + (row0 >> 6) + u * (RB_SUB / 64) + wave * 2;
                out[0] = w_lo;
                out[1] = w_hi;
            }
        }
        if (lane == 0 && cnt) atomicAdd(&tile_counts[t], cnt);
    }

    // ragged tail (< 1024 rows): one row per lane, the ballot is already in row order
    if ((int64_t)blockIdx.x == n_tiles % grid && n_tiles * RB_TILE < n_rows) {
        const int64_t tail0 = n_tiles * RB_TILE;
        uint32_t cnt = 0;
        for (int k = 0; k < RB_TILE / BLOCK; ++k) {
            const int64_t i = tail0 + (int64_t)k * BLOCK + tid;
            bool ok = i < n_rows;
            if (ok) {
#pragma unroll
                for (int p = 0; p < NRANGE; ++p)
                    if (p < n_ranges) {
                        const double x = r32[p] ? (double)*(const BHIP_GLOBAL int32_t*)(rp[p] + i * 4) : *(const BHIP_GLOBAL double*)(rp[p] + i * 8);
                        ok = ok && x >= rlo[p] && x <= rhi[p];
                        if (rvp[p]) ok = ok && ((rvp[p][i >> 6] >> (i & 63)) & 1ull);
                    }
            }
            const uint64_t word = __ballot(ok);
            const int64_t first = tail0 + (int64_t)k * BLOCK + wave * 64;      // first row of this wave's word
            if (lane == 0 && first < n_rows) bitmap[first >> 6] = word;
            cnt += (uint32_t)__popcll(word);
        }
        if (lane == 0 && cnt) atomicAdd(&tile_counts[n_tiles], cnt);
    }
}

template <int NRANGE>
static hipError_t launch_range_t(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, uint64_t* bitmap, uint32_t* tile_counts) {
    auto k = range_bitmap_kernel<NRANGE>;
    const int64_t n_tiles = (S.n_rows + RB_TILE - 1) / RB_TILE;
    hipError_t e = hipMemsetAsync(tile_counts, 0, (size_t)n_tiles * 4, cfg.stream);
    if (e != hipSuccess) return e;
    int64_t grid = (int64_t)cfg.device_cus * 8;
    if (grid > n_tiles) grid = n_tiles;
    if (grid < 1) grid = 1;
    e = hipMemcpyAsync(dprog, &S, sizeof(SopProgram), hipMemcpyHostToDevice, cfg.stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, (const SopProgram*)dprog, bitmap, tile_counts);
    return hipGetLastError();
}

hipError_t launch_range_bitmap(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, uint64_t* bitmap, uint32_t* tile_counts) {
    if (S.n_rows == 0) return hipSuccess;
    return S.n_ranges <= 1 ? launch_range_t<1>(cfg, S, dprog, bitmap, tile_counts) : launch_range_t<4>(cfg, S, dprog, bitmap, tile_counts);
}

}  // namespace bhip
