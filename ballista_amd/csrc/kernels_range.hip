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
#include <type_traits>
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

    // TPI consecutive tiles per pass: every load of the pass is issued before the first comparison (a predicate over one
    // 4-byte column has only two 8-byte loads per thread and tile, too little in flight to cover HBM latency)
    // the loads of one tile
    auto load_tile = [&](const int64_t t, LeanU4 (&dst)[NRANGE][U]) {
        const int64_t row0 = t * RB_TILE;
#pragma unroll
        for (int p = 0; p < NRANGE; ++p)
            if (p < n_ranges) {
                if (r32[p]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const LeanU2 v = lean_ld2(rp[p] + row0 * 4 + u * (RB_SUB * 4) + t8);
                        dst[p][u].x = v.x; dst[p][u].y = v.y; dst[p][u].z = 0; dst[p][u].w = 0;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) dst[p][u] = lean_ld4(rp[p] + row0 * 8 + u * (RB_SUB * 8) + t16);
                }
            }
    };
    // TPI consecutive tiles per pass, already loaded into rv
    auto tiles = [&](auto tpi_c, const int64_t t0, LeanU4 (&rv)[decltype(tpi_c)::value][NRANGE][U]) {
        constexpr int TPI = decltype(tpi_c)::value;
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const int64_t t = t0 + q;
            const int64_t row0 = t * RB_TILE;
            bool live[RB_ROWS];
#pragma unroll
            for (int r = 0; r < RB_ROWS; ++r) live[r] = true;
#pragma unroll
            for (int p = 0; p < NRANGE; ++p)
                if (p < n_ranges) {
                    if (r32[p]) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double a = (double)(int32_t)rv[q][p][u].x, b = (double)(int32_t)rv[q][p][u].y;
                            live[2 * u] = live[2 * u] && a >= rlo[p] && a <= rhi[p];
                            live[2 * u + 1] = live[2 * u + 1] && b >= rlo[p] && b <= rhi[p];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double a = u2d(((uint64_t)rv[q][p][u].y << 32) | rv[q][p][u].x),
                                         b = u2d(((uint64_t)rv[q][p][u].w << 32) | rv[q][p][u].z);
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
    };
    if (NRANGE == 1) {
        // one column: four consecutive tiles per pass, every load issued before the first comparison (two 8-byte loads per thread
        // and tile are too little in flight to cover HBM latency)
        constexpr int TPI_MAIN = 4;
        const int64_t n_packs = n_tiles / TPI_MAIN;                   // packs of TPI_MAIN full tiles, then the full tiles left over
        for (int64_t g = blockIdx.x; g < n_packs; g += grid) {
            LeanU4 rv[TPI_MAIN][NRANGE][U];
#pragma unroll
            for (int q = 0; q < TPI_MAIN; ++q) load_tile(g * TPI_MAIN + q, rv[q]);
            tiles(std::integral_constant<int, TPI_MAIN>{}, g * TPI_MAIN, rv);
        }
        for (int64_t t = n_packs * TPI_MAIN + blockIdx.x; t < n_tiles; t += grid) {
            LeanU4 rv[1][NRANGE][U];
            load_tile(t, rv[0]);
            tiles(std::integral_constant<int, 1>{}, t, rv);
        }
    } else {
        // several columns (Q6's filter: 20 B per row over three): the NEXT tile of this workgroup is loaded while the current one
        // is compared, so the loads never drain between tiles
        LeanU4 cur[1][NRANGE][U], nxt[1][NRANGE][U];
        if ((int64_t)blockIdx.x < n_tiles) load_tile(blockIdx.x, cur[0]);
        for (int64_t t = blockIdx.x; t < n_tiles; t += grid) {
            if (t + grid < n_tiles) load_tile(t + grid, nxt[0]);
            tiles(std::integral_constant<int, 1>{}, t, cur);
#pragma unroll
            for (int p = 0; p < NRANGE; ++p)
#pragma unroll
                for (int u = 0; u < U; ++u) cur[0][p][u] = nxt[0][p][u];
        }
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

// ---- ONE range over ONE Int32 / Date32 column (l_shipdate > d, o_orderdate in [a, b) ...) -----------------------------------
// The kernel above is instruction-bound there (2.7 TB/s on Q3's 600 M-row filter: int -> double conversions and the bit
// interleave of the row-pair mapping per 4 bytes of input).  Here a lane owns rows base + 64 k + lane, so a ballot IS a
// bitmap word, the bounds are integers, and R32_ROWS loads per lane are in flight before the first comparison.
constexpr int R32_ROWS = 8;
__global__ void __launch_bounds__(BLOCK)
range_bitmap32_kernel(const int32_t* __restrict__ col, const uint64_t* __restrict__ validity, int32_t lo, int32_t hi, int64_t n_rows,
                      uint64_t* bitmap, uint32_t* tile_counts) {
    static_assert(SEL_TILE % (64 * R32_ROWS) == 0, "the rows of one pass of a wave lie in one selection tile");
    const int lane = threadIdx.x & 63;
    const int64_t chunk_rows = 64 * R32_ROWS;
    const int64_t n_chunks = (n_rows + chunk_rows - 1) / chunk_rows;
    const int64_t n_words = (n_rows + 63) / 64;
    const int64_t n_waves = (int64_t)gridDim.x * (BLOCK / 64);
    for (int64_t c = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); c < n_chunks; c += n_waves) {
        const int64_t base = c * chunk_rows;
        int32_t x[R32_ROWS];
        if (base + chunk_rows <= n_rows) {
#pragma unroll
            for (int k = 0; k < R32_ROWS; ++k) x[k] = col[base + 64 * k + lane];
        } else {
#pragma unroll
            for (int k = 0; k < R32_ROWS; ++k) {
                const int64_t row = base + 64 * k + lane;
                x[k] = row < n_rows ? col[row] : 0;                  // rows past the end are masked in the ballot
            }
        }
        uint64_t mine = 0;                                           // lane k keeps word k
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < R32_ROWS; ++k) {
            const int64_t row = base + 64 * k + lane;
            uint64_t word = __ballot(row < n_rows && x[k] >= lo && x[k] <= hi);
            const int64_t wi = (base >> 6) + k;
            if (validity != nullptr && wi < n_words) word &= validity[wi];
            cnt += (uint32_t)__popcll(word);
            if (lane == k) mine = word;
        }
        if (lane < R32_ROWS && (base >> 6) + lane < n_words) bitmap[(base >> 6) + lane] = mine;
        if (lane == 0 && cnt) atomicAdd(&tile_counts[base / SEL_TILE], cnt);
    }
}

// the integer image of a double range [lo, hi] over Int32 values (empty: lo > hi)
static void int_bounds(double lo, double hi, int32_t* lo_i, int32_t* hi_i) {
    *lo_i = 1; *hi_i = 0;
    if (lo != lo || hi != hi || lo > 2147483647.0 || hi < -2147483648.0) return;
    const double l = __builtin_ceil(lo), h = __builtin_floor(hi);
    *lo_i = l <= -2147483648.0 ? (int32_t)(-2147483647 - 1) : (int32_t)l;
    *hi_i = h >= 2147483647.0 ? (int32_t)2147483647 : (int32_t)h;
}

static hipError_t launch_range32(const LaunchCfg& cfg, const SopProgram& S, uint64_t* bitmap, uint32_t* tile_counts) {
    const int64_t n_tiles = (S.n_rows + SEL_TILE - 1) / SEL_TILE;
    hipError_t e = hipMemsetAsync(tile_counts, 0, (size_t)n_tiles * 4, cfg.stream);
    if (e != hipSuccess) return e;
    int32_t lo, hi;
    int_bounds(S.ranges[0].lo, S.ranges[0].hi, &lo, &hi);
    const SopColumn& c = S.cols[S.ranges[0].col];
    const int64_t n_chunks = (S.n_rows + 64 * R32_ROWS - 1) / (64 * R32_ROWS);
    int64_t grid = (int64_t)cfg.device_cus * 8;
    if (grid > (n_chunks + BLOCK / 64 - 1) / (BLOCK / 64)) grid = (n_chunks + BLOCK / 64 - 1) / (BLOCK / 64);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(range_bitmap32_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, (const int32_t*)c.data,
                       (const uint64_t*)c.validity, lo, hi, S.n_rows, bitmap, tile_counts);
    return hipGetLastError();
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

// ---- Utf8 column = / != literal -----------------------------------------------------------------------------------------
// (TPC-H Q3's c_mktsegment = 'BUILDING' over 15 M customers: 0.21 ms in the expression VM — per-row interpreter work — for 200 MB of
// offsets and bytes.)  One workgroup per 1024-row tile, a wave takes 4 x 64 rows: the two offsets, the length test (which settles
// most rows), then the bytes; a NULL row is dropped under either operator (the comparison is NULL).  Same bitmap + tile counts as
// scan_pred_bitmap_kernel.
__global__ void __launch_bounds__(256)
utf8_eq_bitmap_kernel(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ data, const uint64_t* __restrict__ validity, int64_t n,
                      const Utf8Literal lit, int negate, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile_base = (int64_t)blockIdx.x * SEL_TILE;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < SEL_TILE / 256; ++k) {
        const int64_t row0 = tile_base + wave * (SEL_TILE / 4) + k * 64, row = row0 + lane;
        bool keep = false;
        if (row < n) {
            const int32_t o0 = offsets[row], len = offsets[row + 1] - o0;
            bool eq = len == lit.len;
            if (eq) {
                const uint8_t* s = data + o0;
                if (lit.len >= 8) {
                    // 8 bytes per load, the last load overlapping the one before it (unaligned loads are one instruction on gfx9+); every
                    // load goes out before the first compare
                    uint64_t diff = 0;
                    for (int b = 0; b + 8 <= lit.len; b += 8) {
                        uint64_t x, y;
                        __builtin_memcpy(&x, s + b, 8);
                        __builtin_memcpy(&y, lit.bytes + b, 8);
                        diff |= x ^ y;
                    }
                    if (lit.len & 7) {
                        uint64_t x, y;
                        __builtin_memcpy(&x, s + lit.len - 8, 8);
                        __builtin_memcpy(&y, lit.bytes + lit.len - 8, 8);
                        diff |= x ^ y;
                    }
                    eq = diff == 0;
                } else {
                    uint32_t diff = 0;
                    for (int b = 0; b < lit.len; ++b) diff |= (uint32_t)(s[b] ^ lit.bytes[b]);
                    eq = diff == 0;
                }
            }
            const bool valid = validity == nullptr || ((validity[row >> 6] >> (row & 63)) & 1ull);
            keep = valid && (eq != (negate != 0));
        }
        const uint64_t w = __ballot(keep);
        if (lane == 0 && row0 < n) bitmap[row0 >> 6] = w;
        cnt += (uint32_t)__popcll(w);
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

hipError_t launch_utf8_eq_bitmap(const LaunchCfg& cfg, const int32_t* offsets, const void* data, const uint64_t* validity, int64_t n, const Utf8Literal& lit,
                                 bool negate, uint64_t* bitmap, uint32_t* tile_counts) {
    if (n == 0) return hipSuccess;
    if (lit.len < 0 || lit.len > (int32_t)sizeof(lit.bytes)) return hipErrorInvalidValue;
    const int64_t n_tiles = (n + SEL_TILE - 1) / SEL_TILE;
    hipLaunchKernelGGL(utf8_eq_bitmap_kernel, dim3((unsigned)n_tiles), dim3(256), 0, cfg.stream, offsets, static_cast<const uint8_t*>(data), validity, n, lit,
                       negate ? 1 : 0, bitmap, tile_counts);
    return hipGetLastError();
}

hipError_t launch_range_bitmap(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, uint64_t* bitmap, uint32_t* tile_counts) {
    if (S.n_rows == 0) return hipSuccess;
    if (S.n_ranges == 1 && S.ranges[0].is32) return launch_range32(cfg, S, bitmap, tile_counts);
    return S.n_ranges <= 1 ? launch_range_t<1>(cfg, S, dprog, bitmap, tile_counts) : launch_range_t<4>(cfg, S, dprog, bitmap, tile_counts);
}

}  // namespace bhip
