// lean_kernel.h — the wide-load variant of the fused scan + aggregate fast path (template;
// instantiated per (GMAX, NSTEP, NRANGE) in kernels_lean_g{1,4}.hip).
//
// Same operator chain and plan table as sop_kernel.h (FilterExec -> HashAggregateExec, reference
// operators at rust/core/src/serde/physical_plan/from_proto.rs:81-92,173-252; table: sop.h), for the
// sub-shape that covers TPC-H Q1 / Q6 and flag / status / small-integer GROUP BYs in general:
//   * every chain factor reads a Float64 column,
//   * at most two group-key parts, each a 32-bit integer / date or a Utf8 string of <= 3 bytes
//     (a longer string raises SCAN_ERR_KEY_TOO_LONG and the host reruns the batch on sop_kernel.h),
//   * at most 4 groups per workgroup (SCAN_OVERFLOW_GROUPS -> sop_kernel.h with 8 -> hash path).
// host/sop.cpp::lean_eligible decides; results are identical by construction (same per-row arithmetic,
// same fixed reduction order), only the instruction count differs.  What changed against sop_kernel.h,
// each point sized on its ISA and measured (profiles/r01_shapes_120M_rows_lean_vs_sop.txt):
//   * a thread owns TWO CONSECUTIVE rows of each 512-row sub-tile, so one global_load_dwordx4 brings
//     both Float64 values of a column (a wave reads 1 KiB contiguous), one dwordx2 both dates, one
//     dwordx2 + dword the three Arrow offsets of both strings, and one unaligned 8-byte load the bytes
//     of both strings: 4.5 load instructions per row instead of 11;
//   * the main loop runs over FULL tiles only — no per-load bounds predicate; the ragged tail is one
//     scalar-style pass by one workgroup;
//   * group accumulators live in LDS, private to each thread ([group][step][thread]); a row adds to
//     its group's slots with ds_add_f64 — NSTEP + 1 LDS instructions per row whatever the number of
//     groups, instead of GMAX exec-masked blocks of NSTEP + 1 VALU adds; the adds of one thread reach
//     its private slots in program order, so sums stay run-to-run deterministic.  Without GROUP BY
//     (GMAX = 1) the accumulators stay in registers;
//   * keys are one 64-bit word ([len][<=3 bytes] or the integer per part): the lookup in the
//     register-cached key table is one v_cmp_eq_u64 per group;
//   * chain steps are evaluated in place, step-major, so each wave-uniform plan flag costs one scalar
//     branch per tile instead of one per row.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "launch_common.h"
#include "reduce_device.h"
#include "sop.h"
#include "vm_device.h"

namespace bhip {

#ifndef LEAN_U_VALUE
#define LEAN_U_VALUE 2
#endif
constexpr int LEAN_U = LEAN_U_VALUE;            // sub-tiles per loop iteration
constexpr int LEAN_SUB = BLOCK * 2;             // rows per sub-tile (two consecutive rows per thread)
constexpr int LEAN_TILE = LEAN_SUB * LEAN_U;    // 1024 rows
constexpr int LEAN_ROWS = 2 * LEAN_U;           // rows per thread and tile
constexpr uint32_t LEAN_MAX_STR = 3;

template <int GMAX, int NSTEP>
struct LeanLds {
    static constexpr int NG = GMAX > 1 ? GMAX : 0;
    double acc[NG * NSTEP * BLOCK + 1];         // [group][step][thread]
    uint32_t cnt[NG * BLOCK + 1];               // [group][thread]
    uint64_t keys[AGG_GMAX];
    uint32_t ng, lock, overflow, pad;
    uint64_t red[4][GMAX * NSTEP];
    uint64_t rowred[4][GMAX];
    uint64_t rowtot[GMAX];
};

struct LeanU3 { uint32_t a, b, c; };
struct alignas(8) LeanU2 { uint32_t x, y; };
struct alignas(16) LeanU4 { uint32_t x, y, z, w; };

// member-wise loads through the global address space (merged into one dwordx2 / dwordx4 by the alignment)
__device__ inline LeanU2 lean_ld2(const BHIP_GLOBAL char* p) {
    const BHIP_GLOBAL LeanU2* q = (const BHIP_GLOBAL LeanU2*)p;
    LeanU2 v;
    v.x = q->x; v.y = q->y;
    return v;
}
__device__ inline LeanU4 lean_ld4(const BHIP_GLOBAL char* p) {
    const BHIP_GLOBAL LeanU4* q = (const BHIP_GLOBAL LeanU4*)p;
    LeanU4 v;
    v.x = q->x; v.y = q->y; v.z = q->z; v.w = q->w;
    return v;
}

__device__ inline uint64_t lean_uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// [len][bytes] image of a short string whose bytes start at bit 0 of `raw`
__device__ inline uint32_t lean_str_word(uint32_t raw, uint32_t len) {
    return (__builtin_amdgcn_ubfe(raw, 0u, len << 3) << 8) | len;
}

template <int GMAX, int NSTEP, int NRANGE>
__global__ void __launch_bounds__(BLOCK, (GMAX == 1 && NSTEP <= 5) ? 4 : 3)
scan_agg_lean_kernel(const SopProgram* __restrict__ Sp, GroupRec* partials, uint32_t* partial_ng, ScanStatus* status) {
    const SopProgram& S = *Sp;
    constexpr int U = LEAN_U;
    constexpr int NKEY = 2;
    __shared__ LeanLds<GMAX, NSTEP> lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- resolve the plan into wave-uniform registers
    const int n_ranges = S.n_ranges, n_keys = S.n_keys, n_steps = S.n_steps;
    const int64_t n_rows = S.n_rows;
    const int64_t n_tiles = n_rows / LEAN_TILE;                 // full tiles
    const int64_t grid = gridDim.x;
    const int64_t row0 = (int64_t)blockIdx.x * LEAN_TILE;
    const int64_t stride = grid * LEAN_TILE;

    const BHIP_GLOBAL char* rp[NRANGE];
    const BHIP_GLOBAL uint64_t* rvp[NRANGE];   // validity of a range column, or null (host: only range columns may have one)
    bool r32[NRANGE];
    double rlo[NRANGE], rhi[NRANGE];
#pragma unroll
    for (int p = 0; p < NRANGE; ++p) {
        rp[p] = nullptr; rvp[p] = nullptr; r32[p] = false; rlo[p] = -__builtin_huge_val(); rhi[p] = __builtin_huge_val();
        if (p < n_ranges) {
            r32[p] = S.ranges[p].is32 != 0;
            rp[p] = (const BHIP_GLOBAL char*)S.cols[S.ranges[p].col].data + row0 * (r32[p] ? 4 : 8);
            if (S.cols[S.ranges[p].col].validity) rvp[p] = (const BHIP_GLOBAL uint64_t*)S.cols[S.ranges[p].col].validity + row0 / 64;
            rlo[p] = S.ranges[p].lo; rhi[p] = S.ranges[p].hi;
        }
    }
    const BHIP_GLOBAL char* kp[NKEY];          // Int32 / Date32 key: values.  Utf8 key: the OFFSETS
    const BHIP_GLOBAL char* kdat[NKEY];        // Utf8 key: bytes (absolute offsets)
    bool kutf[NKEY];
#pragma unroll
    for (int q = 0; q < NKEY; ++q) {
        kp[q] = nullptr; kdat[q] = nullptr; kutf[q] = false;
        if (q < n_keys) {
            const SopColumn c = S.cols[S.keys[q].col];
            kutf[q] = S.keys[q].kind == SOP_KEY_UTF8;
            kp[q] = (kutf[q] ? (const BHIP_GLOBAL char*)c.offsets : (const BHIP_GLOBAL char*)c.data) + row0 * 4;
            kdat[q] = (const BHIP_GLOBAL char*)c.data;
        }
    }
    const BHIP_GLOBAL char* xp[NSTEP];
    bool xstart[NSTEP], xplain[NSTEP];
    uint32_t xflip[NSTEP];                     // sign bit of sgn: f = (sgn * x) + add, sgn = +-1
    double xadd[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        xp[s] = nullptr; xstart[s] = true; xplain[s] = true; xflip[s] = 0; xadd[s] = 0.0;
        if (s < n_steps) {
            const SopStep st = S.steps[s];
            xstart[s] = st.start != 0;
            xplain[s] = st.sgn == 1.0 && st.add == 0.0 && __builtin_signbit(st.add);   // 1.0 * x + (-0.0) == x
            xflip[s] = st.sgn < 0.0 ? 0x80000000u : 0u;
            xadd[s] = st.add;
            xp[s] = (const BHIP_GLOBAL char*)S.cols[st.col].data + row0 * 8;
        }
    }

    // ---- accumulators
    double acc[NSTEP];                         // GMAX == 1 only
    uint32_t rows1 = 0;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) acc[s] = 0.0;
    if constexpr (GMAX > 1) {
#pragma unroll
        for (int j = 0; j < GMAX * NSTEP; ++j) lds.acc[j * BLOCK + tid] = 0.0;
#pragma unroll
        for (int j = 0; j < GMAX; ++j) lds.cnt[j * BLOCK + tid] = 0;
    }
    if (tid == 0) { lds.ng = 0; lds.overflow = 0; lds.lock = 0; }
    if (tid < AGG_GMAX) lds.keys[tid] = 0;
    __syncthreads();
    volatile uint32_t* v_ng = &lds.ng;
    volatile uint32_t* v_over = &lds.overflow;
    uint32_t bad_len = 0;
    int ng_c = 0;                              // register copy of the workgroup's key table
    uint64_t gk[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) gk[g] = 0;

    // group slot of every row (-1: filtered out); appends unknown keys under the LDS lock (rare)
    auto lookup = [&](const uint64_t (&key)[LEAN_ROWS], const bool (&live)[LEAN_ROWS], int (&lg)[LEAN_ROWS]) -> bool {
        if constexpr (GMAX == 1) {
#pragma unroll
            for (int r = 0; r < LEAN_ROWS; ++r) lg[r] = live[r] ? 0 : -1;
            return true;
        } else {
            for (;;) {
                bool pending = false;
#pragma unroll
                for (int r = 0; r < LEAN_ROWS; ++r) {
                    int found = -2;
#pragma unroll
                    for (int g2 = 0; g2 < GMAX; ++g2) found = (g2 < ng_c && key[r] == gk[g2]) ? g2 : found;
                    lg[r] = live[r] ? found : -1;
                    pending |= (lg[r] == -2);
                }
                const uint64_t pmask = __ballot(pending);
                if (pmask == 0) return true;                       // steady state: no LDS access
                if (*v_over) return false;
                if ((int)*v_ng == ng_c) {
                    if (lane == (int)__builtin_ctzll(pmask)) {
                        uint64_t mine = 0;
#pragma unroll
                        for (int r = LEAN_ROWS - 1; r >= 0; --r)
                            if (lg[r] == -2) mine = key[r];
                        while (atomicCAS(&lds.lock, 0u, 1u) != 0u) {}
                        const int n2 = (int)*v_ng;                 // another wave may have appended it meanwhile
                        bool have = false;
                        for (int g2 = 0; g2 < n2; ++g2) have |= (lds.keys[g2] == mine);
                        if (!have) {
                            if (n2 < GMAX) {
                                lds.keys[n2] = mine;
                                __threadfence_block();             // key before count
                                *v_ng = (uint32_t)(n2 + 1);
                            } else {
                                *v_over = 1;
                            }
                        }
                        __threadfence_block();
                        atomicExch(&lds.lock, 0u);
                    }
                }
                ng_c = __builtin_amdgcn_readfirstlane((int)*v_ng);
#pragma unroll
                for (int g2 = 0; g2 < GMAX; ++g2) gk[g2] = lean_uniform_u64(lds.keys[g2]);
            }
        }
    };
    // one row's chain values into its group's accumulators
    auto accumulate = [&](int lgr, const double (&tv)[NSTEP]) {
        if constexpr (GMAX == 1) {
            if (lgr >= 0) {
                rows1 += 1;
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) acc[s] += tv[s];
            }
        } else {
            if (lgr >= 0) {
                double* a = &lds.acc[lgr * (NSTEP * BLOCK) + tid];
#pragma unroll
                for (int s = 0; s < NSTEP; ++s)
                    __hip_atomic_fetch_add(a + s * BLOCK, tv[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&lds.cnt[lgr * BLOCK + tid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

    // ---- registers of one tile (raw bits).  Row j of sub-tile u: tile row u*512 + 2*tid + j.
    LeanU4 rv[NRANGE][U];        // 32-bit column: .x .y = rows 0 1; 64-bit: (.x .y) (.z .w)
    uint64_t rvw[NRANGE][U];    // validity word of this lane's row pair (range columns may carry NULLs: they fail the range)
    uint32_t kv[NKEY][U][2];    // Int32 key: the two values.  Utf8 key: lengths of the two strings
    uint64_t kb[NKEY][U];       // Utf8 key: 8 bytes at the first string's offset
    LeanU3 ko[NKEY][U];         // Utf8 key: the three offsets, ONE tile ahead of kv / kb
    LeanU4 xv[NSTEP][U];         // the two Float64 values
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int p = 0; p < NRANGE; ++p) { rv[p][u] = LeanU4{0, 0, 0, 0}; rvw[p][u] = ~0ull; }
#pragma unroll
        for (int q = 0; q < NKEY; ++q) { kv[q][u][0] = kv[q][u][1] = 0; kb[q][u] = 0; ko[q][u] = LeanU3{0, 0, 0}; }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) xv[s][u] = LeanU4{0, 0, 0, 0};
    }
    const uint32_t t8 = (uint32_t)tid * 8u, t16 = (uint32_t)tid * 16u;

    auto load_ranges = [&]() {
#pragma unroll
        for (int p = 0; p < NRANGE; ++p)
            if (p < n_ranges) {
                if (r32[p]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const LeanU2 v = lean_ld2(rp[p] + u * (LEAN_SUB * 4) + t8);
                        rv[p][u].x = v.x; rv[p][u].y = v.y;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) rv[p][u] = lean_ld4(rp[p] + u * (LEAN_SUB * 8) + t16);
                }
                if (rvp[p]) {           // word of rows [64k, 64k+64) that holds this lane's pair: 32 lanes share it
#pragma unroll
                    for (int u = 0; u < U; ++u) rvw[p][u] = rvp[p][u * (LEAN_SUB / 64) + (tid >> 5)];
                }
            }
    };
    auto load_offsets = [&]() {          // kp of Utf8 keys addresses the tile whose offsets are wanted
#pragma unroll
        for (int q = 0; q < NKEY; ++q)
            if (q < n_keys && kutf[q]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const BHIP_GLOBAL char* a = kp[q] + u * (LEAN_SUB * 4) + t8;
                    const LeanU2 v = lean_ld2(a);
                    ko[q][u].a = v.x; ko[q][u].b = v.y;
                    ko[q][u].c = *(const BHIP_GLOBAL uint32_t*)(a + 8);
                }
            }
    };
    // Utf8: lengths from the offsets loaded a tile ago + the bytes; Int32: the values
    auto load_keys = [&]() {
#pragma unroll
        for (int q = 0; q < NKEY; ++q)
            if (q < n_keys) {
                if (kutf[q]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t l0 = ko[q][u].b - ko[q][u].a, l1 = ko[q][u].c - ko[q][u].b;
                        bad_len |= (l0 | l1);
                        kv[q][u][0] = l0; kv[q][u][1] = l1;
                        kb[q][u] = ((const BHIP_GLOBAL PackedU64*)(kdat[q] + ko[q][u].a))->v;   // buffers carry 16 B of slack
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const LeanU2 v = lean_ld2(kp[q] + u * (LEAN_SUB * 4) + t8);
                        kv[q][u][0] = v.x; kv[q][u][1] = v.y;
                    }
                }
            }
    };
    auto load_steps = [&]() {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
            if (s < n_steps) {
#pragma unroll
                for (int u = 0; u < U; ++u) xv[s][u] = lean_ld4(xp[s] + u * (LEAN_SUB * 8) + t16);
            }
    };
    auto advance_keys_utf8 = [&]() {
#pragma unroll
        for (int q = 0; q < NKEY; ++q)
            if (kutf[q]) kp[q] += stride * 4;
    };
    auto advance_rest = [&]() {
#pragma unroll
        for (int p = 0; p < NRANGE; ++p) {
            rp[p] += stride * (r32[p] ? 4 : 8);
            if (rvp[p]) rvp[p] += stride / 64;
        }
#pragma unroll
        for (int q = 0; q < NKEY; ++q)
            if (!kutf[q]) kp[q] += stride * 4;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) xp[s] += stride * 8;
    };

    // ---- per-tile evaluation on the registers above
    auto eval_live = [&](bool (&live)[LEAN_ROWS]) {
#pragma unroll
        for (int r = 0; r < LEAN_ROWS; ++r) live[r] = true;
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
                if (rvp[p]) {           // a NULL fails every comparison
                    const uint32_t bit = (2u * (uint32_t)tid) & 63u;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        live[2 * u] = live[2 * u] && ((rvw[p][u] >> bit) & 1ull);
                        live[2 * u + 1] = live[2 * u + 1] && ((rvw[p][u] >> (bit + 1u)) & 1ull);
                    }
                }
            }
    };
    auto eval_keys = [&](uint64_t (&key)[LEAN_ROWS]) {
        uint32_t w[NKEY][LEAN_ROWS];
#pragma unroll
        for (int q = 0; q < NKEY; ++q) {
#pragma unroll
            for (int r = 0; r < LEAN_ROWS; ++r) w[q][r] = 0;
            if (q < n_keys) {
                if (kutf[q]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t l0 = kv[q][u][0] > LEAN_MAX_STR ? LEAN_MAX_STR : kv[q][u][0];
                        const uint32_t l1 = kv[q][u][1] > LEAN_MAX_STR ? LEAN_MAX_STR : kv[q][u][1];
                        w[q][2 * u] = lean_str_word((uint32_t)kb[q][u], l0);
                        w[q][2 * u + 1] = lean_str_word((uint32_t)(kb[q][u] >> (l0 << 3)), l1);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) { w[q][2 * u] = kv[q][u][0]; w[q][2 * u + 1] = kv[q][u][1]; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < LEAN_ROWS; ++r) key[r] = ((uint64_t)w[1][r] << 32) | w[0][r];
    };
    // chain values, in place in xv: f = sgn*x + add ; t = (start ? 1 : t_prev) * f
    auto eval_chain = [&]() {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
            if (s < n_steps) {
                if (!xplain[s]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double a = u2d(((uint64_t)(xv[s][u].y ^ xflip[s]) << 32) | xv[s][u].x) + xadd[s];
                        const double b = u2d(((uint64_t)(xv[s][u].w ^ xflip[s]) << 32) | xv[s][u].z) + xadd[s];
                        xv[s][u].x = (uint32_t)d2u(a); xv[s][u].y = (uint32_t)(d2u(a) >> 32);
                        xv[s][u].z = (uint32_t)d2u(b); xv[s][u].w = (uint32_t)(d2u(b) >> 32);
                    }
                }
                if (s > 0 && !xstart[s]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double a = u2d(((uint64_t)xv[s - (s > 0)][u].y << 32) | xv[s - (s > 0)][u].x) * u2d(((uint64_t)xv[s][u].y << 32) | xv[s][u].x);
                        const double b = u2d(((uint64_t)xv[s - (s > 0)][u].w << 32) | xv[s - (s > 0)][u].z) * u2d(((uint64_t)xv[s][u].w << 32) | xv[s][u].z);
                        xv[s][u].x = (uint32_t)d2u(a); xv[s][u].y = (uint32_t)(d2u(a) >> 32);
                        xv[s][u].z = (uint32_t)d2u(b); xv[s][u].w = (uint32_t)(d2u(b) >> 32);
                    }
                }
            }
    };

    // ---- main loop over the full tiles of this workgroup: tile t, loads of t+1 (offsets: t+2) in flight
    bool over = false;
    if ((int64_t)blockIdx.x < n_tiles) {
        load_offsets();                                   // offsets(t0)
        load_ranges();
        load_steps();
        // keys(t0) need offsets(t0): the only exposed dependent load of the kernel
        load_keys();
        advance_keys_utf8();
        if ((int64_t)blockIdx.x + grid < n_tiles) load_offsets();     // offsets(t0 + grid)
        advance_keys_utf8();                                          // kp (Utf8) -> t0 + 2 grid
    }
    for (int64_t t = blockIdx.x; t < n_tiles; t += grid) {
        const bool more1 = t + grid < n_tiles, more2 = t + 2 * grid < n_tiles;
        advance_rest();                                   // rp / kp(Int32) / xp -> tile t+1

        bool live[LEAN_ROWS];
        eval_live(live);
        if (more1) load_ranges();

        uint64_t key[LEAN_ROWS];
        eval_keys(key);
        if (more1) load_keys();                           // consumes offsets(t+1), issues bytes(t+1)
        if (more2) load_offsets();                        // offsets(t+2)
        advance_keys_utf8();

        int lg[LEAN_ROWS];
        if (!lookup(key, live, lg)) { over = true; break; }

        eval_chain();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double ta[NSTEP], tb[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                ta[s] = u2d(((uint64_t)xv[s][u].y << 32) | xv[s][u].x);
                tb[s] = u2d(((uint64_t)xv[s][u].w << 32) | xv[s][u].z);
            }
            accumulate(lg[2 * u], ta);
            accumulate(lg[2 * u + 1], tb);
        }
        if (more1) load_steps();
    }

    // ---- ragged tail (< 1024 rows): the workgroup next in line takes it, one row per thread and pass
    if (!over && (int64_t)blockIdx.x == n_tiles % grid && n_tiles * LEAN_TILE < n_rows) {
        const int64_t tail0 = n_tiles * LEAN_TILE;
        for (int k = 0; k < LEAN_TILE / BLOCK; ++k) {
            const int64_t i = tail0 + (int64_t)k * BLOCK + tid;
            const bool in = i < n_rows;
            bool ok = in;
            uint32_t w[NKEY] = {0, 0};
            double tv[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) tv[s] = 0.0;
            if (in) {
#pragma unroll
                for (int p = 0; p < NRANGE; ++p)
                    if (p < n_ranges) {
                        const BHIP_GLOBAL char* base = (const BHIP_GLOBAL char*)S.cols[S.ranges[p].col].data;
                        const double x = r32[p] ? (double)*(const BHIP_GLOBAL int32_t*)(base + i * 4) : *(const BHIP_GLOBAL double*)(base + i * 8);
                        ok = ok && x >= rlo[p] && x <= rhi[p];
                        const BHIP_GLOBAL uint64_t* vb = (const BHIP_GLOBAL uint64_t*)S.cols[S.ranges[p].col].validity;
                        if (vb) ok = ok && ((vb[i >> 6] >> (i & 63)) & 1ull);
                    }
#pragma unroll
                for (int q = 0; q < NKEY; ++q)
                    if (q < n_keys) {
                        const SopColumn c = S.cols[S.keys[q].col];
                        if (kutf[q]) {
                            const uint32_t o0 = (uint32_t)c.offsets[i];
                            uint32_t len = (uint32_t)c.offsets[i + 1] - o0;
                            bad_len |= len;
                            len = len > LEAN_MAX_STR ? LEAN_MAX_STR : len;
                            w[q] = lean_str_word((uint32_t)((const BHIP_GLOBAL PackedU64*)((const BHIP_GLOBAL char*)c.data + o0))->v, len);
                        } else {
                            w[q] = *(const BHIP_GLOBAL uint32_t*)((const BHIP_GLOBAL char*)c.data + i * 4);
                        }
                    }
#pragma unroll
                for (int s = 0; s < NSTEP; ++s)
                    if (s < n_steps) {
                        const double x = *(const BHIP_GLOBAL double*)((const BHIP_GLOBAL char*)S.cols[S.steps[s].col].data + i * 8);
                        const double f = xplain[s] ? x : u2d(d2u(x) ^ ((uint64_t)xflip[s] << 32)) + xadd[s];
                        tv[s] = (s > 0 && !xstart[s]) ? tv[s - (s > 0)] * f : f;
                    }
            }
            // the lookup is a wave-level protocol: every lane takes part, rows beyond the end as "filtered out"
            uint64_t key[LEAN_ROWS];
            bool live[LEAN_ROWS];
            int lg[LEAN_ROWS];
#pragma unroll
            for (int r = 0; r < LEAN_ROWS; ++r) { key[r] = 0; live[r] = false; }
            key[0] = ((uint64_t)w[1] << 32) | w[0];
            live[0] = ok;
            if (!lookup(key, live, lg)) { over = true; break; }
            accumulate(lg[0], tv);
        }
    }

    // ---- fixed-order workgroup reduction: lanes (shuffle tree) -> waves 0..3
    __syncthreads();
    uint64_t tot_rows = 0;
    double tot_acc = 0.0;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const uint32_t mine = GMAX == 1 ? rows1 : lds.cnt[(GMAX > 1 ? g : 0) * BLOCK + tid];
        const uint64_t v = wave_reduce((uint64_t)mine, ACC_COUNT_ROWS);
        if (lane == 0) lds.rowred[wave][g] = v;
    }
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const double mine = GMAX == 1 ? acc[s] : lds.acc[(GMAX > 1 ? (g * NSTEP + s) : 0) * BLOCK + tid];
            const uint64_t v = wave_reduce(d2u(mine), ACC_SUM_F64);
            if (lane == 0) lds.red[wave][g * NSTEP + s] = v;
        }
    __syncthreads();
    if (tid < GMAX) {
        tot_rows = lds.rowred[0][tid] + lds.rowred[1][tid] + lds.rowred[2][tid] + lds.rowred[3][tid];
        lds.rowtot[tid] = tot_rows;
    }
    if (tid < GMAX * NSTEP)
        tot_acc = ((u2d(lds.red[0][tid]) + u2d(lds.red[1][tid])) + u2d(lds.red[2][tid])) + u2d(lds.red[3][tid]);
    __syncthreads();

    GroupRec* out = partials + (size_t)blockIdx.x * GMAX;
    if (tid < GMAX) {
        // the packed-key layout of sop_kernel.h: part 0 -> word 0, part 1 -> word 1
        out[tid].k0 = lds.keys[tid] & 0xFFFFFFFFull;
        out[tid].k1 = lds.keys[tid] >> 32;
        out[tid].rows = tot_rows;
    }
    if (tid < GMAX * NSTEP) {
        const int g = tid / NSTEP, s = tid % NSTEP;
        if (s < S.n_steps && S.steps[s].acc != 0xFF) {
            out[g].acc[S.steps[s].acc] = d2u(tot_acc);
            out[g].nvalid[S.steps[s].acc] = lds.rowtot[g];
        }
    }
    if (tid == 0) {
        // without GROUP BY the one group (key 0) exists once a row has passed the filter
        partial_ng[blockIdx.x] = GMAX == 1 ? (tot_rows > 0 ? 1u : 0u) : lds.ng;
        if (lds.overflow) atomicOr(&status->flags, SCAN_OVERFLOW_GROUPS);
    }
    if (bad_len > LEAN_MAX_STR) atomicOr(&status->flags, SCAN_ERR_KEY_TOO_LONG);
}

template <int GMAX, int NSTEP, int NRANGE>
static hipError_t launch_lean_t(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                                uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    auto k = scan_agg_lean_kernel<GMAX, NSTEP, NRANGE>;
    const int64_t n_tiles = (S.n_rows + LEAN_TILE - 1) / LEAN_TILE;
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), BLOCK, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    static const int forced_per_cu = [] { const char* v = getenv("BHIP_AGG_BLOCKS_PER_CU"); return v ? atoi(v) : 0; }();
    if (forced_per_cu > 0) per_cu = forced_per_cu;
    int64_t grid = (int64_t)cfg.device_cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    e = hipMemcpyAsync(dprog, &S, sizeof(SopProgram), hipMemcpyHostToDevice, cfg.stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, (const SopProgram*)dprog, partials, partial_ng, status);
    *grid_out = (int)grid;
    return hipGetLastError();
}

template <int GMAX>
static hipError_t launch_lean_g(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                                uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
#define BHIP_LEAN(NS_, NR_) launch_lean_t<GMAX, NS_, NR_>(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out)
    const bool few = S.n_ranges <= 1;
    if (S.n_steps <= 2) return few ? BHIP_LEAN(2, 1) : BHIP_LEAN(2, 4);
    if (S.n_steps <= 5) return few ? BHIP_LEAN(5, 1) : BHIP_LEAN(5, 4);
    return few ? BHIP_LEAN(8, 1) : BHIP_LEAN(8, 4);
#undef BHIP_LEAN
}

}  // namespace bhip
