// sop_kernel.h — the register-resident fused scan + aggregate kernel (template; instantiated per
// (GMAX, NSTEP, NRANGE) in kernels_sop_g{1,4,8}.hip).
//
// Same operator chain as kernels_agg.hip (FilterExec -> HashAggregateExec, reference operators at
// rust/core/src/serde/physical_plan/from_proto.rs:81-92,173-252), for the expression shape that
// TPC-H aggregates have (Q1, Q6, Q3, Q5 ...):
//   predicate  = AND of comparisons   column <op> literal        (-> one [lo, hi] range per column)
//   group keys = up to three integer / date / short Utf8 columns
//   aggregates = SUM / AVG / COUNT over multiplication chains  f0 * f1 * ...  whose factors are
//                column | literal +- column | column +- literal | literal   (-> f = sgn * x + add)
// The host (host/sop.cpp) recognises the shape; anything else runs on the VM kernel.
//
// Structure (each point answers a measured stall, profiles/r01_*):
//  * the plan is resolved ONCE into wave-uniform registers (base pointers, bounds, coefficients);
//    the pointers advance by one grid stride per tile.  Re-reading the plan table per tile cost a
//    dependent scalar-load round trip per site: 77 % of the wave time was s_waitcnt.
//  * the tile body is straight-line code: ranges and coefficients instead of operator switches,
//    loops unrolled over the template bounds with zero-padded entries, every value in a statically
//    indexed register.  The switch-per-operator version ran ~15 cycles per instruction on one wave
//    (throughput grew linearly with waves per CU): scalar branches, not memory, were the latency.
//  * the loads of tile t+1 are issued before tile t is evaluated (register double buffering); the
//    dependent second stage of short-string keys (the bytes) is issued in the middle of the evaluation.
//  * no workgroup barrier in the loop: a wave that meets an unknown group key appends it to the
//    workgroup's key table under an LDS lock (at most GMAX times per workgroup); readers see a
//    consistent prefix because the key is written before the count.
//  * lanes -> waves -> workgroup are reduced in a fixed order: run-to-run deterministic sums.
// Each chain step rounds separately (-ffp-contract=off): per-row values are bit-identical to the
// reference's one-kernel-per-node evaluation.  HBM traffic: every referenced column once.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "kernels.h"
#include "launch_common.h"
#include "reduce_device.h"
#include "sop.h"
#include "vm_device.h"

namespace bhip {

#ifndef SOP_R_VALUE
#define SOP_R_VALUE 2
#endif
constexpr int SOP_R = SOP_R_VALUE;         // rows per thread and tile
constexpr int SOP_TILE = BLOCK * SOP_R;

struct SopLds {
    Key128 keys[AGG_GMAX];
    uint32_t ng, lock, overflow, pad;
    uint64_t red[4][AGG_GMAX * SOP_NSTEP];
    uint64_t rowtot[AGG_GMAX];
};

template <int NSTEP, int NRANGE>
struct SopRegs {                    // one tile's loaded values (raw bits; converted when consumed)
    uint64_t rv[NRANGE][SOP_R];
    uint64_t kv[SOP_NKEY][SOP_R];
    uint32_t kl[SOP_NKEY][SOP_R];   // short strings: length (kv holds the raw 8 bytes until the tile is evaluated)
    uint64_t xv[NSTEP][SOP_R];
};

// raw bits of element i of the tile at `p`: 32-bit columns fill the low half only.  `p` is wave-uniform
// (scalar registers) and the byte offset i*4 / i*8 is the same for every tile, so a load costs no
// address arithmetic in the loop (global_load with scalar base + 32-bit vector offset).
__device__ inline uint64_t sop_load_raw(const BHIP_GLOBAL char* p, bool is32, uint32_t i) {
    if (is32) return (uint64_t)*(const BHIP_GLOBAL uint32_t*)(p + (i << 2));
    return *(const BHIP_GLOBAL uint64_t*)(p + (i << 3));
}
__device__ inline uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// value as double; is32 is wave-uniform, so this is a scalar branch around one conversion
__device__ inline double sop_value(uint64_t raw, bool is32) {
    if (is32) return (double)(int32_t)(uint32_t)raw;
    return u2d(raw);
}

template <int GMAX, int NSTEP, int NRANGE>
// register budget: 3 workgroups per CU (168 VGPRs) up to 4 groups x 5 steps (128 spills), else 2
__global__ void __launch_bounds__(BLOCK, (GMAX * NSTEP <= 20 ? 3 : 2))
scan_agg_sop_kernel(const SopProgram* __restrict__ Sp, GroupRec* partials, uint32_t* partial_ng, ScanStatus* status) {
    const SopProgram& S = *Sp;
    constexpr int R = SOP_R;
    __shared__ SopLds lds;
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- resolve the plan into wave-uniform registers
    const int n_ranges = S.n_ranges, n_keys = S.n_keys, n_steps = S.n_steps;
    const int64_t n_rows = S.n_rows;
    const bool key64 = n_keys <= 1;
    const int64_t first = (int64_t)blockIdx.x * SOP_TILE;
    const int64_t stride = (int64_t)gridDim.x * SOP_TILE;

    const BHIP_GLOBAL char* rp[NRANGE];
    bool r32[NRANGE];
    double rlo[NRANGE], rhi[NRANGE];
#pragma unroll
    for (int p = 0; p < NRANGE; ++p) {
        rp[p] = nullptr; r32[p] = false;
        rlo[p] = -__builtin_huge_val(); rhi[p] = __builtin_huge_val();       // padding: always true
        if (p < n_ranges) {
            const SopColumn c = S.cols[S.ranges[p].col];
            r32[p] = S.ranges[p].is32 != 0;
            rp[p] = (const BHIP_GLOBAL char*)c.data + first * (r32[p] ? 4 : 8);
            rlo[p] = S.ranges[p].lo; rhi[p] = S.ranges[p].hi;
        }
    }
    const BHIP_GLOBAL char* kp[SOP_NKEY];
    const BHIP_GLOBAL int32_t* koff[SOP_NKEY];
    int kkind[SOP_NKEY], kbytes[SOP_NKEY];
#pragma unroll
    for (int q = 0; q < SOP_NKEY; ++q) {
        kp[q] = nullptr; koff[q] = nullptr; kkind[q] = -1; kbytes[q] = 0;
        if (q < n_keys) {
            const SopColumn c = S.cols[S.keys[q].col];
            kkind[q] = S.keys[q].kind;
            if (kkind[q] == SOP_KEY_UTF8) {
                kp[q] = (const BHIP_GLOBAL char*)c.data;                      // bytes: absolute offsets
                koff[q] = (const BHIP_GLOBAL int32_t*)c.offsets + first;
                kbytes[q] = c.data_bytes;
            } else {
                kp[q] = (const BHIP_GLOBAL char*)c.data + first * (kkind[q] == SOP_KEY_I32 ? 4 : 8);
            }
        }
    }
    const BHIP_GLOBAL char* xp[NSTEP];
    bool x32[NSTEP], xhas[NSTEP], xstart[NSTEP], xplain[NSTEP];
    double xsgn[NSTEP], xadd[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        xp[s] = nullptr; x32[s] = false; xhas[s] = false; xstart[s] = true; xplain[s] = false;
        xsgn[s] = 0.0; xadd[s] = 0.0;                                         // padding: t = 1.0 * (0*0 + 0) = 0
        if (s < n_steps) {
            const SopStep st = S.steps[s];
            x32[s] = st.is32 != 0; xhas[s] = st.has_col != 0; xstart[s] = st.start != 0;
            xsgn[s] = st.sgn; xadd[s] = st.add;
            // f = 1.0 * x + (-0.0) == x bit for bit: skip the two operations
            xplain[s] = xhas[s] && !x32[s] && st.sgn == 1.0 && st.add == 0.0 && __builtin_signbit(st.add);
            if (xhas[s]) xp[s] = (const BHIP_GLOBAL char*)S.cols[st.col].data + first * (x32[s] ? 4 : 8);
        }
    }

    double acc[GMAX][NSTEP];
    uint32_t rows[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        rows[g] = 0;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) acc[g][s] = 0.0;
    }
    if (tid == 0) { lds.ng = 0; lds.overflow = 0; lds.lock = 0; }
    if (tid < AGG_GMAX) { lds.keys[tid].k0 = 0; lds.keys[tid].k1 = 0; }
    __syncthreads();
    volatile uint32_t* v_ng = &lds.ng;
    volatile uint32_t* v_over = &lds.overflow;
    uint32_t err = 0;
    int ng_c = 0;                                      // register copy of the key table
    Key128 gk[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) gk[g] = Key128{0, 0};

    // ---- loaders.  Every register is reloaded for tile t+1 right after its last use in tile t, in the
    // order the next iteration consumes it (ranges, keys, chain inputs): loads return in order, so each
    // consumer waits only for what it needs while everything issued later stays in flight, and one
    // register set serves both tiles.
    auto load_ranges = [&](SopRegs<NSTEP, NRANGE>& g, uint32_t left) {
#pragma unroll
        for (int p = 0; p < NRANGE; ++p)
            if (p < n_ranges) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t i = r * BLOCK + tid;
                    g.rv[p][r] = i < left ? sop_load_raw(rp[p], r32[p], i) : 0;
                }
            }
    };
    // fixed-width keys: the value; short strings: the raw 8 bytes at the offset loaded one iteration ago
    // (masking / packing waits until the tile is evaluated — consuming a load right after issuing it would
    // drain every load in flight)
    auto load_keys = [&](SopRegs<NSTEP, NRANGE>& g, const uint64_t (&ko)[SOP_NKEY][R], uint32_t left) {
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q)
            if (q < n_keys) {
                if (kkind[q] == SOP_KEY_UTF8) {
                    const BHIP_GLOBAL uint8_t* data = (const BHIP_GLOBAL uint8_t*)kp[q];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t i = r * BLOCK + tid;
                        uint64_t raw = 0;
                        uint32_t len = 0;
                        if (i < left) {
                            const uint32_t o0 = (uint32_t)ko[q][r];
                            len = (uint32_t)(ko[q][r] >> 32);
                            if (len > 7u) { err |= SCAN_ERR_KEY_TOO_LONG; len = 7u; }
                            if ((int64_t)o0 + 8 <= (int64_t)kbytes[q]) {
                                raw = ((const BHIP_GLOBAL PackedU64*)(data + o0))->v;      // one 8-byte load
                            } else {                                                       // last bytes of the buffer
                                for (uint32_t b = 0; b < len; ++b) raw |= (uint64_t)data[o0 + b] << (8 * b);
                            }
                        }
                        g.kv[q][r] = raw;
                        g.kl[q][r] = len;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t i = r * BLOCK + tid;
                        g.kv[q][r] = i < left ? sop_load_raw(kp[q], kkind[q] == SOP_KEY_I32, i) : 0;
                    }
                }
            }
    };
    // short-string keys: the two Arrow offsets of each row, loaded TWO tiles ahead (pointer koff)
    auto load_offsets = [&](uint64_t (&ko)[SOP_NKEY][R], uint32_t left) {
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q)
            if (q < n_keys && kkind[q] == SOP_KEY_UTF8) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t i = r * BLOCK + tid;
                    uint64_t v = 0;
                    if (i < left) {
                        const int32_t o0 = koff[q][i], o1 = koff[q][i + 1];
                        v = (uint64_t)(uint32_t)o0 | ((uint64_t)(uint32_t)(o1 - o0) << 32);
                    }
                    ko[q][r] = v;
                }
            }
    };
    auto load_step = [&](SopRegs<NSTEP, NRANGE>& g, int s, uint32_t left) {
        if (xhas[s]) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t i = r * BLOCK + tid;
                g.xv[s][r] = i < left ? sop_load_raw(xp[s], x32[s], i) : 0;
            }
        }
    };
    auto advance = [&]() {      // pointers -> next tile of this workgroup
#pragma unroll
        for (int p = 0; p < NRANGE; ++p) rp[p] += stride * (r32[p] ? 4 : 8);
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q) {
            if (kkind[q] == SOP_KEY_UTF8) koff[q] += stride;
            else kp[q] += stride * (kkind[q] == SOP_KEY_I32 ? 4 : 8);
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) xp[s] += stride * (x32[s] ? 4 : 8);
    };
    auto tile_left = [&](int64_t base) -> uint32_t {
        const int64_t l = n_rows - base;
        return l <= 0 ? 0u : (l < SOP_TILE ? (uint32_t)l : (uint32_t)SOP_TILE);
    };

    // prologue: the first tile's values, and the string offsets of the first two tiles
    SopRegs<NSTEP, NRANGE> g{};                        // padding entries stay zero
    uint64_t ko[SOP_NKEY][R] = {};
    {
        const uint32_t l0 = tile_left(first);
        load_offsets(ko, l0);
        load_ranges(g, l0);
        load_keys(g, ko, l0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) load_step(g, s, l0);
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q) koff[q] += stride;
        load_offsets(ko, tile_left(first + stride));
    }
    for (int64_t base = first; base < n_rows; base += stride) {
        const uint32_t left = tile_left(base);
        const uint32_t left_next = tile_left(base + stride);
        advance();                                     // pointers now address tile t+1 (koff: t+2)

        // ---- predicate of each row, then the range columns of tile t+1
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = r * BLOCK + tid;
            bool ok = i < left;
#pragma unroll
            for (int p = 0; p < NRANGE; ++p) {
                const double x = sop_value(g.rv[p][r], r32[p]);
                ok = ok && (x >= rlo[p]) && (x <= rhi[p]);
            }
            live[r] = ok;
        }
        load_ranges(g, left_next);

        // ---- key of each row, then the keys of tile t+1 and the string offsets of tile t+2
        Key128 rk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint64_t kw[SOP_NKEY];
#pragma unroll
            for (int q = 0; q < SOP_NKEY; ++q) {
                kw[q] = g.kv[q][r];
                if (kkind[q] == SOP_KEY_UTF8) {          // [len][bytes] image of the short string
                    const uint32_t len = g.kl[q][r];
                    kw[q] = (uint64_t)len | ((kw[q] & ((1ull << (8 * len)) - 1ull)) << 8);
                }
            }
            rk[r].k0 = kw[0];
            rk[r].k1 = kw[1] | (kw[2] << 32);
        }
        load_keys(g, ko, left_next);
        load_offsets(ko, tile_left(base + 2 * stride));

        // ---- group of each row.  The workgroup's key table is cached in registers; LDS is touched only
        //      when a row's key is unknown (then it is appended under the LDS lock — rare)
        int lg[R];
        for (;;) {
            bool pending = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int found = -2;
#pragma unroll
                for (int g2 = 0; g2 < GMAX; ++g2) {
                    const bool eq = rk[r].k0 == gk[g2].k0 && (key64 || rk[r].k1 == gk[g2].k1);
                    found = (g2 < ng_c && eq) ? g2 : found;
                }
                lg[r] = live[r] ? found : -1;
                pending |= (lg[r] == -2);
            }
            const uint64_t pmask = __ballot(pending);
            if (pmask == 0) break;                       // steady state: no LDS access at all
            if (*v_over) break;
            if ((int)*v_ng == ng_c) {
                // the cached table is current and a key is still unknown: append it under the lock
                if (lane == (int)__builtin_ctzll(pmask)) {
                    Key128 mine{0, 0};
#pragma unroll
                    for (int r = R - 1; r >= 0; --r)
                        if (lg[r] == -2) mine = rk[r];
                    while (atomicCAS(&lds.lock, 0u, 1u) != 0u) {}
                    const int n2 = (int)*v_ng;           // another wave may have appended it meanwhile
                    bool have = false;
                    for (int g2 = 0; g2 < n2; ++g2) have |= (lds.keys[g2].k0 == mine.k0 && lds.keys[g2].k1 == mine.k1);
                    if (!have) {
                        if (n2 < GMAX) {
                            lds.keys[n2].k0 = mine.k0;
                            lds.keys[n2].k1 = mine.k1;
                            __threadfence_block();       // key before count
                            *v_ng = (uint32_t)(n2 + 1);
                        } else {
                            *v_over = 1;
                        }
                    }
                    __threadfence_block();
                    atomicExch(&lds.lock, 0u);
                }
            }
            // refresh the register copy of the workgroup's key table
            ng_c = __builtin_amdgcn_readfirstlane((int)*v_ng);
#pragma unroll
            for (int g2 = 0; g2 < GMAX; ++g2) {            // wave-uniform: keep the copy in scalar registers
                gk[g2].k0 = uniform_u64(lds.keys[g2].k0);
                gk[g2].k1 = key64 ? 0 : uniform_u64(lds.keys[g2].k1);
            }
        }
        if (ng_c >= GMAX && *v_over) break;

        // ---- multiplication chains (f = sgn*x + add ; t = (start ? 1 : t) * f); each input register is
        //      reloaded for tile t+1 as soon as the chain has consumed it
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double tv[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                double f;
                if (xplain[s]) f = u2d(g.xv[s][r]);                     // plain Float64 column: f = x
                else f = xsgn[s] * sop_value(xhas[s] ? g.xv[s][r] : 0ull, x32[s]) + xadd[s];
                if (xstart[s] || s == 0) tv[s] = f;
                else tv[s] = tv[s == 0 ? 0 : s - 1] * f;
            }
            // one exec-masked block of adds per group
#pragma unroll
            for (int g2 = 0; g2 < GMAX; ++g2) {
                if (lg[r] == g2) {
                    rows[g2] += 1;
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) acc[g2][s] += tv[s];
                }
            }
        }
        // the chain inputs of tile t+1
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) load_step(g, s, left_next);
    }

    // ---- fixed-order workgroup reduction: lanes (shuffle tree) -> waves 0..3
    const int wave = tid >> 6;
    __syncthreads();
    uint64_t tot_rows = 0;
    double tot_acc = 0.0;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const uint64_t v = wave_reduce((uint64_t)rows[g], ACC_COUNT_ROWS);
        if (lane == 0) lds.red[wave][g] = v;
    }
    __syncthreads();
    if (tid < GMAX) {
        tot_rows = lds.red[0][tid] + lds.red[1][tid] + lds.red[2][tid] + lds.red[3][tid];
        lds.rowtot[tid] = tot_rows;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const uint64_t v = wave_reduce(d2u(acc[g][s]), ACC_SUM_F64);
            if (lane == 0) lds.red[wave][g * NSTEP + s] = v;
        }
    __syncthreads();
    if (tid < GMAX * NSTEP)
        tot_acc = ((u2d(lds.red[0][tid]) + u2d(lds.red[1][tid])) + u2d(lds.red[2][tid])) + u2d(lds.red[3][tid]);

    GroupRec* out = partials + (size_t)blockIdx.x * GMAX;
    if (tid < GMAX) {
        out[tid].k0 = lds.keys[tid].k0;
        out[tid].k1 = lds.keys[tid].k1;
        out[tid].rows = tot_rows;
    }
    if (tid < GMAX * NSTEP) {
        const int g = tid / NSTEP, s = tid % NSTEP;
        if (s < S.n_steps && S.steps[s].acc != 0xFF) {
            out[g].acc[S.steps[s].acc] = d2u(tot_acc);
            out[g].nvalid[S.steps[s].acc] = lds.rowtot[g];     // no NULLs on this path: every row contributes
        }
    }
    if (tid == 0) {
        partial_ng[blockIdx.x] = lds.ng;
        if (lds.overflow) atomicOr(&status->flags, SCAN_OVERFLOW_GROUPS);
    }
    if (err) atomicOr(&status->flags, err);
}

template <int GMAX, int NSTEP, int NRANGE>
static hipError_t launch_sop_t(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    auto k = scan_agg_sop_kernel<GMAX, NSTEP, NRANGE>;
    const int64_t n_tiles = (S.n_rows + SOP_TILE - 1) / SOP_TILE;
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

// smallest instantiated (NSTEP, NRANGE) that holds the plan
template <int GMAX>
static hipError_t launch_sop_g(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
#define BHIP_SOP(NS_, NR_) launch_sop_t<GMAX, NS_, NR_>(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out)
    const bool few = S.n_ranges <= 1;
    if (S.n_steps <= 2) return few ? BHIP_SOP(2, 1) : BHIP_SOP(2, 4);
    if (S.n_steps <= 5) return few ? BHIP_SOP(5, 1) : BHIP_SOP(5, 4);
    return few ? BHIP_SOP(8, 1) : BHIP_SOP(8, 4);
#undef BHIP_SOP
}

}  // namespace bhip
