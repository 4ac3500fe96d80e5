// sop_kernel.h — the register-resident fused scan + aggregate kernel (template; instantiated per
// (GMAX, NSTEP, NPRED) in kernels_sop_g{1,4,8}.hip).
//
// Same operator chain as kernels_agg.hip (FilterExec -> HashAggregateExec, reference operators at
// rust/core/src/serde/physical_plan/from_proto.rs:81-92,173-252), for the expression shape that
// TPC-H aggregates have (Q1, Q6, Q3, Q5 ...):
//   predicate  = AND of comparisons   column <op> literal
//   group keys = integer / date / short Utf8 columns packing into 16 bytes
//   aggregates = SUM / AVG / COUNT over multiplication chains  f0 * f1 * ...  whose factors are
//                column | literal +- column | column +- literal | literal
// The host (host/sop.cpp) recognises the shape; anything else runs on the VM kernel.
//
// Structure (each point answers a measured stall, profiles/r01_*):
//  * the plan is resolved ONCE into wave-uniform registers (a "site" = base pointer + element type
//    + role parameters); the base pointers advance by one grid stride per tile with two scalar adds.
//    Re-reading the plan table per tile cost a dependent scalar-load round trip per site: 77 % of the
//    wave time was s_waitcnt.
//  * every value lives in a statically indexed register: all loops are unrolled over the template
//    bounds, unused steps are zero padded, so there is no dispatch and no LDS operand traffic.
//  * the loads of tile t+1 are issued before tile t is evaluated (register double buffering); the
//    dependent second stage of short-string keys (the bytes) is issued in the middle of the evaluation.
//  * no workgroup barrier in the loop: a wave that meets an unknown group key appends it to the
//    workgroup's key table under an LDS lock (at most GMAX times per workgroup), readers see a
//    consistent prefix because the key is written before the count.
//  * lanes -> waves -> workgroup are reduced in a fixed order: run-to-run deterministic sums.
// Each chain step rounds separately (-ffp-contract=off): per-row values are bit-identical to the
// reference's one-kernel-per-node evaluation.  HBM traffic: every referenced column once.
#pragma once
#include <hip/hip_runtime.h>
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

struct SopSite {
    const BHIP_GLOBAL char* ptr;      // values of the CURRENT tile (advanced every iteration)
    uint32_t meta;                    // dtype | a << 8 | b << 16 | c << 24 (role specific)
};

__device__ inline int sop_width(uint32_t dtype) {
    return (dtype == DT_INT32 || dtype == DT_DATE32) ? 4 : (dtype == DT_UINT8 ? 1 : 8);
}

// 64-bit image of element `i` of the tile
__device__ inline uint64_t sop_load(const BHIP_GLOBAL char* tile_ptr, uint32_t dtype, uint32_t i) {
    if (dtype == DT_INT32 || dtype == DT_DATE32) return (uint64_t)(int64_t)((const BHIP_GLOBAL int32_t*)tile_ptr)[i];
    if (dtype == DT_UINT8) return ((const BHIP_GLOBAL uint8_t*)tile_ptr)[i];
    return ((const BHIP_GLOBAL uint64_t*)tile_ptr)[i];
}

__device__ inline double sop_as_f64(uint64_t bits, uint32_t dtype) {
    if (dtype == DT_FLOAT64) return u2d(bits);
    if (dtype == DT_UINT64) return (double)bits;
    return (double)(int64_t)bits;                 // CAST(int AS Float64)
}

__device__ inline bool sop_cmp(uint64_t x, uint64_t lit, uint32_t vclass, uint32_t kind) {
    if (vclass == VC_F64) return cmp_vals<double>(u2d(x), u2d(lit), kind);
    if (vclass == 3) return cmp_vals<uint64_t>(x, lit, kind);
    return cmp_vals<int64_t>((int64_t)x, (int64_t)lit, kind);
}

template <int NSTEP, int NPRED>
struct SopRegs {                    // one tile's loaded values
    uint64_t pv[NPRED][SOP_R];
    uint64_t kv[SOP_NKEY][SOP_R];
    uint64_t xv[NSTEP][SOP_R];
};

template <int GMAX, int NSTEP, int NPRED>
__global__ void __launch_bounds__(BLOCK, 2)
scan_agg_sop_kernel(const SopProgram* __restrict__ Sp, GroupRec* partials, uint32_t* partial_ng, ScanStatus* status) {
    const SopProgram& S = *Sp;
    constexpr int R = SOP_R;
    __shared__ SopLds lds;
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- resolve the plan into wave-uniform registers
    const int n_pred = S.n_pred, n_keys = S.n_keys, n_steps = S.n_steps;
    const int64_t n_rows = S.n_rows;
    const bool key64 = S.key_bytes <= 8;
    const int64_t first = (int64_t)blockIdx.x * SOP_TILE;
    const int64_t stride = (int64_t)gridDim.x * SOP_TILE;
    SopSite ps[NPRED], ks[SOP_NKEY], xs[NSTEP];
    const BHIP_GLOBAL int32_t* koff[SOP_NKEY];
    int32_t kbytes[SOP_NKEY];
    uint64_t plit[NPRED];
    double xlit[NSTEP];
#pragma unroll
    for (int p = 0; p < NPRED; ++p) {
        ps[p] = SopSite{nullptr, 0};
        plit[p] = 0;
        if (p < n_pred) {
            const SopColumn c = S.cols[S.pred[p].col];
            ps[p].meta = (uint32_t)c.dtype | ((uint32_t)S.pred[p].cmp << 8) | ((uint32_t)S.pred[p].vclass << 16);
            ps[p].ptr = (const BHIP_GLOBAL char*)c.data + first * sop_width(c.dtype);
            plit[p] = S.pred[p].lit;
        }
    }
#pragma unroll
    for (int q = 0; q < SOP_NKEY; ++q) {
        ks[q] = SopSite{nullptr, 0};
        koff[q] = nullptr;
        kbytes[q] = 0;
        if (q < n_keys) {
            const SopColumn c = S.cols[S.keys[q].col];
            ks[q].meta = (uint32_t)c.dtype | ((uint32_t)S.keys[q].width << 8) | ((uint32_t)S.keys[q].pos << 16);
            if (c.dtype == DT_UTF8) {
                ks[q].ptr = (const BHIP_GLOBAL char*)c.data;                  // bytes: absolute offsets
                koff[q] = (const BHIP_GLOBAL int32_t*)c.offsets + first;
                kbytes[q] = c.data_bytes;
            } else {
                ks[q].ptr = (const BHIP_GLOBAL char*)c.data + first * sop_width(c.dtype);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        xs[s] = SopSite{nullptr, (uint32_t)DT_FLOAT64 | ((uint32_t)SOP_F_LIT << 8) | ((uint32_t)SOP_OP_START << 16)};
        xlit[s] = 0.0;                                    // padding step: t = 0.0, no load
        if (s < n_steps) {
            const SopColumn c = S.cols[S.steps[s].col];
            xs[s].meta = (uint32_t)c.dtype | ((uint32_t)S.steps[s].mode << 8) | ((uint32_t)S.steps[s].op << 16);
            xs[s].ptr = (const BHIP_GLOBAL char*)c.data + first * sop_width(c.dtype);
            xlit[s] = S.steps[s].lit;
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

    // stage A: everything whose address is known (values, Utf8 offsets) of the tile the sites point at
    auto issue_a = [&](SopRegs<NSTEP, NPRED>& g, uint32_t left) {
#pragma unroll
        for (int p = 0; p < NPRED; ++p)
            if (p < n_pred) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t i = r * BLOCK + tid;
                    g.pv[p][r] = i < left ? sop_load(ps[p].ptr, ps[p].meta & 0xFF, i) : 0;
                }
            }
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q)
            if (q < n_keys) {
                if ((ks[q].meta & 0xFF) == DT_UTF8) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t i = r * BLOCK + tid;
                        uint64_t v = 0;
                        if (i < left) {
                            const int32_t o0 = koff[q][i], o1 = koff[q][i + 1];
                            v = (uint64_t)(uint32_t)o0 | ((uint64_t)(uint32_t)(o1 - o0) << 32);
                        }
                        g.kv[q][r] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t i = r * BLOCK + tid;
                        g.kv[q][r] = i < left ? sop_load(ks[q].ptr, ks[q].meta & 0xFF, i) : 0;
                    }
                }
            }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
            if (((xs[s].meta >> 8) & 0xFF) != SOP_F_LIT) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t i = r * BLOCK + tid;
                    g.xv[s][r] = i < left ? sop_load(xs[s].ptr, xs[s].meta & 0xFF, i) : 0;
                }
            }
    };
    // stage B: the bytes of short-string keys -> [len][bytes...] image
    auto issue_b = [&](SopRegs<NSTEP, NPRED>& g, uint32_t left) {
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q)
            if (q < n_keys && (ks[q].meta & 0xFF) == DT_UTF8) {
                const BHIP_GLOBAL uint8_t* data = (const BHIP_GLOBAL uint8_t*)ks[q].ptr;
                const uint32_t width = (ks[q].meta >> 8) & 0xFF;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t i = r * BLOCK + tid;
                    uint64_t packed = 0;
                    if (i < left) {
                        const uint32_t o0 = (uint32_t)g.kv[q][r];
                        uint32_t len = (uint32_t)(g.kv[q][r] >> 32);
                        if (len > width - 1u) { err |= SCAN_ERR_KEY_TOO_LONG; len = width - 1u; }
                        uint64_t bytes = 0;
                        if ((int64_t)o0 + 8 <= (int64_t)kbytes[q]) bytes = ((const BHIP_GLOBAL PackedU64*)(data + o0))->v;
                        else for (uint32_t b = 0; b < len; ++b) bytes |= (uint64_t)data[o0 + b] << (8 * b);
                        if (len < 8) bytes &= (1ull << (8 * len)) - 1ull;
                        packed = (uint64_t)len | (bytes << 8);
                    }
                    g.kv[q][r] = packed;
                }
            }
    };
    auto advance = [&]() {      // sites -> next tile of this workgroup
#pragma unroll
        for (int p = 0; p < NPRED; ++p) ps[p].ptr += stride * sop_width(ps[p].meta & 0xFF);
#pragma unroll
        for (int q = 0; q < SOP_NKEY; ++q) {
            if ((ks[q].meta & 0xFF) == DT_UTF8) koff[q] += stride;
            else ks[q].ptr += stride * sop_width(ks[q].meta & 0xFF);
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) xs[s].ptr += stride * sop_width(xs[s].meta & 0xFF);
    };
    auto tile_left = [&](int64_t base) -> uint32_t {
        const int64_t l = n_rows - base;
        return l <= 0 ? 0u : (l < SOP_TILE ? (uint32_t)l : (uint32_t)SOP_TILE);
    };

    SopRegs<NSTEP, NPRED> nxt;
    {
        const uint32_t l0 = tile_left(first);
        issue_a(nxt, l0);
        issue_b(nxt, l0);
    }
    for (int64_t base = first; base < n_rows; base += stride) {
        const uint32_t left = tile_left(base);
        const uint32_t left_next = tile_left(base + stride);
        SopRegs<NSTEP, NPRED> cur = nxt;               // waits for the prefetched loads
        advance();
        issue_a(nxt, left_next);                       // tile t+1 in flight during the evaluation of t

        // ---- predicate and key of each row
        Key128 rk[R];
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = r * BLOCK + tid;
            bool ok = i < left;
#pragma unroll
            for (int p = 0; p < NPRED; ++p)
                if (p < n_pred) ok = ok && sop_cmp(cur.pv[p][r], plit[p], (ps[p].meta >> 16) & 0xFF, (ps[p].meta >> 8) & 0xFF);
            live[r] = ok;
            Key128 k{0, 0};
#pragma unroll
            for (int q = 0; q < SOP_NKEY; ++q)
                if (q < n_keys) key_put(k, (ks[q].meta >> 16) & 0xFF, cur.kv[q][r], (ks[q].meta >> 8) & 0xFF);
            rk[r] = k;
        }
        // ---- group of each row; unknown keys are appended under the LDS lock (rare)
        int lg[R];
        for (;;) {
            const int ng = (int)*v_ng;
            Key128 gk[GMAX];
#pragma unroll
            for (int g = 0; g < GMAX; ++g) { gk[g].k0 = lds.keys[g].k0; gk[g].k1 = key64 ? 0 : lds.keys[g].k1; }
            bool pending = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int found = -2;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) {
                    const bool eq = rk[r].k0 == gk[g].k0 && (key64 || rk[r].k1 == gk[g].k1);
                    found = (g < ng && eq) ? g : found;
                }
                lg[r] = live[r] ? found : -1;
                pending |= (lg[r] == -2);
            }
            const uint64_t pmask = __ballot(pending);
            if (pmask == 0 || *v_over) break;
            if (lane == (int)__builtin_ctzll(pmask)) {
                Key128 mine{0, 0};
#pragma unroll
                for (int r = R - 1; r >= 0; --r)
                    if (lg[r] == -2) mine = rk[r];
                while (atomicCAS(&lds.lock, 0u, 1u) != 0u) {}
                const int n2 = (int)*v_ng;               // another wave may have appended it meanwhile
                bool have = false;
                for (int g = 0; g < n2; ++g) have |= (lds.keys[g].k0 == mine.k0 && lds.keys[g].k1 == mine.k1);
                if (!have) {
                    if (n2 < GMAX) {
                        lds.keys[n2].k0 = mine.k0;
                        lds.keys[n2].k1 = mine.k1;
                        __threadfence_block();           // key before count
                        *v_ng = (uint32_t)(n2 + 1);
                    } else {
                        *v_over = 1;
                    }
                }
                __threadfence_block();
                atomicExch(&lds.lock, 0u);
            }
        }
        if (*v_over) break;

        issue_b(nxt, left_next);                       // tile t+1's string bytes (its offsets have landed)

        // ---- multiplication chains, one exec-masked block of adds per group
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double tv[NSTEP];
            double run = 0.0;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const uint32_t mode = (xs[s].meta >> 8) & 0xFF, op = (xs[s].meta >> 16) & 0xFF;
                const double x = mode == SOP_F_LIT ? 0.0 : sop_as_f64(cur.xv[s][r], xs[s].meta & 0xFF);
                double f;
                switch (mode) {
                    case SOP_F_COL: f = x; break;
                    case SOP_F_LIT_MINUS_COL: f = xlit[s] - x; break;
                    case SOP_F_LIT_PLUS_COL: f = xlit[s] + x; break;
                    case SOP_F_COL_MINUS_LIT: f = x - xlit[s]; break;
                    case SOP_F_COL_PLUS_LIT: f = x + xlit[s]; break;
                    default: f = xlit[s]; break;
                }
                run = op == SOP_OP_START ? f : (op == SOP_OP_MUL ? run * f : run / f);
                tv[s] = run;
            }
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (lg[r] == g) {
                    rows[g] += 1;
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) acc[g][s] += tv[s];
                }
            }
        }
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

template <int GMAX, int NSTEP, int NPRED>
static hipError_t launch_sop_t(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    auto k = scan_agg_sop_kernel<GMAX, NSTEP, NPRED>;
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

// smallest instantiated (NSTEP, NPRED) that holds the plan
template <int GMAX>
static hipError_t launch_sop_g(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
#define BHIP_SOP(NS_, NP_) launch_sop_t<GMAX, NS_, NP_>(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out)
    const bool small_pred = S.n_pred <= 2;
    if (S.n_steps <= 2) return small_pred ? BHIP_SOP(2, 2) : BHIP_SOP(2, 6);
    if (S.n_steps <= 5) return small_pred ? BHIP_SOP(5, 2) : BHIP_SOP(5, 6);
    return small_pred ? BHIP_SOP(8, 2) : BHIP_SOP(8, 6);
#undef BHIP_SOP
}

}  // namespace bhip
