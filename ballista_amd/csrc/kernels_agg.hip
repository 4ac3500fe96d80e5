// kernels_agg.hip — fused scan + low-cardinality aggregate, and the merge of per-workgroup
// partial states.
//
// Replaces the reference's FilterExec -> [CoalesceBatches] -> HashAggregateExec(Partial)
// chain (operators built at rust/core/src/serde/physical_plan/from_proto.rs:81-92,122-128,
// 173-252; executed from rust/executor/src/flight_service.rs:117-121) for group-bys with few
// groups (TPC-H Q1: 4, Q6: 1, Q5: 5).  HBM traffic = each referenced input column read once
// (coalesced, one row per lane) + a few KB of per-workgroup partial state => HBM-bound.
//
// Keys live in LDS, accumulators in registers (static indices; updates predicated on the
// row's group id); lanes -> waves -> workgroup -> grid are reduced in a fixed order, so sums
// are run-to-run deterministic.  A workgroup that meets more than GMAX groups raises
// SCAN_OVERFLOW_GROUPS and the host re-runs the aggregate on the hash path.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "vm_device.h"
#include "reduce_device.h"
#include "launch_common.h"

namespace bhip {

// =============================================================================================
// Sink: low-cardinality aggregate
// =============================================================================================
template <int GMAX>
struct AggLowCardArgs {
    GroupRec* partials;      // [grid][GMAX]
    uint32_t* partial_ng;    // [grid]
    ScanStatus* status;
};

constexpr int AGG_DEFAULT_R = 2;
constexpr bool AGG_DEFAULT_PREFETCH = true;

struct AggLds {
    Key128 keys[AGG_GMAX];
    uint32_t ng;
    uint32_t winner;
    uint32_t overflow;
    uint32_t pad;
    uint64_t red[4][AGG_GMAX * VM_MAX_ACC];
};

// The scan parameters (program, column pointers) are read through a pointer to a per-launch copy
// in device memory: the addresses are wave-uniform, so they are fetched with scalar loads.  (As a
// by-value kernel argument the 2 KB struct was copied to scratch once the next tile's loads were
// issued from inside the loop.)
// NACC_: accumulator registers per group.  8 for everything that scans; 16 only for tiny inputs (the Final aggregate over a few
// partial-state rows per rank: Q1 carries 11 accumulators), where one launch instead of the hash path's dozen is what counts
__device__ inline uint64_t readfirstlane_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// (a real call: inlined, the end-of-kernel reduction is GMAX x NACC copies of six shuffle steps around the combine's switch)
__device__ __noinline__ uint64_t wave_reduce_call(uint64_t v, int kind) { return wave_reduce(v, kind); }

// workgroups per CU the register allocator plans for.  The accumulators are 2 x GMAX x NACC VGPRs (+ a valid count each with
// NULLS): 128 of them get the whole SIMD's file (AGPRs included), 64 a cap of 256, fewer one of 168 — no variant spills
constexpr int lowcard_min_blocks(int gmax, int nacc, bool nulls) {
    const int w = gmax * nacc;
    return w >= 64 ? 1 : (w >= 32 || (nulls && w >= 8)) ? 2 : 3;
}

template <int R, bool NULLS, int GMAX, bool PREFETCH, int NACC_ = AGG_NACC>
__global__ void __launch_bounds__(BLOCK, lowcard_min_blocks(GMAX, NACC_, NULLS))
scan_agg_lowcard_kernel(const ScanParams* __restrict__ Pp, const AggLowCardArgs<GMAX> A) {
    // blockIdx.y = the input batch (one ScanParams each: several small batches — the partial states of N ranks in front of a Final
    // aggregate — go through ONE launch; this kernel's fixed cost, the 16-accumulator epilogue above all, is ~0.1 ms per launch)
    const ScanParams& P = Pp[blockIdx.y];
    constexpr int TILE = BLOCK * R;
    constexpr int NACC = NACC_;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const TileLds L = carve_tile_lds<R, NULLS>(lds_raw, P.prog);
    BHIP_LDS AggLds* S = (BHIP_LDS AggLds*)((lds_u8*)lds_raw + tile_lds_bytes<R>(P.prog.n_vslots, P.prog.n_bslots, NULLS));
    const int tid = threadIdx.x;
    const int n_acc = P.n_acc;
    // all accumulators are SUM(Float64) (TPC-H Q1/Q3/Q5/Q6): no per-accumulator dispatch in the hot loop
    bool all_sum_f64 = true;
    for (int a = 0; a < n_acc; ++a) all_sum_f64 &= (P.acc[a].kind == ACC_SUM_F64);
    const bool key64 = P.key_bytes <= 8;

    uint64_t acc[GMAX][NACC];
    // without NULLS every accumulated input is known, so nvalid == rows
    uint32_t nvalid[NULLS ? GMAX : 1][NULLS ? NACC : 1];
    uint32_t rows[GMAX];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const uint64_t id = (a < n_acc) ? acc_identity(P.acc[a].kind) : 0;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            acc[g][a] = id;
            if (NULLS) nvalid[NULLS ? g : 0][NULLS ? a : 0] = 0;
        }
    }
#pragma unroll
    for (int g = 0; g < GMAX; ++g) rows[g] = 0;
    if (tid == 0) { S->ng = 0; S->overflow = 0; S->winner = 0xFFFFFFFFu; }
    if (tid < GMAX) { S->keys[tid].k0 = 0; S->keys[tid].k1 = 0; }
    __syncthreads();

    uint32_t err = 0;
    const int64_t n_tiles = (P.n_rows + TILE - 1) / TILE;
    // register double buffering: the loads of the next tile are in flight while this one is
    // interpreted and accumulated (first LOAD_GROUP loads; any further ones load blocking)
    LoadRegs<PREFETCH ? R : 1, NULLS> pre;
    if (PREFETCH && (int64_t)blockIdx.x < n_tiles) {
        vm_load_issue_a<PREFETCH ? R : 1, NULLS>(P, (int64_t)blockIdx.x * TILE, 0, pre);
        vm_load_issue_b<PREFETCH ? R : 1, NULLS>(P, (int64_t)blockIdx.x * TILE, 0, pre, err);
    }
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t base = t * TILE;
        const int64_t next = t + gridDim.x;
        if (PREFETCH) {
            vm_load_commit<PREFETCH ? R : 1, NULLS>(P, L, 0, pre);
            if (next < n_tiles) vm_load_issue_a<PREFETCH ? R : 1, NULLS>(P, next * TILE, 0, pre);
            if (P.prog.n_loads > LOAD_GROUP) vm_load_tile<R, NULLS>(P, L, base, err, LOAD_GROUP);
        } else {
            vm_load_tile<R, NULLS>(P, L, base, err);
        }
        vm_execute<R, NULLS>(P, L, base, err);
        if (PREFETCH && next < n_tiles) vm_load_issue_b<PREFETCH ? R : 1, NULLS>(P, next * TILE, 0, pre, err);

        // ---- resolve each row's group among this workgroup's keys
        int ng = S->ng;
        Key128 gk[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {          // the same for every lane: scalar registers
            gk[g].k0 = readfirstlane_u64(S->keys[g].k0);
            gk[g].k1 = key64 ? 0 : readfirstlane_u64(S->keys[g].k1);
        }
        Key128 rk[R];
        int lg[R];
        bool pending = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = r * BLOCK + tid;
            bool live = (base + idx) < P.n_rows;
            if (P.pred_slot >= 0) live = live && (L.bvals[P.pred_slot * TILE + idx] & 1);
            rk[r] = Key128{0, 0};
            if (P.n_keyparts > 0) rk[r] = pack_key<R, NULLS>(P, L, base, r, err);
            int g_found = -2;   // live, group unknown
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                const bool eq = rk[r].k0 == gk[g].k0 && (key64 || rk[r].k1 == gk[g].k1);
                g_found = (g < ng && eq) ? g : g_found;
            }
            lg[r] = live ? g_found : -1;
            pending |= (lg[r] == -2);
        }
        // insertion rounds: rare (at most GMAX times per workgroup)
        while (__syncthreads_or(pending ? 1 : 0)) {
            if (pending) {
                uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
                for (int r = R - 1; r >= 0; --r)
                    if (lg[r] == -2) mine = (uint32_t)(r * BLOCK + tid);
                atomicMin((uint32_t*)&S->winner, mine);
            }
            __syncthreads();
            const uint32_t w = S->winner;
            const int cur = S->ng;
            __syncthreads();   // everyone has read (winner, ng) before the winner updates them
            if (cur >= GMAX) {
                if (tid == 0) S->overflow = 1;
            } else if ((w % BLOCK) == (uint32_t)tid) {
                Key128 wk{0, 0};
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if ((uint32_t)r == w / BLOCK) wk = rk[r];
                S->keys[cur].k0 = wk.k0;
                S->keys[cur].k1 = wk.k1;
                S->ng = cur + 1;
            }
            __syncthreads();
            if (tid == 0) S->winner = 0xFFFFFFFFu;
            if (S->overflow) break;
            const Key128 nk{S->keys[cur].k0, S->keys[cur].k1};
            pending = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (lg[r] == -2 && rk[r] == nk) lg[r] = cur;
                pending |= (lg[r] == -2);
            }
        }
        if (S->overflow) break;

        // ---- accumulate (registers, static indices, one predicated block per group)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = r * BLOCK + tid;
            if (all_sum_f64 && !NULLS) {
                // fast path: fetch the row's inputs once, then one exec-masked block of adds per group
                double v[NACC];
#pragma unroll
                for (int a = 0; a < NACC; ++a) v[a] = (a < n_acc) ? u2d(L.vals[P.acc[a].slot * TILE + idx]) : 0.0;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) {
                    if (lg[r] == g) {
                        rows[g] += 1;
#pragma unroll
                        for (int a = 0; a < NACC; ++a)
                            if (a < n_acc) acc[g][a] = d2u(u2d(acc[g][a]) + v[a]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < GMAX; ++g)
                    if (lg[r] == g) rows[g] += 1;
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    if (a < n_acc && lg[r] >= 0) {
                        const AccSpec sp = P.acc[a];
                        uint64_t v = 1;
                        bool k = true;
                        if (sp.kind == ACC_COUNT_VALID_B) {
                            k = L.bvals[sp.slot * TILE + idx] >> 1;
                        } else if (sp.kind != ACC_COUNT_ROWS) {
                            v = L.vals[sp.slot * TILE + idx];
                            if (NULLS) k = L.vvalid[sp.slot * TILE + idx];
                        }
                        if (sp.kind == ACC_COUNT_VALID || sp.kind == ACC_COUNT_VALID_B) { v = k ? 1 : 0; k = true; }
                        if (k) {
                            // the row's group picked and written back with static indices, ONE combine between (a combine per
                            // group is GMAX x NACC x R copies of its switch: past the unroller's budget, and the arrays go to scratch)
                            uint64_t cur = 0;
#pragma unroll
                            for (int g = 0; g < GMAX; ++g) cur = (lg[r] == g) ? acc[g][a] : cur;
                            cur = acc_combine(cur, v, sp.kind);
#pragma unroll
                            for (int g = 0; g < GMAX; ++g) {
                                acc[g][a] = (lg[r] == g) ? cur : acc[g][a];
                                if (NULLS) nvalid[NULLS ? g : 0][NULLS ? a : 0] += (lg[r] == g) ? 1u : 0u;
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- workgroup reduction in a fixed order: lanes (shuffle tree) -> waves 0..3
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    const int ng = S->ng;
    uint64_t tot_rows = 0, tot_acc = 0, tot_nv = 0;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const uint64_t v = wave_reduce((uint64_t)rows[g], ACC_COUNT_ROWS);
        if (lane == 0) S->red[wave][g] = v;
    }
    __syncthreads();
    if (tid < GMAX) tot_rows = S->red[0][tid] + S->red[1][tid] + S->red[2][tid] + S->red[3][tid];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int kind = (a < n_acc) ? P.acc[a].kind : ACC_COUNT_ROWS;
            const uint64_t v = wave_reduce_call(acc[g][a], kind);
            if (lane == 0) S->red[wave][g * NACC + a] = v;
        }
    __syncthreads();
    if (tid < GMAX * NACC) {
        const int a = tid % NACC;
        const int kind = (a < n_acc) ? P.acc[a].kind : ACC_COUNT_ROWS;
        tot_acc = S->red[0][tid];
        tot_acc = acc_combine(tot_acc, S->red[1][tid], kind);
        tot_acc = acc_combine(tot_acc, S->red[2][tid], kind);
        tot_acc = acc_combine(tot_acc, S->red[3][tid], kind);
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const uint64_t v = wave_reduce(NULLS ? (uint64_t)nvalid[NULLS ? g : 0][NULLS ? a : 0] : (uint64_t)rows[g], ACC_COUNT_ROWS);
            if (lane == 0) S->red[wave][g * NACC + a] = v;
        }
    __syncthreads();
    if (tid < GMAX * NACC) tot_nv = S->red[0][tid] + S->red[1][tid] + S->red[2][tid] + S->red[3][tid];

    const size_t part = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    GroupRec* out = A.partials + part * GMAX;
    if (tid < GMAX) {
        out[tid].k0 = S->keys[tid].k0;
        out[tid].k1 = S->keys[tid].k1;
        out[tid].rows = tot_rows;
    }
    if (tid < GMAX * NACC) {
        out[tid / NACC].acc[tid % NACC] = tot_acc;
        out[tid / NACC].nvalid[tid % NACC] = tot_nv;
    }
    if (tid == 0) {
        A.partial_ng[part] = (uint32_t)ng;
        if (S->overflow) atomicOr(&A.status->flags, SCAN_OVERFLOW_GROUPS);
    }
    if (err) atomicOr(&A.status->flags, err);
}

// ---- merge of per-workgroup partials -------------------------------------------------------
// Step 1 (one workgroup): give every partial record a group index in `table` (LDS hash of the
// packed key; insertion in rounds separated by barriers, so compares only see finished keys).
constexpr int MERGE_SLOTS = 2048;

__global__ void __launch_bounds__(1024)
merge_assign_kernel(const GroupRec* partials, const uint32_t* partial_ng, int n_part, int gmax,
                    GroupRec* table, int cap, uint32_t* entry_group, ScanStatus* status) {
    __shared__ Key128 skeys[MERGE_SLOTS];
    __shared__ uint32_t sgroup[MERGE_SLOTS];   // 0 = empty, else group index + 1
    __shared__ uint32_t s_ng, s_over;
    const int tid = threadIdx.x;
    for (int i = tid; i < MERGE_SLOTS; i += blockDim.x) sgroup[i] = 0;
    if (tid == 0) { s_ng = 0; s_over = 0; }
    __syncthreads();
    const int n_entries = n_part * gmax;
    for (int e0 = 0; e0 < n_entries; e0 += blockDim.x) {
        const int e = e0 + tid;
        bool live = false;
        Key128 k{0, 0};
        if (e < n_entries) {
            const int p = e / gmax, g = e % gmax;
            live = (uint32_t)g < partial_ng[p];
            if (live) k = Key128{partials[e].k0, partials[e].k1};
            else entry_group[e] = 0xFFFFFFFFu;
        }
        uint32_t slot = (uint32_t)(hash_key(k) & (MERGE_SLOTS - 1));
        bool pending = live;
        // sgroup[slot]: 0 = empty, CLAIMED = being written this round, else group index + 1
        constexpr uint32_t CLAIMED = 0xFFFFFFFFu;
        while (__syncthreads_or(pending ? 1 : 0)) {
            if (s_over) break;
            if (pending) {
                for (;;) {
                    const uint32_t gidx = sgroup[slot];
                    if (gidx == 0 || gidx == CLAIMED) break;   // CLAIMED: finished by the next round, look again then
                    if (skeys[slot] == k) { entry_group[e] = gidx - 1; pending = false; break; }
                    slot = (slot + 1) & (MERGE_SLOTS - 1);
                }
                if (pending && atomicCAS(&sgroup[slot], 0u, CLAIMED) == 0u) {
                    const uint32_t gi = atomicAdd(&s_ng, 1u);
                    if ((int)gi >= cap || gi >= MERGE_SLOTS / 2) {
                        s_over = 1;
                    } else {
                        skeys[slot] = k;
                        table[gi].k0 = k.k0;
                        table[gi].k1 = k.k1;
                        entry_group[e] = gi;
                        __threadfence_block();
                        sgroup[slot] = gi + 1;
                    }
                    pending = false;
                }
            }
        }
        if (s_over) break;
    }
    __syncthreads();
    if (tid == 0) {
        status->n_groups = s_over ? 0 : s_ng;
        if (s_over) atomicOr(&status->flags, SCAN_OVERFLOW_GROUPS);
    }
}

// Step 2: one workgroup per (group, accumulator): threads stride over the partial records in record order, a fixed
// shuffle tree per wave, then the four wave results in wave order => deterministic.
struct MergeAccSpecs { AccSpec acc[VM_MAX_ACC]; int n_acc; };

__global__ void __launch_bounds__(BLOCK)
merge_reduce_kernel(const GroupRec* partials, const uint32_t* entry_group, int n_entries,
                    MergeAccSpecs specs, GroupRec* table, const ScanStatus* status) {
    __shared__ uint64_t s_v[BLOCK / 64], s_nv[BLOCK / 64];
    const int n_groups = (int)status->n_groups;
    const int per_group = specs.n_acc + 1;       // + rows
    const int n_pairs = n_groups * per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const int g = pair / per_group, a = pair % per_group;
        const bool is_rows = (a == specs.n_acc);
        const int kind = is_rows ? (int)ACC_COUNT_ROWS : (int)specs.acc[a].kind;
        uint64_t v = is_rows ? 0 : acc_identity(kind);
        uint64_t nv = 0;
        // eight records per thread at a time: their group indices first, then every value load at once (unconditional, from a
        // safe address) — two memory latencies per eight records instead of sixteen; the combine order stays the record order
        constexpr int U = 8;
        for (int e0 = threadIdx.x; e0 < n_entries; e0 += BLOCK * U) {
            uint32_t eg[U];
            uint64_t x[U], xn[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { const int e = e0 + k * BLOCK; eg[k] = e < n_entries ? entry_group[e] : 0xFFFFFFFFu; }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + k * BLOCK;
                const GroupRec& p = partials[eg[k] == (uint32_t)g ? e : 0];
                x[k] = is_rows ? p.rows : p.acc[is_rows ? 0 : a];
                xn[k] = is_rows ? 0 : p.nvalid[is_rows ? 0 : a];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (eg[k] == (uint32_t)g) {
                    if (is_rows) v += x[k];
                    else { v = acc_combine(v, x[k], kind); nv += xn[k]; }   // empty partials hold the identity
                }
            }
        }
        v = wave_reduce(v, kind);
        nv = wave_reduce(nv, ACC_COUNT_ROWS);
        if (lane == 0) { s_v[wave] = v; s_nv[wave] = nv; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t tv = s_v[0], tn = s_nv[0];
            for (int w = 1; w < BLOCK / 64; ++w) {
                tv = is_rows ? tv + s_v[w] : acc_combine(tv, s_v[w], kind);
                tn += s_nv[w];
            }
            if (is_rows) table[g].rows = tv;
            else { table[g].acc[a] = tv; table[g].nvalid[a] = tn; }
        }
        __syncthreads();
    }
}

int scan_agg_lowcard_max_grid(const LaunchCfg& cfg) { return cfg.device_cus * 8; }

template <int R, bool NULLS, int GMAX, bool PREFETCH, int NACC_ = AGG_NACC>
static hipError_t launch_lowcard_t(const LaunchCfg& cfg, const ScanParams& P, ScanParams* dparams, GroupRec* partials,
                                   uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out, int n_batches = 1) {
    // n_batches > 1: &P is the first of n_batches ScanParams (the same program bound to different batches), dparams has room for all
    constexpr int TILE = BLOCK * R;
    const size_t lds = host_tile_bytes<R>(P.prog) + sizeof(AggLds);
    if (lds > LDS_PER_CU) return hipErrorInvalidValue;
    int64_t n_tiles = 0;
    for (int b = 0; b < n_batches; ++b) n_tiles = std::max<int64_t>(n_tiles, ((&P)[b].n_rows + TILE - 1) / TILE);
    auto k = scan_agg_lowcard_kernel<R, NULLS, GMAX, PREFETCH, NACC_>;
    hipError_t e = set_lds(k, lds);
    if (e != hipSuccess) return e;
    // one grid-stride wave of resident workgroups: every workgroup gets the same number of tiles
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), BLOCK, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    static const int forced_per_cu = [] { const char* v = getenv("BHIP_AGG_BLOCKS_PER_CU"); return v ? atoi(v) : 0; }();
    if (forced_per_cu > 0) per_cu = forced_per_cu;
    int64_t grid = (int64_t)cfg.device_cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    e = hipMemcpyAsync(dparams, &P, sizeof(ScanParams) * (size_t)n_batches, hipMemcpyHostToDevice, cfg.stream);
    if (e != hipSuccess) return e;
    AggLowCardArgs<GMAX> A{partials, partial_ng, status};
    hipLaunchKernelGGL(k, dim3((unsigned)grid, (unsigned)n_batches), dim3(BLOCK), lds, cfg.stream, (const ScanParams*)dparams, A);
    *grid_out = (int)grid * n_batches;
    return hipGetLastError();
}

// rows per thread R and register prefetch per variant.  Defaults come from tuning on MI355X
// (DESIGN.md "Kernel tuning"); BHIP_SCAN_R / BHIP_PREFETCH select other variants in a -DBHIP_TUNE build.
template <bool NULLS>
static hipError_t launch_lowcard_n(const LaunchCfg& cfg, const ScanParams& P, ScanParams* dparams, int gmax,
                                   GroupRec* partials, uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out, int n_batches) {
#define BHIP_LC(R_, G_, PF_) launch_lowcard_t<R_, NULLS, G_, PF_>(cfg, P, dparams, partials, partial_ng, max_grid, status, grid_out, n_batches)
#ifdef BHIP_TUNE
    static const int r_env = [] { const char* v = getenv("BHIP_SCAN_R"); return v ? atoi(v) : 0; }();
    static const int pf_env = [] { const char* v = getenv("BHIP_PREFETCH"); return v ? atoi(v) : -1; }();
    if (!NULLS && gmax == 4 && (r_env > 0 || pf_env >= 0)) {
        const int r = r_env > 0 ? r_env : AGG_DEFAULT_R;
        const bool pf = pf_env >= 0 ? pf_env != 0 : AGG_DEFAULT_PREFETCH;
        if (r == 1) return pf ? BHIP_LC(1, 4, true) : BHIP_LC(1, 4, false);
        if (r == 2) return pf ? BHIP_LC(2, 4, true) : BHIP_LC(2, 4, false);
        if (r == 4) return pf ? BHIP_LC(4, 4, true) : BHIP_LC(4, 4, false);
    }
#endif
    if (P.n_acc > AGG_NACC) {                      // tiny inputs only (host/ops_agg.cpp): 16 accumulator registers per group
        if (gmax == 4) return launch_lowcard_t<1, NULLS, 4, false, VM_MAX_ACC>(cfg, P, dparams, partials, partial_ng, max_grid, status, grid_out, n_batches);
        if (gmax == 1) return launch_lowcard_t<1, NULLS, 1, false, VM_MAX_ACC>(cfg, P, dparams, partials, partial_ng, max_grid, status, grid_out, n_batches);
        return hipErrorInvalidValue;
    }
    // 8 groups x 8 accumulators take 128 VGPRs: no room for prefetch registers
    if (P.n_acc <= 4) {                            // (TPC-H Q3 / Q5 / Q6 carry one SUM): a quarter / half of the accumulator registers
        if (gmax == 8) return launch_lowcard_t<2, NULLS, 8, false, 4>(cfg, P, dparams, partials, partial_ng, max_grid, status, grid_out, n_batches);
        if (gmax == 4) return launch_lowcard_t<AGG_DEFAULT_R, NULLS, 4, AGG_DEFAULT_PREFETCH, 4>(cfg, P, dparams, partials, partial_ng, max_grid, status, grid_out, n_batches);
    }
    if (gmax == 8) return BHIP_LC(2, 8, false);
    if (gmax == 4) return BHIP_LC(AGG_DEFAULT_R, 4, AGG_DEFAULT_PREFETCH);
    if (gmax == 1) return BHIP_LC(AGG_DEFAULT_R, 1, AGG_DEFAULT_PREFETCH);
    return hipErrorInvalidValue;
#undef BHIP_LC
}

hipError_t launch_scan_agg_lowcard(const LaunchCfg& cfg, const ScanParams& P, ScanParams* dparams, int gmax,
                                   GroupRec* partials, uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out, int n_batches) {
    if (n_batches < 1 || n_batches > 65535) return hipErrorInvalidValue;
    return P.prog.nullable ? launch_lowcard_n<true>(cfg, P, dparams, gmax, partials, partial_ng, max_grid, status, grid_out, n_batches)
                           : launch_lowcard_n<false>(cfg, P, dparams, gmax, partials, partial_ng, max_grid, status, grid_out, n_batches);
}

hipError_t launch_merge_partials(const LaunchCfg& cfg, const GroupRec* partials, const uint32_t* partial_ng,
                                 int n_part, int gmax, const AccSpec* acc_host, int n_acc, GroupRec* table, int cap,
                                 uint32_t* entry_group, ScanStatus* status) {
    hipLaunchKernelGGL(merge_assign_kernel, dim3(1), dim3(1024), 0, cfg.stream, partials, partial_ng, n_part, gmax,
                       table, cap, entry_group, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    MergeAccSpecs specs;
    specs.n_acc = n_acc;
    for (int i = 0; i < VM_MAX_ACC; ++i) specs.acc[i] = i < n_acc ? acc_host[i] : AccSpec{ACC_COUNT_ROWS, 0, {0, 0}};
    hipLaunchKernelGGL(merge_reduce_kernel, dim3(256), dim3(BLOCK), 0, cfg.stream, partials, entry_group,
                       n_part * gmax, specs, table, status);
    return hipGetLastError();
}

}  // namespace bhip
