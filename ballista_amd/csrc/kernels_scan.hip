// kernels_scan.hip — the fused scan kernels: column loads -> expression VM -> sink.
//
// One kernel replaces the reference's FilterExec -> [CoalesceBatches] -> ProjectionExec ->
// HashAggregateExec(Partial) chain (operators built at rust/core/src/serde/physical_plan/
// from_proto.rs:69-92,122-128,173-252; executed from rust/executor/src/flight_service.rs:
// 117-121).  HBM traffic = each referenced input column read once (coalesced, one row per
// lane) + a few KB of per-workgroup partial state.  Bound: HBM bandwidth.
//
// Sinks:
//   AggLowCard<GMAX>  group-by with <= GMAX groups per workgroup: keys in LDS, accumulators
//                     in registers (static indices; updates predicated on the group id), a
//                     fixed-order lane -> wave -> workgroup reduction => run-to-run
//                     deterministic sums.  More groups => SCAN_OVERFLOW_GROUPS, the host
//                     re-runs on the hash path.
//   Project           writes result columns (Boolean / validity as ballot words).
//   PredBitmap        selection bitmap + per-tile popcounts (FilterExec compaction step 1).
//   Keys              packed 16-byte keys and/or 64-bit row hashes (+ selection bitmap).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "vm_device.h"
#include "reduce_device.h"
#include "launch_common.h"

namespace bhip {

// =============================================================================================
// Sink: projection
// =============================================================================================
template <int R, bool NULLS>
__global__ void __launch_bounds__(BLOCK)
scan_project_kernel(const ScanParams P, const ProjectOut O, ScanStatus* status) {
    constexpr int TILE = BLOCK * R;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const TileLds L = carve_tile_lds<R, NULLS>(lds_raw, P.prog);
    const int tid = threadIdx.x;
    uint32_t err = 0;
    const int64_t n_tiles = (P.n_rows + TILE - 1) / TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t base = t * TILE;
        vm_load_tile<R, NULLS>(P, L, base, err);
        vm_execute<R, NULLS>(P, L, base, err);
        for (int j = 0; j < P.n_out; ++j) {
            const int slot = P.out_slot[j];
            const int dt = P.out_dtype[j];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = r * BLOCK + tid;
                const int64_t row = base + idx;
                const bool in = row < P.n_rows;
                bool known = true;
                if (dt == DT_BOOLEAN) {
                    const uint8_t b = in ? L.bvals[slot * TILE + idx] : 0;
                    known = b >> 1;
                    const uint64_t word = __ballot(b & 1);
                    if ((tid & 63) == 0 && (row - (tid & 63)) < P.n_rows)
                        gptr_w<uint64_t>(O.data[j])[row >> 6] = word;
                } else {
                    const uint64_t v = L.vals[slot * TILE + idx];
                    if (NULLS) known = L.vvalid[slot * TILE + idx];
                    if (in) dt_store(dt, O.data[j], row, v);
                }
                if (O.validity[j] != nullptr) {
                    const uint64_t vw = __ballot(in && known);
                    if ((tid & 63) == 0 && (row - (tid & 63)) < P.n_rows) gptr_w<uint64_t>(O.validity[j])[row >> 6] = vw;
                }
            }
        }
    }
    if (err) atomicOr(&status->flags, err);
}

// =============================================================================================
// Sink: predicate -> selection bitmap + per-1024-row counts
// =============================================================================================
template <int R, bool NULLS>
__global__ void __launch_bounds__(BLOCK)
scan_pred_bitmap_kernel(const ScanParams P, uint64_t* bitmap, uint32_t* tile_counts, ScanStatus* status) {
    constexpr int TILE = BLOCK * R;
    static_assert(TILE == SEL_TILE, "selection tiles are 1024 rows");
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const TileLds L = carve_tile_lds<R, NULLS>(lds_raw, P.prog);
    __shared__ uint32_t s_cnt;
    const int tid = threadIdx.x;
    uint32_t err = 0;
    const int64_t n_tiles = (P.n_rows + TILE - 1) / TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t base = t * TILE;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        vm_load_tile<R, NULLS>(P, L, base, err);
        vm_execute<R, NULLS>(P, L, base, err);
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = r * BLOCK + tid;
            const int64_t row = base + idx;
            bool sel = row < P.n_rows;
            if (sel && P.pred_slot >= 0) sel = L.bvals[P.pred_slot * TILE + idx] & 1;
            const uint64_t word = __ballot(sel);
            if ((tid & 63) == 0) {
                if ((row) < P.n_rows) gptr_w<uint64_t>(bitmap)[row >> 6] = word;
                cnt += (uint32_t)__popcll(word);
            }
        }
        if ((tid & 63) == 0) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[t] = s_cnt;
        __syncthreads();
    }
    if (err) atomicOr(&status->flags, err);
}

// =============================================================================================
// Sink: packed keys / row hashes
// =============================================================================================
// Row hash (RepartitionExec(Hash), join-key hashing): h = 0; per key column h = mix64(h ^ bits)
// with bits = the value as sign/zero-extended 64-bit integer, f64 bit pattern (-0.0 -> +0.0),
// FNV-1a-64 of Utf8 bytes, "nullnull" for NULL.  Restated in oracle/engine.py::row_hash.
template <int R, bool NULLS>
__device__ inline uint64_t hash_row(const ScanParams& P, const TileLds& L, int64_t base, int r) {
    constexpr int TILE = BLOCK * R;
    const int idx = r * BLOCK + threadIdx.x;
    const int64_t row = base + idx;
    uint64_t h = 0;
    for (int i = 0; i < P.n_keyparts; ++i) {
        const KeyPart kp = P.keyparts[i];
        uint64_t bits = 0x6E756C6C6E756C6Cull;
        if (kp.kind == KP_VSLOT || kp.kind == KP_VSLOT_F64) {
            const bool valid = NULLS ? (bool)L.vvalid[kp.src * TILE + idx] : true;
            if (valid) {
                bits = L.vals[kp.src * TILE + idx];
                if (kp.kind == KP_VSLOT_F64 && bits == 0x8000000000000000ull) bits = 0;
            }
        } else if (kp.kind == KP_BSLOT) {
            const uint8_t b = L.bvals[kp.src * TILE + idx];
            if (b >> 1) bits = b & 1;
        } else if (row < P.n_rows) {
            const ColumnRef& c = P.cols[kp.src];
            if (!NULLS || column_valid_bit(c, row)) {
                const int32_t o0 = gptr<int32_t>(c.offsets)[row], o1 = gptr<int32_t>(c.offsets)[row + 1];
                const BHIP_GLOBAL uint8_t* s = gptr<uint8_t>(c.data);
                uint64_t f = 0xCBF29CE484222325ull;
                for (int32_t o = o0; o < o1; ++o) f = (f ^ s[o]) * 0x100000001B3ull;
                bits = f;
            }
        }
        h = mix64(h ^ bits);
    }
    return h;
}

template <int R, bool NULLS>
__global__ void __launch_bounds__(BLOCK)
scan_keys_kernel(const ScanParams P, uint64_t* keys128, uint64_t* hashes, uint64_t* bitmap, ScanStatus* status) {
    constexpr int TILE = BLOCK * R;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const TileLds L = carve_tile_lds<R, NULLS>(lds_raw, P.prog);
    const int tid = threadIdx.x;
    uint32_t err = 0;
    const int64_t n_tiles = (P.n_rows + TILE - 1) / TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t base = t * TILE;
        vm_load_tile<R, NULLS>(P, L, base, err);
        vm_execute<R, NULLS>(P, L, base, err);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = r * BLOCK + tid;
            const int64_t row = base + idx;
            const bool in = row < P.n_rows;
            bool sel = in;
            if (sel && P.pred_slot >= 0) sel = L.bvals[P.pred_slot * TILE + idx] & 1;
            if (keys128 != nullptr) {
                const Key128 k = pack_key<R, NULLS>(P, L, base, r, err);
                if (in) { gptr_w<uint64_t>(keys128)[2 * row] = k.k0; gptr_w<uint64_t>(keys128)[2 * row + 1] = k.k1; }
            }
            if (hashes != nullptr) {
                const uint64_t h = hash_row<R, NULLS>(P, L, base, r);
                if (in) gptr_w<uint64_t>(hashes)[row] = h;
            }
            if (bitmap != nullptr) {
                const uint64_t word = __ballot(sel);
                if ((tid & 63) == 0 && row < P.n_rows) gptr_w<uint64_t>(bitmap)[row >> 6] = word;
            }
        }
    }
    if (err) atomicOr(&status->flags, err);
}

template <bool NULLS>
static hipError_t launch_project_n(const LaunchCfg& cfg, const ScanParams& P, const ProjectOut& out, ScanStatus* status) {
    const int r = choose_r(P.prog, 0);
    const size_t lds = r == 4 ? host_tile_bytes<4>(P.prog) : host_tile_bytes<2>(P.prog);
    const int64_t n_tiles = (P.n_rows + BLOCK * r - 1) / (BLOCK * r);
    const int grid = pick_grid(cfg, n_tiles, lds, 4) ;
    hipError_t e;
    if (r == 4) {
        auto k = scan_project_kernel<4, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, out, status);
    } else {
        auto k = scan_project_kernel<2, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, out, status);
    }
    return hipGetLastError();
}

hipError_t launch_scan_project(const LaunchCfg& cfg, const ScanParams& P, const ProjectOut& out, ScanStatus* status) {
    return P.prog.nullable ? launch_project_n<true>(cfg, P, out, status) : launch_project_n<false>(cfg, P, out, status);
}

hipError_t launch_scan_pred_bitmap(const LaunchCfg& cfg, const ScanParams& P, uint64_t* bitmap, uint32_t* tile_counts,
                                   ScanStatus* status) {
    const size_t lds = host_tile_bytes<4>(P.prog);
    if (lds > LDS_PER_CU) return hipErrorInvalidValue;
    const int64_t n_tiles = (P.n_rows + SEL_TILE - 1) / SEL_TILE;
    const int grid = pick_grid(cfg, n_tiles, lds, 4);
    hipError_t e;
    if (P.prog.nullable) {
        auto k = scan_pred_bitmap_kernel<4, true>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, bitmap, tile_counts, status);
    } else {
        auto k = scan_pred_bitmap_kernel<4, false>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, bitmap, tile_counts, status);
    }
    return hipGetLastError();
}

template <bool NULLS>
static hipError_t launch_keys_n(const LaunchCfg& cfg, const ScanParams& P, uint64_t* keys128, uint64_t* hashes,
                                uint64_t* bitmap, ScanStatus* status) {
    const int r = choose_r(P.prog, 0);
    const size_t lds = r == 4 ? host_tile_bytes<4>(P.prog) : host_tile_bytes<2>(P.prog);
    const int64_t n_tiles = (P.n_rows + BLOCK * r - 1) / (BLOCK * r);
    const int grid = pick_grid(cfg, n_tiles, lds, 4);
    hipError_t e;
    if (r == 4) {
        auto k = scan_keys_kernel<4, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, keys128, hashes, bitmap, status);
    } else {
        auto k = scan_keys_kernel<2, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, keys128, hashes, bitmap, status);
    }
    return hipGetLastError();
}

hipError_t launch_scan_keys(const LaunchCfg& cfg, const ScanParams& P, uint64_t* keys128, uint64_t* hashes,
                            uint64_t* bitmap, ScanStatus* status) {
    return P.prog.nullable ? launch_keys_n<true>(cfg, P, keys128, hashes, bitmap, status)
                           : launch_keys_n<false>(cfg, P, keys128, hashes, bitmap, status);
}

}  // namespace bhip
