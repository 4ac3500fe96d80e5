// kernels_dagg.hip — SUM(Float64) per group in a fixed order (the hash path of HashAggregateExec,
// rust/core/src/serde/physical_plan/from_proto.rs:173-252).  See host/hash_kernels.h "Float64 sums in a fixed order".
//
// The reference folds each group's values sequentially per batch and then across batches — not
// reproducible across its own partitionings; this path is reproducible by construction: every group's addends are combined in
// row order, the association being a function of the row positions alone (64-row chunks, 1024-row tiles), never of the
// scheduling of the waves.  Algorithmic bytes: 4 B slot + 8 B per summed accumulator per row, read once.
#include <hip/hip_runtime.h>
#include "host/hash_kernels.h"
#include "launch_common.h"

namespace bhip {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int DS_TILE = 1024;
constexpr int DS_MAX = 4;          // accumulators summed per pass of the segment kernel (more: several passes over the same runs)

// one wave per tile; k0 .. k0 + nk - 1 = the accumulators of this pass
template <int NK>
__global__ void __launch_bounds__(BLOCK)
det_segments_kernel(const DetSum D, int k0, int first_pass) {
    const int lane = threadIdx.x & 63;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    const uint64_t n = D.total_rows;
    const uint32_t n_tiles = (uint32_t)((n + DS_TILE - 1) / DS_TILE);
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    const size_t stage_n = (size_t)n_tiles * DS_TILE;
    for (uint32_t t = wave_id; t < n_tiles; t += n_waves) {
        const uint64_t tile_base = (uint64_t)t * DS_TILE;
        uint32_t n_out = 0;
        // the run that is still open at the end of the previous chunk (wave-uniform)
        uint32_t c_slot = NONE, c_first = 0, c_last = 0;
        double c_sum[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) c_sum[k] = 0.0;
        auto emit_carry = [&]() {
            if (c_slot == NONE) return;
            if (lane == 0) {
                const size_t at = tile_base + n_out;
                if (first_pass) {
                    D.seg_slot[at] = c_slot;
                    D.seg_first[at] = c_first;
                    atomicAdd(&D.runs[c_slot], 1u);
                    atomicAdd(reinterpret_cast<unsigned long long*>(&D.rows[c_slot]), (unsigned long long)(c_last - c_first + 1u));
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) D.seg_sum[(size_t)(k0 + k) * stage_n + at] = c_sum[k];
            }
            n_out += 1;
            c_slot = NONE;
        };
        for (int c = 0; c < DS_TILE / 64; ++c) {
            const uint64_t row = tile_base + 64ull * c + lane;
            const bool in = row < n;
            const uint32_t slot = in ? D.rowslot[row] : NONE;
            double v[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) v[k] = (in && slot != NONE) ? D.fvals[(size_t)(k0 + k) * n + row] : 0.0;
            const uint32_t prev = __shfl_up(slot, 1, 64);
            const uint32_t first_slot = __shfl(slot, 0, 64);
            // the open run continues into this chunk: its sum goes in front of the chunk's first row
            const bool continues = c_slot != NONE && first_slot == c_slot;
            if (!continues) emit_carry();
            const bool head = lane == 0 || prev != slot;
            uint32_t first_row = (uint32_t)row;
            if (continues && lane == 0) {
                first_row = c_first;
#pragma unroll
                for (int k = 0; k < NK; ++k) v[k] = c_sum[k] + v[k];
            }
            c_slot = NONE;
            // inclusive segmented scan over runs of equal slot (fixed tree); the head's first row travels along
            const uint64_t heads = __ballot(head);
            const int my_head = 63 - __clzll(heads & (lane_lt | (1ull << lane)));      // lane of my run's head
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                double pv[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) pv[k] = __shfl_up(v[k], off, 64);
                if (lane - off >= my_head) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) v[k] = pv[k] + v[k];
                }
            }
            first_row = __shfl(first_row, my_head, 64);
            // run ends: the lane before the next head, or lane 63 (that run stays open)
            const bool last_of_run = lane == 63 || ((heads >> (lane + 1)) & 1ull);
            const bool closes = last_of_run && lane != 63 && slot != NONE;
            const uint64_t cw = __ballot(closes);
            if (closes) {
                const size_t at = tile_base + n_out + (uint32_t)__popcll(cw & lane_lt);
                if (first_pass) {
                    D.seg_slot[at] = slot;
                    D.seg_first[at] = first_row;
                    atomicAdd(&D.runs[slot], 1u);
                    atomicAdd(reinterpret_cast<unsigned long long*>(&D.rows[slot]), (unsigned long long)((uint32_t)row - first_row + 1u));
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) D.seg_sum[(size_t)(k0 + k) * stage_n + at] = v[k];
            }
            n_out += (uint32_t)__popcll(cw);
            // lane 63's run is the new carry
            c_slot = __shfl(slot, 63, 64);
            c_first = __shfl(first_row, 63, 64);
            c_last = (uint32_t)(tile_base + 64ull * c + 63);
#pragma unroll
            for (int k = 0; k < NK; ++k) c_sum[k] = __shfl(v[k], 63, 64);
        }
        emit_carry();
        if (first_pass && lane == 0) D.tile_nseg[t] = n_out;
    }
}

// slots with one run take its sums; the runs of the others are listed for the ordered combine
__global__ void __launch_bounds__(BLOCK)
det_apply_kernel(const DetSum D) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_tiles = (uint32_t)((D.total_rows + DS_TILE - 1) / DS_TILE);
    const size_t stage_n = (size_t)n_tiles * DS_TILE;
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    for (uint32_t t = wave_id; t < n_tiles; t += n_waves) {
        const uint32_t cnt = D.tile_nseg[t];
        for (uint32_t j = lane; j < cnt; j += 64) {
            const size_t at = (size_t)t * DS_TILE + j;
            const uint32_t slot = D.seg_slot[at];
            if (D.runs[slot] == 1u) {
                for (int k = 0; k < D.n_fsum; ++k)
                    D.acc[(size_t)slot * D.n_acc + D.acc_of_fsum[k]] = (uint64_t)__double_as_longlong(D.seg_sum[(size_t)k * stage_n + at]);
            } else {
                const uint32_t pos = atomicAdd(D.spill_count, 1u);
                D.spill_key[pos] = ((uint64_t)slot << 32) | D.seg_first[at];
                D.spill_seg[pos] = (uint32_t)at;
                D.spill_next[pos] = atomicExch(&D.spill_head[slot], pos);
            }
        }
    }
}

// One thread per spill entry walks its group's list (at most DET_LIST_MAX entries: first row + run index of each).  The entry with
// the SMALLEST first row orders the list by first row (insertion into a sorted prefix: the arrays stay in registers, every index
// is static) and adds the runs' sums left to right; the others have nothing to do.  The result does not depend on the order in
// which the entries were linked.  A longer list raises the flag for the sorted combine.
__global__ void __launch_bounds__(BLOCK)
det_spill_lists_kernel(const DetSum D) {
    const uint32_t n_spill = D.spill_count[0];
    const uint32_t n_tiles = (uint32_t)((D.total_rows + DS_TILE - 1) / DS_TILE);
    const size_t stage_n = (size_t)n_tiles * DS_TILE;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_spill; i += gridDim.x * BLOCK) {
        const uint32_t slot = (uint32_t)(D.spill_key[i] >> 32), mine = (uint32_t)D.spill_key[i];
        uint32_t first[DET_LIST_MAX], seg[DET_LIST_MAX];
#pragma unroll
        for (int q = 0; q < DET_LIST_MAX; ++q) { first[q] = NONE; seg[q] = 0; }
        uint32_t e = D.spill_head[slot];
        int n = 0;
        bool leader = true;
        while (e != NONE && n <= DET_LIST_MAX) {
            const uint32_t f = (uint32_t)D.spill_key[e], sg = D.spill_seg[e];
            leader = leader && f >= mine;
            if (n < DET_LIST_MAX) {
                // insert (f, sg) into the sorted prefix: everything greater moves one place up
                uint32_t cf = f, cs = sg;
#pragma unroll
                for (int q = 0; q < DET_LIST_MAX; ++q) {
                    const bool sw = cf < first[q];
                    const uint32_t tf = first[q], ts = seg[q];
                    first[q] = sw ? cf : tf; seg[q] = sw ? cs : ts;
                    cf = sw ? tf : cf; cs = sw ? ts : cs;
                }
            }
            ++n;
            e = D.spill_next[e];
        }
        if (n > DET_LIST_MAX) { D.spill_count[1] = 1u; continue; }           // too many runs for the walk: the sorted combine takes over
        if (!leader) continue;
        for (int k = 0; k < D.n_fsum; ++k) {
            double total = 0.0;
#pragma unroll
            for (int q = 0; q < DET_LIST_MAX; ++q) {
                if (q < n) {
                    const double v = D.seg_sum[(size_t)k * stage_n + seg[q]];
                    total = q == 0 ? v : total + v;
                }
            }
            D.acc[(size_t)slot * D.n_acc + D.acc_of_fsum[k]] = (uint64_t)__double_as_longlong(total);
        }
    }
}

// the spill list sorted by (slot, first row): the thread at the head of a slot's runs adds them up left to right
__global__ void __launch_bounds__(BLOCK)
det_spill_combine_kernel(const DetSum D, const uint64_t* __restrict__ key, const uint32_t* __restrict__ seg, uint32_t n_spill) {
    const uint32_t n_tiles = (uint32_t)((D.total_rows + DS_TILE - 1) / DS_TILE);
    const size_t stage_n = (size_t)n_tiles * DS_TILE;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_spill; i += gridDim.x * BLOCK) {
        const uint32_t slot = (uint32_t)(key[i] >> 32);
        if (i > 0 && (uint32_t)(key[i - 1] >> 32) == slot) continue;
        for (int k = 0; k < D.n_fsum; ++k) {
            double total = D.seg_sum[(size_t)k * stage_n + seg[i]];
            for (uint32_t j = i + 1; j < n_spill && (uint32_t)(key[j] >> 32) == slot; ++j) total = total + D.seg_sum[(size_t)k * stage_n + seg[j]];
            D.acc[(size_t)slot * D.n_acc + D.acc_of_fsum[k]] = (uint64_t)__double_as_longlong(total);
        }
    }
}

int tile_grid(const LaunchCfg& cfg, uint64_t n_rows) {
    const int64_t n_tiles = (int64_t)((n_rows + DS_TILE - 1) / DS_TILE);
    int64_t grid = (int64_t)cfg.device_cus * 8;
    const int64_t need = (n_tiles + BLOCK / 64 - 1) / (BLOCK / 64);
    if (grid > need) grid = need;
    return grid < 1 ? 1 : (int)grid;
}

}  // namespace

hipError_t launch_det_segments(const LaunchCfg& cfg, const DetSum& D) {
    if (D.total_rows == 0) return hipSuccess;
    const int grid = tile_grid(cfg, D.total_rows);
    for (int k0 = 0; k0 < D.n_fsum; k0 += DS_MAX) {
        const int nk = D.n_fsum - k0 < DS_MAX ? D.n_fsum - k0 : DS_MAX;
        const int first = k0 == 0 ? 1 : 0;
        switch (nk) {
            case 1: hipLaunchKernelGGL(det_segments_kernel<1>, dim3(grid), dim3(BLOCK), 0, cfg.stream, D, k0, first); break;
            case 2: hipLaunchKernelGGL(det_segments_kernel<2>, dim3(grid), dim3(BLOCK), 0, cfg.stream, D, k0, first); break;
            case 3: hipLaunchKernelGGL(det_segments_kernel<3>, dim3(grid), dim3(BLOCK), 0, cfg.stream, D, k0, first); break;
            default: hipLaunchKernelGGL(det_segments_kernel<4>, dim3(grid), dim3(BLOCK), 0, cfg.stream, D, k0, first); break;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_det_apply(const LaunchCfg& cfg, const DetSum& D) {
    if (D.total_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(det_apply_kernel, dim3(tile_grid(cfg, D.total_rows)), dim3(BLOCK), 0, cfg.stream, D);
    return hipGetLastError();
}

hipError_t launch_det_spill_lists(const LaunchCfg& cfg, const DetSum& D) {
    if (D.total_rows == 0) return hipSuccess;
    // the number of entries is on the device: a modest grid, grid-stride over the list
    hipLaunchKernelGGL(det_spill_lists_kernel, dim3((unsigned)cfg.device_cus * 2), dim3(BLOCK), 0, cfg.stream, D);
    return hipGetLastError();
}

hipError_t launch_det_spill_combine(const LaunchCfg& cfg, const DetSum& D, const uint64_t* sorted_key, const uint32_t* sorted_seg, uint32_t n_spill) {
    if (n_spill == 0) return hipSuccess;
    size_t g = ((size_t)n_spill + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(det_spill_combine_kernel, dim3((unsigned)g), dim3(BLOCK), 0, cfg.stream, D, sorted_key, sorted_seg, n_spill);
    return hipGetLastError();
}

}  // namespace bhip
