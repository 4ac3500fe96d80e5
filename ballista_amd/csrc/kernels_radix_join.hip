// kernels_radix_join.hip — the radix-partitioned hash join with LDS-resident tables: the textbook GPU design the project's brief
// names ("LDS-staged hash tables"), kept as the measured alternative to the rank map of kernels_join.hip (BHIP_JOIN_RADIX=1;
// profiles/r02_join_ab.txt holds the A/B).  HashJoinExec (rust/core/src/serde/physical_plan/from_proto.rs:253-276), Inner, ONE
// Int32 / Date32 key, unique NULL-free build side.
//
//   partition  both sides are brought into partition order by the stable 8-bit radix passes of kernels_sort.hip (in-workgroup
//              ranks, no global atomics) over the sort key  (partition id << 32) | key,  partition id = high bits of mix64(key),
//              so that a partition of the build side fits one workgroup's LDS table (4096 slots of 8 bytes, <= 75 % full);
//   join       one workgroup per partition: its build rows go into the LDS table (LDS atomics), its probe rows are streamed
//              against it — every table access is an LDS access, HBM only sees sequential reads and the result writes.
//
// Why it loses here: the probe side must be MOVED before it can be probed — key + row id written and read twice (two 8-bit
// passes for 2^13..2^16 partitions) is >= 64 bytes of HBM traffic per probe row, against 8 bytes per row for the one-pass probe
// of kernels_join.hip whose random accesses (one bit, rarely one rank word) mostly hit L2 / MALL.
#include <hip/hip_runtime.h>
#include "host/hash_kernels.h"
#include "launch_common.h"

namespace bhip {

namespace {

constexpr int RJ_SLOTS = 4096;                 // LDS table slots per partition (32 KiB)
constexpr int RJ_MAX_BUILD = 3072;             // rows of a build partition the table takes (75 % load)

__device__ inline uint32_t rj_pid(uint32_t key, int log2p) { return (uint32_t)(mix64((uint64_t)key) >> 40) & ((1u << log2p) - 1u); }

// sort key = (partition id << 32) | key, payload = the row's index (through `gather`: its position in the filtered probe side)
__global__ void __launch_bounds__(BLOCK)
radix_join_keys_kernel(const uint32_t* __restrict__ keys, uint32_t n, int log2p, uint64_t* __restrict__ out_keys, uint32_t* __restrict__ out_rows) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t k = keys[i];
        out_keys[i] = ((uint64_t)rj_pid(k, log2p) << 32) | k;
        out_rows[i] = i;
    }
}

// first[p] = first position of partition p in the sorted side, first[P] = n
__global__ void __launch_bounds__(BLOCK)
radix_join_bounds_kernel(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t n_parts, uint32_t* __restrict__ first) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t cur = i < n ? (uint32_t)(sorted[i] >> 32) : n_parts;
        const uint32_t prev = i > 0 ? (uint32_t)(sorted[i - 1] >> 32) + 1u : 0u;
        for (uint32_t p = prev; p <= cur && p <= n_parts; ++p) first[p] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(BLOCK)
radix_join_lds_kernel(const uint64_t* __restrict__ bkeys, const uint32_t* __restrict__ brows, const uint32_t* __restrict__ bfirst,
                      const uint64_t* __restrict__ pkeys, const uint32_t* __restrict__ prows, const uint32_t* __restrict__ pfirst, uint32_t n_parts,
                      uint32_t* __restrict__ partner, unsigned long long* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, uint32_t* flags) {
    __shared__ unsigned long long table[RJ_SLOTS];          // key | (build row + 1) << 32, 0 = empty
    for (uint32_t p = blockIdx.x; p < n_parts; p += gridDim.x) {
        const uint32_t b0 = bfirst[p], b1 = bfirst[p + 1], p0 = pfirst[p], p1 = pfirst[p + 1];
        if (p0 == p1 || b0 == b1) continue;                 // uniform per workgroup
        if (b1 - b0 > RJ_MAX_BUILD) { if (threadIdx.x == 0) atomicOr(&flags[0], 1u); continue; }
        for (int i = threadIdx.x; i < RJ_SLOTS; i += BLOCK) table[i] = 0ull;
        __syncthreads();
        for (uint32_t i = b0 + threadIdx.x; i < b1; i += BLOCK) {
            const uint32_t key = (uint32_t)bkeys[i];
            const unsigned long long mine = (unsigned long long)key | ((unsigned long long)(brows[i] + 1u) << 32);
            uint32_t slot = (uint32_t)mix64((uint64_t)key) & (RJ_SLOTS - 1);
            for (;;) {
                const unsigned long long v = atomicCAS(&table[slot], 0ull, mine);
                if (v == 0ull) break;
                if ((uint32_t)v == key) { atomicOr(&flags[1], 1u); break; }          // duplicate build key
                slot = (slot + 1) & (RJ_SLOTS - 1);
            }
        }
        __syncthreads();
        for (uint32_t i = p0 + threadIdx.x; i < p1; i += BLOCK) {
            const uint32_t key = (uint32_t)pkeys[i];
            uint32_t slot = (uint32_t)mix64((uint64_t)key) & (RJ_SLOTS - 1);
            for (;;) {
                const unsigned long long v = table[slot];
                if (v == 0ull) break;
                if ((uint32_t)v == key) {
                    const uint32_t row = prows[i];
                    partner[row] = (uint32_t)(v >> 32) - 1u;
                    atomicOr(&bitmap[row >> 6], 1ull << (row & 63));
                    atomicAdd(&tile_counts[row / SEL_TILE], 1u);
                    break;
                }
                slot = (slot + 1) & (RJ_SLOTS - 1);
            }
        }
        __syncthreads();
    }
}

int rows_grid(const LaunchCfg& cfg, size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

hipError_t launch_radix_join_keys(const LaunchCfg& cfg, const uint32_t* keys, uint32_t n, int log2p, uint64_t* out_keys, uint32_t* out_rows) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(radix_join_keys_kernel, dim3(rows_grid(cfg, n)), dim3(BLOCK), 0, cfg.stream, keys, n, log2p, out_keys, out_rows);
    return hipGetLastError();
}
hipError_t launch_radix_join_bounds(const LaunchCfg& cfg, const uint64_t* sorted, uint32_t n, uint32_t n_parts, uint32_t* first) {
    hipLaunchKernelGGL(radix_join_bounds_kernel, dim3(rows_grid(cfg, (size_t)n + 1)), dim3(BLOCK), 0, cfg.stream, sorted, n, n_parts, first);
    return hipGetLastError();
}
hipError_t launch_radix_join_lds(const LaunchCfg& cfg, const uint64_t* bkeys, const uint32_t* brows, const uint32_t* bfirst, const uint64_t* pkeys,
                                 const uint32_t* prows, const uint32_t* pfirst, uint32_t n_parts, uint32_t* partner, uint64_t* bitmap,
                                 uint32_t* tile_counts, uint32_t* flags) {
    if (n_parts == 0) return hipSuccess;
    uint32_t grid = (uint32_t)cfg.device_cus * 4;            // 32 KiB of LDS each: four workgroups per CU
    if (grid > n_parts) grid = n_parts;
    hipLaunchKernelGGL(radix_join_lds_kernel, dim3(grid), dim3(BLOCK), 0, cfg.stream, bkeys, brows, bfirst, pkeys, prows, pfirst, n_parts, partner,
                       reinterpret_cast<unsigned long long*>(bitmap), tile_counts, flags);
    return hipGetLastError();
}

}  // namespace bhip
