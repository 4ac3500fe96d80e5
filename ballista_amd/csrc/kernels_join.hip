// kernels_join.hip — HashJoinExec (rust/core/src/serde/physical_plan/from_proto.rs:253-276) for ONE integer key and a unique
// build side — every join of TPC-H Q3 / Q5 but Q5's two-column supplier join.
//
// BUILD.  `join_key_stats` reads the build keys once: min, max, "strictly increasing" (= sorted and unique).
//   * Keys inside a window of <= 2^36 values (dense surrogate keys: o_orderkey, c_custkey, ...) get a RANK MAP instead of a hash
//     table: bits[] = the exact key set, one bit per value of the window; prefix[w] = set bits before 64-bit word w.  The rank of
//     a present key (prefix + popcount of the lower bits of its word) IS its build row when the build side arrives sorted by key
//     (a table stored in key order stays sorted under FilterExec); otherwise perm[rank] = build row.  No CAS, no collisions, no
//     load factor: a sorted build side is written with plain stores (a wave combines the bits of a word; only words that straddle
//     two waves take an atomicOr), an unsorted one with one atomicOr per key (which also finds duplicate keys).  A probe that
//     cannot match ends at the bit — the same single random read the round-1 key-set bitmap cost — and a probe that matches reads
//     one more 4-byte word (and perm[] for unsorted builds) instead of walking a 2x-oversized table of 8-byte slots.
//   * Sparse keys keep the CAS table of kernels_hash.hip (join_build_narrow_kernel).
//
// PROBE.  `join_filter_probe_kernel` walks the UNFILTERED probe batch once: AND of integer ranges (the FilterExec under the join:
// l_shipdate > d, o_orderdate < d, ...) -> key-set bit -> rank / table.  One wave owns a 1024-row tile: the per-tile counts are
// plain stores, the build rows of the emitted rows are staged compacted per tile, results are deterministic.  Inside a 256-row
// pass the rows that survive the bit test (Q3: ~5 % of lineitem) are packed into consecutive lanes through LDS before the
// dependent reads, so a pass pays ONE round trip for all of them instead of one per row slot.
// Algorithmic bytes per probe row: predicate columns + key column once, + 1 bit; per emitted row 4 B partner twice.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "host/hash_kernels.h"
#include "launch_common.h"

namespace bhip {

namespace {

__device__ inline bool jbit_at(const uint64_t* bm, uint64_t i) { return bm == nullptr || ((bm[i >> 6] >> (i & 63)) & 1ull); }

#ifndef BHIP_PROBE_WAVES
#define BHIP_PROBE_WAVES
#endif
template <int KW> struct KeyT;
template <> struct KeyT<4> { using type = uint32_t; };
template <> struct KeyT<8> { using type = uint64_t; };

// key -> offset in the window [kmin, kmin + range]; keys are compared in SIGNED order (Int32 / Int64 / Date32; UInt64 keys above
// 2^63 wrap, which only makes the window test fail safely into "absent" for keys the build side cannot hold either)
template <int KW>
__device__ inline uint64_t key_offset(typename KeyT<KW>::type key, uint64_t kmin) {
    if constexpr (KW == 4) return (uint64_t)(uint32_t)(key - (uint32_t)kmin);
    else return key - kmin;
}

// offset inside the window?  (4-byte keys: offsets are < 2^32 and so is the window's last offset — a 32-bit compare)
template <int KW>
__device__ inline bool in_window(uint64_t off, uint64_t krange) {
    if constexpr (KW == 4) return (uint32_t)off <= (uint32_t)krange;
    else return off <= krange;
}

// ---- build ---------------------------------------------------------------------------------------------------------------
// stats[0] = min, stats[1] = max of (key ^ sign bit) as unsigned 64-bit (seeded ~0 / 0); stats[2] |= 1 when some key is not
// greater than its predecessor; rows with a NULL key (sel) are skipped and make the side "unsorted"
template <int KW>
__global__ void __launch_bounds__(BLOCK)
join_key_stats_kernel(const void* __restrict__ keys_v, const uint64_t* __restrict__ sel, uint32_t n, unsigned long long* stats) {
    using K = typename KeyT<KW>::type;
    const K* __restrict__ keys = static_cast<const K*>(keys_v);
    const uint64_t bias = KW == 4 ? 0x80000000ull : (1ull << 63);
    uint64_t lo = ~0ull, hi = 0;
    uint32_t unsorted = sel != nullptr ? 1u : 0u;
    const uint64_t kmask = KW == 4 ? 0xFFFFFFFFull : ~0ull;
    if (sel == nullptr) {
        // eight rows of a thread in flight (the grid is small — one set of atomics per workgroup — so a load per iteration was a
        // chain of ~110 exposed round trips: 0.09 ms for Q3's 14.6 M build keys); indices clamped, no branch around the loads
        constexpr int U = 8;
        const uint32_t stride = gridDim.x * BLOCK, last = n - 1;
        for (uint64_t row0 = blockIdx.x * BLOCK + threadIdx.x; row0 < n; row0 += (uint64_t)stride * U) {
            K k[U], kn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t r = row0 + (uint64_t)u * stride;
                const uint32_t rc = r < last ? (uint32_t)r : last, rn = r + 1 < last ? (uint32_t)(r + 1) : last;
                k[u] = keys[rc];
                kn[u] = keys[rn];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t r = row0 + (uint64_t)u * stride;
                const uint64_t b = ((uint64_t)k[u] ^ bias) & kmask, nb = ((uint64_t)kn[u] ^ bias) & kmask;
                if (r < n) {
                    lo = b < lo ? b : lo;
                    hi = b > hi ? b : hi;
                    if (r + 1 < n && nb <= b) unsorted = 1u;
                }
            }
        }
    } else {
        for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
            if (!jbit_at(sel, row)) continue;
            const K k = keys[row];
            const uint64_t b = ((uint64_t)k ^ bias) & kmask;
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
            if (row + 1 < n) {
                const uint64_t nb = ((uint64_t)keys[row + 1] ^ bias) & kmask;
                if (nb <= b) unsorted = 1u;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t l2 = __shfl_down((unsigned long long)lo, d, 64), h2 = __shfl_down((unsigned long long)hi, d, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    const bool any_unsorted = __ballot(unsorted != 0) != 0ull;
    __shared__ uint64_t s_lo[BLOCK / 64], s_hi[BLOCK / 64];
    __shared__ uint32_t s_un[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; s_un[threadIdx.x >> 6] = any_unsorted; }
    __syncthreads();
    if (threadIdx.x == 0) {      // one set of atomics per workgroup
        uint32_t un = s_un[0];
        for (int w = 1; w < BLOCK / 64; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; un |= s_un[w]; }
        atomicMin(&stats[0], (unsigned long long)lo);
        atomicMax(&stats[1], (unsigned long long)hi);
        if (un) atomicOr(&stats[2], 1ull);
    }
}

// ---- a whole rank-map build of at most TINY_BUILD_ROWS keys in ONE launch of one workgroup ---------------------------------------------
// (a dimension table's keys — Q5's nation: 25, region: 5: key statistics -> host decides -> fill -> key set -> packed map [-> rank ->
// row permutation] were four launches and two or three host waits, ~80 us, for a few bytes of map.)  Statistics as join_key_stats; when
// the keys span at most TINY_BUILD_WINDOW values the key set is built in LDS (an atomicOr per key finds duplicates), its prefix
// popcounts make the packed words, an unsorted side gets its rank -> row permutation; `out` = {min, max (biased), flags}: the host
// reads it ONCE and either takes the map or carries on with the statistics on the general path.
constexpr int TINY_BUILD_ROWS = 1024;
constexpr uint32_t TINY_BUILD_WINDOW = 1u << 16;                 // values: 2048 granules, 16 KiB of packed map
template <int KW>
__global__ void __launch_bounds__(TINY_BUILD_ROWS)
tiny_rank_build_kernel(const void* __restrict__ keys_v, const uint64_t* __restrict__ sel, uint32_t n, uint64_t* __restrict__ rpack,
                       uint32_t* __restrict__ rperm, unsigned long long* __restrict__ out) {
    using K = typename KeyT<KW>::type;
    const K* __restrict__ keys = static_cast<const K*>(keys_v);
    constexpr int N_GRAN_MAX = (int)(TINY_BUILD_WINDOW / 32);
    __shared__ uint32_t s_bits[N_GRAN_MAX + 2];
    __shared__ uint32_t s_before[N_GRAN_MAX + 2];
    __shared__ uint64_t s_lo[TINY_BUILD_ROWS / 64], s_hi[TINY_BUILD_ROWS / 64];
    __shared__ uint32_t s_flag[TINY_BUILD_ROWS / 64], s_wave[TINY_BUILD_ROWS / 64];
    __shared__ uint32_t s_dup;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t bias = KW == 4 ? 0x80000000ull : (1ull << 63), kmask = KW == 4 ? 0xFFFFFFFFull : ~0ull;
    const bool live = tid < n && (sel == nullptr || jbit_at(sel, tid));
    const uint64_t b = live ? (((uint64_t)keys[tid] ^ bias) & kmask) : 0;
    uint64_t lo = live ? b : ~0ull, hi = live ? b : 0;
    // "strictly increasing, no NULL" as join_key_stats has it: a NULL-able column counts as unsorted
    uint32_t unsorted = sel != nullptr ? 1u : 0u;
    if (live && tid + 1 < n && ((((uint64_t)keys[tid + 1] ^ bias) & kmask) <= b)) unsorted = 1u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t l2 = __shfl_down((unsigned long long)lo, d, 64), h2 = __shfl_down((unsigned long long)hi, d, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    const bool any_unsorted = __ballot(unsorted != 0) != 0ull;
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_flag[wave] = any_unsorted ? 1u : 0u; }
    if (tid == 0) s_dup = 0;
    for (uint32_t g = tid; g < (uint32_t)N_GRAN_MAX + 2; g += TINY_BUILD_ROWS) s_bits[g] = 0;
    __syncthreads();
    uint32_t un = 0;
    lo = ~0ull; hi = 0;
    for (int w = 0; w < TINY_BUILD_ROWS / 64; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; un |= s_flag[w]; }
    const bool any_key = lo <= hi;
    const uint64_t range = any_key ? hi - lo : 0;
    const bool build = any_key && range < TINY_BUILD_WINDOW;
    const uint32_t n_gran = 2u * ((uint32_t)(range >> 6) + 1u);                      // as the host sizes the map: whole 64-bit words
    const uint32_t off = (uint32_t)(b - lo);
    if (build && live) {
        const uint32_t bit = 1u << (off & 31u);
        if (atomicOr(&s_bits[off >> 5], bit) & bit) s_dup = 1u;
    }
    __syncthreads();
    if (build) {
        // set bits before each granule: two granules per thread, a workgroup scan of the pair sums
        const uint32_t g0 = 2 * tid, c0 = g0 < n_gran ? (uint32_t)__popc(s_bits[g0]) : 0u, c1 = g0 + 1 < n_gran ? (uint32_t)__popc(s_bits[g0 + 1]) : 0u;
        uint32_t x = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= (uint32_t)d) x += y; }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        uint32_t before = x - (c0 + c1);
        for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
        if (g0 < n_gran) { s_before[g0] = before; rpack[g0] = (uint64_t)s_bits[g0] | ((uint64_t)before << 32); }
        if (g0 + 1 < n_gran) { s_before[g0 + 1] = before + c0; rpack[g0 + 1] = (uint64_t)s_bits[g0 + 1] | ((uint64_t)(before + c0) << 32); }
        if (tid < 2) rpack[n_gran + tid] = 0;                                       // NarrowJoinTable::rzero and the word behind it
        __syncthreads();
        if (live && rperm != nullptr && s_dup == 0u)
            rperm[s_before[off >> 5] + (uint32_t)__popc(s_bits[off >> 5] & ((1u << (off & 31u)) - 1u))] = tid;
    }
    if (tid == 0) {
        out[0] = lo;
        out[1] = hi;
        out[2] = (un ? 1ull : 0ull) | (s_dup ? 2ull : 0ull) | (build ? 4ull : 0ull);
    }
}

// sorted, unique build keys: lanes whose keys fall into the same 32-bit piece of the bitmap are neighbours; the last lane of each
// run writes the combined bits — with a plain store when the run lies strictly inside the wave (then no other wave holds a key
// of that piece), with an atomicOr when it touches the wave's first or last lane.  (32-bit pieces: the segmented scan moves two
// 32-bit values per step instead of two 64-bit ones, and the kernel is bound by those instructions.)
template <int KW>
__global__ void __launch_bounds__(BLOCK)
rank_bits_sorted_kernel(const void* __restrict__ keys_v, uint32_t n, uint64_t kmin, unsigned long long* __restrict__ bits64) {
    using K = typename KeyT<KW>::type;
    const K* __restrict__ keys = static_cast<const K*>(keys_v);
    uint32_t* __restrict__ bits = reinterpret_cast<uint32_t*>(bits64);        // little-endian: piece 2w / 2w + 1 = low / high half of word w
    const int lane = threadIdx.x & 63;
    const uint32_t n_round = (n + 63u) & ~63u;
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_round; row += gridDim.x * BLOCK) {
        const bool in = row < n;
        const uint64_t d = in ? key_offset<KW>(keys[row], kmin) : 0;
        const uint32_t wi = in ? (uint32_t)(d >> 5) : 0xFFFFFFFFu;              // (the window holds <= 2^36 values)
        uint32_t m = in ? (1u << (d & 31)) : 0u;
        // inclusive segmented OR over runs of equal wi (runs are contiguous: the keys increase)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t pm = __shfl_up(m, off, 64);
            const uint32_t pw = __shfl_up(wi, off, 64);
            if (lane >= off && pw == wi) m |= pm;
        }
        const uint32_t next_wi = __shfl_down(wi, 1, 64);
        const uint32_t first_wi = __shfl(wi, 0, 64);     // outside the branch: every lane takes part in a shuffle
        const bool last_of_run = in && (lane == 63 || next_wi != wi);
        if (last_of_run) {
            const bool touches_edge = lane == 63 || first_wi == wi || row + 1 >= n;
            if (touches_edge) atomicOr(&bits[wi], m);
            else bits[wi] = m;
        }
    }
}

// (r03 tried "a key whose two neighbours fall into other pieces stores its bit plainly, the others atomicOr" — two shuffles instead of
// twelve: SLOWER, 0.154 -> 0.207 ms for Q3's 14.6 M keys, 0.085 -> 0.139 ms for Q5's builds: at ~0.8 keys per piece half the keys
// share a piece and their atomics cost more than the scan saves.  Reverted.)
// any order: one atomicOr per key; a bit that was already set is a duplicate key
template <int KW>
__global__ void __launch_bounds__(BLOCK)
rank_bits_any_kernel(const void* __restrict__ keys_v, const uint64_t* __restrict__ sel, uint32_t n, uint64_t kmin,
                     unsigned long long* __restrict__ bits, uint32_t* dup_flag) {
    using K = typename KeyT<KW>::type;
    const K* __restrict__ keys = static_cast<const K*>(keys_v);
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        if (!jbit_at(sel, row)) continue;
        const uint64_t d = key_offset<KW>(keys[row], kmin);
        const unsigned long long bit = 1ull << (d & 63);
        if (atomicOr(&bits[d >> 6], bit) & bit) *dup_flag = 1u;
    }
}

// unsorted build side: perm[rank of key] = build row
template <int KW>
__global__ void __launch_bounds__(BLOCK)
rank_perm_kernel(const void* __restrict__ keys_v, const uint64_t* __restrict__ sel, uint32_t n, uint64_t kmin,
                 const uint64_t* __restrict__ rpack, uint32_t* __restrict__ perm) {
    using K = typename KeyT<KW>::type;
    const K* __restrict__ keys = static_cast<const K*>(keys_v);
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        if (!jbit_at(sel, row)) continue;
        const uint64_t d = key_offset<KW>(keys[row], kmin);
        const uint64_t w = rpack[d >> 5];
        perm[(uint32_t)(w >> 32) + (uint32_t)__popc((uint32_t)w & ((1u << (d & 31)) - 1u))] = row;
    }
}

// ---- probe ---------------------------------------------------------------------------------------------------------------
// FP_ROWS rows per lane and pass: their loads are in flight together (8 rows and a three-deep pipeline were measured and dropped:
// profiles/r02_probe_variants_q3_sf100.txt)

template <int KW>
__device__ inline uint32_t table_lookup(const NarrowJoinTable& T, typename KeyT<KW>::type key) {
    if constexpr (KW == 4) {
        uint64_t slot = mix64((uint64_t)key) & T.mask;
        for (;;) {
            const uint64_t v = T.slots[slot];
            if (v == 0) return 0xFFFFFFFFu;
            if ((uint32_t)v == key) return (uint32_t)(v >> 32) - 1u;
            slot = (slot + 1) & T.mask;
        }
    } else {
        const ulonglong2* slots = reinterpret_cast<const ulonglong2*>(T.slots);
        uint64_t slot = mix64(key) & T.mask;
        for (;;) {
            const ulonglong2 v = slots[slot];
            const uint32_t r = (uint32_t)v.y;
            if (r == 0) return 0xFFFFFFFFu;
            if (v.x == key) return r - 1u;
            slot = (slot + 1) & T.mask;
        }
    }
}

// per-wave LDS scratch of the lookup compaction: up to FP_CHUNK packed items
template <int FP_CHUNK>
struct WaveScratchT {
    uint64_t a[FP_CHUNK];        // rank map: the rank; table: the key
    uint32_t b[FP_CHUNK];        // the result (build row)
    uint32_t c[FP_CHUNK];        // residual key (second key column of a two-column join) of the probe row
};

// NF: filter columns compiled in (0, 1 or JOIN_FILTER_MAX): registers for the ones a plan does not have would only cost occupancy.
// Every streamed load and the key-set word load are UNCONDITIONAL (row index clamped to the last row, word index 0 for rows that
// do not need it): a predicated load costs a saved-exec branch of its own — the first version of this kernel spent as many scalar
// as vector instructions on them (rocprofv3 SQ_INSTS_SALU 293 M vs SQ_INSTS_VALU 275 M per launch, profiles/r02_probe_variants_q3_sf100.txt).
// RESID: a second 4-byte key column on both sides (ON a = c AND b = d with the build side unique on `a` alone): the lookup goes by
// the first key, a match stands only if the second keys are equal too (T.resid_build[build row] vs the probe row's value).
// (The common configuration — rank map, no NULL keys on the probe side, not a left join — has a kernel of its own below.)
template <int KW, int NF, int FP_ROWS, bool RESID>
__global__ void __launch_bounds__(BLOCK)
join_filter_probe_kernel(const NarrowJoinTable T, const ProbeFilter F, const void* __restrict__ rkeys_v, const uint64_t* __restrict__ rsel,
                         uint32_t n_right, int right_outer, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts,
                         uint32_t* __restrict__ staging, uint32_t* matched, const uint32_t* __restrict__ resid_probe,
                         uint32_t* __restrict__ staging_rows) {
    using K = typename KeyT<KW>::type;
    constexpr int NFR = NF > 0 ? NF : 1;
    constexpr int FP_CHUNK = 64 * FP_ROWS;           // rows of one pass of a wave
    static_assert(SEL_TILE % FP_CHUNK == 0, "a tile is a whole number of passes");
    using WaveScratch = WaveScratchT<FP_CHUNK>;
    __shared__ WaveScratch scratch[BLOCK / 64];
    WaveScratch& S = scratch[threadIdx.x >> 6];
    const K* __restrict__ rkeys = static_cast<const K*>(rkeys_v);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)n_right + SEL_TILE - 1) / SEL_TILE);
    const uint32_t last_row = n_right - 1;
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    const bool ranked = T.rpack != nullptr;
    const bool key_set = ranked || T.present != nullptr;

    // the streamed inputs of one pass; the NEXT pass's are loaded before this pass walks its dependent reads.
    // Rows are 32-bit (a batch holds < 2^32 - 16 rows and a pass starts at a multiple of 256: base + 255 does not wrap).
    struct Regs { K key[FP_ROWS]; int32_t f[NFR][FP_ROWS]; uint32_t g[RESID ? FP_ROWS : 1]; };
    auto load = [&](uint32_t base, Regs& r) {
#pragma unroll
        for (int k = 0; k < FP_ROWS; ++k) {
            const uint32_t row = base + 64u * k + lane;
            const uint32_t rc = row < last_row ? row : last_row;
            r.key[k] = rkeys[rc];
            if (RESID) r.g[RESID ? k : 0] = resid_probe[rc];
#pragma unroll
            for (int j = 0; j < NF; ++j) r.f[j][k] = F.col[j < F.n ? j : 0][rc];
        }
    };
    uint32_t flo[NFR], fspan[NFR];
    uint32_t alive = 1u;
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
        const bool have = j < NF && j < F.n;
        flo[j] = have ? (uint32_t)F.lo[j] : 0u;
        fspan[j] = have ? (uint32_t)F.hi[j] - (uint32_t)F.lo[j] : 0xFFFFFFFFu;
        if (have && F.hi[j] < F.lo[j]) alive = 0u;
    }
    Regs cur, nxt;
    if (wave_id < n_tiles) load(wave_id * SEL_TILE, cur);
    for (uint32_t t = wave_id; t < n_tiles; t += n_waves) {
        const uint32_t tile_base = t * SEL_TILE;
        uint32_t tile_cnt = 0;
#pragma unroll 1
        for (int c = 0; c < SEL_TILE / FP_CHUNK; ++c) {
            const uint32_t base = tile_base + (uint32_t)c * FP_CHUNK;
            // The next pass's streamed loads are issued AFTER this pass's key-set word loads: loads return in issue order, so a wait
            // for the words (an L2 round trip) would otherwise also wait for the streamed loads issued before them (an HBM round trip)
            auto prefetch = [&]() {
                const bool last = c == SEL_TILE / FP_CHUNK - 1;
                // (after the wave's last tile the prefetch re-reads the final rows: harmless, and no branch around the loads)
                const uint32_t nt = t + n_waves < n_tiles ? t + n_waves : n_tiles - 1;
                load(last ? nt * SEL_TILE : base + FP_CHUNK, nxt);
            };
            if (!key_set) prefetch();
            bool pass[FP_ROWS], live[FP_ROWS];
            uint32_t m[FP_ROWS], d[FP_ROWS];
            uint64_t word[FP_ROWS];
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) {
                const uint32_t row = base + 64u * k + lane;
                uint32_t ok = (uint32_t)(row < n_right) & alive;                        // branch-free: see join_rank_probe_kernel
#pragma unroll
                for (int j = 0; j < NF; ++j) ok &= (uint32_t)(((uint32_t)cur.f[j][k] - flo[j]) <= fspan[j]);
                const bool p = ok != 0u;
                pass[k] = p;
                m[k] = 0xFFFFFFFFu;
                live[k] = p;
                if (rsel != nullptr) {                            // NULL keys never match (wave-uniform test of the pointer)
                    const uint32_t rc = row < last_row ? row : last_row;
                    live[k] = live[k] && ((rsel[rc >> 6] >> (rc & 63)) & 1ull);
                }
            }
            // ---- the exact key set: one bit per value of the window (rank map: rbits, CAS table: present) ----------------
            if (key_set) {
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) {
                    const uint64_t off = key_offset<KW>(cur.key[k], T.kmin64);
                    live[k] = live[k] && in_window<KW>(off, T.krange64);
                    const uint32_t g = live[k] ? (uint32_t)(off >> 5) : 0u;               // the window holds <= 2^36 values (key-set bitmap: 2^30)
                    d[k] = (uint32_t)off & 31u;
                    if (ranked) word[k] = T.rpack[g];                                     // key set of the granule | keys before it << 32
                    else word[k] = (uint64_t)T.present[g];
                }
                __builtin_amdgcn_sched_barrier(0);
                prefetch();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) live[k] = live[k] && (((uint32_t)word[k] >> (d[k] & 31)) & 1u);
            }
            // ---- pack the surviving row slots into consecutive lanes, one round of dependent reads for all of them ---------
            uint64_t lw[FP_ROWS];
            uint32_t before[FP_ROWS], total = 0;
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) { lw[k] = __ballot(live[k]); before[k] = total; total += (uint32_t)__popcll(lw[k]); }
            if (total) {                                                // wave-uniform
                uint32_t idx[FP_ROWS];
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) {
                    idx[k] = before[k] + (uint32_t)__popcll(lw[k] & lane_lt);
                    if (live[k]) {
                        if (ranked) S.a[idx[k]] = (word[k] >> 32) + (uint32_t)__popc((uint32_t)word[k] & ((1u << (d[k] & 31)) - 1u));   // the rank
                        else S.a[idx[k]] = (uint64_t)cur.key[k];
                        if (RESID) S.c[idx[k]] = cur.g[RESID ? k : 0];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t j = lane; j < total; j += 64) {
                    uint32_t r;
                    if (ranked) {
                        r = (uint32_t)S.a[j];
                        if (T.rperm) r = T.rperm[r];                      // (unsorted build side; a uniform test once per lookup round)
                    } else {
                        r = table_lookup<KW>(T, (K)S.a[j]);
                    }
                    if (RESID && r != 0xFFFFFFFFu && T.resid_build[r] != S.c[j]) r = 0xFFFFFFFFu;
                    S.b[j] = r;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) {
                    if (live[k]) {
                        m[k] = S.b[idx[k]];
                        if (matched && m[k] != 0xFFFFFFFFu) atomicOr(&matched[m[k] >> 5], 1u << (m[k] & 31));
                    }
                }
                __builtin_amdgcn_wave_barrier();                        // the next pass overwrites the scratch
            }
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) {
                const bool emit = pass[k] && (right_outer || m[k] != 0xFFFFFFFFu);
                const uint64_t wd = __ballot(emit);
                // (bitmap words past the last row's word exist: the bitmap is allocated for whole tiles)
                if (lane == 0) bitmap[(base >> 6) + k] = wd;
                if (emit && (staging || staging_rows)) {
                    const uint32_t at = tile_base + tile_cnt + (uint32_t)__popcll(wd & lane_lt);
                    if (staging) staging[at] = m[k];
                    if (staging_rows) staging_rows[at] = base + 64u * k + lane;            // the emitted row itself: no bitmap -> index pass later
                }
                tile_cnt += (uint32_t)__popcll(wd);
            }
            cur = nxt;
        }
        if (lane == 0) tile_counts[t] = tile_cnt;
    }
}

// The common configuration — rank map, no NULL keys on the probe side, not a left join — without the lookup round: the packed
// word of the rank map (NarrowJoinTable::rpack) holds the membership bit AND the rank, so a row is decided by one dependent
// 8-byte read and 32-bit arithmetic; no ballot / LDS compaction of the surviving rows, no second read.  The general kernel above
// issues ~235 vector and ~236 scalar instructions per 256-row pass of a wave, which is what bounds it: at four cycles per wave64
// vector instruction that is one row per cycle and CU — the 570 rows/ns it measures on 600 M rows.
// RESID: the second key column of every match is compared in place (T.resid_build[build row]: a second dependent read for the
// rows that passed the first).  An unsorted build side adds the rank -> row read (T.rperm).
// Instruction issue is what bounds this kernel (rocprofv3: SQ_ACTIVE_INST_ANY x resident waves = 90 % of the SIMD cycles, 10 % of a
// wave's cycles waiting on memory), so everything wave-uniform is kept scalar: the wave's tile index goes through readfirstlane,
// interior tiles (all but a batch's last) run a variant without row-bound tests and address clamps, the streamed loads and the
// staging stores address `scalar base + lane offset`.
// MAPBUF: the packed map is shorter than 2 GiB (windows up to 2^33 key values: every join of TPC-H up to SF1000 with dense keys,
// SF300 with dbgen's sparse ones): it is read through a buffer descriptor too, and the descriptor's range check IS the window
// test and the "dropped rows read zeros" rule — an offset past the map's last granule returns 0 without touching memory.
// BITS: a semi-join (no build column is read: nothing is staged, no rank is ever computed) against a sorted one-column build side reads
// the key-set words themselves (NarrowJoinTable::rbits, 4 bytes per 32 key values) — half the packed map: Q3's customers, 15 M key
// values, are 1.9 MB instead of 3.75 MB against the 4 MB of L2 an XCD has, under 150 M random lookups.
template <int KW, int NF, int FP_ROWS, bool RESID, bool PERM, bool MAPBUF, bool BITS = false>
__global__ void __launch_bounds__(BLOCK) BHIP_PROBE_WAVES
join_rank_probe_kernel(const NarrowJoinTable T, const ProbeFilter F, const void* __restrict__ rkeys_v, uint32_t n_right,
                       uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, uint32_t* __restrict__ staging,
                       const uint32_t* __restrict__ resid_probe, uint32_t* __restrict__ staging_rows) {
    using K = typename KeyT<KW>::type;
    constexpr int NFR = NF > 0 ? NF : 1;
    constexpr int FP_CHUNK = 64 * FP_ROWS;
    constexpr int PASSES = SEL_TILE / FP_CHUNK;
    static_assert(SEL_TILE % FP_CHUNK == 0, "a tile is a whole number of passes");
    static_assert(FP_ROWS <= 64, "one bitmap word per row slot, written by the first FP_ROWS lanes");
    const K* __restrict__ rkeys = static_cast<const K*>(rkeys_v);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)n_right + SEL_TILE - 1) / SEL_TILE);
    const uint32_t full_tiles = n_right / SEL_TILE;                  // tiles before this one hold SEL_TILE rows each
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)));
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    const uint64_t* __restrict__ rpack = T.rpack;
    const bool staged = staging != nullptr;            // (staging and staging_rows come together: host/ops_join.cpp process_fused)
    const bool scalar_map = T.scalar_map != 0;         // (BHIP_PROBE_NO_SCALAR_MAP=1 clears it: the A/B partner)

    struct Regs { K key[FP_ROWS]; int32_t f[NFR][FP_ROWS]; uint32_t g[RESID ? FP_ROWS : 1]; };
    // rows [base, base + FP_CHUNK) of the streamed columns, as BUFFER loads: a 128-bit descriptor over [column + base, end of the
    // column) built from wave-uniform values (scalar instructions), the lane's byte offset in one VGPR shared by every load, the
    // row slot's 256 / 512-byte step in the instruction's immediate offset — no vector instruction computes an address (the flat
    // loads here cost one v_lshl_add_u64 each: 16-24 per pass of a kernel bound by vector issue), and the hardware's range check
    // returns 0 for rows past the end of the column, so the last tile needs no index clamping (the row < n test stays a predicate).
    // A descriptor holds a 32-bit byte count: it is rebuilt per pass from the pass's first row, so columns beyond 4 GiB
    // (Int64 keys of an SF100+ lineitem) need nothing special.
    const uint32_t lane_b4 = lane * 4u, lane_bk = lane * (uint32_t)KW;
    auto rsrc_of = [&](const void* col, uint32_t base, uint32_t width) {
        const uint64_t remain = base < n_right ? (uint64_t)(n_right - base) * width : 0;   // (a prefetch may start past the last row: zero records, every lane reads 0)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(static_cast<const uint8_t*>(col)) + (uint64_t)base * width, 0,
                                                 (int)(remain > 0x7FFFFFFFull ? 0x7FFFFFFFull : remain), 0x00020000);
    };
    auto load_any = [&](uint32_t base, Regs& r) {
        const auto rk = rsrc_of(rkeys_v, base, KW);
#pragma unroll
        for (int k = 0; k < FP_ROWS; ++k) {
            // (the row slot's step rides in the VECTOR offset: a scalar offset is left out of the descriptor's range check, and the last
            // tile relies on that check to read zeros, not memory, past the end of the column)
            if constexpr (KW == 4) r.key[k] = (K)__builtin_amdgcn_raw_buffer_load_b32(rk, (int)(lane_bk + 256u * k), 0, 0);
            else {
                typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
                const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rk, (int)(lane_bk + 512u * k), 0, 0);
                r.key[k] = (K)(((uint64_t)v.y << 32) | v.x);
            }
        }
        if (RESID) {
            const auto rg = rsrc_of(resid_probe, base, 4);
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) r.g[RESID ? k : 0] = __builtin_amdgcn_raw_buffer_load_b32(rg, (int)(lane_b4 + 256u * k), 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const auto rf = rsrc_of(F.col[j < F.n ? j : 0], base, 4);
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) r.f[j][k] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rf, (int)(lane_b4 + 256u * k), 0, 0);
        }
    };
    // The range predicates, branch-free: lo <= f <= hi  <=>  (uint32)(f - lo) <= (uint32)(hi - lo) — one subtraction and one compare
    // per column, and the conditions of a row are combined with `&` on their lane masks.  (Written as `p = p && f >= lo && f <= hi`
    // hipcc lowered every `&&` to an exec-mask branch — s_and_saveexec / s_xor / s_or around each compare: ~20 scalar and vector
    // instructions per row slot and column, half of this kernel's issue slots, in a kernel that is bound by instruction issue.)
    // A column the plan does not have gets the full span: every value passes.  An empty range (lo > hi) kills every row.
    uint32_t flo[NFR], fspan[NFR];
    uint32_t alive = 1u;
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
        const bool have = j < NF && j < F.n;
        flo[j] = have ? (uint32_t)F.lo[j] : 0u;
        fspan[j] = have ? (uint32_t)F.hi[j] - (uint32_t)F.lo[j] : 0xFFFFFFFFu;
        if (have && F.hi[j] < F.lo[j]) alive = 0u;
    }
    // granules [0, rzero) of the map (the host only picks MAPBUF when rzero * 8 < 2^31)
    const auto rmap = BITS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(T.rbits), 0, (int)(T.rzero * 4u), 0x00020000)
                           : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(rpack), 0, MAPBUF ? (int)(T.rzero * 8u) : 0, 0x00020000);
    // Emitted (build row, probe row) pairs collect in a wave-private LDS strip and go out in whole-wave stores when the strip fills up
    // and at the end of the tile: two scattered global stores per row SLOT with a match (a handful of active lanes each, 16 store
    // instructions per pass when most slots hold one — Q5's lineitem probe: 3 % of the rows match, 86 % of the slots) become one LDS
    // write per slot and two coalesced stores per tile.
    constexpr uint32_t STRIP = 192;                                   // >= 64 more than the flush threshold
    __shared__ uint2 s_strip[BLOCK / 64][STRIP];
    uint2* __restrict__ strip = s_strip[threadIdx.x >> 6];
    // The streamed rows are loaded TWO passes ahead: the passes of a tile cycle through SETS register sets, and a pass refills its own
    // set — for the pass SETS later — as soon as its keys have become map offsets.  (One pass ahead, a wave had its next rows in flight
    // only while it waited for its dependent map reads: a pass took one full memory round trip with the SIMD a quarter busy — rocprofv3
    // on Q5's lineitem launch: 4 waves per SIMD, 188 SIMD cycles per 64-row slot for ~33 instructions.)  -DBHIP_PROBE_SETS=4: a whole
    // tile ahead with 4 rows per lane — 84-116 registers instead of 62-78, and 2-4 % slower (profiles/r03_probe_kernel_pmc_and_variants.txt).
#ifndef BHIP_PROBE_SETS
#define BHIP_PROBE_SETS 2
#endif
    constexpr int SETS = BHIP_PROBE_SETS < PASSES ? BHIP_PROBE_SETS : PASSES;
    static_assert(PASSES % SETS == 0, "the passes of a tile cycle through the register sets");
    Regs sets[SETS];
    if (wave_id < n_tiles) {
#pragma unroll
        for (int q = 0; q < SETS; ++q) load_any(wave_id * SEL_TILE + (uint32_t)q * FP_CHUNK, sets[q]);
    }
    for (uint32_t t = wave_id; t < n_tiles; t += n_waves) {
        const uint32_t tile_base = t * SEL_TILE;
        uint32_t* __restrict__ st_m = staging + tile_base;            // (null + offset when not staged: never dereferenced)
        uint32_t* __restrict__ st_r = staging_rows + tile_base;
        uint32_t tile_cnt = 0;
        uint32_t strip_cnt = 0, flushed = 0;                          // wave-uniform: entries in the strip, entries of this tile already stored
        auto flush = [&]() {
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (uint32_t i = lane; i < strip_cnt; i += 64u) {                            // (inlined at every row slot: keep it small)
                const uint2 e = strip[i];
                st_m[flushed + i] = e.x;
                st_r[flushed + i] = e.y;
            }
            __builtin_amdgcn_wave_barrier();
            flushed += strip_cnt;
            strip_cnt = 0;
        };
        auto pass_body = [&](int c, Regs& cur, auto edge) {
            constexpr bool EDGE = decltype(edge)::value;
            const uint32_t base = tile_base + (uint32_t)c * FP_CHUNK;
            // The refill goes out BEHIND this pass's last dependent load and nothing of this pass waits on a load issued after it:
            // loads return in issue order, so the streamed rows stay in flight until the pass after next picks them up.
            auto prefetch = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                // (after the wave's last tile the prefetch re-reads the final rows: harmless, and no branch around the loads)
                const uint32_t nt = t + n_waves < n_tiles ? t + n_waves : n_tiles - 1;
                const bool wraps = c + SETS >= PASSES;
                load_any((wraps ? nt * SEL_TILE : tile_base) + (uint32_t)(wraps ? c + SETS - PASSES : c + SETS) * FP_CHUNK, cur);
                __builtin_amdgcn_sched_barrier(0);
            };
            bool live[FP_ROWS];
            uint32_t d[FP_ROWS], m[FP_ROWS], lv[FP_ROWS];
            uint64_t pk[FP_ROWS];
            // d = the key's offset in the window (32 bits: a map read through a descriptor is shorter than 1 GiB, its window below 2^32
            // values), all ones for a dropped row; the byte offset of the row's map word is derived from it wherever it is needed
            // (one register per row slot instead of two: this kernel's occupancy is set by its registers)
            // 8-byte keys keep the word's byte offset in a register of its own: their windows reach 2^33 values (TPC-H SF1000 order keys in
            // dbgen's layout: 6 x 10^9, a 1.5 GB map) and those variants take 4 rows per lane anyway
            constexpr uint32_t MAP_WORD = BITS ? 4u : 8u;
            uint32_t vo8[KW == 8 ? FP_ROWS : 1];
            auto vo_of = [&](int k) { return KW == 8 ? vo8[KW == 8 ? k : 0] : (BITS ? ((d[k] >> 3) & ~3u) : ((d[k] >> 2) & ~7u)); };
            // the offset a dropped row carries: past the end of any map its variant reads through a descriptor (4-byte keys: what
            // d = all ones yields; 8-byte keys: the last word a 32-bit offset can name)
            constexpr uint32_t DROPPED = KW == 8 ? (0u - MAP_WORD) : (BITS ? 0x1FFFFFFCu : 0x3FFFFFF8u);
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) {
                uint32_t ok = EDGE ? (uint32_t)((base + 64u * k + lane) < n_right) & alive : alive;
#pragma unroll
                for (int j = 0; j < NF; ++j) ok &= (uint32_t)(((uint32_t)cur.f[j][k] - flo[j]) <= fspan[j]);
                const uint64_t off = key_offset<KW>(cur.key[k], T.kmin64);
                d[k] = (uint32_t)off;                                                     // (the bit-field extract below reads its low five bits)
                // a row the filter dropped, or whose key lies outside the window, reads an all-zero granule: the bit test below is
                // then the whole decision — ONE compare, whose lane mask is the ballot (a ballot of an AND of conditions costs two
                // more vector instructions per row slot)
                if constexpr (BITS || MAPBUF) {
                    // byte offset of the granule's word = (off >> 5) * 8 = (off >> 2) & ~7 (key-set words alone: (off >> 5) * 4); 8-byte keys
                    // beyond 2^32 must not wrap into the map
                    if constexpr (KW == 8) {
                        ok &= (uint32_t)((off >> 34) == 0);                  // (the word's byte offset fits 32 bits below 2^34; beyond that a key must not wrap into the map)
                        vo8[k] = ok ? (BITS ? ((uint32_t)(off >> 3) & ~3u) : ((uint32_t)(off >> 2) & ~7u)) : DROPPED;
                    } else {
                        d[k] = ok ? (uint32_t)off : 0xFFFFFFFFu;
                    }
                } else {
                    ok &= (uint32_t)in_window<KW>(off, T.krange64);
                    // granule index (the window holds <= 2^36 values: < 2^31 granules; 4-byte keys stay in 32-bit arithmetic)
                    const uint32_t g = KW == 4 ? ((uint32_t)off >> 5) : (uint32_t)(off >> 5);
                    pk[k] = rpack[ok ? g : T.rzero];
                }
            }
            if constexpr (BITS || MAPBUF) {
                // Probe keys clustered like the probe order (lineitem by order key): the 64 rows of a slot fall into one or two
                // neighbouring granules.  A gather instruction costs the L1 a tag lookup per four lanes whatever the addresses —
                // rocprofv3 on Q5's lineitem launch: 16.6 TCP accesses per map read against 4 per streamed key read, the L1 busy or
                // stalled 0.62 of the launch's 0.69 ms — so when EVERY slot of the pass is that narrow (wave-uniform test) the two words
                // of each slot come through the SCALAR cache (s_load) and the lanes pick theirs; one slot wider than that and the
                // pass gathers as before (random keys: orders by customer key).
                uint32_t g0[FP_ROWS];
                bool narrow = scalar_map;
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) {
                    const uint32_t vo = vo_of(k);
                    const uint64_t okm = __builtin_amdgcn_ballot_w64(vo != DROPPED);
                    const uint32_t first = okm ? (uint32_t)__builtin_amdgcn_readlane((int)vo, (int)__builtin_ctzll(okm)) : 0u;   // wave-uniform
                    g0[k] = first;
                    // ... and inside the map (the gather's descriptor range-checks; a scalar read does not): first + one word <= the zero granule
                    narrow = narrow && first < T.rzero * MAP_WORD && __builtin_amdgcn_ballot_w64(vo != DROPPED && (vo - first) > MAP_WORD) == 0ull;
                }
                if (narrow) {
                    uint64_t w0[FP_ROWS], w1[FP_ROWS];
#pragma unroll
                    for (int k = 0; k < FP_ROWS; ++k) {
                        if constexpr (BITS) {
                            typedef const uint32_t __attribute__((address_space(4)))* cptr;
                            const cptr mp = (cptr)(uintptr_t)(reinterpret_cast<const uint8_t*>(T.rbits) + g0[k]);
                            w0[k] = mp[0]; w1[k] = mp[1];
                        } else {
                            typedef const uint64_t __attribute__((address_space(4)))* cptr;
                            const cptr mp = (cptr)(uintptr_t)(reinterpret_cast<const uint8_t*>(rpack) + g0[k]);
                            w0[k] = mp[0]; w1[k] = mp[1];
                        }
                    }
                    if (!PERM && !RESID) prefetch();
#pragma unroll
                    for (int k = 0; k < FP_ROWS; ++k) {
                        // (DROPPED, not "d is all ones": a key up to 31 below the window's first maps to the same out-of-range word and
                        // was left out of the narrowness test like a dropped row — it must read zeros like one)
                        const uint32_t vo = vo_of(k);
                        pk[k] = vo == DROPPED ? 0ull : (vo == g0[k] ? w0[k] : w1[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < FP_ROWS; ++k) {
                        if constexpr (BITS) pk[k] = __builtin_amdgcn_raw_buffer_load_b32(rmap, (int)vo_of(k), 0, 0);
                        else {
                            typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
                            const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(rmap, (int)vo_of(k), 0, 0);
                            pk[k] = ((uint64_t)w.y << 32) | w.x;
                        }
                    }
                    if (!PERM && !RESID) prefetch();
                }
            } else {
                if (!PERM && !RESID) prefetch();
            }
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) {
                const uint32_t bits = (uint32_t)pk[k];
                lv[k] = __builtin_amdgcn_ubfe(bits, d[k], 1u);    // v_bfe_u32: bit (d & 31) of the key set
                live[k] = lv[k] != 0u;
                // the rank: needed at once where another dependent read goes by it (unsorted build side, second key column); else
                // only by the row slots in which some lane emits (Q3's lineitem probe: 0.5 % of the rows) — computed there
                if (PERM || RESID) m[k] = (uint32_t)(pk[k] >> 32) + (uint32_t)__popc(__builtin_amdgcn_ubfe(bits, 0u, d[k] & 31u));
            }
            if (PERM) {
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) m[k] = T.rperm[live[k] ? m[k] : 0u];
                if (!RESID) prefetch();
            }
            if (RESID) {
                uint32_t second[FP_ROWS], mine[FP_ROWS];
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) { second[k] = T.resid_build[live[k] ? m[k] : 0u]; mine[k] = cur.g[RESID ? k : 0]; }
                prefetch();                                     // (refills `cur`: this pass's second keys were copied out above)
#pragma unroll
                for (int k = 0; k < FP_ROWS; ++k) live[k] = live[k] && second[k] == mine[k];
            }
            uint64_t my_word = 0;
#pragma unroll
            for (int k = 0; k < FP_ROWS; ++k) {
                const bool emit = live[k];                                                // (inner join: a right join takes the general kernel)
                const uint64_t wd = __builtin_amdgcn_ballot_w64(emit);
                my_word = lane == (uint32_t)k ? wd : my_word;
                const uint32_t n_emit = (uint32_t)__popcll(wd);
                if (staged && wd != 0ull) {                                               // wave-uniform: a slot without a match costs one scalar test
                    if (__builtin_expect(strip_cnt + n_emit > STRIP, 0)) flush();         // wave-uniform, rare: laid out of line
                    if (emit) {
                        uint32_t mk;
                        if (PERM || RESID) mk = m[k];
                        else {
                            const uint32_t bits = (uint32_t)pk[k];
                            mk = (uint32_t)(pk[k] >> 32) + (uint32_t)__popc(__builtin_amdgcn_ubfe(bits, 0u, d[k] & 31u));     // set bits below the key's
                        }
                        // strip_cnt + (emitting lanes below this one): two mbcnt instructions, the count riding along as their addend
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(wd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wd, strip_cnt));
                        strip[at] = make_uint2(mk, base + 64u * k + lane);
                    }
                    strip_cnt += n_emit;
                }
                tile_cnt += n_emit;
            }
            // the pass's FP_ROWS selection words in one store (bitmap words past the last row's word exist: whole tiles)
            if (lane < FP_ROWS) (bitmap + (base >> 6))[lane] = my_word;
        };
        if (t < full_tiles) {                                                             // wave-uniform
#pragma unroll 1
            for (int c = 0; c < PASSES; c += SETS) {
#pragma unroll
                for (int q = 0; q < SETS; ++q) pass_body(c + q, sets[q], std::false_type{});
            }
        } else {
#pragma unroll 1
            for (int c = 0; c < PASSES; c += SETS) {
#pragma unroll
                for (int q = 0; q < SETS; ++q) pass_body(c + q, sets[q], std::true_type{});
            }
        }
        if (staged && strip_cnt != 0u) flush();
        if (lane == 0) tile_counts[t] = tile_cnt;
    }
}

// out[tile_off[t] + j] = staging[t * 1024 + j] for the tile's first (tile_off[t + 1] - tile_off[t]) entries.
// A wave takes 64 tiles at a time: their offsets in one coalesced load, then FOUR tiles per step, whose loads do not wait for
// each other (one tile after the other is a chain of dependent loads per tile: 0.15 ms for the 586 K tiles of Q5's lineitem probe)
constexpr int COMPACT_TILES = 4;
__global__ void __launch_bounds__(BLOCK)
join_compact_staged_kernel(const uint32_t* __restrict__ staging, const uint64_t* __restrict__ tile_off, uint64_t total, uint32_t n_tiles,
                           uint32_t* __restrict__ out, const uint32_t* __restrict__ staging2, uint32_t* __restrict__ out2) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    // every wave the same number of consecutive tiles
    const uint64_t per_wave = ((uint64_t)n_tiles + n_waves - 1) / n_waves;
    const uint64_t first = (uint64_t)wave_id * per_wave;
    const uint64_t last = first + per_wave < n_tiles ? first + per_wave : n_tiles;
    for (uint64_t t0 = first; t0 < last; t0 += 64) {
        const uint64_t t = t0 + lane;
        const uint64_t my_off = t < last ? tile_off[t] : total;
        const uint64_t my_end = t < last ? (t + 1 < n_tiles ? tile_off[t + 1] : total) : total;
        const uint32_t my_cnt = (uint32_t)(my_end - my_off);
        for (int j = 0; j < 64 && t0 + j < last; j += COMPACT_TILES) {
            uint64_t off[COMPACT_TILES];
            uint32_t cnt[COMPACT_TILES], most = 0;
#pragma unroll
            for (int q = 0; q < COMPACT_TILES; ++q) {
                off[q] = ((uint64_t)(uint32_t)__shfl((int)(my_off >> 32), j + q, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)my_off, j + q, 64);
                cnt[q] = (uint32_t)__shfl((int)my_cnt, j + q, 64);        // (0 past the last tile)
                most = cnt[q] > most ? cnt[q] : most;
            }
            for (uint32_t k = lane; k < most; k += 64) {
                uint32_t v[COMPACT_TILES], w[COMPACT_TILES];
#pragma unroll
                for (int q = 0; q < COMPACT_TILES; ++q) {
                    const uint64_t src = (t0 + j + q) * SEL_TILE + k;
                    v[q] = (staging && k < cnt[q]) ? staging[src] : 0;
                    w[q] = (staging2 && k < cnt[q]) ? staging2[src] : 0;
                }
#pragma unroll
                for (int q = 0; q < COMPACT_TILES; ++q) {
                    if (k < cnt[q]) {
                        if (staging) out[off[q] + k] = v[q];
                        if (staging2) out2[off[q] + k] = w[q];
                    }
                }
            }
        }
    }
}

// two 4-byte integer key columns -> ONE 8-byte key (first column in the high half): a two-column equi-join runs the
// single-key machinery; a row is valid when both parts are
__global__ void __launch_bounds__(BLOCK)
pack_key_pair_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int64_t n, uint64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        out[i] = ((uint64_t)a[i] << 32) | (uint64_t)b[i];
}
__global__ void __launch_bounds__(BLOCK)
and_bitmaps_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t n_words, uint64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * BLOCK)
        out[i] = (a ? a[i] : ~0ull) & (b ? b[i] : ~0ull);
}

int rows_grid(const LaunchCfg& cfg, size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

hipError_t launch_join_key_stats(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t* stats) {
    if (n == 0) return hipSuccess;
    auto st = reinterpret_cast<unsigned long long*>(stats);
    // every workgroup ends with three atomics on the SAME three words: few, long-running workgroups (4096 of them cost ~0.1 ms of
    // serialised atomics on a 5 M-row build side)
    size_t g = ((size_t)n + BLOCK * 8 - 1) / (BLOCK * 8);
    if (g > (size_t)cfg.device_cus * 2) g = (size_t)cfg.device_cus * 2;
    const unsigned stats_grid = (unsigned)(g < 1 ? 1 : g);
    if (key_width == 4) hipLaunchKernelGGL(join_key_stats_kernel<4>, dim3(stats_grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, st);
    else hipLaunchKernelGGL(join_key_stats_kernel<8>, dim3(stats_grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, st);
    return hipGetLastError();
}

int tiny_rank_build_max_rows() { return TINY_BUILD_ROWS; }
size_t tiny_rank_build_map_words() { return (size_t)(TINY_BUILD_WINDOW / 32) + 2; }
hipError_t launch_tiny_rank_build(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t* rpack, uint32_t* rperm,
                                  uint64_t* out) {
    if (n == 0 || n > (uint32_t)TINY_BUILD_ROWS || (key_width != 4 && key_width != 8)) return hipErrorInvalidValue;
    if (key_width == 4) hipLaunchKernelGGL(tiny_rank_build_kernel<4>, dim3(1), dim3(TINY_BUILD_ROWS), 0, cfg.stream, keys, sel, n, rpack, rperm, (unsigned long long*)out);
    else hipLaunchKernelGGL(tiny_rank_build_kernel<8>, dim3(1), dim3(TINY_BUILD_ROWS), 0, cfg.stream, keys, sel, n, rpack, rperm, (unsigned long long*)out);
    return hipGetLastError();
}

hipError_t launch_rank_bits(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t kmin, bool sorted,
                            uint64_t* bits, uint32_t* dup_flag) {
    if (n == 0) return hipSuccess;
    auto b = reinterpret_cast<unsigned long long*>(bits);
    const int grid = rows_grid(cfg, n);
    if (sorted) {
        if (key_width == 4) hipLaunchKernelGGL(rank_bits_sorted_kernel<4>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, n, kmin, b);
        else hipLaunchKernelGGL(rank_bits_sorted_kernel<8>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, n, kmin, b);
    } else {
        if (key_width == 4) hipLaunchKernelGGL(rank_bits_any_kernel<4>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, b, dup_flag);
        else hipLaunchKernelGGL(rank_bits_any_kernel<8>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, b, dup_flag);
    }
    return hipGetLastError();
}

hipError_t launch_rank_perm(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t kmin,
                            const uint64_t* rpack, uint32_t* perm) {
    if (n == 0) return hipSuccess;
    const int grid = rows_grid(cfg, n);
    if (key_width == 4) hipLaunchKernelGGL(rank_perm_kernel<4>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, rpack, perm);
    else hipLaunchKernelGGL(rank_perm_kernel<8>, dim3(grid), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, rpack, perm);
    return hipGetLastError();
}

template <int KW, int NF, int ROWS, bool RESID, bool PERM>
static void launch_rank_probe(hipStream_t st, unsigned grid, bool map_buf, bool bits_only, const NarrowJoinTable& T, const ProbeFilter& F, const void* rkeys,
                              uint32_t n_right, uint64_t* bitmap, uint32_t* tile_counts, uint32_t* staging, const uint32_t* resid_probe,
                              uint32_t* staging_rows) {
    if constexpr (!RESID && !PERM) {
        if (bits_only) {
            hipLaunchKernelGGL((join_rank_probe_kernel<KW, NF, ROWS, false, false, true, true>), dim3(grid), dim3(BLOCK), 0, st, T, F, rkeys, n_right, bitmap,
                               tile_counts, staging, resid_probe, staging_rows);
            return;
        }
    }
    if (map_buf)
        hipLaunchKernelGGL((join_rank_probe_kernel<KW, NF, ROWS, RESID, PERM, true>), dim3(grid), dim3(BLOCK), 0, st, T, F, rkeys, n_right, bitmap, tile_counts,
                           staging, resid_probe, staging_rows);
    else
        hipLaunchKernelGGL((join_rank_probe_kernel<KW, NF, ROWS, RESID, PERM, false>), dim3(grid), dim3(BLOCK), 0, st, T, F, rkeys, n_right, bitmap, tile_counts,
                           staging, resid_probe, staging_rows);
}

hipError_t launch_join_filter_probe(const LaunchCfg& cfg, const NarrowJoinTable& T, const ProbeFilter& F, const void* rkeys, int key_width,
                                    const uint64_t* rsel, uint32_t n_right, bool right_outer, uint64_t* bitmap, uint32_t* tile_counts,
                                    uint32_t* staging, uint32_t* matched, const uint32_t* resid_probe, uint32_t* staging_rows) {
    if (n_right == 0) return hipSuccess;
    if ((resid_probe != nullptr) != (T.resid_build != nullptr)) return hipErrorInvalidValue;
    const int64_t n_tiles = ((int64_t)n_right + SEL_TILE - 1) / SEL_TILE;
    static const int per_cu = [] { const char* v = getenv("BHIP_PROBE_BLOCKS_PER_CU"); return v && atoi(v) > 0 ? atoi(v) : 12; }();   // profiles/r02_probe_variants_q3_sf100.txt
    int64_t grid = (int64_t)cfg.device_cus * per_cu;
    const int64_t need = (n_tiles + BLOCK / 64 - 1) / (BLOCK / 64);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    // rows per lane and pass (4-byte keys).  r02: 8 (profiles/r02_probe_variants_q3_sf100.txt, at 62 registers either way); with two passes
    // of streamed rows in flight 8 rows cost 105-135 registers (3-4 waves per SIMD), 4 rows 62-78 (6-7 waves): Q3's probes 1.52 -> 1.42 ms,
    // Q5's 1.31 -> 1.28 (profiles/r03_probe_kernel_pmc_and_variants.txt); BHIP_PROBE_ROWS=8 for the A/B
    static const int probe_rows = [] { const char* v = getenv("BHIP_PROBE_ROWS"); return v ? atoi(v) : 4; }();
    // the rank map without NULL probe keys and without a left join: the one-read kernel
    const bool direct = T.rpack != nullptr && rsel == nullptr && matched == nullptr && !right_outer && (staging != nullptr) == (staging_rows != nullptr);
    // the packed map through a buffer descriptor when its granules fit one (BHIP_PROBE_MAP_FLAT=1: the flat-load variant, the A/B partner)
    static const bool map_flat = [] { const char* v = getenv("BHIP_PROBE_MAP_FLAT"); return v && atoi(v) != 0; }();
    // the packed map through a descriptor: shorter than 1 GiB for 4-byte keys (the kernel keeps their 32-bit offsets only), 2 GiB for 8-byte keys
    const bool map_buf = !map_flat && T.rpack != nullptr && (uint64_t)T.rzero * 8u < (key_width == 8 ? 0x7FFFFFF0ull : 0x3FFFFFF0ull);
    // nothing staged (a semi-join), sorted one-column build side: the key-set words alone (BHIP_PROBE_NO_BITS=1: the packed map, the A/B partner)
    static const bool no_bits = [] { const char* v = getenv("BHIP_PROBE_NO_BITS"); return v && atoi(v) != 0; }();
    const bool bits_only = !no_bits && map_buf && T.rbits != nullptr && staging == nullptr && T.rperm == nullptr && resid_probe == nullptr;
#define BHIP_PROBE_R(KW_, NF_, ROWS_, RESID_, PERM_)                                                                                  \
    launch_rank_probe<KW_, NF_, ROWS_, RESID_, PERM_>(cfg.stream, (unsigned)grid, map_buf, bits_only, T, F, rkeys, n_right, bitmap, tile_counts, staging, \
                                                      resid_probe, staging_rows)
#define BHIP_PROBE_L(KW_, NF_, RESID_)                                                                                                \
    do {                                                                                                                              \
        if (direct && T.rperm) BHIP_PROBE_R(KW_, NF_, 4, RESID_, true);                                                               \
        else if (direct && probe_rows == 8 && KW_ == 4) BHIP_PROBE_R(4, NF_, 8, RESID_, false);                                       \
        else if (direct) BHIP_PROBE_R(KW_, NF_, 4, RESID_, false);                                                                    \
        else                                                                                                                          \
            hipLaunchKernelGGL((join_filter_probe_kernel<KW_, NF_, 4, RESID_>), dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, T, F, rkeys, \
                               rsel, n_right, right_outer ? 1 : 0, bitmap, tile_counts, staging, matched, resid_probe, staging_rows); \
    } while (0)
#define BHIP_PROBE(KW_, NF_)                                                                                                          \
    do {                                                                                                                              \
        if (resid_probe) BHIP_PROBE_L(KW_, NF_, true);                                                                                \
        else BHIP_PROBE_L(KW_, NF_, false);                                                                                           \
    } while (0)
    if (key_width == 4) {
        if (F.n == 0) BHIP_PROBE(4, 0);
        else if (F.n == 1) BHIP_PROBE(4, 1);
        else BHIP_PROBE(4, JOIN_FILTER_MAX);
    } else {
        if (F.n == 0) BHIP_PROBE(8, 0);
        else if (F.n == 1) BHIP_PROBE(8, 1);
        else BHIP_PROBE(8, JOIN_FILTER_MAX);
    }
#undef BHIP_PROBE
#undef BHIP_PROBE_L
#undef BHIP_PROBE_R
    return hipGetLastError();
}

hipError_t launch_and_bitmaps(const LaunchCfg& cfg, const uint64_t* a, const uint64_t* b, int64_t n_bits, uint64_t* out) {
    if (n_bits == 0) return hipSuccess;
    hipLaunchKernelGGL(and_bitmaps_kernel, dim3(rows_grid(cfg, (size_t)(n_bits + 63) / 64)), dim3(BLOCK), 0, cfg.stream, a, b, (n_bits + 63) / 64, out);
    return hipGetLastError();
}
hipError_t launch_pack_key_pair(const LaunchCfg& cfg, const void* a, const void* b, const uint64_t* va, const uint64_t* vb, int64_t n, uint64_t* out,
                                uint64_t* validity_out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_key_pair_kernel, dim3(rows_grid(cfg, (size_t)n)), dim3(BLOCK), 0, cfg.stream, static_cast<const uint32_t*>(a),
                       static_cast<const uint32_t*>(b), n, out);
    if (validity_out)
        hipLaunchKernelGGL(and_bitmaps_kernel, dim3(rows_grid(cfg, (size_t)(n + 63) / 64)), dim3(BLOCK), 0, cfg.stream, va, vb, (n + 63) / 64, validity_out);
    return hipGetLastError();
}

hipError_t launch_join_compact_staged(const LaunchCfg& cfg, const uint32_t* staging, const uint64_t* tile_off, uint64_t total,
                                      int64_t n_tiles, uint32_t* out, const uint32_t* staging2, uint32_t* out2) {
    if (n_tiles == 0 || total == 0) return hipSuccess;
    int64_t grid = (int64_t)cfg.device_cus * 8;
    const int64_t need = (n_tiles + 4 * (BLOCK / 64) - 1) / (4 * (BLOCK / 64));   // at least one step of four tiles per wave
    if (grid > need) grid = need;
    hipLaunchKernelGGL(join_compact_staged_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, cfg.stream, staging, tile_off, total,
                       (uint32_t)n_tiles, out, staging2, out2);
    return hipGetLastError();
}

}  // namespace bhip
