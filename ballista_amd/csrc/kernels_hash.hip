// kernels_hash.hip — device-wide hash tables: high-cardinality HashAggregateExec and
// HashJoinExec build / probe (rust/core/src/serde/physical_plan/from_proto.rs:173-276).
//
// Design (MI355X: 8 XCDs whose L2s are not coherent with each other, per-CU L1s never refreshed):
// a slot is claimed by ONE 32-bit CAS that stores the claiming row's id; key equality is decided
// by comparing against that row's packed key in a key array written by an EARLIER kernel
// (scan_keys), so everything a prober reads besides the CAS word is immutable during the kernel —
// no in-kernel release/acquire protocol, no stale-cache hazard.  Accumulators are only ever
// touched by atomics (executed at the memory side).  Bound: random 16-B key reads + atomic traffic;
// the algorithmic-bytes roofline deliberately does not credit it.
#include <hip/hip_runtime.h>
#include "host/hash_kernels.h"
#include "launch_common.h"
#include "reduce_device.h"
#include "vm_device.h"

namespace bhip {

__device__ inline void atomic_acc(uint64_t* p, uint64_t v, int kind) {
    switch (kind) {
        case ACC_SUM_F64: unsafeAtomicAdd(reinterpret_cast<double*>(p), u2d(v)); break;
        case ACC_MIN_F64: atomicMin(reinterpret_cast<double*>(p), u2d(v)); break;
        case ACC_MAX_F64: atomicMax(reinterpret_cast<double*>(p), u2d(v)); break;
        case ACC_MIN_I64: atomicMin(reinterpret_cast<long long*>(p), (long long)v); break;
        case ACC_MAX_I64: atomicMax(reinterpret_cast<long long*>(p), (long long)v); break;
        default: atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); break;
    }
}

// find (or claim) the slot of `key`; `row` = global id of the probing row (index into keys128)
__device__ inline uint32_t table_upsert(uint32_t* owner, uint64_t mask, const uint64_t* keys128, const Key128& key,
                                        uint32_t row) {
    uint64_t slot = hash_key(key) & mask;
    for (;;) {
        uint32_t o = owner[slot];
        if (o == 0) {
            o = atomicCAS(&owner[slot], 0u, row + 1u);
            if (o == 0) return (uint32_t)slot;       // claimed: this row's key defines the slot
        }
        const uint64_t* k = keys128 + 2ull * (o - 1u);
        if (k[0] == key.k0 && k[1] == key.k1) return (uint32_t)slot;
        slot = (slot + 1) & mask;
    }
}

// T.fsum_of_acc[a] without indexing the by-value argument dynamically (that copies the array to scratch: a global-memory read per
// accumulator and row): the 16 bytes as two scalars, shifted
__device__ inline uint32_t fsum_index(const HashAggTable& T, int a) {
    static_assert(VM_MAX_ACC == 16, "two 8-byte halves");
    uint64_t w0, w1;
    __builtin_memcpy(&w0, T.fsum_of_acc, 8);
    __builtin_memcpy(&w1, T.fsum_of_acc + 8, 8);
    return (uint32_t)(((a < 8 ? w0 : w1) >> (8 * (a & 7))) & 0xFFu);
}

template <int R, bool NULLS>
__global__ void __launch_bounds__(BLOCK)
scan_agg_hash_kernel(const ScanParams P, const HashAggTable T, uint32_t row_base, ScanStatus* status) {
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
            bool live = row < P.n_rows;
            const uint32_t grow = row_base + (uint32_t)row;
            if (live && P.pred_slot >= 0) live = L.bvals[P.pred_slot * TILE + idx] & 1;
            if (!live) {
                if (T.n_fsum && !T.slots_given && row < P.n_rows) T.rowslot[grow] = 0xFFFFFFFFu;
                continue;
            }
            uint32_t slot;
            if (T.slots_given) {
                slot = T.rowslot[grow];                                 // clustered input: the row's run is its group (run_* kernels below)
            } else {
                const Key128 key{T.keys128[2ull * grow], T.keys128[2ull * grow + 1]};
                slot = table_upsert(T.owner, T.mask, T.keys128, key, grow);
                if (T.n_fsum) T.rowslot[grow] = slot;
            }
            // (with fixed-order sums the row counts come from the segment kernel: one atomic per run of rows, kernels_dagg.hip)
            if (!T.n_fsum) atomicAdd(reinterpret_cast<unsigned long long*>(&T.rows[slot]), 1ull);
            for (int a = 0; a < P.n_acc; ++a) {
                const AccSpec sp = P.acc[a];
                uint64_t v = 1;
                bool k = true;
                if (sp.kind == ACC_COUNT_VALID_B) k = L.bvals[sp.slot * TILE + idx] >> 1;
                else if (sp.kind != ACC_COUNT_ROWS) {
                    v = L.vals[sp.slot * TILE + idx];
                    if (NULLS) k = L.vvalid[sp.slot * TILE + idx];
                }
                if (sp.kind == ACC_COUNT_VALID || sp.kind == ACC_COUNT_VALID_B) { v = k ? 1 : 0; k = true; }
                const uint32_t fs = fsum_index(T, a);
                if (T.n_fsum && fs != 0xFF) {       // summed later, in row order: record the addend (NULL adds +0.0)
                    T.fvals[(size_t)fs * T.total_rows + grow] = k ? u2d(v) : 0.0;
                    if (NULLS && k) atomicAdd(reinterpret_cast<unsigned long long*>(&T.nvalid[(size_t)slot * T.n_acc + a]), 1ull);
                    continue;
                }
                if (!k) continue;
                atomic_acc(&T.acc[(size_t)slot * T.n_acc + a], v, sp.kind);
                if (NULLS) atomicAdd(reinterpret_cast<unsigned long long*>(&T.nvalid[(size_t)slot * T.n_acc + a]), 1ull);
            }
        }
    }
    if (err) atomicOr(&status->flags, err);
}

// ---- input clustered by group key ------------------------------------------------------------------------------------------
// When the rows of a group tend to be consecutive (lineitem joined to its orders and grouped by the order key: TPC-H Q3, 2.6
// rows per group), the table is consulted once per RUN of equal packed keys instead of once per row, and the slot space is the
// dense space of runs:
//   run_heads   flags[i] = row i starts a run                                   (then an exclusive scan: runs before row i)
//   run_slots   rowslot[i] = run of row i; head[run] = its first row
//   run_claim   every run looks its key up in a table keyed by head rows and records the SMALLEST head row among the runs with
//               that key (a key that comes back later — nothing is assumed about the input — joins the earlier run)
//   run_winner  winner[run] = the run that owns the key; owner[run] = head row + 1 for an owning run, 0 for one that joined another
//   run_remap   rowslot[i] = winner[run of row i]
// after which the scan adds into slot rowslot[i] (scan_agg_hash_kernel with slots_given) and the compaction runs over the run
// space.  The group of every row is decided by key equality alone; deterministic whatever the schedule (the minimum).
__global__ void __launch_bounds__(BLOCK)
run_heads_kernel(const uint64_t* keys128, uint32_t n, uint32_t* flags, uint64_t first_mask, unsigned long long* not_ascending) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const ulonglong2 k = reinterpret_cast<const ulonglong2*>(keys128)[i];
        uint32_t head = 1;
        if (i > 0) {
            const ulonglong2 p = reinterpret_cast<const ulonglong2*>(keys128)[i - 1];
            head = (k.x != p.x || k.y != p.y) ? 1u : 0u;
            // first key part (the low bytes of the packed key) strictly increasing from run to run => no key comes back in a later run:
            // every run is a group of its own and no table has to be consulted (lineitem rows of a join arrive in order-key order).
            // info[1] counts the places where it does not increase, info[2] = the first such row: ONE such place (a table whose
            // tail wraps around, two sorted batches back to back) still needs no table — launch_run_tail below
            if (head && (k.x & first_mask) <= (p.x & first_mask)) {
                atomicAdd(&not_ascending[0], 1ull);
                atomicMin(&not_ascending[1], (unsigned long long)i);
            }
        }
        flags[i] = head;
    }
}

// TWO ascending stretches of runs (exactly one place where the first key part does not increase, at row `split_row`): the runs
// before it are distinct, the runs from it on are distinct among themselves, and a later run can only repeat a key of the first
// stretch — found by binary search over the first stretch's head keys (they ascend), no table.
//   resolve  head2[r] = head[r] for the first stretch; a later run r: match[r - t0] = the run of the first stretch with its key
//            (0xFFFFFFFF: none), fresh[r - t0] = 1 when it is a new group                      (t0 = the run of split_row)
//   (an exclusive scan of `fresh` by the caller)
//   remap    rows from split_row on: rowslot = the matched run, or t0 + (new groups before the row's run); head2 of the new groups;
//            *n_groups_out = t0 + new groups
__global__ void __launch_bounds__(BLOCK)
run_tail_resolve_kernel(const uint64_t* keys128, const uint32_t* head, const uint64_t* n_runs_dev, const uint32_t* rowslot, uint32_t split_row,
                        uint64_t first_mask, uint32_t* head2, uint32_t* match, uint32_t* fresh) {
    const uint32_t n_runs = (uint32_t)*n_runs_dev, t0 = rowslot[split_row];
    for (uint32_t r = blockIdx.x * BLOCK + threadIdx.x; r < n_runs; r += gridDim.x * BLOCK) {
        if (r < t0) { head2[r] = head[r]; continue; }
        const uint32_t h = head[r];
        const uint64_t k0 = keys128[2ull * h], k1 = keys128[2ull * h + 1], want = k0 & first_mask;
        uint32_t lo = 0, hi = t0;                                       // first run of the first stretch whose first part is >= want
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if ((keys128[2ull * head[mid]] & first_mask) < want) lo = mid + 1;
            else hi = mid;
        }
        uint32_t m = 0xFFFFFFFFu;
        if (lo < t0) {
            const uint32_t hm = head[lo];
            if (keys128[2ull * hm] == k0 && keys128[2ull * hm + 1] == k1) m = lo;
        }
        match[r - t0] = m;
        fresh[r - t0] = m == 0xFFFFFFFFu ? 1u : 0u;
    }
}
__global__ void __launch_bounds__(BLOCK)
run_tail_remap_kernel(const uint32_t* head, const uint64_t* n_runs_dev, uint32_t n_rows, uint32_t split_row, const uint32_t* match,
                      const uint32_t* fresh_before, uint32_t* rowslot, uint32_t* head2, unsigned long long* n_groups_out) {
    const uint32_t n_runs = (uint32_t)*n_runs_dev;
    // (rowslot[split_row] is rewritten below: every thread takes t0 from the run index of the row in front of it)
    const uint32_t t0 = split_row > 0 ? rowslot[split_row - 1] + 1u : 0u;
    for (uint32_t i = split_row + blockIdx.x * BLOCK + threadIdx.x; i < n_rows; i += gridDim.x * BLOCK) {
        const uint32_t r = rowslot[i], t = r - t0, m = match[t];
        const uint32_t g = m != 0xFFFFFFFFu ? m : t0 + fresh_before[t];
        rowslot[i] = g;
        if (m == 0xFFFFFFFFu && head[r] == i) head2[g] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t n_tail = n_runs - t0;
        *n_groups_out = (unsigned long long)t0 + (n_tail ? fresh_before[n_tail] : 0u);
    }
}

// distinct runs (run_heads found the first key part ascending): the group records straight from the runs — slot r IS group r
__global__ void __launch_bounds__(BLOCK)
run_compact_kernel(HashAggTable T, const uint32_t* head, uint32_t n_runs, int nulls, GroupRec* out) {
    for (uint32_t r = blockIdx.x * BLOCK + threadIdx.x; r < n_runs; r += gridDim.x * BLOCK) {
        const uint32_t o = head[r];
        GroupRec& g = out[r];
        g.k0 = T.keys128[2ull * o];
        g.k1 = T.keys128[2ull * o + 1];
        g.rows = T.rows[r];
        for (int a = 0; a < T.n_acc; ++a) {
            g.acc[a] = T.acc[(size_t)r * T.n_acc + a];
            g.nvalid[a] = nulls ? T.nvalid[(size_t)r * T.n_acc + a] : g.rows;
        }
    }
}
__global__ void __launch_bounds__(BLOCK)
run_slots_kernel(const uint32_t* flags, const uint32_t* runs_before, uint32_t n, uint32_t* rowslot, uint32_t* head) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t run = runs_before[i] + flags[i] - 1u;
        rowslot[i] = run;
        if (flags[i]) head[run] = i;
    }
}
__global__ void __launch_bounds__(BLOCK)
run_claim_kernel(const uint64_t* keys128, const uint32_t* head, const uint64_t* n_runs_dev, uint32_t* table, uint64_t mask, uint32_t* min_head,
                 uint32_t* slot_of_run) {
    const uint32_t n_runs = (uint32_t)*n_runs_dev;
    for (uint32_t r = blockIdx.x * BLOCK + threadIdx.x; r < n_runs; r += gridDim.x * BLOCK) {
        const uint32_t h = head[r];
        const Key128 key{keys128[2ull * h], keys128[2ull * h + 1]};
        const uint32_t slot = table_upsert(table, mask, keys128, key, h);
        atomicMin(&min_head[slot], h);
        slot_of_run[r] = slot;
    }
}
__global__ void __launch_bounds__(BLOCK)
run_winner_kernel(const uint32_t* head, const uint64_t* n_runs_dev, const uint32_t* min_head, const uint32_t* slot_of_run, const uint32_t* run_of_row,
                  uint32_t* winner, uint32_t* owner) {
    const uint32_t n_runs = (uint32_t)*n_runs_dev;
    for (uint32_t r = blockIdx.x * BLOCK + threadIdx.x; r < n_runs; r += gridDim.x * BLOCK) {
        const uint32_t w = min_head[slot_of_run[r]];           // first row of the earliest run with this key
        winner[r] = run_of_row[w];
        owner[r] = w == head[r] ? w + 1u : 0u;
    }
}
__global__ void __launch_bounds__(BLOCK)
run_remap_kernel(const uint32_t* winner, uint32_t n, uint32_t* rowslot) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) rowslot[i] = winner[rowslot[i]];
}

// accumulator identities (min/max) must be in place before the first atomic
__global__ void __launch_bounds__(BLOCK)
hash_agg_init_kernel(HashAggTable T, MergeAccKinds kinds) {
    const size_t n = (size_t)(T.mask + 1) * T.n_acc;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
        T.acc[i] = acc_identity(kinds.kind[i % T.n_acc]);
}

// used slots -> flags (for the prefix sum), then dense GroupRec records
__global__ void __launch_bounds__(BLOCK)
hash_agg_flags_kernel(HashAggTable T, uint32_t* flags) {
    const size_t n = (size_t)(T.mask + 1);
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
        flags[i] = T.owner[i] != 0 ? 1u : 0u;
}

__global__ void __launch_bounds__(BLOCK)
hash_agg_compact_kernel(HashAggTable T, const uint64_t* dense_index, int nulls, GroupRec* out) {
    const size_t n = (size_t)(T.mask + 1);
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const uint32_t o = T.owner[i];
        if (o == 0) continue;
        GroupRec& g = out[dense_index ? dense_index[i] : i];
        g.k0 = T.keys128[2ull * (o - 1u)];
        g.k1 = T.keys128[2ull * (o - 1u) + 1];
        g.rows = T.rows[i];
        for (int a = 0; a < T.n_acc; ++a) {
            g.acc[a] = T.acc[i * T.n_acc + a];
            g.nvalid[a] = nulls ? T.nvalid[i * T.n_acc + a] : g.rows;
        }
    }
}

// =============================================================================================
// HashJoinExec: build (left side) and probe (right side)
// =============================================================================================
__device__ inline bool bit_at(const uint64_t* bm, uint32_t i) { return bm == nullptr || ((bm[i >> 6] >> (i & 63)) & 1ull); }

// JoinTable slot word = (claiming row + 1) | (high half of the key's hash) << 32, claimed by ONE 64-bit CAS.  The tag lets
// a prober pass a slot of another key without touching that key in keys128: the probe is bound by the number of cache
// lines it pulls out of L2 / MALL (Q5: 18 M probes, 96 % of them misses walking ~2.5 slots, two lines per slot before).
__device__ inline uint32_t join_table_upsert(const JoinTable& T, const Key128& key, uint32_t row) {
    const uint64_t h = hash_key(key);
    const uint64_t tag = h >> 32;
    const unsigned long long want = (unsigned long long)(row + 1u) | (tag << 32);
    uint64_t slot = h & T.mask;
    for (;;) {
        unsigned long long o = T.owner[slot];
        if (o == 0) {
            o = atomicCAS(reinterpret_cast<unsigned long long*>(&T.owner[slot]), 0ull, want);
            if (o == 0) return (uint32_t)slot;       // claimed: this row's key defines the slot
        }
        if ((o >> 32) == tag) {
            const uint64_t* k = T.keys128 + 2ull * ((uint32_t)o - 1u);
            if (k[0] == key.k0 && k[1] == key.k1) return (uint32_t)slot;
        }
        slot = (slot + 1) & T.mask;
    }
}

__global__ void __launch_bounds__(BLOCK)
join_build_kernel(JoinTable T, const uint64_t* sel, uint32_t n_left) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_left; row += gridDim.x * BLOCK) {
        if (!bit_at(sel, row)) continue;                       // NULL keys never match
        const Key128 key{T.keys128[2ull * row], T.keys128[2ull * row + 1]};
        const uint32_t slot = join_table_upsert(T, key, row);
        const uint32_t prev = atomicExch(&T.head[slot], row + 1u);     // push on the slot's chain
        T.next[row] = prev;
        if (prev != 0 && T.dup_flag) *T.dup_flag = 1u;                 // a second row with this key: not a unique build side
    }
}

// slot holding `key`, or 0xFFFFFFFF (read-only: the table was finished by the build kernel)
__device__ inline uint32_t table_find(const JoinTable& T, const Key128& key) {
    const uint64_t h = hash_key(key);
    const uint64_t tag = h >> 32;
    uint64_t slot = h & T.mask;
    for (;;) {
        const uint64_t o = T.owner[slot];
        if (o == 0) return 0xFFFFFFFFu;
        if ((o >> 32) == tag) {
            const uint64_t* k = T.keys128 + 2ull * ((uint32_t)o - 1u);
            if (k[0] == key.k0 && k[1] == key.k1) return (uint32_t)slot;
        }
        slot = (slot + 1) & T.mask;
    }
}

__global__ void __launch_bounds__(BLOCK)
join_probe_count_kernel(JoinTable T, const uint64_t* rkeys128, const uint64_t* rsel, uint32_t n_right, int right_outer,
                        uint32_t* counts) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_right; row += gridDim.x * BLOCK) {
        uint32_t c = 0;
        if (bit_at(rsel, row)) {
            const Key128 key{rkeys128[2ull * row], rkeys128[2ull * row + 1]};
            const uint32_t slot = table_find(T, key);
            if (slot != 0xFFFFFFFFu)
                for (uint32_t l = T.head[slot]; l != 0; l = T.next[l - 1u]) ++c;
        }
        counts[row] = (right_outer && c == 0) ? 1u : c;
    }
}

__global__ void __launch_bounds__(BLOCK)
join_probe_emit_kernel(JoinTable T, const uint64_t* rkeys128, const uint64_t* rsel, uint32_t n_right, int right_outer,
                       const uint64_t* offsets, uint32_t* left_idx, uint32_t* right_idx, uint32_t* matched) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_right; row += gridDim.x * BLOCK) {
        uint64_t pos = offsets[row];
        uint32_t c = 0;
        if (bit_at(rsel, row)) {
            const Key128 key{rkeys128[2ull * row], rkeys128[2ull * row + 1]};
            const uint32_t slot = table_find(T, key);
            if (slot != 0xFFFFFFFFu)
                for (uint32_t l = T.head[slot]; l != 0; l = T.next[l - 1u]) {
                    left_idx[pos] = l - 1u;
                    right_idx[pos] = row;
                    if (matched) atomicOr(&matched[(l - 1u) >> 5], 1u << ((l - 1u) & 31));
                    ++pos;
                    ++c;
                }
        }
        if (right_outer && c == 0) { left_idx[pos] = 0xFFFFFFFFu; right_idx[pos] = row; }
    }
}

// ---- unique build keys (primary-key side: every TPC-H join): a probe row has at most one partner, so the
// table is probed ONCE and the result is a selection: partner[row] (build row id, 0xFFFFFFFF = none), the bitmap
// of emitting rows and its per-1024-row counts — the index pass of FilterExec (select_indices) then yields the
// probe-side indices in row order and one gather of partner[] the build-side indices.  With unique keys the
// slot's owner IS its only row: the chain head is never read (two random reads per probe instead of three).
__global__ void __launch_bounds__(BLOCK)
join_probe_match_kernel(JoinTable T, const uint64_t* rkeys128, const uint64_t* rsel, uint32_t n_right, int right_outer,
                        uint32_t* partner, uint64_t* bitmap, uint32_t* tile_counts, uint32_t* matched) {
    // four rows per lane and pass, as the narrow probe: the packed keys, then the slot owners, then the owners' keys of all
    // four are in flight together (the chain key -> owner -> owner's key is latency; Q5: 18 M probes took 0.72 ms one by one)
    constexpr int PROBE_ROWS = 4;
    static_assert(SEL_TILE % (64 * PROBE_ROWS) == 0, "the rows of one pass of a wave lie in one selection tile");
    const int lane = threadIdx.x & 63;
    const uint64_t n_round = ((uint64_t)n_right + 63u) & ~(uint64_t)63;
    const uint64_t wave_rows = 64ull * PROBE_ROWS;
    const uint64_t wave_id = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    const ulonglong2* rk = reinterpret_cast<const ulonglong2*>(rkeys128);
    const ulonglong2* lk = reinterpret_cast<const ulonglong2*>(T.keys128);
    for (uint64_t wbase = wave_id * wave_rows; wbase < n_round; wbase += n_waves * wave_rows) {
        ulonglong2 key[PROBE_ROWS];
        uint64_t slot[PROBE_ROWS], tag[PROBE_ROWS], owner[PROBE_ROWS];
        uint32_t m[PROBE_ROWS];
        bool in[PROBE_ROWS], live[PROBE_ROWS];
#pragma unroll
        for (int k = 0; k < PROBE_ROWS; ++k) {
            const uint64_t row64 = wbase + 64ull * k + lane;
            in[k] = row64 < n_right;
            live[k] = in[k] && bit_at(rsel, (uint32_t)row64);
            key[k] = live[k] ? rk[row64] : ulonglong2{0ull, 0ull};
            m[k] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < PROBE_ROWS; ++k) {
            const uint64_t h = hash_key(Key128{key[k].x, key[k].y});
            slot[k] = h & T.mask;
            tag[k] = h >> 32;
            owner[k] = live[k] ? T.owner[slot[k]] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < PROBE_ROWS; ++k) {
            while (owner[k] != 0) {
                if ((owner[k] >> 32) == tag[k]) {
                    const ulonglong2 ok = lk[(uint32_t)owner[k] - 1u];
                    if (ok.x == key[k].x && ok.y == key[k].y) { m[k] = (uint32_t)owner[k] - 1u; break; }
                }
                slot[k] = (slot[k] + 1) & T.mask;
                owner[k] = T.owner[slot[k]];
            }
            if (matched && m[k] != 0xFFFFFFFFu) atomicOr(&matched[m[k] >> 5], 1u << (m[k] & 31));
        }
        uint32_t emitted = 0;
#pragma unroll
        for (int k = 0; k < PROBE_ROWS; ++k) {
            const uint64_t row64 = wbase + 64ull * k + lane;
            const bool emit = in[k] && (right_outer || m[k] != 0xFFFFFFFFu);
            if (emit) partner[(uint32_t)row64] = m[k];
            const uint64_t word = __ballot(emit);
            if (lane == 0 && wbase + 64ull * k < n_round) bitmap[(wbase >> 6) + k] = word;
            emitted += (uint32_t)__popcll(word);
        }
        if (lane == 0 && emitted) atomicAdd(&tile_counts[wbase / SEL_TILE], emitted);
    }
}

// ---- narrow keys: ONE Int32 / Date32 key column and a unique build side (the primary-key joins of TPC-H).
// The slot holds the key and the build row together (key | (row + 1) << 32), so a probe step is ONE random
// 8-byte read instead of a slot read followed by a dependent read of the 16-byte packed key; no packed keys are
// materialised on either side.  A second build row with the same key raises dup_flag and the host rebuilds
// with the general table.
__device__ inline uint64_t narrow_hash(uint32_t key) { return mix64((uint64_t)key); }

__global__ void __launch_bounds__(BLOCK)
join_build_narrow_kernel(NarrowJoinTable T, const uint32_t* keys, const uint64_t* sel, uint32_t n_left) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_left; row += gridDim.x * BLOCK) {
        if (!bit_at(sel, row)) continue;                       // NULL keys never match
        const uint32_t key = keys[row];
        const unsigned long long mine = (unsigned long long)key | ((unsigned long long)(row + 1u) << 32);
        uint64_t slot = narrow_hash(key) & T.mask;
        for (;;) {
            unsigned long long v = T.slots[slot];
            if (v == 0) {
                v = atomicCAS(reinterpret_cast<unsigned long long*>(&T.slots[slot]), 0ull, mine);
                if (v == 0) break;                             // claimed
            }
            if ((uint32_t)v == key) { *T.dup_flag = 1u; break; }
            slot = (slot + 1) & T.mask;
        }
    }
}

__global__ void __launch_bounds__(BLOCK)
join_key_present64_kernel(const uint64_t* keys, const uint64_t* sel, uint32_t n, uint64_t kmin, uint32_t* present) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK)
        if (bit_at(sel, row)) { const uint64_t d = keys[row] - kmin; atomicOr(&present[d >> 5], 1u << (d & 31)); }
}

__global__ void __launch_bounds__(BLOCK)
join_key_present_kernel(const uint32_t* keys, const uint64_t* sel, uint32_t n, uint32_t kmin, uint32_t* present) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK)
        if (bit_at(sel, row)) { const uint32_t d = keys[row] - kmin; atomicOr(&present[d >> 5], 1u << (d & 31)); }
}

// ---- group keys of any width: every row gets a REPRESENTATIVE row with an equal key -------------------------------
// (HashAggregateExec over keys that do not fit the 16-byte packed key: c_name, c_address, c_comment ... of TPC-H Q10.)
// The table holds row ids claimed by one 32-bit CAS; equality is decided on the immutable key COLUMNS of the two rows
// (NULL == NULL, as GROUP BY wants), so nothing a prober reads besides the CAS word changes during the kernel.
// Which of the equal rows becomes the representative is decided by the race; the groups are the same either way.
__device__ inline bool wide_keys_equal(const WideKeyCols& K, uint32_t a, uint32_t b) {
    for (int c = 0; c < K.n; ++c) {
        const ColumnRef& r = K.col[c];
        const bool va = bit_at(r.validity, a), vb = bit_at(r.validity, b);
        if (va != vb) return false;
        if (!va) continue;
        switch (r.dtype) {
            case DT_UTF8: {
                const int32_t a0 = r.offsets[a], a1 = r.offsets[a + 1], b0 = r.offsets[b], b1 = r.offsets[b + 1];
                if (a1 - a0 != b1 - b0) return false;
                const uint8_t* d = static_cast<const uint8_t*>(r.data);
                for (int32_t i = 0; i < a1 - a0; ++i)
                    if (d[a0 + i] != d[b0 + i]) return false;
            } break;
            case DT_BOOLEAN: {
                const uint8_t* d = static_cast<const uint8_t*>(r.data);
                if (((d[a >> 3] >> (a & 7)) & 1) != ((d[b >> 3] >> (b & 7)) & 1)) return false;
            } break;
            default:                                                  // fixed width: by bits
                if (dt_load(r.dtype, r.data, a) != dt_load(r.dtype, r.data, b)) return false;
                break;
        }
    }
    return true;
}

__global__ void __launch_bounds__(BLOCK)
wide_key_assign_kernel(const WideKeyCols K, const uint64_t* hashes, uint32_t* table, uint64_t mask, uint32_t n, uint32_t* rep) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        uint64_t slot = mix64(hashes[row]) & mask;
        for (;;) {
            uint32_t o = table[slot];
            if (o == 0) {
                o = atomicCAS(&table[slot], 0u, row + 1u);
                if (o == 0) { rep[row] = row; break; }
            }
            if (wide_keys_equal(K, o - 1u, row)) { rep[row] = o - 1u; break; }
            slot = (slot + 1) & mask;
        }
    }
}

// ---- the same for ONE Int64 / UInt64 key (TPC-H at SF1000: l_orderkey / o_orderkey are Int64): 16-byte slots
// {key, build row + 1}.  The build claims a slot with a 32-bit CAS on the row word and compares against the key
// COLUMN of the claiming row (immutable input), so no reader ever depends on a half-written slot; the key word is
// stored after the claim and is complete when the build kernel ends.  The probe reads a slot with one 16-byte load.
struct alignas(16) NarrowSlot64 { uint64_t key; uint32_t row1; uint32_t pad; };

__global__ void __launch_bounds__(BLOCK)
join_build_narrow64_kernel(NarrowJoinTable T, const uint64_t* keys, const uint64_t* sel, uint32_t n_left) {
    NarrowSlot64* slots = reinterpret_cast<NarrowSlot64*>(T.slots);
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_left; row += gridDim.x * BLOCK) {
        if (!bit_at(sel, row)) continue;
        const uint64_t key = keys[row];
        uint64_t slot = mix64(key) & T.mask;
        for (;;) {
            uint32_t r = slots[slot].row1;
            if (r == 0) {
                r = atomicCAS(&slots[slot].row1, 0u, row + 1u);
                if (r == 0) { slots[slot].key = key; break; }      // claimed
            }
            if (keys[r - 1u] == key) { *T.dup_flag = 1u; break; }
            slot = (slot + 1) & T.mask;
        }
    }
}

__global__ void __launch_bounds__(BLOCK)
join_unmatched_flags_kernel(const uint32_t* matched, uint32_t n_left, uint32_t* flags) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n_left; row += gridDim.x * BLOCK)
        flags[row] = ((matched[row >> 5] >> (row & 31)) & 1u) ? 0u : 1u;
}

__global__ void __launch_bounds__(BLOCK)
compact_flags_kernel(const uint32_t* flags, const uint64_t* offsets, uint32_t n, uint32_t* out) {
    for (uint32_t row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK)
        if (flags[row]) out[offsets[row]] = row;
}

static int grid_rows(const LaunchCfg& cfg, size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_join_build(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* sel, uint32_t n_left) {
    if (n_left == 0) return hipSuccess;
    hipLaunchKernelGGL(join_build_kernel, dim3(grid_rows(cfg, n_left)), dim3(BLOCK), 0, cfg.stream, T, sel, n_left);
    return hipGetLastError();
}
hipError_t launch_join_probe_count(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                   uint32_t n_right, bool right_outer, uint32_t* counts) {
    if (n_right == 0) return hipSuccess;
    hipLaunchKernelGGL(join_probe_count_kernel, dim3(grid_rows(cfg, n_right)), dim3(BLOCK), 0, cfg.stream, T, rkeys128, rsel,
                       n_right, right_outer ? 1 : 0, counts);
    return hipGetLastError();
}
hipError_t launch_join_probe_emit(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                  uint32_t n_right, bool right_outer, const uint64_t* offsets, uint32_t* left_idx,
                                  uint32_t* right_idx, uint32_t* matched) {
    if (n_right == 0) return hipSuccess;
    hipLaunchKernelGGL(join_probe_emit_kernel, dim3(grid_rows(cfg, n_right)), dim3(BLOCK), 0, cfg.stream, T, rkeys128, rsel,
                       n_right, right_outer ? 1 : 0, offsets, left_idx, right_idx, matched);
    return hipGetLastError();
}
hipError_t launch_join_probe_match(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                   uint32_t n_right, bool right_outer, uint32_t* partner, uint64_t* bitmap, uint32_t* tile_counts,
                                   uint32_t* matched) {
    if (n_right == 0) return hipSuccess;
    const size_t n_tiles = ((size_t)n_right + SEL_TILE - 1) / SEL_TILE;
    hipError_t e = hipMemsetAsync(tile_counts, 0, n_tiles * 4, cfg.stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(join_probe_match_kernel, dim3(grid_rows(cfg, n_right)), dim3(BLOCK), 0, cfg.stream, T, rkeys128, rsel,
                       n_right, right_outer ? 1 : 0, partner, bitmap, tile_counts, matched);
    return hipGetLastError();
}
hipError_t launch_join_build_narrow(const LaunchCfg& cfg, const NarrowJoinTable& T, const void* keys, int key_width,
                                    const uint64_t* sel, uint32_t n_left) {
    if (n_left == 0) return hipSuccess;
    if (key_width == 4)
        hipLaunchKernelGGL(join_build_narrow_kernel, dim3(grid_rows(cfg, n_left)), dim3(BLOCK), 0, cfg.stream, T, (const uint32_t*)keys, sel, n_left);
    else
        hipLaunchKernelGGL(join_build_narrow64_kernel, dim3(grid_rows(cfg, n_left)), dim3(BLOCK), 0, cfg.stream, T, (const uint64_t*)keys, sel, n_left);
    return hipGetLastError();
}
hipError_t launch_wide_key_assign(const LaunchCfg& cfg, const WideKeyCols& K, const uint64_t* hashes, uint32_t* table, uint64_t mask,
                                  uint32_t n, uint32_t* rep) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(wide_key_assign_kernel, dim3(grid_rows(cfg, n)), dim3(BLOCK), 0, cfg.stream, K, hashes, table, mask, n, rep);
    return hipGetLastError();
}
hipError_t launch_join_key_present64(const LaunchCfg& cfg, const uint64_t* keys, const uint64_t* sel, uint32_t n, uint64_t kmin,
                                     uint32_t* present) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(join_key_present64_kernel, dim3(grid_rows(cfg, n)), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, present);
    return hipGetLastError();
}
hipError_t launch_join_key_present(const LaunchCfg& cfg, const uint32_t* keys, const uint64_t* sel, uint32_t n, uint32_t kmin,
                                   uint32_t* present) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(join_key_present_kernel, dim3(grid_rows(cfg, n)), dim3(BLOCK), 0, cfg.stream, keys, sel, n, kmin, present);
    return hipGetLastError();
}
hipError_t launch_join_unmatched_flags(const LaunchCfg& cfg, const uint32_t* matched, uint32_t n_left, uint32_t* flags) {
    if (n_left == 0) return hipSuccess;
    hipLaunchKernelGGL(join_unmatched_flags_kernel, dim3(grid_rows(cfg, n_left)), dim3(BLOCK), 0, cfg.stream, matched, n_left, flags);
    return hipGetLastError();
}
hipError_t launch_compact_flags(const LaunchCfg& cfg, const uint32_t* flags, const uint64_t* offsets, uint32_t n, uint32_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_flags_kernel, dim3(grid_rows(cfg, n)), dim3(BLOCK), 0, cfg.stream, flags, offsets, n, out);
    return hipGetLastError();
}

template <bool NULLS>
static hipError_t launch_agg_hash_n(const LaunchCfg& cfg, const ScanParams& P, const HashAggTable& T, uint32_t row_base,
                                    ScanStatus* status) {
    const int r = choose_r(P.prog, 0);
    const size_t lds = r == 4 ? host_tile_bytes<4>(P.prog) : host_tile_bytes<2>(P.prog);
    const int64_t n_tiles = (P.n_rows + BLOCK * r - 1) / (BLOCK * r);
    const int grid = pick_grid(cfg, n_tiles, lds, 4);
    hipError_t e;
    if (r == 4) {
        auto k = scan_agg_hash_kernel<4, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, T, row_base, status);
    } else {
        auto k = scan_agg_hash_kernel<2, NULLS>;
        if ((e = set_lds(k, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, cfg.stream, P, T, row_base, status);
    }
    return hipGetLastError();
}

hipError_t launch_scan_agg_hash(const LaunchCfg& cfg, const ScanParams& P, const HashAggTable& T, uint32_t row_base,
                                ScanStatus* status) {
    return P.prog.nullable ? launch_agg_hash_n<true>(cfg, P, T, row_base, status)
                           : launch_agg_hash_n<false>(cfg, P, T, row_base, status);
}

static unsigned run_grid(const LaunchCfg& cfg, size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    if (g > (size_t)cfg.device_cus * 16) g = (size_t)cfg.device_cus * 16;
    return (unsigned)(g < 1 ? 1 : g);
}
hipError_t launch_run_heads(const LaunchCfg& cfg, const uint64_t* keys128, uint32_t n, uint32_t* flags, uint64_t first_mask, uint64_t* not_ascending) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(run_heads_kernel, dim3(run_grid(cfg, n)), dim3(BLOCK), 0, cfg.stream, keys128, n, flags, first_mask, (unsigned long long*)not_ascending);
    return hipGetLastError();
}
hipError_t launch_run_tail_resolve(const LaunchCfg& cfg, const uint64_t* keys128, const uint32_t* head, const uint64_t* n_runs_dev, const uint32_t* rowslot,
                                   uint32_t n_rows, uint32_t split_row, uint64_t first_mask, uint32_t* head2, uint32_t* match, uint32_t* fresh) {
    hipLaunchKernelGGL(run_tail_resolve_kernel, dim3(run_grid(cfg, n_rows)), dim3(BLOCK), 0, cfg.stream, keys128, head, n_runs_dev, rowslot, split_row, first_mask,
                       head2, match, fresh);
    return hipGetLastError();
}
hipError_t launch_run_tail_remap(const LaunchCfg& cfg, const uint32_t* head, const uint64_t* n_runs_dev, uint32_t n_rows, uint32_t split_row, const uint32_t* match,
                                 const uint32_t* fresh_before, uint32_t* rowslot, uint32_t* head2, uint64_t* n_groups_out) {
    hipLaunchKernelGGL(run_tail_remap_kernel, dim3(run_grid(cfg, n_rows - split_row)), dim3(BLOCK), 0, cfg.stream, head, n_runs_dev, n_rows, split_row, match,
                       fresh_before, rowslot, head2, (unsigned long long*)n_groups_out);
    return hipGetLastError();
}
hipError_t launch_run_compact(const LaunchCfg& cfg, const HashAggTable& T, const uint32_t* head, uint32_t n_runs, bool nulls, GroupRec* out) {
    if (n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(run_compact_kernel, dim3(run_grid(cfg, n_runs)), dim3(BLOCK), 0, cfg.stream, T, head, n_runs, nulls ? 1 : 0, out);
    return hipGetLastError();
}
hipError_t launch_run_slots(const LaunchCfg& cfg, const uint32_t* flags, const uint32_t* runs_before, uint32_t n, uint32_t* rowslot, uint32_t* head) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(run_slots_kernel, dim3(run_grid(cfg, n)), dim3(BLOCK), 0, cfg.stream, flags, runs_before, n, rowslot, head);
    return hipGetLastError();
}
hipError_t launch_run_groups(const LaunchCfg& cfg, const uint64_t* keys128, uint32_t n, const uint32_t* head, const uint64_t* n_runs_dev, uint32_t* table,
                             uint64_t mask, uint32_t* min_head, uint32_t* slot_of_run, uint32_t* winner, uint32_t* owner, uint32_t* rowslot) {
    if (n == 0) return hipSuccess;
    const unsigned g = run_grid(cfg, n);                         // the run count is only known on the device: sized for n, bounded by it there
    hipLaunchKernelGGL(run_claim_kernel, dim3(g), dim3(BLOCK), 0, cfg.stream, keys128, head, n_runs_dev, table, mask, min_head, slot_of_run);
    hipLaunchKernelGGL(run_winner_kernel, dim3(g), dim3(BLOCK), 0, cfg.stream, head, n_runs_dev, min_head, slot_of_run, rowslot, winner, owner);
    hipLaunchKernelGGL(run_remap_kernel, dim3(g), dim3(BLOCK), 0, cfg.stream, winner, n, rowslot);
    return hipGetLastError();
}

static int grid_n(const LaunchCfg& cfg, size_t n) {
    size_t g = (n + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_hash_agg_init(const LaunchCfg& cfg, const HashAggTable& T, const MergeAccKinds& kinds) {
    hipLaunchKernelGGL(hash_agg_init_kernel, dim3(grid_n(cfg, (size_t)(T.mask + 1) * T.n_acc)), dim3(BLOCK), 0, cfg.stream, T, kinds);
    return hipGetLastError();
}
hipError_t launch_hash_agg_flags(const LaunchCfg& cfg, const HashAggTable& T, uint32_t* flags) {
    hipLaunchKernelGGL(hash_agg_flags_kernel, dim3(grid_n(cfg, (size_t)(T.mask + 1))), dim3(BLOCK), 0, cfg.stream, T, flags);
    return hipGetLastError();
}
hipError_t launch_hash_agg_compact(const LaunchCfg& cfg, const HashAggTable& T, const uint64_t* dense_index, bool nulls,
                                   GroupRec* out) {
    hipLaunchKernelGGL(hash_agg_compact_kernel, dim3(grid_n(cfg, (size_t)(T.mask + 1))), dim3(BLOCK), 0, cfg.stream, T,
                       dense_index, nulls ? 1 : 0, out);
    return hipGetLastError();
}

}  // namespace bhip
