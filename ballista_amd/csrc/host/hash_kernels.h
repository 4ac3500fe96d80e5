// hash_kernels.h — launchers of kernels_hash.hip (device-wide hash tables: high-cardinality
// aggregation, join build/probe) — internal C++ interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../kernels.h"

namespace bhip {

// ---- aggregation ------------------------------------------------------------------------------
struct HashAggTable {
    uint32_t* owner;          // [capacity] 0 = empty, else id+1 of the row whose key defines the slot
    uint64_t mask;            // capacity - 1 (capacity is a power of two)
    const uint64_t* keys128;  // packed keys of every input row (all batches, 2 x u64 per row)
    uint64_t* acc;            // [capacity][n_acc]
    uint64_t* nvalid;         // [capacity][n_acc]   (only with NULLs)
    uint64_t* rows;           // [capacity]
    int32_t n_acc;
    int32_t n_fsum;           // SUM(Float64) accumulators summed in a fixed order (kernels_dagg.hip); 0: all accumulators are atomics
    // fixed-order Float64 sums: the scan records every row's slot and addends instead of adding atomically
    uint32_t* rowslot;        // [total rows] slot of the row's group, 0xFFFFFFFF = the row reaches no group
    double* fvals;            // [n_fsum][total rows]
    uint64_t total_rows;
    uint8_t fsum_of_acc[VM_MAX_ACC];   // accumulator -> index into fvals, 0xFF = an atomic accumulator
    int32_t slots_given;      // 1: rowslot is an INPUT (clustered input: slot = run of equal keys, launch_run_*); owner / mask: run table
};
struct MergeAccKinds { uint8_t kind[VM_MAX_ACC]; };

// ---- Float64 sums in a fixed order (kernels_dagg.hip) ------------------------------------------------------------------------
// Atomic adds make a SUM(Float64) depend on the scheduling of the waves; here every group's addends are combined in ROW ORDER:
//   segments  one wave per 1024-row tile: runs of consecutive rows with the same slot are summed by a segmented scan (a fixed
//             tree), a run that continues into the next 64-row chunk carries its sum along; every run end leaves a record
//             (slot, first row, sums) in the tile's staging area and bumps the slot's run count (an integer atomic);
//   apply     a slot with ONE run takes its sums with a plain store; the runs of the others go to a spill list;
//   spill     per group a linked list of its runs: a few runs are ordered by first row in registers and combined left to right
//             by the thread of the earliest run; only when some group has more than DET_LIST_MAX runs is the whole list sorted
//             by (slot, first row) and combined one thread per slot.
// Input clustered by group (lineitem by order key: Q3) spills only the runs that straddle a tile; unclustered input spills
// everything and pays a sort — still the same sums, run to run and whatever the grid.
struct DetSum {
    const uint32_t* rowslot;
    const double* fvals;
    uint64_t total_rows;
    int32_t n_fsum;
    int32_t n_acc;
    uint8_t acc_of_fsum[VM_MAX_ACC];
    // staging, [n_tiles * 1024] each
    uint32_t* seg_slot;
    uint32_t* seg_first;
    double* seg_sum;          // [n_fsum][n_tiles * 1024]
    uint32_t* tile_nseg;      // [n_tiles]
    uint32_t* runs;           // [capacity] runs per slot
    uint64_t* acc;            // the table's accumulators [capacity][n_acc]
    uint64_t* rows;           // the table's row counts [capacity]: every run adds its length (one atomic per run, not per row)
    // spill
    uint64_t* spill_key;      // (slot << 32) | first row
    uint32_t* spill_seg;      // index of the run in the staging area
    uint32_t* spill_count;    // [0] entries of the spill list; [1] != 0: some group has more runs than the list walk holds
    uint32_t* spill_next;     // per spill entry: the group's previous entry (a linked list per group, in arrival order), NONE ends it
    uint32_t* spill_head;     // [capacity] last entry of the group's list, NONE = none
};
hipError_t launch_det_segments(const LaunchCfg& cfg, const DetSum& D);
hipError_t launch_det_apply(const LaunchCfg& cfg, const DetSum& D);
// groups with a handful of runs (a clustered input's runs straddling a tile): every group's runs are found through its list,
// ordered by first row in registers and added up left to right — no sort, no host round trip; groups with more than
// DET_LIST_MAX runs raise spill_count[1] and are left to the sorted combine below
constexpr int DET_LIST_MAX = 16;
hipError_t launch_det_spill_lists(const LaunchCfg& cfg, const DetSum& D);
// sorted_key / sorted_seg: the spill list after a stable sort by key
hipError_t launch_det_spill_combine(const LaunchCfg& cfg, const DetSum& D, const uint64_t* sorted_key, const uint32_t* sorted_seg, uint32_t n_spill);

hipError_t launch_scan_agg_hash(const LaunchCfg& cfg, const ScanParams& P, const HashAggTable& T, uint32_t row_base,
                                ScanStatus* status);
// clustered input (kernels_hash.hip "input clustered by group key"): flags[i] = row i starts a run of equal keys;
// runs_before = exclusive scan of flags -> rowslot[i] = run of row i, head[run] = its first row;
// launch_run_groups: rowslot[i] = the run that owns row i's key, owner[run] = head row + 1 for owning runs, 0 otherwise
// (table: 2 x n entries zeroed, min_head: as many entries set to 0xFFFFFFFF; *n_runs_dev = the scan's total)
// not_ascending[0] (zeroed by the caller) counts the places where the first key part (k0 & first_mask) does not strictly increase
// from one run to the next, not_ascending[1] (set to ~0 by the caller) = the first such row.  0 places: every run is a distinct group
// (launch_run_compact instead of launch_run_groups + flags + scan + compact); 1 place: launch_run_tail_* (kernels_hash.hip)
hipError_t launch_run_tail_resolve(const LaunchCfg& cfg, const uint64_t* keys128, const uint32_t* head, const uint64_t* n_runs_dev, const uint32_t* rowslot,
                                   uint32_t n_rows, uint32_t split_row, uint64_t first_mask, uint32_t* head2, uint32_t* match, uint32_t* fresh);
hipError_t launch_run_tail_remap(const LaunchCfg& cfg, const uint32_t* head, const uint64_t* n_runs_dev, uint32_t n_rows, uint32_t split_row, const uint32_t* match,
                                 const uint32_t* fresh_before, uint32_t* rowslot, uint32_t* head2, uint64_t* n_groups_out);
hipError_t launch_run_heads(const LaunchCfg& cfg, const uint64_t* keys128, uint32_t n, uint32_t* flags, uint64_t first_mask, uint64_t* not_ascending);
hipError_t launch_run_compact(const LaunchCfg& cfg, const HashAggTable& T, const uint32_t* head, uint32_t n_runs, bool nulls, GroupRec* out);
hipError_t launch_run_slots(const LaunchCfg& cfg, const uint32_t* flags, const uint32_t* runs_before, uint32_t n, uint32_t* rowslot, uint32_t* head);
hipError_t launch_run_groups(const LaunchCfg& cfg, const uint64_t* keys128, uint32_t n, const uint32_t* head, const uint64_t* n_runs_dev, uint32_t* table,
                             uint64_t mask, uint32_t* min_head, uint32_t* slot_of_run, uint32_t* winner, uint32_t* owner, uint32_t* rowslot);
hipError_t launch_hash_agg_init(const LaunchCfg& cfg, const HashAggTable& T, const MergeAccKinds& kinds);
hipError_t launch_hash_agg_flags(const LaunchCfg& cfg, const HashAggTable& T, uint32_t* flags);
hipError_t launch_hash_agg_compact(const LaunchCfg& cfg, const HashAggTable& T, const uint64_t* dense_index, bool nulls,
                                   GroupRec* out);

// group keys of any width: rep[row] = a row of the same batch whose key columns equal row's (NULL == NULL);
// table = zeroed [mask + 1] words, mask + 1 a power of two >= 2 n; hashes = 64-bit row hashes of the key columns
struct WideKeyCols { ColumnRef col[VM_MAX_COLS]; int32_t n; int32_t pad; };
hipError_t launch_wide_key_assign(const LaunchCfg& cfg, const WideKeyCols& K, const uint64_t* hashes, uint32_t* table, uint64_t mask,
                                  uint32_t n, uint32_t* rep);

// ---- join -------------------------------------------------------------------------------------
struct JoinTable {
    uint64_t* owner;          // [capacity] (id+1 of the build row whose key defines the slot) | (high half of the key hash) << 32
    uint32_t* head;           // [capacity] id+1 of the most recently inserted build row of the slot's key
    uint32_t* next;           // [n_left]   id+1 of the next build row with the same key
    uint64_t mask;
    const uint64_t* keys128;  // packed keys of the build rows
    uint32_t* dup_flag;       // set by the build when two build rows share a key (may be null)
};
hipError_t launch_join_build(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* sel, uint32_t n_left);
hipError_t launch_join_probe_count(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                   uint32_t n_right, bool right_outer, uint32_t* counts);
hipError_t launch_join_probe_emit(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                  uint32_t n_right, bool right_outer, const uint64_t* offsets, uint32_t* left_idx,
                                  uint32_t* right_idx, uint32_t* matched);
// unique build keys: one table probe per row -> partner[] (build row id, 0xFFFFFFFF = none) + the selection bitmap of
// emitting rows and its counts per SEL_TILE rows (the inputs of launch_select_indices)
hipError_t launch_join_probe_match(const LaunchCfg& cfg, const JoinTable& T, const uint64_t* rkeys128, const uint64_t* rsel,
                                   uint32_t n_right, bool right_outer, uint32_t* partner, uint64_t* bitmap, uint32_t* tile_counts,
                                   uint32_t* matched);
// one Int32 / Date32 key column, unique build side: key and build row share the slot (kernels_hash.hip)
struct NarrowJoinTable {
    // CAS table (sparse keys): key width 4: [capacity] key | (build row + 1) << 32, 0 = empty;
    // key width 8: [capacity] {key, build row + 1 (low half of the second word)}, 16 bytes per slot.  Null in rank mode.
    uint64_t* slots;
    uint64_t mask;
    uint32_t* dup_flag;       // set when two build rows share a key: the host falls back to JoinTable
    // optional with the CAS table: the exact set of build keys, one bit per value of [kmin64, kmin64 + krange64] (null: absent)
    const uint32_t* present;
    uint32_t kmin;
    uint64_t krange64;        // last offset of the window: keys in [kmin64, kmin64 + krange64] (rank map: <= 2^36; key-set bitmap: <= 2^30)
    uint64_t kmin64;          // first key of the window as raw key bits (32-bit keys: zero- or sign-extension does not matter, the
                              // offset is taken modulo 2^32)
    // rank map (kernels_join.hip; keys inside a window of <= 2^36 values — TPC-H SF1000 order keys reach 6 x 10^9): rpack[g] = the key set of granule g (32 key values,
    // low half) | the number of build keys before the granule (high half): ONE 8-byte read gives a probe row its membership bit and
    // its rank; rank -> build row through rperm (null: the build side is sorted by key, rank = row)
    const uint64_t* rpack;
    // the key-set words alone, rbits[g] = low half of rpack[g] (null: not kept — a window beyond 2^31 values): what a semi-join reads
    const uint32_t* rbits;
    uint32_t scalar_map;      // probe kernel: slots whose rows fall into two neighbouring granules read them through the scalar cache
    uint32_t rzero;           // index of an all-zero granule behind the map (rows that need no lookup read it)
    const uint32_t* rperm;
    // two-column join whose build side is unique on the first column: second key of every build row (null: one-column join)
    const uint32_t* resid_build;
};
// one pass over the build keys (kernels_join.hip): stats[0] / [1] = min / max of (key ^ sign bit) as unsigned (seed ~0 / 0),
// stats[2] != 0 when the keys are not strictly increasing (or some are NULL)
hipError_t launch_join_key_stats(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t* stats);
// at most tiny_rank_build_max_rows() build keys: statistics AND — when the keys span at most 2^16 values — the packed map (rpack:
// tiny_rank_build_map_words() words) and the rank -> row permutation (rperm, n entries; may be null) in ONE launch.
// out[0] / out[1] = min / max of (key ^ sign bit) as for launch_join_key_stats, out[2] = unsorted | duplicate keys << 1 | map built << 2
int tiny_rank_build_max_rows();
size_t tiny_rank_build_map_words();
hipError_t launch_tiny_rank_build(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t* rpack, uint32_t* rperm,
                                  uint64_t* out);
// rank map build: key-set bits (zeroed by the caller; 64 keys per word = two granules), then launch_rank_pack (util_kernels.h), perm
hipError_t launch_rank_bits(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t kmin, bool sorted,
                            uint64_t* bits, uint32_t* dup_flag);
hipError_t launch_rank_perm(const LaunchCfg& cfg, const void* keys, int key_width, const uint64_t* sel, uint32_t n, uint64_t kmin,
                            const uint64_t* rpack, uint32_t* perm);
// the key-set bitmap in front of the CAS table (round-1 design; kept as the A/B partner of the rank map, BHIP_JOIN_TABLE=1)
hipError_t launch_join_key_present64(const LaunchCfg& cfg, const uint64_t* keys, const uint64_t* sel, uint32_t n, uint64_t kmin,
                                     uint32_t* present);
hipError_t launch_join_key_present(const LaunchCfg& cfg, const uint32_t* keys, const uint64_t* sel, uint32_t n, uint32_t kmin,
                                   uint32_t* present);
hipError_t launch_join_build_narrow(const LaunchCfg& cfg, const NarrowJoinTable& T, const void* keys, int key_width,
                                    const uint64_t* sel, uint32_t n_left);
// the probe side as one pass over the UNFILTERED batch (kernels_join.hip): AND of integer ranges over NULL-free
// Int32 / Date32 columns (n = 0: no filter) -> key-set bitmap -> table.  One wave owns a whole 1024-row tile: tile_counts[] are
// plain stores (no memset needed), staging[tile * 1024 + j] = build row of the tile's j-th emitted row (staging may be null:
// a semi-join needs no partners).  Then: exclusive scan of tile_counts, launch_select_indices for the row indices,
// launch_join_compact_staged for the partners.
constexpr int JOIN_FILTER_MAX = 3;
struct ProbeFilter {
    int32_t n;
    int32_t lo[JOIN_FILTER_MAX], hi[JOIN_FILTER_MAX];
    const int32_t* col[JOIN_FILTER_MAX];
};
hipError_t launch_join_filter_probe(const LaunchCfg& cfg, const NarrowJoinTable& T, const ProbeFilter& F, const void* rkeys, int key_width,
                                    const uint64_t* rsel, uint32_t n_right, bool right_outer, uint64_t* bitmap, uint32_t* tile_counts,
                                    uint32_t* staging, uint32_t* matched, const uint32_t* resid_probe = nullptr, uint32_t* staging_rows = nullptr);
hipError_t launch_and_bitmaps(const LaunchCfg& cfg, const uint64_t* a, const uint64_t* b, int64_t n_bits, uint64_t* out);
// out[i] = a[i] << 32 | b[i] (4-byte integer columns); validity_out (when not null) = va & vb (a null input bitmap = all valid)
hipError_t launch_pack_key_pair(const LaunchCfg& cfg, const void* a, const void* b, const uint64_t* va, const uint64_t* vb, int64_t n, uint64_t* out,
                                uint64_t* validity_out);
hipError_t launch_join_compact_staged(const LaunchCfg& cfg, const uint32_t* staging, const uint64_t* tile_off, uint64_t total,
                                      int64_t n_tiles, uint32_t* out, const uint32_t* staging2 = nullptr, uint32_t* out2 = nullptr);
// ---- radix-partitioned join with LDS-resident tables (kernels_radix_join.hip): the measured alternative, BHIP_JOIN_RADIX=1 ----
// sort keys (partition id << 32) | key + row ids; partition bounds of a sorted side; the per-partition LDS build + probe.
// flags[0] != 0: a build partition outgrew the LDS table; flags[1] != 0: duplicate build keys.
hipError_t launch_radix_join_keys(const LaunchCfg& cfg, const uint32_t* keys, uint32_t n, int log2p, uint64_t* out_keys, uint32_t* out_rows);
hipError_t launch_radix_join_bounds(const LaunchCfg& cfg, const uint64_t* sorted, uint32_t n, uint32_t n_parts, uint32_t* first);
hipError_t launch_radix_join_lds(const LaunchCfg& cfg, const uint64_t* bkeys, const uint32_t* brows, const uint32_t* bfirst, const uint64_t* pkeys,
                                 const uint32_t* prows, const uint32_t* pfirst, uint32_t n_parts, uint32_t* partner, uint64_t* bitmap,
                                 uint32_t* tile_counts, uint32_t* flags);
hipError_t launch_join_unmatched_flags(const LaunchCfg& cfg, const uint32_t* matched, uint32_t n_left, uint32_t* flags);
hipError_t launch_compact_flags(const LaunchCfg& cfg, const uint32_t* flags, const uint64_t* offsets, uint32_t n, uint32_t* out);

}  // namespace bhip
