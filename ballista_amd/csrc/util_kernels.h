// util_kernels.h — launchers of kernels_util.hip / kernels_gen.hip / kernels_hash.hip /
// kernels_sort.hip (internal C++ interface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

// ---- prefix sums ------------------------------------------------------------------------
size_t exclusive_scan_temp_bytes(int64_t n);
// rank map of the narrow join: pack[g] = bits32[g] | (set bits before granule g) << 32; temp: exclusive_scan_temp_bytes(n)
hipError_t launch_rank_pack(hipStream_t st, const uint32_t* bits32, int64_t n, uint64_t* pack, uint64_t* total_out, void* temp);
// out[i] = sum in[0..i) ; write_total: also out[n] = grand total ; total_out (device u64) optional
hipError_t exclusive_scan_u32_u64(hipStream_t st, const uint32_t* in, int64_t n, uint64_t* out, bool write_total,
                                  uint64_t* total_out, void* temp);
hipError_t exclusive_scan_u32_i32(hipStream_t st, const uint32_t* in, int64_t n, int32_t* out, bool write_total,
                                  uint64_t* total_out, void* temp);
hipError_t exclusive_scan_u32_u32(hipStream_t st, const uint32_t* in, int64_t n, uint32_t* out, bool write_total,
                                  uint64_t* total_out, void* temp);

// ---- selection ---------------------------------------------------------------------------
hipError_t launch_select_indices(const LaunchCfg& cfg, const uint64_t* bitmap, const uint64_t* tile_offsets,
                                 int64_t n_rows, uint32_t* indices);

// ---- take ----------------------------------------------------------------------------------
// up to TAKE_MANY_MAX gathers by the same index vector in one launch (width 1 / 4 / 8 bytes, 0 = bitmap)
constexpr int TAKE_MANY_MAX = 32;
struct TakeMany {
    int32_t n;
    const void* src[TAKE_MANY_MAX];
    void* dst[TAKE_MANY_MAX];
    int32_t width[TAKE_MANY_MAX];
};
hipError_t launch_take_many(const LaunchCfg& cfg, const TakeMany& d, const uint32_t* idx, int64_t n);
hipError_t launch_take_fixed(const LaunchCfg& cfg, const void* src, int width, const uint32_t* idx, int64_t n, void* dst);
hipError_t launch_take_bitmap(const LaunchCfg& cfg, const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst);
hipError_t launch_take_utf8_lengths(const LaunchCfg& cfg, const int32_t* offsets, const uint32_t* idx, int64_t n,
                                    uint32_t* lengths);
hipError_t launch_narrow_i32(const LaunchCfg& cfg, const int32_t* src, int64_t n, int width, void* dst);
hipError_t launch_compose_indices(const LaunchCfg& cfg, const uint32_t* inner, const uint32_t* idx, int64_t n, uint32_t* out);
hipError_t launch_take_utf8_copy(const LaunchCfg& cfg, const int32_t* src_off, const uint8_t* src, int64_t src_bytes, const uint32_t* idx,
                                 int64_t n, const int32_t* dst_off, uint8_t* dst);

hipError_t launch_iota_u32(const LaunchCfg& cfg, uint32_t* out, int64_t n, uint32_t start);
// up to FILL_MANY_MAX regions (4-byte aligned, a multiple of 4 bytes long), each set to its own 32-bit pattern, one launch
constexpr int FILL_MANY_MAX = 8;
struct FillMany {
    int n = 0;
    void* ptr[FILL_MANY_MAX];
    uint64_t bytes[FILL_MANY_MAX];
    uint32_t value[FILL_MANY_MAX];
    void add(void* p, uint64_t nbytes, uint32_t v = 0) { ptr[n] = p; bytes[n] = nbytes; value[n] = v; ++n; }
};
hipError_t launch_fill_many(const LaunchCfg& cfg, const FillMany& F);
hipError_t launch_rebase_offsets(const LaunchCfg& cfg, const int32_t* src_off, int64_t n_plus_1, int32_t add, int32_t* dst_off);
hipError_t launch_copy_bits(const LaunchCfg& cfg, const uint64_t* src, int64_t src_bit0, uint64_t* dst, int64_t dst_bit0,
                            int64_t n_bits);

// ---- group table -> columns ----------------------------------------------------------------
struct EmitKeySpec {
    int32_t pos;        // byte position of the part in the packed key
    int32_t width;      // bytes (incl. null byte / length byte)
    int32_t nullable;
    int32_t dtype;
};
enum EmitValueKind : int32_t {
    EMIT_VALUE = 0,     // acc[a], NULL when no input reached it (SUM / MIN / MAX)
    EMIT_COUNT = 1,     // number of non-NULL inputs of acc[a] (COUNT(x), AVG count)
    EMIT_ROWS = 2,      // rows of the group (COUNT(*))
    EMIT_RAW = 3,       // acc[a] as is (merged counts in Final mode)
    EMIT_AVG = 4,       // acc[a] / count(a)
    EMIT_AVG_ACC = 5    // acc[a] / acc[b]   (Final mode: merged sum / merged count)
};
struct EmitValueSpec {
    int32_t kind;
    int32_t acc_a, acc_b;
    int32_t count_is_rows;   // program has no NULLs: count(a) == rows
    int32_t dtype;
};
// packed-key image of one NULL-free Int32 / Date32 / Int64 / UInt64 column (width 4 or 8)
hipError_t launch_widen_key(const LaunchCfg& cfg, const void* src, int width, int64_t n, uint64_t* keys128);

// up to PACK_MAX small device buffers -> one contiguous block (offsets chosen by the host)
constexpr int PACK_MAX = 64;
struct PackDesc {
    int32_t n;
    const void* src[PACK_MAX];
    uint32_t bytes[PACK_MAX];
    uint32_t dst[PACK_MAX];
};
// ---- Parquet value decode (host/parquet.cpp) ---------------------------------------------------------------------------------
// one run of the RLE / bit-packed hybrid (the host parses the run headers, the device expands the runs)
struct PqRun {
    uint32_t out_start;     // index of the run's first value in the page
    uint32_t count;
    uint32_t packed;        // > 0: bit-packed at `packed` bits per value, `value` = byte offset of the run's bits in the page bytes; 0: RLE, `value` = the repeated value
    uint32_t value;
};
// out[i] = i-th value of the run table (one thread per value: binary search of its run, bit extraction at the run's own width — the
// pages of a column chunk go through ONE launch even though the index width grows with the dictionary); values >= limit
// (a dictionary index outside the dictionary: a corrupt page) are clamped to 0xFFFFFFFF
hipError_t launch_pq_expand_runs(const LaunchCfg& cfg, const PqRun* runs, uint32_t n_runs, const uint8_t* page, int bit_width, uint32_t n_values,
                                 uint32_t limit, uint32_t* out);
// dense values -> rows: dst[i] = valid(i) ? src[rank of i among the valid rows] : (null_index ? 0xFFFFFFFF : 0); width 4 or 8
hipError_t launch_pq_scatter_valid(const LaunchCfg& cfg, const uint64_t* validity, const uint32_t* word_prefix, const void* src, int width,
                                   int64_t n, void* dst, int null_index);

// one-thread kernel: n_words 4-byte words from src to the slot's payload (byte 8 on), then its first 8 bytes = seq with system-scope release (pinned host slot)
hipError_t launch_publish(hipStream_t stream, const void* src, int n_words, void* slot, uint64_t seq);
hipError_t launch_pack_buffers(const LaunchCfg& cfg, const PackDesc& d, uint8_t* out);

constexpr int EMIT_BATCH_MAX = 16;
struct EmitValueBatch {
    int32_t n;
    EmitValueSpec spec[EMIT_BATCH_MAX];
    void* data[EMIT_BATCH_MAX];
    uint64_t* validity[EMIT_BATCH_MAX];
};
// dev_n != nullptr (all three launchers below): the group count is read on the device from dev_n->n_groups and `n_groups` only
// bounds it (the size the outputs were allocated for) — the host queues the emit before it has read the count back
hipError_t launch_emit_group_values(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitValueBatch& batch,
                                    const ScanStatus* dev_n = nullptr);
hipError_t launch_emit_group_key(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                 void* data, uint64_t* validity, uint32_t* utf8_lengths, const ScanStatus* dev_n = nullptr);
// a whole group table of at most EMIT_ALL_MAX_GROUPS groups -> every output column, ONE launch of one workgroup (thread = group);
// the count is read on the device (status->n_groups): the launch is queued right behind the merge (host/ops_agg.cpp)
constexpr int EMIT_ALL_MAX_GROUPS = 1024, EMIT_ALL_MAX_KEYS = 8, EMIT_ALL_MAX_VALUES = 24;
struct EmitAllArgs {
    const GroupRec* table;
    const ScanStatus* status;
    int32_t n_keys, n_values;
    EmitKeySpec key[EMIT_ALL_MAX_KEYS];
    void* key_data[EMIT_ALL_MAX_KEYS];
    int32_t* key_offsets[EMIT_ALL_MAX_KEYS];      // Utf8 keys
    uint64_t* key_validity[EMIT_ALL_MAX_KEYS];
    uint64_t* key_total[EMIT_ALL_MAX_KEYS];       // Utf8 keys: value bytes written
    EmitValueSpec value[EMIT_ALL_MAX_VALUES];
    void* value_data[EMIT_ALL_MAX_VALUES];
    uint64_t* value_validity[EMIT_ALL_MAX_VALUES];
};
hipError_t launch_emit_all(const LaunchCfg& cfg, const EmitAllArgs& A);
// the packed 16-byte key of every row from plain NULL-free integer / date columns: the low `width` bytes of part p at byte `pos` (the
// layout the expression VM packs: vm_device.h key_put) — a streaming pass with a few hundred bytes of code where the VM kernel's launch
// alone costs ~0.06 ms of instruction fetch (Q3's aggregate keys: three Int32 / Date32 columns of 3.2 M rows)
constexpr int FIXED_KEY_PARTS_MAX = 8;
struct FixedKeyParts { int32_t n; const void* src[FIXED_KEY_PARTS_MAX]; uint8_t width[FIXED_KEY_PARTS_MAX], pos[FIXED_KEY_PARTS_MAX]; };
hipError_t launch_pack_fixed_keys(const LaunchCfg& cfg, const FixedKeyParts& K, int64_t n, uint64_t* keys128);
// The same columns for ANY number of groups straight from the run slots of a clustered hash aggregate (host/ops_agg.cpp: distinct
// runs — slot g IS group g): key of the run's first row, accumulators of slot g; no GroupRec table is written and read back
// (Q3's 1.13 M groups: run_compact + four emit launches were 0.22 ms).  Fixed-width, non-Boolean keys only; A.table / A.status unused.
struct SlotSource {
    const uint64_t* keys128;   // packed keys of every input row
    const uint32_t* head;      // first row of run g
    const uint64_t* acc;       // [groups][n_acc]
    const uint64_t* nvalid;    // [groups][n_acc], or null: every accumulator saw every row of its group
    const uint64_t* rows;      // [groups]
    int32_t n_acc;
    int32_t valid;             // host side: this source is set
};
hipError_t launch_emit_slots(const LaunchCfg& cfg, const SlotSource& S, int64_t n_groups, const EmitAllArgs& A);

// n_groups <= EMIT_UTF8_SMALL_MAX: lengths + prefix sum + offsets (n + 1) + bytes + validity + byte total in one launch
constexpr int64_t EMIT_UTF8_SMALL_MAX = 4096;
hipError_t launch_emit_group_utf8_small(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                        uint64_t* validity, int32_t* offsets, uint8_t* bytes, uint64_t* total_out,
                                        const ScanStatus* dev_n = nullptr);
hipError_t launch_emit_group_utf8(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups, const EmitKeySpec& spec,
                                  const int32_t* offsets, uint8_t* bytes);
hipError_t launch_emit_group_value(const LaunchCfg& cfg, const GroupRec* table, int64_t n_groups,
                                   const EmitValueSpec& spec, void* data, uint64_t* validity);

// ---- synthetic TPC-H generator (kernels_gen.hip) -------------------------------------------
struct GenLineitemOut {
    int32_t* l_orderkey; int64_t* l_orderkey_i64; int32_t* l_suppkey;
    double* l_quantity; double* l_extendedprice; double* l_discount; double* l_tax;
    int32_t* l_shipdate; int32_t* l_commitdate; int32_t* l_receiptdate;
    int32_t* flag_off; uint8_t* flag_data; int32_t* status_off; uint8_t* status_data;
};
// order key of order index `ord` (0-based): key_base + 1 + (sparse ? dbgen's layout — the low 3 bits kept, the rest shifted up by
// two: 8 of every 32 values used, SF1000 keys reach 6 x 10^9 — : ord)
struct GenKeyLayout { int64_t key_base; int32_t sparse; };
hipError_t launch_gen_lineitem(const LaunchCfg& cfg, uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_orders,
                               uint64_t n_parts, uint64_t n_supp, const GenLineitemOut& out, GenKeyLayout keys = GenKeyLayout{0, 0});
struct GenOrdersOut {
    int32_t* o_orderkey; int64_t* o_orderkey_i64; int32_t* o_custkey; int32_t* o_orderdate; int32_t* o_shippriority;
};
hipError_t launch_gen_orders(const LaunchCfg& cfg, uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_cust,
                             const GenOrdersOut& out, GenKeyLayout keys = GenKeyLayout{0, 0});

}  // namespace bhip
