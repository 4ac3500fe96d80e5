// sop.h — plan table of the register-resident aggregate fast path (sop_kernel.h); built by
// host/sop.cpp from the same expressions the VM program is built from.
//
// The table is branch-free by construction — every entry is data for straight-line code:
//   predicate : AND of ranges   lo <= (double)x <= hi      (one range per column: comparisons on the
//               same column are intersected; strict bounds move to the neighbouring double / integer)
//   chain step: f = sgn * x + add ;  t = (start ? 1.0 : t_prev) * f
//               (column: sgn 1, add -0.0 | lit - col: -1, lit | lit + col: 1, lit | col - lit: 1, -lit |
//                literal: no column, sgn 0, add lit) — each is ONE correctly rounded add of the same two
//               operands the reference adds, multiplication by +-1 / 1.0 is exact
//   keys      : part 0 -> key word 0, part 1 -> key word 1, part 2 -> high half of word 1 (32-bit parts)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

constexpr int SOP_NRANGE = 4;     // predicate ranges (distinct columns)
constexpr int SOP_NKEY = 3;       // group-key columns
constexpr int SOP_NSTEP = 8;      // chain steps (= accumulators)
constexpr int SOP_NCOL = 16;

struct SopColumn {
    const void* data;
    const int32_t* offsets;
    const uint64_t* validity;   // only a column the predicate constrains may carry one (lean_kernel.h, kernels_range.hip)
    int32_t dtype;
    int32_t data_bytes;
};

struct SopRange {
    uint8_t col;        // index into SopProgram::cols (Float64, Int32 or Date32 column)
    uint8_t is32;       // 1: 32-bit integer column (converted exactly to double)
    uint8_t pad[6];
    double lo, hi;
};

struct SopStep {
    uint8_t col;
    uint8_t is32;       // 1: Int32/Date32 column under CAST(... AS Float64)
    uint8_t has_col;    // 0: literal factor
    uint8_t start;      // 1: starts a new chain
    uint8_t acc;        // accumulator (GroupRec::acc index) that sums t, 0xFF = none
    uint8_t pad[3];
    double sgn, add;
};

enum SopKeyKind : uint8_t { SOP_KEY_I32 = 0, SOP_KEY_I64 = 1, SOP_KEY_UTF8 = 2 };
struct SopKey {
    uint8_t col;
    uint8_t kind;       // SopKeyKind
    uint8_t pad[2];
};

struct SopProgram {
    int64_t n_rows;
    int32_t n_cols, n_ranges, n_keys, n_steps;
    SopColumn cols[SOP_NCOL];
    SopRange ranges[SOP_NRANGE];
    SopKey keys[SOP_NKEY];
    SopStep steps[SOP_NSTEP];
};

// dprog: device scratch for the per-launch copy of S
hipError_t launch_scan_agg_sop(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, int gmax, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out);

// wide-load variant (lean_kernel.h): gmax 1 or 4, plans accepted by host/sop.cpp::lean_eligible
hipError_t launch_scan_agg_lean(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, int gmax, GroupRec* partials,
                                uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out);

// FilterExec's predicate pass for AND-of-ranges predicates (kernels_range.hip): selection bitmap + kept rows per
// 1024-row tile, the interface of launch_scan_pred_bitmap
hipError_t launch_range_bitmap(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, uint64_t* bitmap, uint32_t* tile_counts);
// Utf8 column = literal (negate: !=), NULL rows dropped: the same bitmap + tile counts (kernels_range.hip)
struct Utf8Literal { uint8_t bytes[64]; int32_t len; };
hipError_t launch_utf8_eq_bitmap(const LaunchCfg& cfg, const int32_t* offsets, const void* data, const uint64_t* validity, int64_t n, const Utf8Literal& lit,
                                 bool negate, uint64_t* bitmap, uint32_t* tile_counts);

}  // namespace bhip
