// kernels.h — host-callable launchers of the HIP kernels (internal C++ interface between
// host/*.cpp and kernels_*.hip; the public C ABI is include/ballista_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vm_isa.h"

namespace bhip {

// ---- group table: result of an aggregation (low-cardinality or hash path) --------------
constexpr int AGG_NACC = 8;     // accumulators per group on the register (low-cardinality) path
constexpr int AGG_GMAX = 8;     // groups a workgroup can hold in registers

struct GroupRec {               // one group of one workgroup's partial result
    uint64_t k0, k1;            // packed key
    uint64_t rows;              // rows that reached the group
    uint64_t acc[VM_MAX_ACC];
    uint64_t nvalid[VM_MAX_ACC];  // non-NULL inputs per accumulator
};

struct ScanStatus {
    uint32_t flags;             // SCAN_ERR_* | SCAN_OVERFLOW_GROUPS
    uint32_t n_groups;          // merge kernels: number of distinct groups
    uint64_t count;             // generic counter (selected rows, ...)
};

// what the host reads back after a small aggregate, in one piece (<= 56 bytes: one pinned host slot, host/core.cpp)
constexpr int TAIL_TOTALS = 5;
struct TailInfo {
    ScanStatus st;
    uint64_t totals[TAIL_TOTALS];   // value bytes of the Utf8 key columns
};

struct LaunchCfg {
    int device_cus;             // multiprocessor count
    hipStream_t stream;
};

// low-cardinality fused scan+aggregate; gmax in {1,4,8}.  partials: grid*gmax GroupRec,
// partial_ng: grid uint32.  Returns the grid size used in *grid_out.
// dparams: device scratch for the per-launch copy of P (sizeof(ScanParams)), must outlive the kernel.
// n_batches > 1: &P is the first of n_batches ScanParams (one program bound to several small batches), dparams holds as many: ONE launch,
// blockIdx.y = batch; *grid_out = workgroups over all batches (the partial tables written)
hipError_t launch_scan_agg_lowcard(const LaunchCfg& cfg, const ScanParams& P, ScanParams* dparams, int gmax,
                                   GroupRec* partials, uint32_t* partial_ng, int max_grid,
                                   ScanStatus* status, int* grid_out, int n_batches = 1);
int scan_agg_lowcard_max_grid(const LaunchCfg& cfg);

// merge per-workgroup partials (n_part*gmax records) into `table` (capacity cap groups)
hipError_t launch_merge_partials(const LaunchCfg& cfg, const GroupRec* partials, const uint32_t* partial_ng,
                                 int n_part, int gmax, const AccSpec* acc_host, int n_acc,
                                 GroupRec* table, int cap, uint32_t* entry_group, ScanStatus* status);

// projection: evaluate P and write P.n_out output columns (+ validity bitmaps, may be null)
struct ProjectOut {
    void* data[VM_MAX_OUT];
    uint64_t* validity[VM_MAX_OUT];
};
hipError_t launch_scan_project(const LaunchCfg& cfg, const ScanParams& P, const ProjectOut& out, ScanStatus* status);

// predicate -> selection bitmap (bit i = row i selected) + per-tile counts (tile = 1024 rows)
hipError_t launch_scan_pred_bitmap(const LaunchCfg& cfg, const ScanParams& P, uint64_t* bitmap,
                                   uint32_t* tile_counts, ScanStatus* status);
constexpr int SEL_TILE = 1024;

// keys: packed key (16 B/row, may be null), row hash (8 B/row, may be null), and the
// selection bitmap of the fused predicate (may be null)
hipError_t launch_scan_keys(const LaunchCfg& cfg, const ScanParams& P, uint64_t* keys128, uint64_t* hashes,
                            uint64_t* bitmap, ScanStatus* status);

}  // namespace bhip
