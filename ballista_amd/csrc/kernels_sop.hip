// kernels_sop.hip — dispatcher of the register-resident scan + aggregate fast path
// (kernel: sop_kernel.h; instantiations: kernels_sop_g{1,4,8}.hip; plan table: sop.h).
#include <hip/hip_runtime.h>
#include "sop.h"

namespace bhip {

hipError_t launch_scan_agg_sop_g1(const LaunchCfg&, const SopProgram&, SopProgram*, GroupRec*, uint32_t*, int, ScanStatus*, int*);
hipError_t launch_scan_agg_sop_g4(const LaunchCfg&, const SopProgram&, SopProgram*, GroupRec*, uint32_t*, int, ScanStatus*, int*);
hipError_t launch_scan_agg_sop_g8(const LaunchCfg&, const SopProgram&, SopProgram*, GroupRec*, uint32_t*, int, ScanStatus*, int*);

hipError_t launch_scan_agg_sop(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, int gmax, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    switch (gmax) {
        case 1: return launch_scan_agg_sop_g1(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
        case 4: return launch_scan_agg_sop_g4(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
        case 8: return launch_scan_agg_sop_g8(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_agg_lean_g1(const LaunchCfg&, const SopProgram&, SopProgram*, GroupRec*, uint32_t*, int, ScanStatus*, int*);
hipError_t launch_scan_agg_lean_g4(const LaunchCfg&, const SopProgram&, SopProgram*, GroupRec*, uint32_t*, int, ScanStatus*, int*);

// wide-load variant (lean_kernel.h; instantiations: kernels_lean_g{1,4}.hip)
hipError_t launch_scan_agg_lean(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, int gmax, GroupRec* partials,
                                uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    switch (gmax) {
        case 1: return launch_scan_agg_lean_g1(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
        case 4: return launch_scan_agg_lean_g4(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace bhip
