// kernels_lean_g1.hip — instantiations of the wide-load scan + aggregate kernel (lean_kernel.h)
// for workgroups holding up to 1 group(s).
#include "lean_kernel.h"

namespace bhip {

hipError_t launch_scan_agg_lean_g1(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, GroupRec* partials,
                                   uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out) {
    return launch_lean_g<1>(cfg, S, dprog, partials, partial_ng, max_grid, status, grid_out);
}

}  // namespace bhip
