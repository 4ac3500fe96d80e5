// launch_common.h — launch-geometry helpers shared by the kernels_*.hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "vm_device.h"

namespace bhip {

template <int R>
static size_t host_tile_bytes(const VmProgram& G) {
    constexpr int TILE = BLOCK * R;
    size_t b = (size_t)G.n_vslots * TILE * 8;
    if (G.nullable) b += (size_t)G.n_vslots * TILE;
    b += (size_t)G.n_bslots * TILE;
    return (b + 15) & ~(size_t)15;
}

constexpr size_t LDS_PER_CU = 160 * 1024;

template <class K>
static hipError_t set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// rows per thread: 4 when at least two workgroups still fit a CU's LDS, else 2.
// BHIP_SCAN_R=2|4 overrides (tuning experiments).
static int choose_r(const VmProgram& G, size_t extra) {
    static const int forced = [] { const char* e = getenv("BHIP_SCAN_R"); return e ? atoi(e) : 0; }();
    if (forced == 2) return 2;
    if (forced == 4 && host_tile_bytes<4>(G) + extra <= LDS_PER_CU) return 4;
    return (host_tile_bytes<4>(G) + extra) * 2 <= LDS_PER_CU ? 4 : 2;
}

static int pick_grid(const LaunchCfg& cfg, int64_t n_tiles, size_t lds_bytes, int vgpr_blocks_per_cu) {
    int by_lds = (int)(LDS_PER_CU / (lds_bytes ? lds_bytes : 1));
    if (by_lds < 1) by_lds = 1;
    int per_cu = by_lds < vgpr_blocks_per_cu ? by_lds : vgpr_blocks_per_cu;
    int64_t g = (int64_t)cfg.device_cus * per_cu;
    if (g > n_tiles) g = n_tiles;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace bhip
