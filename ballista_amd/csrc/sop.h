// sop.h — the "chain of products" plan table of the register-resident aggregate fast path
// (kernels_sop.hip); built by host/sop.cpp from the same expressions the VM program is built from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

constexpr int SOP_NPRED = 6;      // comparisons in the conjunction
constexpr int SOP_NKEY = 3;       // group-key columns
constexpr int SOP_NSTEP = 8;      // chain steps (= accumulators)
constexpr int SOP_NCOL = 16;

struct SopColumn {
    const void* data;
    const int32_t* offsets;
    int32_t dtype;
    int32_t data_bytes;
    // the same column starting `base` rows further (fixed-width types): lets the kernel index with 32 bits
    __host__ __device__ SopColumn at(int64_t base) const {
        SopColumn c = *this;
        int w = 8;
        if (dtype == DT_INT32 || dtype == DT_DATE32) w = 4;
        else if (dtype == DT_UINT8) w = 1;
        c.data = reinterpret_cast<const char*>(data) + base * w;
        return c;
    }
};

struct SopCmp {
    uint8_t col;        // index into SopProgram::cols
    uint8_t cmp;        // CmpKind:  value(col) <cmp> lit
    uint8_t vclass;     // VC_I64, VC_F64, 3 = unsigned 64
    uint8_t pad[5];
    uint64_t lit;       // literal bits
};

enum SopFactor : uint8_t {
    SOP_F_COL = 0, SOP_F_LIT_MINUS_COL = 1, SOP_F_LIT_PLUS_COL = 2, SOP_F_COL_MINUS_LIT = 3, SOP_F_COL_PLUS_LIT = 4,
    SOP_F_LIT = 5
};
enum SopOp : uint8_t { SOP_OP_START = 0, SOP_OP_MUL = 1, SOP_OP_DIV = 2 };

struct SopStep {        // t_s = START ? f : t_(s-1) (*|/) f ,  f = factor(col, lit)
    uint8_t col;
    uint8_t mode;       // SopFactor
    uint8_t op;         // SopOp
    uint8_t acc;        // accumulator (GroupRec::acc index) that sums t_s, 0xFF = none
    uint8_t pad[4];
    double lit;
};

struct SopKey {
    uint8_t col;
    uint8_t width;      // bytes in the packed key (Utf8: 1 length byte + chars)
    uint8_t pos;        // byte position in the packed key
    uint8_t pad;
};

struct SopProgram {
    int64_t n_rows;
    int32_t n_cols, n_pred, n_keys, n_steps;
    int32_t key_bytes, pad;
    SopColumn cols[SOP_NCOL];
    SopCmp pred[SOP_NPRED];
    SopKey keys[SOP_NKEY];
    SopStep steps[SOP_NSTEP];
};

// dprog: device scratch for the per-launch copy of S
hipError_t launch_scan_agg_sop(const LaunchCfg& cfg, const SopProgram& S, SopProgram* dprog, int gmax, GroupRec* partials,
                               uint32_t* partial_ng, int max_grid, ScanStatus* status, int* grid_out);

}  // namespace bhip
