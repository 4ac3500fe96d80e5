// tbl_kernels.h — launchers of kernels_tbl.hip ('|'-separated TPC-H text -> Arrow columns on the device)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

constexpr int TBL_CHUNK = 16384;          // bytes of text per workgroup in the line passes
constexpr int TBL_MAX_FIELDS = 32;

enum TblErr : uint32_t {
    TBL_ERR_MISSING_FIELD = 1u,           // a line has fewer fields than the schema
    TBL_ERR_BAD_VALUE = 2u,               // not a number / date of the declared type
    TBL_ERR_PRECISION = 4u,               // a decimal with more than 15-16 significant digits (not converted exactly)
    TBL_ERR_BLANK_LINE = 8u
};

// what to do with each field of a line
struct TblPlan {
    int32_t n_fields;
    int32_t dtype[TBL_MAX_FIELDS];        // DType of the field
    int32_t out[TBL_MAX_FIELDS];          // output slot, or -1: skipped
    void* data[TBL_MAX_FIELDS];           // [slot] fixed-width values
    uint32_t* str_start[TBL_MAX_FIELDS];  // [slot] Utf8: offset of the field in the text
    uint32_t* str_len[TBL_MAX_FIELDS];    // [slot] Utf8: its length
};

hipError_t launch_tbl_count(const LaunchCfg& cfg, const uint8_t* text, int64_t n_bytes, uint32_t* chunk_lines);
hipError_t launch_tbl_starts(const LaunchCfg& cfg, const uint8_t* text, int64_t n_bytes, const uint64_t* chunk_base, uint64_t* starts);
hipError_t launch_tbl_parse(const LaunchCfg& cfg, const uint8_t* text, const uint64_t* starts, int64_t n_lines, int64_t n_bytes,
                            const TblPlan& plan, uint32_t* flags);
hipError_t launch_tbl_copy_strings(const LaunchCfg& cfg, const uint8_t* text, const uint32_t* str_start, const uint32_t* str_len,
                                   const int32_t* offsets, int64_t n, uint8_t* out);

}  // namespace bhip
