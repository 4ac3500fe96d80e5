// str_kernels.h — launchers of kernels_str.hip (internal C++ interface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace bhip {

enum StrFn : int { STR_LOWER = 0, STR_UPPER = 1, STR_TRIM = 2, STR_LTRIM = 3, STR_RTRIM = 4 };

// lower / upper (ASCII letters; *non_ascii is set when a byte >= 0x80 went through unchanged) and the three trims
// (Unicode White_Space, as Rust's str::trim / trim_start / trim_end): lengths first, then — after a scan — the bytes
hipError_t launch_str_transform_lengths(const LaunchCfg& cfg, int kind, const ColumnRef& c, int64_t n, uint32_t* lengths);
hipError_t launch_str_transform_write(const LaunchCfg& cfg, int kind, const ColumnRef& c, int64_t n, const int32_t* out_offsets, uint8_t* out,
                                      uint32_t* non_ascii);

// a string literal as a column of n equal values
constexpr int STR_LITERAL_MAX = 240;
struct StrLiteral { int32_t len; uint8_t bytes[STR_LITERAL_MAX]; };
hipError_t launch_str_broadcast(const LaunchCfg& cfg, const StrLiteral& lit, int64_t n, int32_t* offsets, uint8_t* out);

// CASE WHEN cond[0] THEN val[0] ... [ELSE val[n_when]] END over Utf8 values
constexpr int STR_SELECT_MAX = 8;
struct StrSelectArgs {
    int32_t n_when, has_else;
    ColumnRef cond[STR_SELECT_MAX];          // Boolean columns
    ColumnRef val[STR_SELECT_MAX + 1];       // Utf8 columns; val[n_when] = the ELSE value
};
hipError_t launch_str_select_lengths(const LaunchCfg& cfg, const StrSelectArgs& A, int64_t n, uint32_t* lengths, uint64_t* validity);
hipError_t launch_str_select_write(const LaunchCfg& cfg, const StrSelectArgs& A, int64_t n, const int32_t* out_offsets, uint8_t* out);

// sha224 / sha256 / sha384 / sha512 of every string: n digests of bits / 8 bytes at out + row * (bits / 8), offsets written too
hipError_t launch_sha2(const LaunchCfg& cfg, int bits, const ColumnRef& c, int64_t n, int32_t* out_offsets, uint8_t* out);

// MIN / MAX over Utf8 through sort ranks
hipError_t launch_invert_perm(const LaunchCfg& cfg, const uint32_t* perm, int64_t n, int64_t* rank);
hipError_t launch_rank_to_row(const LaunchCfg& cfg, const int64_t* rank, const uint64_t* validity, const uint32_t* perm, int64_t n, uint32_t* idx);

}  // namespace bhip
