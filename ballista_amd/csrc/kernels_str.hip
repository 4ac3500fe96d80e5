// kernels_str.hip — expressions that PRODUCE Utf8 values: lower / upper / trim / ltrim / rtrim
// (rust/core/src/serde/logical_plan/from_proto.rs:910-918 ships them), CASE ... THEN <string>, string literals as columns.
//
// The expression VM (vm_device.h) keeps 64-bit values per row; a string result is a new Arrow column instead: every kernel
// here runs as lengths -> exclusive scan (offsets) -> bytes, one thread per row, over the offsets / bytes / validity buffers
// of its inputs.  The host (host/utf8_exprs.cpp) materialises such nodes as temporary columns and hands the rest of the
// expression to the VM, which compares / hashes / LIKEs Utf8 columns in place.
#include <hip/hip_runtime.h>
#include "str_kernels.h"

namespace bhip {

namespace {
constexpr int BLOCK = 256;

__device__ inline bool bit_at(const uint64_t* bits, int64_t i) { return bits == nullptr || ((bits[i >> 6] >> (i & 63)) & 1ull); }

// Unicode White_Space (what Rust's str::trim strips): one code point starting at s[0]; returns its byte length or 0
__device__ inline int ws_prefix(const uint8_t* s, int len) {
    if (len <= 0) return 0;
    const uint8_t b0 = s[0];
    if (b0 == 0x20 || (b0 >= 0x09 && b0 <= 0x0D)) return 1;
    if (len >= 2 && b0 == 0xC2 && (s[1] == 0x85 || s[1] == 0xA0)) return 2;                         // U+0085, U+00A0
    if (len >= 3) {
        const uint8_t b1 = s[1], b2 = s[2];
        if (b0 == 0xE1 && b1 == 0x9A && b2 == 0x80) return 3;                                        // U+1680
        if (b0 == 0xE2 && b1 == 0x80 && ((b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF)) return 3;  // U+2000-200A, 2028, 2029, 202F
        if (b0 == 0xE2 && b1 == 0x81 && b2 == 0x9F) return 3;                                        // U+205F
        if (b0 == 0xE3 && b1 == 0x80 && b2 == 0x80) return 3;                                        // U+3000
    }
    return 0;
}
// the same, for the code point that ENDS at s[len-1]
__device__ inline int ws_suffix(const uint8_t* s, int len) {
    if (len <= 0) return 0;
    const uint8_t e0 = s[len - 1];
    if (e0 < 0x80) return (e0 == 0x20 || (e0 >= 0x09 && e0 <= 0x0D)) ? 1 : 0;
    if (len >= 2 && (s[len - 2] & 0xC0) == 0xC0) return ws_prefix(s + len - 2, 2) == 2 ? 2 : 0;
    if (len >= 3 && (s[len - 3] & 0xC0) == 0xC0) return ws_prefix(s + len - 3, 3) == 3 ? 3 : 0;
    return 0;
}

// [begin, end) of the transformed value inside the source value
__device__ inline void trimmed_range(int kind, const uint8_t* s, int len, int& begin, int& end) {
    begin = 0;
    end = len;
    if (kind == STR_TRIM || kind == STR_LTRIM)
        for (int w; (w = ws_prefix(s + begin, end - begin)) > 0;) begin += w;
    if (kind == STR_TRIM || kind == STR_RTRIM)
        for (int w; (w = ws_suffix(s + begin, end - begin)) > 0;) end -= w;
}

__global__ void __launch_bounds__(BLOCK)
str_transform_lengths_kernel(int kind, ColumnRef c, int64_t n, uint32_t* lengths) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int32_t o0 = c.offsets[i];
        int len = c.offsets[i + 1] - o0;
        if (!bit_at(c.validity, i)) len = 0;
        else if (kind >= STR_TRIM) {
            int b, e;
            trimmed_range(kind, reinterpret_cast<const uint8_t*>(c.data) + o0, len, b, e);
            len = e - b;
        }
        lengths[i] = (uint32_t)len;
    }
}

__global__ void __launch_bounds__(BLOCK)
str_transform_write_kernel(int kind, ColumnRef c, int64_t n, const int32_t* out_offsets, uint8_t* out, uint32_t* non_ascii) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int32_t o0 = c.offsets[i], d0 = out_offsets[i];
        const int len = out_offsets[i + 1] - d0;
        const uint8_t* s = reinterpret_cast<const uint8_t*>(c.data) + o0;
        int b = 0, e = 0;
        if (kind >= STR_TRIM) trimmed_range(kind, s, c.offsets[i + 1] - o0, b, e);
        for (int k = 0; k < len; ++k) {
            uint8_t ch = s[b + k];
            if (kind == STR_LOWER) { if (ch >= 'A' && ch <= 'Z') ch += 32; bad |= ch >= 0x80; }
            else if (kind == STR_UPPER) { if (ch >= 'a' && ch <= 'z') ch -= 32; bad |= ch >= 0x80; }
            out[d0 + k] = ch;
        }
    }
    if (bad) atomicOr(non_ascii, 1u);
}

__global__ void __launch_bounds__(BLOCK)
str_broadcast_kernel(StrLiteral lit, int64_t n, int32_t* offsets, uint8_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i <= n; i += (int64_t)gridDim.x * BLOCK) {
        offsets[i] = (int32_t)(i * lit.len);
        if (i < n)
            for (int k = 0; k < lit.len; ++k) out[i * lit.len + k] = lit.bytes[k];
    }
}

// CASE: the first branch whose condition is TRUE (valid and set); none -> the ELSE value or NULL
__device__ inline int case_branch(const StrSelectArgs& A, int64_t i) {
    for (int b = 0; b < A.n_when; ++b) {
        const ColumnRef& c = A.cond[b];
        const uint64_t* bits = reinterpret_cast<const uint64_t*>(c.data);
        if (bit_at(c.validity, i) && ((bits[i >> 6] >> (i & 63)) & 1ull)) return b;
    }
    return A.has_else ? A.n_when : -1;
}

__global__ void __launch_bounds__(BLOCK)
str_select_lengths_kernel(StrSelectArgs A, int64_t n, uint32_t* lengths, uint64_t* validity) {
    const int64_t n_round = (n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
        bool valid = false;
        if (i < n) {
            const int b = case_branch(A, i);
            uint32_t len = 0;
            if (b >= 0) {
                const ColumnRef& v = A.val[b];
                valid = bit_at(v.validity, i);
                if (valid) len = (uint32_t)(v.offsets[i + 1] - v.offsets[i]);
            }
            lengths[i] = len;
        }
        const uint64_t w = __ballot(valid);
        if ((threadIdx.x & 63) == 0) validity[i >> 6] = w;
    }
}

__global__ void __launch_bounds__(BLOCK)
str_select_write_kernel(StrSelectArgs A, int64_t n, const int32_t* out_offsets, uint8_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int32_t d0 = out_offsets[i];
        const int len = out_offsets[i + 1] - d0;
        if (len == 0) continue;
        const ColumnRef& v = A.val[case_branch(A, i)];
        const uint8_t* s = reinterpret_cast<const uint8_t*>(v.data) + v.offsets[i];
        for (int k = 0; k < len; ++k) out[d0 + k] = s[k];
    }
}

// rank[perm[i]] = i  (the position of every row in a sort order: MIN / MAX over Utf8 become MIN / MAX over Int64 ranks)
__global__ void __launch_bounds__(BLOCK)
invert_perm_kernel(const uint32_t* perm, int64_t n, int64_t* rank) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) rank[perm[i]] = i;
}
// idx[g] = valid(g) ? perm[rank[g]] : NULL_INDEX
__global__ void __launch_bounds__(BLOCK)
rank_to_row_kernel(const int64_t* rank, const uint64_t* validity, const uint32_t* perm, int64_t n, uint32_t* idx) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        idx[i] = bit_at(validity, i) ? perm[rank[i]] : 0xFFFFFFFFu;
}

inline int grid_of(const LaunchCfg& cfg, int64_t n) {
    const int64_t want = (n + BLOCK - 1) / BLOCK, cap = (int64_t)cfg.device_cus * 8;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}
}  // namespace

hipError_t launch_str_transform_lengths(const LaunchCfg& cfg, int kind, const ColumnRef& c, int64_t n, uint32_t* lengths) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(str_transform_lengths_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, kind, c, n, lengths);
    return hipGetLastError();
}
hipError_t launch_str_transform_write(const LaunchCfg& cfg, int kind, const ColumnRef& c, int64_t n, const int32_t* out_offsets, uint8_t* out,
                                      uint32_t* non_ascii) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(str_transform_write_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, kind, c, n, out_offsets, out, non_ascii);
    return hipGetLastError();
}
hipError_t launch_str_broadcast(const LaunchCfg& cfg, const StrLiteral& lit, int64_t n, int32_t* offsets, uint8_t* out) {
    hipLaunchKernelGGL(str_broadcast_kernel, dim3(grid_of(cfg, n + 1)), dim3(BLOCK), 0, cfg.stream, lit, n, offsets, out);
    return hipGetLastError();
}
hipError_t launch_str_select_lengths(const LaunchCfg& cfg, const StrSelectArgs& A, int64_t n, uint32_t* lengths, uint64_t* validity) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(str_select_lengths_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, A, n, lengths, validity);
    return hipGetLastError();
}
hipError_t launch_str_select_write(const LaunchCfg& cfg, const StrSelectArgs& A, int64_t n, const int32_t* out_offsets, uint8_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(str_select_write_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, A, n, out_offsets, out);
    return hipGetLastError();
}
hipError_t launch_invert_perm(const LaunchCfg& cfg, const uint32_t* perm, int64_t n, int64_t* rank) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(invert_perm_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, perm, n, rank);
    return hipGetLastError();
}
hipError_t launch_rank_to_row(const LaunchCfg& cfg, const int64_t* rank, const uint64_t* validity, const uint32_t* perm, int64_t n, uint32_t* idx) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rank_to_row_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, rank, validity, perm, n, idx);
    return hipGetLastError();
}

}  // namespace bhip
