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

// ---- SHA-2 (FIPS 180-4) of every string: sha224 / sha256 / sha384 / sha512 (rust/core/src/serde/logical_plan/from_proto.rs:924-927) --
// one thread per row, the message schedule kept as a 16-word ring; the digest (28 / 32 / 48 / 64 bytes, big-endian words) goes to
// out + row * digest_bytes: fixed-stride Binary values (a NULL row's bytes are never read: validity says so)
__device__ inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
__constant__ uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__constant__ uint64_t SHA512_K[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
    0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
    0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
    0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
    0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
    0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
    0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
    0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
    0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
    0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
    0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

// byte `i` of the padded message: the string, 0x80, zeros, the bit length as a big-endian integer in the last `len_bytes` bytes
__device__ inline uint32_t sha_msg_byte(const uint8_t* s, uint64_t len, uint64_t total, int len_bytes, uint64_t i) {
    if (i < len) return s[i];
    if (i == len) return 0x80u;
    if (i + (uint64_t)len_bytes < total) return 0u;
    const uint64_t k = total - 1 - i;                       // 0 = least significant byte of the length field
    return k < 8 ? (uint32_t)(((len * 8) >> (8 * k)) & 0xFF) : 0u;
}

__global__ void __launch_bounds__(BLOCK)
sha256_kernel(int bits, ColumnRef c, int64_t n, uint8_t* out) {
    const int digest = bits / 8;
    for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < n; row += (int64_t)gridDim.x * BLOCK) {
        if (!bit_at(c.validity, row)) continue;
        const uint8_t* s = reinterpret_cast<const uint8_t*>(c.data) + c.offsets[row];
        const uint64_t len = (uint64_t)(c.offsets[row + 1] - c.offsets[row]);
        const uint64_t total = ((len + 1 + 8 + 63) / 64) * 64;
        uint32_t h[8];
        if (bits == 224) { h[0] = 0xc1059ed8; h[1] = 0x367cd507; h[2] = 0x3070dd17; h[3] = 0xf70e5939; h[4] = 0xffc00b31; h[5] = 0x68581511; h[6] = 0x64f98fa7; h[7] = 0xbefa4fa4; }
        else { h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19; }
        for (uint64_t b0 = 0; b0 < total; b0 += 64) {
            uint32_t w[16];
            for (int t = 0; t < 16; ++t) {
                uint32_t x = 0;
                for (int k = 0; k < 4; ++k) x = (x << 8) | sha_msg_byte(s, len, total, 8, b0 + 4 * t + k);
                w[t] = x;
            }
            uint32_t a = h[0], b = h[1], cc = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
            for (int t = 0; t < 64; ++t) {
                if (t >= 16) {
                    const uint32_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                    const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3), s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                    w[t & 15] = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
                }
                const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
                const uint32_t t1 = hh + S1 + ch + SHA256_K[t] + w[t & 15];
                const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & cc) ^ (b & cc);
                const uint32_t t2 = S0 + mj;
                hh = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
            }
            h[0] += a; h[1] += b; h[2] += cc; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
        }
        uint8_t* o = out + row * digest;
        for (int k = 0; k < digest; ++k) o[k] = (uint8_t)(h[k >> 2] >> (8 * (3 - (k & 3))));
    }
}

__global__ void __launch_bounds__(BLOCK)
sha512_kernel(int bits, ColumnRef c, int64_t n, uint8_t* out) {
    const int digest = bits / 8;
    for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < n; row += (int64_t)gridDim.x * BLOCK) {
        if (!bit_at(c.validity, row)) continue;
        const uint8_t* s = reinterpret_cast<const uint8_t*>(c.data) + c.offsets[row];
        const uint64_t len = (uint64_t)(c.offsets[row + 1] - c.offsets[row]);
        const uint64_t total = ((len + 1 + 16 + 127) / 128) * 128;
        uint64_t h[8];
        if (bits == 384) {
            h[0] = 0xcbbb9d5dc1059ed8ull; h[1] = 0x629a292a367cd507ull; h[2] = 0x9159015a3070dd17ull; h[3] = 0x152fecd8f70e5939ull;
            h[4] = 0x67332667ffc00b31ull; h[5] = 0x8eb44a8768581511ull; h[6] = 0xdb0c2e0d64f98fa7ull; h[7] = 0x47b5481dbefa4fa4ull;
        } else {
            h[0] = 0x6a09e667f3bcc908ull; h[1] = 0xbb67ae8584caa73bull; h[2] = 0x3c6ef372fe94f82bull; h[3] = 0xa54ff53a5f1d36f1ull;
            h[4] = 0x510e527fade682d1ull; h[5] = 0x9b05688c2b3e6c1full; h[6] = 0x1f83d9abfb41bd6bull; h[7] = 0x5be0cd19137e2179ull;
        }
        for (uint64_t b0 = 0; b0 < total; b0 += 128) {
            uint64_t w[16];
            for (int t = 0; t < 16; ++t) {
                uint64_t x = 0;
                for (int k = 0; k < 8; ++k) x = (x << 8) | sha_msg_byte(s, len, total, 16, b0 + 8 * t + k);
                w[t] = x;
            }
            uint64_t a = h[0], b = h[1], cc = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
            for (int t = 0; t < 80; ++t) {
                if (t >= 16) {
                    const uint64_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                    const uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7), s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
                    w[t & 15] = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
                }
                const uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41), ch = (e & f) ^ (~e & g);
                const uint64_t t1 = hh + S1 + ch + SHA512_K[t] + w[t & 15];
                const uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39), mj = (a & b) ^ (a & cc) ^ (b & cc);
                const uint64_t t2 = S0 + mj;
                hh = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
            }
            h[0] += a; h[1] += b; h[2] += cc; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
        }
        uint8_t* o = out + row * digest;
        for (int k = 0; k < digest; ++k) o[k] = (uint8_t)(h[k >> 3] >> (8 * (7 - (k & 7))));
    }
}

// offsets of n fixed-length values
__global__ void __launch_bounds__(BLOCK)
fixed_offsets_kernel(int32_t stride, int64_t n, int32_t* offsets) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i <= n; i += (int64_t)gridDim.x * BLOCK) offsets[i] = (int32_t)(i * stride);
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
hipError_t launch_sha2(const LaunchCfg& cfg, int bits, const ColumnRef& c, int64_t n, int32_t* out_offsets, uint8_t* out) {
    if (bits != 224 && bits != 256 && bits != 384 && bits != 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fixed_offsets_kernel, dim3(grid_of(cfg, n + 1)), dim3(BLOCK), 0, cfg.stream, bits / 8, n, out_offsets);
    if (n == 0) return hipGetLastError();
    if (bits <= 256) hipLaunchKernelGGL(sha256_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, bits, c, n, out);
    else hipLaunchKernelGGL(sha512_kernel, dim3(grid_of(cfg, n)), dim3(BLOCK), 0, cfg.stream, bits, c, n, out);
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
