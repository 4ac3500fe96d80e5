// kernels_tbl.hip — TPC-H `.tbl` text (dbgen: '|'-separated fields, '|' before the newline) -> Arrow columns on the
// device.  Stands where the reference's scan leaf stands: CsvExec with delimiter '|', no header, explicit schema
// (rust/benchmarks/tpch/src/main.rs:129-150, schemas :267-360; rust/core/src/serde/physical_plan/from_proto.rs:93-110).
// with `--format tbl` the CPU spends its time here, upstream of every operator.
//
// Byte work, three passes over the text:
//   1. newlines per 16 KiB chunk                          (count)  -> exclusive scan
//   2. start offset of every line                          (stable ranks inside a chunk: thread-local counts + LDS scan)
//   3. one thread per line walks its fields; projected fields are converted in place:
//        Int32 / Int64   [-]digits
//        Float64         [-]digits[.digits]  =  M / 10^k with M < 2^53 and k <= 22: both exact in double, so the one
//                        division is the correctly rounded value of the decimal text (what str::parse::<f64> returns)
//        Date32          YYYY-MM-DD -> days since 1970-01-01 (proleptic Gregorian)
//        Utf8            (offset, length) of the field; a second pass copies the bytes behind an exclusive scan
// Anything else in the text (exponents, > 15 significant digits, missing fields, blank lines) raises a flag and the
// host reports BHIP_EEXEC / BHIP_ENOTIMPL: the caller keeps its CPU reader for that file.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "tbl_kernels.h"
#include "vm_device.h"
#include "vm_isa.h"

namespace bhip {

constexpr int TBL_THREAD_BYTES = TBL_CHUNK / BLOCK;      // 64 bytes per thread and chunk

// newlines among the four bytes of `w` (byte by byte: the subtract-and-mask zero-byte trick can flag a 0x0B byte that
// sits right above a newline, and these counts must equal the line count exactly)
__device__ inline uint32_t newlines_exact(uint32_t w) {
    uint32_t c = 0;
    c += (w & 0xFFu) == 0x0Au;
    c += ((w >> 8) & 0xFFu) == 0x0Au;
    c += ((w >> 16) & 0xFFu) == 0x0Au;
    c += (w >> 24) == 0x0Au;
    return c;
}

// pass 1: newlines per chunk.  Lanes read adjacent 16-byte pieces (coalesced), order does not matter for a count.
__global__ void __launch_bounds__(BLOCK)
tbl_count_kernel(const uint8_t* text, int64_t n_bytes, uint32_t* chunk_lines) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t chunk0 = (int64_t)blockIdx.x * TBL_CHUNK;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < TBL_CHUNK / (BLOCK * 16); ++k) {
        const int64_t p = chunk0 + ((int64_t)k * BLOCK + threadIdx.x) * 16;
        if (p + 16 <= n_bytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(text + p);            // text is 256-byte aligned, p a multiple of 16
            c += newlines_exact(v.x) + newlines_exact(v.y) + newlines_exact(v.z) + newlines_exact(v.w);
        } else {
            for (int64_t q = p; q < n_bytes && q < p + 16; ++q) c += text[q] == '\n';
        }
    }
    if (c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) chunk_lines[blockIdx.x] = s_cnt;
}

// pass 2: starts[i] = offset of the first byte of line i (starts[0] = 0 is written by the host).  The chunk is staged
// in LDS with coalesced 16-byte loads; each thread then owns 64 CONSECUTIVE bytes (rows of 16 dwords padded to 17,
// so the 64 lanes of a wave read 64 different banks), which keeps the newline ranks in text order.
__global__ void __launch_bounds__(BLOCK)
tbl_starts_kernel(const uint8_t* text, int64_t n_bytes, const uint64_t* chunk_base, uint64_t* starts) {
    __shared__ uint32_t s_text[BLOCK * 17];
    __shared__ uint32_t s_scan[BLOCK];
    const int tid = threadIdx.x;
    const int64_t chunk0 = (int64_t)blockIdx.x * TBL_CHUNK;
#pragma unroll
    for (int k = 0; k < TBL_CHUNK / (BLOCK * 16); ++k) {
        const int piece = k * BLOCK + tid;                       // 16-byte piece of the chunk
        const int64_t p = chunk0 + (int64_t)piece * 16;
        uint4 v = make_uint4(0, 0, 0, 0);                        // bytes past the text read as 0: never a newline
        if (p + 16 <= n_bytes) v = *reinterpret_cast<const uint4*>(text + p);
        else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int64_t q = p; q < n_bytes && q < p + 16; ++q) w[(q - p) >> 2] |= (uint32_t)text[q] << (8 * ((q - p) & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        const int row = piece >> 2, col = (piece & 3) * 4;       // row = owning thread (64 bytes = 4 pieces)
        s_text[row * 17 + col + 0] = v.x; s_text[row * 17 + col + 1] = v.y;
        s_text[row * 17 + col + 2] = v.z; s_text[row * 17 + col + 3] = v.w;
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int d = 0; d < 16; ++d) c += newlines_exact(s_text[tid * 17 + d]);
    s_scan[tid] = c;
    __syncthreads();
    for (int d = 1; d < BLOCK; d <<= 1) {                 // inclusive Hillis-Steele scan of the thread counts
        const uint32_t v = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    if (c == 0) return;
    uint64_t rank = chunk_base[blockIdx.x] + (s_scan[tid] - c);
    const int64_t base = chunk0 + (int64_t)tid * TBL_THREAD_BYTES;
    for (int d = 0; d < 16; ++d) {
        const uint32_t w = s_text[tid * 17 + d];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (((w >> (8 * b)) & 0xFFu) == 0x0Au) starts[++rank] = (uint64_t)(base + d * 4 + b) + 1;
    }
}

__device__ inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

__constant__ double TBL_POW10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                     1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// byte sources of the field walk: the text in HBM, or the lines of one workgroup staged in LDS
struct TblGlobalReader {
    const uint8_t* text;
    __device__ uint8_t operator()(int64_t pos) const { return text[pos]; }
};
struct TblLdsReader {
    const uint8_t* buf;          // LDS copy of text[origin, origin + ...)
    int64_t origin;
    __device__ uint8_t operator()(int64_t pos) const { return buf[pos - origin]; }
};

// one line [p, e): walk the fields, convert the projected ones.  Returns the error flags.
template <class R>
__device__ inline uint32_t tbl_parse_line(const R& rd, int64_t p, int64_t e, int64_t i, const TblPlan& plan) {
    uint32_t err = 0;
    if (e > p && rd(e - 1) == '\r') --e;
    if (e <= p) return TBL_ERR_BLANK_LINE;
    for (int f = 0; f < plan.n_fields; ++f) {
        if (p > e) { err |= TBL_ERR_MISSING_FIELD; break; }
        int64_t q = p;
        while (q < e && rd(q) != '|') ++q;              // field = [p, q)
        const int out = plan.out[f];
        if (out >= 0) {
            const int dt = plan.dtype[f];
            if (dt == DT_UTF8) {
                plan.str_start[out][i] = (uint32_t)p;
                plan.str_len[out][i] = (uint32_t)(q - p);
            } else if (dt == DT_DATE32) {
                // YYYY-MM-DD
                bool ok = (q - p) == 10 && rd(p + 4) == '-' && rd(p + 7) == '-';
                int v[8];
                const int pos[8] = {0, 1, 2, 3, 5, 6, 8, 9};
                for (int k = 0; k < 8 && ok; ++k) {
                    const int c = (int)rd(p + pos[k]) - '0';
                    ok = c >= 0 && c <= 9;
                    v[k] = c;
                }
                int32_t days = 0;
                if (ok) {
                    const int y = v[0] * 1000 + v[1] * 100 + v[2] * 10 + v[3], m = v[4] * 10 + v[5], d = v[6] * 10 + v[7];
                    ok = m >= 1 && m <= 12 && d >= 1 && d <= 31;
                    days = (int32_t)days_from_civil(y, (unsigned)m, (unsigned)d);
                }
                if (!ok) err |= TBL_ERR_BAD_VALUE;
                reinterpret_cast<int32_t*>(plan.data[out])[i] = days;
            } else {
                int64_t r = p;
                bool neg = false;
                if (r < q && (rd(r) == '-' || rd(r) == '+')) { neg = rd(r) == '-'; ++r; }
                uint64_t m = 0;
                int digits = 0, frac = 0;
                bool seen_dot = false, ok = r < q;
                for (; r < q; ++r) {
                    const uint8_t ch = rd(r);
                    if (ch >= '0' && ch <= '9') {
                        if (digits >= 19) {                                  // 19 digits still fit 64 bits
                            if (dt == DT_FLOAT64) { err |= TBL_ERR_PRECISION; m = 0; frac = 0; r = q; break; }
                            ok = false;
                            break;
                        }
                        m = m * 10 + (uint64_t)(ch - '0');
                        if (m != 0 || seen_dot) ++digits;             // leading zeros of the integer part are free
                        if (seen_dot) ++frac;
                    } else if (ch == '.' && !seen_dot && dt == DT_FLOAT64) {
                        seen_dot = true;
                    } else { ok = false; break; }
                }
                if (dt == DT_FLOAT64) {
                    if (!ok) err |= TBL_ERR_BAD_VALUE;
                    else if (m >= (1ull << 53) || frac > 22) { err |= TBL_ERR_PRECISION; ok = false; }
                    double v = ok ? (double)m / TBL_POW10[frac] : 0.0;
                    reinterpret_cast<double*>(plan.data[out])[i] = neg ? -v : v;
                } else {
                    if (!ok || seen_dot) err |= TBL_ERR_BAD_VALUE;
                    if (m > (neg ? (1ull << 63) : (1ull << 63) - 1ull)) err |= TBL_ERR_BAD_VALUE;      // beyond Int64
                    const int64_t v = neg ? (int64_t)(0ull - m) : (int64_t)m;
                    if (dt == DT_INT32) {
                        if (v > 2147483647ll || v < -2147483648ll) err |= TBL_ERR_BAD_VALUE;
                        reinterpret_cast<int32_t*>(plan.data[out])[i] = (int32_t)v;
                    } else {
                        reinterpret_cast<int64_t*>(plan.data[out])[i] = v;
                    }
                }
            }
        }
        p = q + 1;
    }
    return err;
}

// pass 3: a workgroup takes 256 consecutive lines.  Their text is one contiguous span: it is staged in LDS with
// coalesced 16-byte loads (a thread walking its line byte by byte in HBM issues one dependent load per byte), and
// every thread then walks its own line in LDS.  A span that does not fit (very long lines) is walked in HBM.
constexpr int TBL_STAGE = 48 * 1024;
__global__ void __launch_bounds__(BLOCK)
tbl_parse_kernel(const uint8_t* text, const uint64_t* starts, int64_t n_lines, int64_t n_bytes, TblPlan plan, uint32_t* flags) {
    __shared__ __align__(16) uint8_t s_buf[TBL_STAGE];
    uint32_t err = 0;
    const int tid = threadIdx.x;
    for (int64_t i0 = (int64_t)blockIdx.x * BLOCK; i0 < n_lines; i0 += (int64_t)gridDim.x * BLOCK) {
        const int64_t n_here = n_lines - i0 < BLOCK ? n_lines - i0 : BLOCK;
        const int64_t span0 = (int64_t)starts[i0] & ~(int64_t)15;                 // 16-byte aligned (the text buffer is)
        int64_t span1 = (int64_t)starts[i0 + n_here];
        if (span1 > n_bytes) span1 = n_bytes;
        const bool staged = span1 - span0 <= TBL_STAGE - 16;       // the copy below moves whole 16-byte pieces
        if (staged) {
            for (int64_t k = (int64_t)tid * 16; k < span1 - span0; k += BLOCK * 16) {
                const int64_t g = span0 + k;
                if (g + 16 <= n_bytes) *reinterpret_cast<uint4*>(s_buf + k) = *reinterpret_cast<const uint4*>(text + g);
                else
                    for (int64_t b = g; b < n_bytes; ++b) s_buf[b - span0] = text[b];
            }
        }
        __syncthreads();
        if (tid < n_here) {
            const int64_t i = i0 + tid;
            const int64_t p = (int64_t)starts[i];
            const int64_t e = (int64_t)starts[i + 1] - 1;       // the newline (or one past the text for an unterminated last line)
            if (staged) err |= tbl_parse_line(TblLdsReader{s_buf, span0}, p, e, i, plan);
            else err |= tbl_parse_line(TblGlobalReader{text}, p, e, i, plan);
        }
        __syncthreads();
    }
    if (err) atomicOr(flags, err);
}

__global__ void __launch_bounds__(BLOCK)
tbl_copy_strings_kernel(const uint8_t* text, const uint32_t* str_start, const uint32_t* str_len, const int32_t* offsets, int64_t n,
                        uint8_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint8_t* s = text + str_start[i];
        uint8_t* d = out + offsets[i];
        const uint32_t len = str_len[i];
        for (uint32_t b = 0; b < len; ++b) d[b] = s[b];
    }
}

static int grid_rows(const LaunchCfg& cfg, int64_t n) {
    int64_t g = (n + BLOCK - 1) / BLOCK;
    const int64_t cap = (int64_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_tbl_count(const LaunchCfg& cfg, const uint8_t* text, int64_t n_bytes, uint32_t* chunk_lines) {
    const int64_t n_chunks = (n_bytes + TBL_CHUNK - 1) / TBL_CHUNK;
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(tbl_count_kernel, dim3((unsigned)n_chunks), dim3(BLOCK), 0, cfg.stream, text, n_bytes, chunk_lines);
    return hipGetLastError();
}
hipError_t launch_tbl_starts(const LaunchCfg& cfg, const uint8_t* text, int64_t n_bytes, const uint64_t* chunk_base, uint64_t* starts) {
    const int64_t n_chunks = (n_bytes + TBL_CHUNK - 1) / TBL_CHUNK;
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(tbl_starts_kernel, dim3((unsigned)n_chunks), dim3(BLOCK), 0, cfg.stream, text, n_bytes, chunk_base, starts);
    return hipGetLastError();
}
hipError_t launch_tbl_parse(const LaunchCfg& cfg, const uint8_t* text, const uint64_t* starts, int64_t n_lines, int64_t n_bytes,
                            const TblPlan& plan, uint32_t* flags) {
    if (n_lines == 0) return hipSuccess;
    hipLaunchKernelGGL(tbl_parse_kernel, dim3(grid_rows(cfg, n_lines)), dim3(BLOCK), 0, cfg.stream, text, starts, n_lines, n_bytes, plan,
                       flags);
    return hipGetLastError();
}
hipError_t launch_tbl_copy_strings(const LaunchCfg& cfg, const uint8_t* text, const uint32_t* str_start, const uint32_t* str_len,
                                   const int32_t* offsets, int64_t n, uint8_t* out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tbl_copy_strings_kernel, dim3(grid_rows(cfg, n)), dim3(BLOCK), 0, cfg.stream, text, str_start, str_len, offsets, n, out);
    return hipGetLastError();
}

}  // namespace bhip
