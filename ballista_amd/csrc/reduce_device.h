// reduce_device.h — fixed-order (deterministic) lane/wave reductions of 64-bit accumulators.
#pragma once
#include <hip/hip_runtime.h>
#include "vm_device.h"

namespace bhip {

// ---- fixed-order reductions ----------------------------------------------------------------
__device__ inline uint64_t shfl_down_u64(uint64_t v, int delta) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_down(lo, delta, 64);
    hi = __shfl_down(hi, delta, 64);
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline uint64_t acc_identity(int kind) {
    switch (kind) {
        case ACC_MIN_F64: return d2u(__builtin_huge_val());
        case ACC_MAX_F64: return d2u(-__builtin_huge_val());
        case ACC_MIN_I64: return (uint64_t)INT64_MAX;
        case ACC_MAX_I64: return (uint64_t)INT64_MIN;
        default: return 0;
    }
}

__device__ inline uint64_t acc_combine(uint64_t a, uint64_t b, int kind) {
    switch (kind) {
        case ACC_SUM_F64: return d2u(u2d(a) + u2d(b));
        case ACC_MIN_F64: return u2d(b) < u2d(a) ? b : a;
        case ACC_MAX_F64: return u2d(b) > u2d(a) ? b : a;
        case ACC_MIN_I64: return (int64_t)b < (int64_t)a ? b : a;
        case ACC_MAX_I64: return (int64_t)b > (int64_t)a ? b : a;
        default: return a + b;   // integer sums and counts
    }
}

__device__ inline uint64_t wave_reduce(uint64_t v, int kind) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = acc_combine(v, shfl_down_u64(v, d), kind);
    return v;   // lane 0 holds the result
}

}  // namespace bhip
