// kernels_gen.hip — on-device generator of synthetic TPC-H-shaped Arrow columns.
//
// The reference's benchmark reads dbgen files (rust/benchmarks/tpch/src/main.rs:128-155); dbgen
// needs a network fetch, so SF100 inputs (27.6 GB for Q1) are generated straight into HBM.
// Spec (DESIGN.md "Synthetic data spec", after SURVEY.md Appendix B): every attribute of row i
// is  f(rnd(seed, stream, i))  with rnd a splitmix64 counter hash — integer arithmetic only, so
// the result is bit-identical to the CPU twin in oracle/tpch_gen.c (tests/test_generator.py).
// Schema / types: rust/benchmarks/tpch/src/main.rs:267-360.
#include <hip/hip_runtime.h>
#include "util_kernels.h"

namespace bhip {

namespace gen {

__device__ inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

enum Stream : uint32_t {
    QTY = 1, PART = 2, DISC = 3, TAX = 4, SHIP = 5, COMMIT = 6, RECEIPT = 7, RFLAG = 8, SUPP = 9,
    ODATE = 10, OCUST = 11, ROT = 15
};

__device__ inline uint64_t rnd(uint64_t seed, uint32_t stream, uint64_t idx) {
    return mix(mix(seed + 0xD1342543DE82EF95ull * (uint64_t)stream) ^ (idx * 0x2545F4914F6CDD1Dull));
}

// lineitem row -> order: groups of 7 orders with 1..7 lines (28 lines), rotated per group
__device__ inline uint64_t order_of_line(uint64_t seed, uint64_t row, uint64_t n_orders) {
    const uint64_t group = row / 28u;
    const uint32_t within = (uint32_t)(row % 28u);
    const uint32_t rot = (uint32_t)(rnd(seed, ROT, group) % 7u);
    uint32_t first = 0, j = 0;
    while (j < 7) {
        const uint32_t lines = 1u + ((j + rot) % 7u);
        if (within < first + lines) break;
        first += lines;
        ++j;
    }
    return (group * 7u + j) % n_orders;
}

__device__ inline int64_t order_key(uint64_t ord, GenKeyLayout L) {
    const uint64_t k = L.sparse ? (((ord >> 3) << 5) | (ord & 7u)) : ord;
    return L.key_base + (int64_t)(k + 1u);
}

__device__ inline int32_t order_date(uint64_t seed, uint64_t order) {
    return 8035 + (int32_t)(rnd(seed, ODATE, order) % 2406u);   // 1992-01-01 .. 1998-08-02
}

}  // namespace gen

__global__ void __launch_bounds__(256)
gen_lineitem_kernel(uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_orders, uint64_t n_parts, uint64_t n_supp,
                    GenLineitemOut o, GenKeyLayout keys) {
    using namespace gen;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) {
        const uint64_t row = row0 + k;
        const uint64_t ord = order_of_line(seed, row, n_orders);
        const int32_t odate = order_date(seed, ord);
        const uint32_t qty = 1u + (uint32_t)(rnd(seed, QTY, row) % 50u);
        const uint64_t part = 1u + rnd(seed, PART, row) % n_parts;
        const uint64_t retail = 90000u + ((part / 10u) % 20001u) + 100u * (part % 1000u);   // cents
        const int32_t ship = odate + 1 + (int32_t)(rnd(seed, SHIP, row) % 121u);
        const int32_t receipt = ship + 1 + (int32_t)(rnd(seed, RECEIPT, row) % 30u);
        if (o.l_orderkey) o.l_orderkey[k] = (int32_t)order_key(ord, keys);
        if (o.l_orderkey_i64) o.l_orderkey_i64[k] = order_key(ord, keys);
        if (o.l_suppkey) o.l_suppkey[k] = (int32_t)(1u + rnd(seed, SUPP, row) % n_supp);
        if (o.l_quantity) o.l_quantity[k] = (double)qty;
        if (o.l_extendedprice) o.l_extendedprice[k] = (double)((uint64_t)qty * retail) / 100.0;
        if (o.l_discount) o.l_discount[k] = (double)(rnd(seed, DISC, row) % 11u) / 100.0;
        if (o.l_tax) o.l_tax[k] = (double)(rnd(seed, TAX, row) % 9u) / 100.0;
        if (o.l_shipdate) o.l_shipdate[k] = ship;
        if (o.l_commitdate) o.l_commitdate[k] = odate + 30 + (int32_t)(rnd(seed, COMMIT, row) % 61u);
        if (o.l_receiptdate) o.l_receiptdate[k] = receipt;
        if (o.flag_data) {
            uint8_t f = 'N';
            if (receipt <= 9298) f = (rnd(seed, RFLAG, row) & 1u) ? 'R' : 'A';
            o.flag_data[k] = f;
        }
        if (o.status_data) o.status_data[k] = ship > 9298 ? 'O' : 'F';
        // one character per row: Utf8 offsets are the row numbers
        if (o.flag_off) { o.flag_off[k] = (int32_t)k; if (k == n - 1) o.flag_off[n] = (int32_t)n; }
        if (o.status_off) { o.status_off[k] = (int32_t)k; if (k == n - 1) o.status_off[n] = (int32_t)n; }
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (o.flag_off) o.flag_off[0] = 0;
        if (o.status_off) o.status_off[0] = 0;
    }
}

__global__ void __launch_bounds__(256)
gen_orders_kernel(uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_cust, GenOrdersOut o, GenKeyLayout keys) {
    using namespace gen;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) {
        const uint64_t ord = row0 + k;
        uint64_t cust = 1u + rnd(seed, OCUST, ord) % n_cust;
        if (cust % 3u == 0) cust -= 1u;            // TPC-H: a third of the customers never order
        if (o.o_orderkey) o.o_orderkey[k] = (int32_t)order_key(ord, keys);
        if (o.o_orderkey_i64) o.o_orderkey_i64[k] = order_key(ord, keys);
        if (o.o_custkey) o.o_custkey[k] = (int32_t)cust;
        if (o.o_orderdate) o.o_orderdate[k] = order_date(seed, ord);
        if (o.o_shippriority) o.o_shippriority[k] = 0;
    }
}

static int gen_grid(const LaunchCfg& cfg, uint64_t n) {
    uint64_t g = (n + 255) / 256;
    const uint64_t cap = (uint64_t)cfg.device_cus * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_gen_lineitem(const LaunchCfg& cfg, uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_orders,
                               uint64_t n_parts, uint64_t n_supp, const GenLineitemOut& out, GenKeyLayout keys) {
    hipLaunchKernelGGL(gen_lineitem_kernel, dim3(gen_grid(cfg, n)), dim3(256), 0, cfg.stream, seed, row0, n, n_orders,
                       n_parts, n_supp, out, keys);
    return hipGetLastError();
}

hipError_t launch_gen_orders(const LaunchCfg& cfg, uint64_t seed, uint64_t row0, uint64_t n, uint64_t n_cust,
                             const GenOrdersOut& out, GenKeyLayout keys) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gen_orders_kernel, dim3(gen_grid(cfg, n)), dim3(256), 0, cfg.stream, seed, row0, n, n_cust, out, keys);
    return hipGetLastError();
}

}  // namespace bhip
