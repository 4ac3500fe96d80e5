// vm_device.h — device side of the expression VM (see vm_isa.h for the design).
//
// Geometry: a workgroup is 256 threads (4 wave64); a tile is 256*R rows; thread `tid`
// owns rows  tile_base + r*256 + tid  (r < R), so consecutive lanes touch consecutive
// rows: global loads coalesce (a wave reads 64 consecutive values) and LDS slot accesses
// are bank-conflict free (lane stride = element size).  Every slot element is read and
// written only by its owning thread, so the interpreter needs no barriers.
#pragma once
#include <hip/hip_runtime.h>
#include "vm_isa.h"

namespace bhip {

constexpr int BLOCK = 256;

// The slot pointers carry the LDS address space in their TYPE: every access compiles to ds_read /
// ds_write.  (Plain pointers into the dynamic LDS array were treated as generic by the compiler
// once they crossed a function boundary: flat_load/flat_store, which also force s_waitcnt
// vmcnt(0) + lgkmcnt(0) and so serialise against the global loads in flight.)
#define BHIP_LDS __attribute__((address_space(3)))
// Column pointers reach the kernels inside a struct, so the compiler sees generic pointers and
// would emit flat_load (which also counts on lgkmcnt and so blocks every LDS wait on the HBM
// loads in flight).  gptr<T>() states what they are: global memory.
#define BHIP_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ inline const BHIP_GLOBAL T* gptr(const void* p) { return (const BHIP_GLOBAL T*)p; }
template <class T> __device__ inline BHIP_GLOBAL T* gptr_w(void* p) { return (BHIP_GLOBAL T*)p; }
struct __attribute__((packed)) PackedU64 { uint64_t v; };
typedef BHIP_LDS uint64_t lds_u64;
typedef BHIP_LDS uint8_t lds_u8;

struct TileLds {
    lds_u64* vals;      // [n_vslots][TILE]
    lds_u8* vvalid;     // [n_vslots][TILE]   (only when NULLS)
    lds_u8* bvals;      // [n_bslots][TILE]   bit0 = value, bit1 = known (not NULL)
};

template <int R>
__device__ inline size_t tile_lds_bytes(int n_vslots, int n_bslots, bool nulls) {
    constexpr int TILE = BLOCK * R;
    size_t b = (size_t)n_vslots * TILE * 8;
    if (nulls) b += (size_t)n_vslots * TILE;
    b += (size_t)n_bslots * TILE;
    return (b + 15) & ~(size_t)15;
}

template <int R, bool NULLS>
__device__ inline TileLds carve_tile_lds(uint8_t* base, const VmProgram& G) {
    constexpr int TILE = BLOCK * R;
    TileLds L;
    lds_u8* b = (lds_u8*)base;
    L.vals = (lds_u64*)b;
    lds_u8* p = b + G.n_vslots * TILE * 8;
    L.vvalid = p;
    if (NULLS) p += G.n_vslots * TILE;
    L.bvals = p;
    return L;
}

__device__ inline double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ inline uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }

__device__ inline bool column_valid_bit(const ColumnRef& c, int64_t row) {
    if (c.validity == nullptr) return true;
    return (gptr<uint64_t>(c.validity)[row >> 6] >> (row & 63)) & 1ull;
}

// ---- hoisted column loads ------------------------------------------------------------------
// All global loads of a tile are issued before any is consumed, in two dependent stages:
//   A: fixed-width values and, for short-string packs (DT_UTF8 loads), the two Arrow offsets;
//   B: the string bytes (one 8-byte load per row) — needs A's offsets.
// The results sit in registers (statically indexed: the loops are unrolled over LOAD_GROUP) until
// `commit` writes them to the LDS slots.  The fused-aggregate kernel issues A/B for the NEXT tile
// before it interprets the current one (register double buffering); the other kernels call
// vm_load_tile, which runs the three steps back to back.
constexpr int LOAD_GROUP = 8;

template <int R, bool NULLS>
struct LoadRegs {
    uint64_t x[LOAD_GROUP][R];
    uint8_t ok[NULLS ? LOAD_GROUP : 1][NULLS ? R : 1];   // without NULLS every loaded value is known
};

// Two passes over the group's (load, row) pairs: FIRST every global load, raw and unconditional (row index clamped to the last
// row; narrow values zero-extended, which needs no data), THEN the conversions (sign extension, float -> double, the Boolean's
// bit, the validity bit).  With the conversion next to its load — one loop — every load was followed by `s_waitcnt vmcnt(0)`:
// R x n_loads exposed round trips per tile, ~30 us per 1024-row tile in the kernels that load blocking (rocprofv3: 87 % of the
// wave cycles of scan_agg_hash_kernel waiting on memory, 26 vector loads per tile).
template <int R, bool NULLS>
__device__ inline void vm_load_issue_a(const ScanParams& P, int64_t tile_base, int g0, LoadRegs<R, NULLS>& g) {
    const VmProgram& G = P.prog;
    const int tid = threadIdx.x;
    const int64_t last_row = P.n_rows - 1;                       // (a tile is only loaded when the batch has rows)
    // (raw bits go straight into g.x / g.ok and are converted in place: no second set of registers)
    // TWO load shapes only (more cases and the compiler's chain of flag-guarded blocks makes it wait before loads that write a
    // register some other case may have written): an 8-byte value as one dwordx2; everything narrower as the ALIGNED 4-byte word
    // that holds it (1- / 2- / 4-byte values, the Boolean's byte), a Utf8 row as its two 4-byte offsets.  (An aligned word that
    // holds a valid byte lies inside the buffer's allocation granule.)
#pragma unroll
    for (int j = 0; j < LOAD_GROUP; ++j) {
        if (g0 + j < G.n_loads) {
            const VmLoad ld = G.loads[g0 + j];
            const ColumnRef& c = P.cols[ld.col];
            const int w = dt_width(ld.dtype);                                           // wave-uniform; 0: Boolean / Utf8
            const bool utf8 = ld.dtype == DT_UTF8;
            const BHIP_GLOBAL uint8_t* base = utf8 ? (const BHIP_GLOBAL uint8_t*)gptr<int32_t>(c.offsets) : gptr<uint8_t>(c.data);
            const int shift = utf8 ? 2 : (w == 2 ? 1 : w == 4 ? 2 : 0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row0 = tile_base + r * BLOCK + tid;
                const int64_t row = row0 < last_row ? row0 : last_row;
                uint32_t a, b = 0;
                if (w == 8) {
                    const uint64_t v2 = gptr<uint64_t>(c.data)[row];
                    a = (uint32_t)v2; b = (uint32_t)(v2 >> 32);
                } else {
                    const int64_t byte = ld.dtype == DT_BOOLEAN ? (row >> 3) : (row << shift);
                    const BHIP_GLOBAL uint32_t* word = (const BHIP_GLOBAL uint32_t*)(base + (byte & ~(int64_t)3));
                    a = word[0];
                    if (utf8) b = word[1];
                }
                g.x[j][r] = (uint64_t)a | ((uint64_t)b << 32);
                if (NULLS) g.ok[NULLS ? j : 0][NULLS ? r : 0] = c.validity ? gptr<uint8_t>(c.validity)[row >> 3] : (uint8_t)0xFF;     // the byte that holds the row's bit
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LOAD_GROUP; ++j) {
        if (g0 + j < G.n_loads) {
            const VmLoad ld = G.loads[g0 + j];
            const int w = dt_width(ld.dtype);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = tile_base + r * BLOCK + tid;
                const int64_t rowc = row < last_row ? row : last_row;
                const uint32_t lo = (uint32_t)g.x[j][r], b = (uint32_t)(g.x[j][r] >> 32);
                // the value's bits inside its aligned word
                const uint32_t a = w == 1 ? (lo >> (8 * (rowc & 3))) & 0xFFu
                                 : w == 2 ? (lo >> (16 * (rowc & 1))) & 0xFFFFu
                                 : ld.dtype == DT_BOOLEAN ? (lo >> (8 * ((rowc >> 3) & 3) + (rowc & 7))) & 1u
                                 : lo;
                uint64_t v;
                switch (ld.dtype) {
                    case DT_INT32:
                    case DT_DATE32: v = (uint64_t)(int64_t)(int32_t)a; break;
                    case DT_INT8: v = (uint64_t)(int64_t)(int8_t)a; break;
                    case DT_INT16: v = (uint64_t)(int64_t)(int16_t)a; break;
                    case DT_FLOAT32: v = d2u((double)__uint_as_float(a)); break;
                    case DT_UTF8: v = (uint64_t)a | ((uint64_t)(b - a) << 32); break;   // stage A of a short-string pack: offset | length << 32
                    default: v = (uint64_t)a | ((uint64_t)b << 32); break;               // Boolean, unsigned narrow types (b = 0), every 8-byte type
                }
                const bool in = row < P.n_rows;
                g.x[j][r] = in ? v : 0;
                if (NULLS) g.ok[NULLS ? j : 0][NULLS ? r : 0] = in ? (uint8_t)((g.ok[NULLS ? j : 0][NULLS ? r : 0] >> (row & 7)) & 1u) : (uint8_t)0;
            }
        }
    }
}

// stage B: [len][bytes...] image of each short string (at most width-1 <= 7 chars)
template <int R, bool NULLS>
__device__ inline void vm_load_issue_b(const ScanParams& P, int64_t tile_base, int g0, LoadRegs<R, NULLS>& g, uint32_t& err) {
    const VmProgram& G = P.prog;
    if (!G.has_utf8_loads) return;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LOAD_GROUP; ++j) {
        if (g0 + j < G.n_loads && G.loads[g0 + j].dtype == DT_UTF8) {
            const VmLoad ld = G.loads[g0 + j];
            const ColumnRef& c = P.cols[ld.col];
            const BHIP_GLOBAL uint8_t* data = gptr<uint8_t>(c.data);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = tile_base + r * BLOCK + tid;
                uint64_t packed = 0;
                if (row < P.n_rows) {
                    const uint32_t o0 = (uint32_t)g.x[j][r];
                    uint32_t len = (uint32_t)(g.x[j][r] >> 32);
                    if (len > (uint32_t)ld.width - 1u) { err |= SCAN_ERR_KEY_TOO_LONG; len = ld.width - 1u; }
                    uint64_t bytes = 0;
                    if ((int64_t)o0 + 8 <= (int64_t)c.data_bytes) {
                        bytes = ((const BHIP_GLOBAL PackedU64*)(data + o0))->v;   // one unaligned 8-byte load
                    } else {
                        for (uint32_t b = 0; b < len; ++b) bytes |= (uint64_t)data[o0 + b] << (8 * b);
                    }
                    if (len < 8) bytes &= (1ull << (8 * len)) - 1ull;
                    packed = (uint64_t)len | (bytes << 8);
                }
                g.x[j][r] = packed;
            }
        }
    }
}

template <int R, bool NULLS>
__device__ inline void vm_load_commit(const ScanParams& P, const TileLds& L, int g0, const LoadRegs<R, NULLS>& g) {
    constexpr int TILE = BLOCK * R;
    const VmProgram& G = P.prog;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < LOAD_GROUP; ++j) {
        if (g0 + j < G.n_loads) {
            const VmLoad ld = G.loads[g0 + j];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = r * BLOCK + tid;
                const uint8_t k = NULLS ? g.ok[NULLS ? j : 0][NULLS ? r : 0] : (uint8_t)1;
                if (ld.to_bool) {
                    L.bvals[ld.dst * TILE + idx] = (uint8_t)((g.x[j][r] & k) | (k << 1));
                } else {
                    L.vals[ld.dst * TILE + idx] = g.x[j][r];
                    if (NULLS) L.vvalid[ld.dst * TILE + idx] = k;
                }
            }
        }
    }
}

// loads [from, n_loads) of a tile, blocking (issue A, issue B, commit per group of LOAD_GROUP)
template <int R, bool NULLS>
__device__ inline void vm_load_tile(const ScanParams& P, const TileLds& L, int64_t tile_base, uint32_t& err, int from = 0) {
    for (int g0 = from; g0 < P.prog.n_loads; g0 += LOAD_GROUP) {
        LoadRegs<R, NULLS> g;
        vm_load_issue_a<R, NULLS>(P, tile_base, g0, g);
        vm_load_issue_b<R, NULLS>(P, tile_base, g0, g, err);
        vm_load_commit<R, NULLS>(P, L, g0, g);
    }
}

// ---- Utf8 helpers --------------------------------------------------------------------
// three-way compare of a row's string with `lit[0..litlen)` (bytes; UTF-8 byte order)
template <class PA, class PB>
__device__ inline int str_cmp3(PA s, int len, PB t, int tlen) {
    const int m = len < tlen ? len : tlen;
    for (int i = 0; i < m; ++i) {
        const int d = (int)s[i] - (int)t[i];
        if (d != 0) return d < 0 ? -1 : 1;
    }
    return len < tlen ? -1 : (len > tlen ? 1 : 0);
}

__device__ inline bool cmp3_to_bool(int c, int kind) {
    switch (kind) {
        case CMP_EQ: return c == 0;
        case CMP_NE: return c != 0;
        case CMP_LT: return c < 0;
        case CMP_LE: return c <= 0;
        case CMP_GT: return c > 0;
        default: return c >= 0;
    }
}

// general LIKE: '%' = any sequence of characters, '_' = exactly one (UTF-8) character; iterative matching with one
// backtrack point (the last '%'), which is complete for this pattern language
template <class PA, class PB>
__device__ inline bool str_like_general(PA s, int len, PB p, int plen) {
    int si = 0, pi = 0, star_p = -1, star_s = 0;
    while (si < len) {
        if (pi < plen && p[pi] == '%') { star_p = ++pi; star_s = si; continue; }
        if (pi < plen && p[pi] == '_') {
            ++pi; ++si;
            while (si < len && (s[si] & 0xC0) == 0x80) ++si;                  // the rest of a multi-byte character
            continue;
        }
        if (pi < plen && p[pi] == s[si]) { ++pi; ++si; continue; }
        if (star_p < 0) return false;
        pi = star_p;                                                        // let the last '%' take one more character
        ++star_s;
        while (star_s < len && (s[star_s] & 0xC0) == 0x80) ++star_s;
        si = star_s;
    }
    while (pi < plen && p[pi] == '%') ++pi;
    return pi == plen;
}

template <class PA, class PB>
__device__ inline bool str_like(PA s, int len, PB p, int plen, int kind) {
    if (kind == LIKE_GENERAL) return str_like_general(s, len, p, plen);
    if (kind == LIKE_EXACT) return len == plen && str_cmp3(s, len, p, plen) == 0;
    if (len < plen) return false;
    if (kind == LIKE_PREFIX) return str_cmp3(s, plen, p, plen) == 0;
    if (kind == LIKE_SUFFIX) return str_cmp3(s + (len - plen), plen, p, plen) == 0;
    for (int i = 0; i + plen <= len; ++i)
        if (str_cmp3(s + i, plen, p, plen) == 0) return true;
    return false;
}

template <class T>
__device__ inline bool cmp_vals(T a, T b, int kind) {
    switch (kind) {
        case CMP_EQ: return a == b;
        case CMP_NE: return a != b;
        case CMP_LT: return a < b;
        case CMP_LE: return a <= b;
        case CMP_GT: return a > b;
        default: return a >= b;
    }
}

__device__ __attribute__((noinline)) double math_f64(double x, int fn) {
    switch (fn) {
        case FN_SQRT: return sqrt(x);
        case FN_ABS: return fabs(x);
        case FN_FLOOR: return floor(x);
        case FN_CEIL: return ceil(x);
        case FN_ROUND: return round(x);
        case FN_TRUNC: return trunc(x);
        case FN_SIGNUM: return isnan(x) ? x : copysign(1.0, x);
        case FN_EXP: return exp(x);
        case FN_LN: return log(x);
        case FN_LOG2: return log2(x);
        case FN_LOG10: return log10(x);
        case FN_SIN: return sin(x);
        case FN_COS: return cos(x);
        case FN_TAN: return tan(x);
        case FN_ASIN: return asin(x);
        case FN_ACOS: return acos(x);
        default: return atan(x);
    }
}

__device__ inline bool int_in_range(int64_t v, int dtype) {
    int64_t lo, hi;
    dt_int_range(dtype, lo, hi);
    return v >= lo && v <= hi;
}

// ---- the interpreter -----------------------------------------------------------------
// `err` accumulates SCAN_ERR_* bits for this thread (caller ORs them into global status).
template <int R, bool NULLS>
__device__ inline void vm_execute(const ScanParams& P, const TileLds& L, int64_t tile_base, uint32_t& err) {
    constexpr int TILE = BLOCK * R;
    const VmProgram& G = P.prog;
    const int tid = threadIdx.x;

#define BHIP_FOR_R _Pragma("unroll") for (int r = 0; r < R; ++r)
#define IDX (r * BLOCK + tid)
// operand fetch without branches: a literal operand still reads slot 0 and the (wave-uniform)
// select picks the literal — one v_cndmask pair instead of a scalar branch per row
#define VA (a_lit ? lit_a : L.vals[slot_a * TILE + IDX])
#define VB (b_lit ? lit_b : L.vals[slot_b * TILE + IDX])
#define KA ((uint8_t)(NULLS ? (a_lit ? (uint8_t)1 : L.vvalid[slot_a * TILE + IDX]) : (uint8_t)1))
#define KB ((uint8_t)(NULLS ? (b_lit ? (uint8_t)1 : L.vvalid[slot_b * TILE + IDX]) : (uint8_t)1))
#define PUT_V(val, known)                                   \
    do {                                                    \
        L.vals[I.dst * TILE + IDX] = (val);                 \
        if (NULLS) L.vvalid[I.dst * TILE + IDX] = (known);  \
    } while (0)
#define PUT_B(val, known)                                                            \
    do {                                                                             \
        const uint8_t k_ = (known) ? 1 : 0;                                          \
        const uint8_t v_ = ((val) ? 1 : 0) & k_;                                     \
        L.bvals[I.dst * TILE + IDX] = (uint8_t)(v_ | (k_ << 1));                     \
    } while (0)

    for (int pc = 0; pc < G.n_instr; ++pc) {
        const VmInstr I = G.instr[pc];
        const bool a_lit = I.flags & VF_A_LIT, b_lit = I.flags & VF_B_LIT;
        const uint64_t lit_a = G.lits[a_lit ? I.a : 0];
        const uint64_t lit_b = G.lits[b_lit ? I.b : 0];
        const int slot_a = a_lit ? 0 : I.a, slot_b = b_lit ? 0 : I.b;
        switch (I.op) {
            case OP_ADD_F64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(d2u(u2d(VA) + u2d(VB)), k); } break;
            case OP_SUB_F64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(d2u(u2d(VA) - u2d(VB)), k); } break;
            case OP_MUL_F64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(d2u(u2d(VA) * u2d(VB)), k); } break;
            case OP_DIV_F64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(d2u(u2d(VA) / u2d(VB)), k); } break;
            case OP_ADD_I64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(VA + VB, k); } break;
            case OP_SUB_I64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(VA - VB, k); } break;
            case OP_MUL_I64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_V(VA * VB, k); } break;
            case OP_DIV_I64:
                BHIP_FOR_R {
                    const uint8_t k = KA & KB;
                    const int64_t a = (int64_t)VA, b = (int64_t)VB;
                    // a row the fused predicate (slot c) rejects was never evaluated by the reference
                    const bool live = (tile_base + IDX < P.n_rows) && k &&
                                      (I.c == 0xFF || (L.bvals[I.c * TILE + IDX] & 1));
                    if (b == 0 && live) err |= SCAN_ERR_DIV_ZERO;
                    uint64_t q;
                    if (I.flags & VF_SRC_U64) q = b == 0 ? 0 : VA / VB;                      // UInt64 / UInt64
                    else q = (uint64_t)((b == 0 || (a == INT64_MIN && b == -1)) ? 0 : a / b);
                    PUT_V(q, k);
                }
                break;
            case OP_NEG_F64: BHIP_FOR_R { PUT_V(d2u(-u2d(VA)), KA); } break;
            case OP_NEG_I64: BHIP_FOR_R { PUT_V((uint64_t)0 - VA, KA); } break;
            case OP_CMP_F64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_B(cmp_vals<double>(u2d(VA), u2d(VB), I.aux), k); } break;
            case OP_CMP_I64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_B(cmp_vals<int64_t>((int64_t)VA, (int64_t)VB, I.aux), k); } break;
            case OP_CMP_U64: BHIP_FOR_R { const uint8_t k = KA & KB; PUT_B(cmp_vals<uint64_t>(VA, VB, I.aux), k); } break;
            case OP_AND:
                BHIP_FOR_R {
                    const uint8_t a = L.bvals[I.a * TILE + IDX], b = L.bvals[I.b * TILE + IDX];
                    const uint8_t va = a & 1, ka = a >> 1, vb = b & 1, kb = b >> 1;
                    const uint8_t v = va & vb;
                    const uint8_t k = (ka & kb) | (ka & (va ^ 1)) | (kb & (vb ^ 1));
                    L.bvals[I.dst * TILE + IDX] = (uint8_t)(v | (k << 1));
                }
                break;
            case OP_OR:
                BHIP_FOR_R {
                    const uint8_t a = L.bvals[I.a * TILE + IDX], b = L.bvals[I.b * TILE + IDX];
                    const uint8_t va = a & 1, ka = a >> 1, vb = b & 1, kb = b >> 1;
                    const uint8_t v = va | vb;
                    const uint8_t k = (ka & kb) | va | vb;
                    L.bvals[I.dst * TILE + IDX] = (uint8_t)(v | (k << 1));
                }
                break;
            case OP_NOT:
                BHIP_FOR_R {
                    const uint8_t a = L.bvals[I.a * TILE + IDX];
                    const uint8_t k = a >> 1;
                    L.bvals[I.dst * TILE + IDX] = (uint8_t)((((a & 1) ^ 1) & k) | (k << 1));
                }
                break;
            case OP_IS_NULL_V: BHIP_FOR_R { const uint8_t k = KA; PUT_B(I.aux ? k : (k ^ 1), 1); } break;
            case OP_IS_NULL_B:
                BHIP_FOR_R { const uint8_t k = L.bvals[I.a * TILE + IDX] >> 1; PUT_B(I.aux ? k : (k ^ 1), 1); }
                break;
            case OP_I64_TO_F64: BHIP_FOR_R { PUT_V(d2u((double)(int64_t)VA), KA); } break;
            case OP_U64_TO_F64: BHIP_FOR_R { PUT_V(d2u((double)VA), KA); } break;
            case OP_F64_TO_I64:
                BHIP_FOR_R {
                    const double d = u2d(VA);
                    const double t = trunc(d);
                    bool ok;
                    uint64_t out = 0;
                    if (I.aux == DT_UINT64) {
                        ok = (t >= 0.0) && (t < 18446744073709551616.0);
                        if (ok) out = (uint64_t)t;
                    } else {
                        ok = (t >= -9223372036854775808.0) && (t < 9223372036854775808.0);
                        if (ok) { const int64_t v = (int64_t)t; ok = int_in_range(v, I.aux); out = ok ? (uint64_t)v : 0; }
                    }
                    PUT_V(out, (uint8_t)(KA & (ok ? 1 : 0)));
                }
                break;
            case OP_I64_TO_F32: BHIP_FOR_R { PUT_V(d2u((double)(float)(int64_t)VA), KA); } break;
            case OP_U64_TO_F32: BHIP_FOR_R { PUT_V(d2u((double)(float)VA), KA); } break;
            case OP_ROUND_F32: BHIP_FOR_R { PUT_V(d2u((double)(float)u2d(VA)), KA); } break;
            case OP_I64_NARROW:
                BHIP_FOR_R {
                    const int64_t v = (int64_t)VA;
                    // a UInt64 source above 2^63 reads as negative: out of range for every other integer type
                    const bool ok = (I.flags & VF_SRC_U64) ? (v >= 0 && int_in_range(v, I.aux)) : int_in_range(v, I.aux);
                    PUT_V(ok ? (uint64_t)v : 0, (uint8_t)(KA & (ok ? 1 : 0)));
                }
                break;
            case OP_WRAP_I64: BHIP_FOR_R { PUT_V(dt_wrap(I.aux, VA), KA); } break;
            case OP_B_TO_I64:
                BHIP_FOR_R { const uint8_t a = L.bvals[I.a * TILE + IDX]; PUT_V((uint64_t)(a & 1), (uint8_t)(a >> 1)); }
                break;
            case OP_I64_TO_B: BHIP_FOR_R { PUT_B(VA != 0, KA); } break;
            case OP_SELECT_V:
                BHIP_FOR_R {
                    const bool c = L.bvals[I.c * TILE + IDX] & 1;
                    const uint64_t v = c ? VA : VB;
                    const uint8_t k = c ? KA : KB;
                    PUT_V(v, k);
                }
                break;
            case OP_SELECT_B:
                BHIP_FOR_R {
                    const bool c = L.bvals[I.c * TILE + IDX] & 1;
                    L.bvals[I.dst * TILE + IDX] = c ? L.bvals[I.a * TILE + IDX] : L.bvals[I.b * TILE + IDX];
                }
                break;
            case OP_MOV_V: BHIP_FOR_R { PUT_V(VA, (uint8_t)(I.aux ? 0 : KA)); } break;   // aux=1: typed NULL
            case OP_MOV_B: BHIP_FOR_R { L.bvals[I.dst * TILE + IDX] = L.bvals[I.a * TILE + IDX]; } break;
            case OP_LIT_B: BHIP_FOR_R { L.bvals[I.dst * TILE + IDX] = (uint8_t)(I.aux & 3); } break;
            case OP_STR_CMP_LIT:
            case OP_STR_LIKE_LIT: {
                const ColumnRef& c = P.cols[I.c];
                const uint8_t* lit = G.strlits + I.a;
                BHIP_FOR_R {
                    const int64_t row = tile_base + IDX;
                    bool v = false, k = false;
                    if (row < P.n_rows) {
                        k = !NULLS || column_valid_bit(c, row);
                        const int32_t o0 = gptr<int32_t>(c.offsets)[row], o1 = gptr<int32_t>(c.offsets)[row + 1];
                        const BHIP_GLOBAL uint8_t* s = gptr<uint8_t>(c.data) + o0;
                        if (I.op == OP_STR_CMP_LIT) v = cmp3_to_bool(str_cmp3(s, o1 - o0, lit, I.b), I.aux);
                        else v = str_like(s, o1 - o0, lit, I.b, I.aux);
                        if (I.flags & VF_NEGATE) v = !v;
                    }
                    PUT_B(v, k);
                }
            } break;
            case OP_STR_CMP_COL: {
                const ColumnRef& ca = P.cols[I.a];
                const ColumnRef& cb = P.cols[I.b];
                BHIP_FOR_R {
                    const int64_t row = tile_base + IDX;
                    bool v = false, k = false;
                    if (row < P.n_rows) {
                        k = !NULLS || (column_valid_bit(ca, row) && column_valid_bit(cb, row));
                        const int32_t a0 = gptr<int32_t>(ca.offsets)[row], a1 = gptr<int32_t>(ca.offsets)[row + 1];
                        const int32_t b0 = gptr<int32_t>(cb.offsets)[row], b1 = gptr<int32_t>(cb.offsets)[row + 1];
                        v = cmp3_to_bool(str_cmp3(gptr<uint8_t>(ca.data) + a0, a1 - a0, gptr<uint8_t>(cb.data) + b0, b1 - b0), I.aux);
                    }
                    PUT_B(v, k);
                }
            } break;
            case OP_STR_IS_NULL: {
                const ColumnRef& c = P.cols[I.c];
                BHIP_FOR_R {
                    const int64_t row = tile_base + IDX;
                    const bool k = row < P.n_rows ? column_valid_bit(c, row) : true;
                    PUT_B(I.aux ? k : !k, 1);
                }
            } break;
            case OP_STR_LEN: {
                const ColumnRef& c = P.cols[I.c];
                BHIP_FOR_R {
                    const int64_t row = tile_base + IDX;
                    const bool in = row < P.n_rows;
                    const uint64_t len = in ? (uint64_t)(gptr<int32_t>(c.offsets)[row + 1] - gptr<int32_t>(c.offsets)[row]) : 0;
                    PUT_V(len, (uint8_t)(in ? column_valid_bit(c, row) : 1));
                }
            } break;
            case OP_MATH_F64: BHIP_FOR_R { PUT_V(d2u(math_f64(u2d(VA), I.aux)), KA); } break;
            default: break;
        }
    }
#undef BHIP_FOR_R
#undef IDX
#undef VA
#undef VB
#undef KA
#undef KB
#undef PUT_V
#undef PUT_B
}

// ---- packed keys ------------------------------------------------------------------------
// A key tuple packs into 16 bytes (k0 = bytes 0-7, k1 = bytes 8-15, little-endian).
// Parts are laid out in order; a nullable part is preceded by one byte (0 = NULL, 1 = valid,
// value bytes zeroed when NULL); a Utf8 part is [len][bytes...] padded with zeros.
struct Key128 {
    uint64_t k0, k1;
    __device__ bool operator==(const Key128& o) const { return k0 == o.k0 && k1 == o.k1; }
};

__device__ inline void key_put(Key128& k, int pos, uint64_t v, int width) {
    // little-endian insert of the low `width` bytes of v at byte position pos
    if (width < 8) v &= ((1ull << (8 * width)) - 1ull);
    if (pos < 8) {
        k.k0 |= v << (8 * pos);
        if (pos + width > 8) k.k1 |= v >> (8 * (8 - pos));
    } else {
        k.k1 |= v << (8 * (pos - 8));
    }
}

template <int R, bool NULLS>
__device__ inline Key128 pack_key(const ScanParams& P, const TileLds& L, int64_t tile_base, int r, uint32_t& err) {
    constexpr int TILE = BLOCK * R;
    const int idx = r * BLOCK + threadIdx.x;
    const int64_t row = tile_base + idx;
    Key128 k{0, 0};
    int pos = 0;
    for (int i = 0; i < P.n_keyparts; ++i) {
        const KeyPart kp = P.keyparts[i];
        int width = kp.width;
        bool valid = true;
        if (kp.kind == KP_VSLOT || kp.kind == KP_VSLOT_F64) {
            if (NULLS) valid = L.vvalid[kp.src * TILE + idx];
            if (kp.nullable) { key_put(k, pos, valid ? 1 : 0, 1); pos += 1; width -= 1; }
            if (valid) key_put(k, pos, L.vals[kp.src * TILE + idx], width);
        } else if (kp.kind == KP_BSLOT) {
            const uint8_t b = L.bvals[kp.src * TILE + idx];
            valid = b >> 1;
            if (kp.nullable) { key_put(k, pos, valid ? 1 : 0, 1); pos += 1; width -= 1; }
            if (valid) key_put(k, pos, b & 1, width);
        } else {  // KP_UTF8_COL
            const ColumnRef& c = P.cols[kp.src];
            if (row < P.n_rows) {
                if (NULLS) valid = column_valid_bit(c, row);
                if (kp.nullable) { key_put(k, pos, valid ? 1 : 0, 1); pos += 1; width -= 1; }
                if (valid) {
                    const int32_t o0 = gptr<int32_t>(c.offsets)[row], o1 = gptr<int32_t>(c.offsets)[row + 1];
                    int len = o1 - o0;
                    if (len > width - 1) { err |= SCAN_ERR_KEY_TOO_LONG; len = width - 1; }
                    key_put(k, pos, (uint64_t)len, 1);
                    const BHIP_GLOBAL uint8_t* s = gptr<uint8_t>(c.data) + o0;
                    if ((int64_t)o0 + 16 <= (int64_t)c.data_bytes) {
                        // two unaligned 8-byte loads, cut to the length (a load per byte is a chain of `len` dependent round
                        // trips: 40 us for the one tile per workgroup of a small aggregate)
                        uint64_t b0 = ((const BHIP_GLOBAL PackedU64*)s)->v, b1 = ((const BHIP_GLOBAL PackedU64*)(s + 8))->v;
                        if (len < 8) { b0 &= (1ull << (8 * len)) - 1ull; b1 = 0; }
                        else if (len < 16) b1 &= (1ull << (8 * (len - 8))) - 1ull;
                        const int sh = 8 * (pos + 1);              // the key's bits from here on are still zero
                        if (sh < 64) { k.k0 |= b0 << sh; k.k1 |= (b0 >> (64 - sh)) | (b1 << sh); }
                        else k.k1 |= b0 << (sh - 64);
                    } else {
                        for (int j = 0; j < len; ++j) key_put(k, pos + 1 + j, s[j], 1);
                    }
                }
            } else if (kp.nullable) { pos += 1; width -= 1; }
        }
        pos += width;
    }
    return k;
}

__device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline uint64_t hash_key(const Key128& k) { return mix64(mix64(k.k0) ^ k.k1); }

}  // namespace bhip
