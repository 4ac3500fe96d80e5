// vm_isa.h — the expression-VM instruction set shared by the host compiler (host/compile.cpp)
// and the device interpreter (vm_device.h).
//
// Why a VM: a Ballista executor receives arbitrary PhysicalExpr trees over the wire
// (rust/core/src/serde/physical_plan/from_proto.rs:348-364) and DataFusion evaluates each
// node as its own arrow kernel with a materialised intermediate array.  On MI355X the
// whole Filter -> Projection -> (aggregate | join-key | partition-id) chain is ONE kernel:
// a wave-uniform interpreter walks a flat register program held in kernel arguments
// (scalar loads, scalar branches) while the per-row operands live in LDS "slots" that work
// as a dynamically indexable register file private to each thread — no intermediate ever
// touches HBM.  Every arithmetic node still rounds separately (kernels are built with
// -ffp-contract=off), so per-row values are bit-identical to the reference's.
#pragma once
#include <stdint.h>

namespace bhip {

// ---- Arrow types at the boundary (SURVEY §8(b) "Data types at the edge") -------------
enum DType : int32_t {
    DT_INT32 = 1, DT_INT64 = 2, DT_UINT8 = 3, DT_UINT64 = 4, DT_FLOAT64 = 5,
    DT_DATE32 = 6, DT_BOOLEAN = 7, DT_UTF8 = 8,
    // the other primitive types the serde can ship (rust/core/proto/ballista.proto:755-790)
    DT_INT8 = 9, DT_INT16 = 10, DT_UINT16 = 11, DT_UINT32 = 12, DT_FLOAT32 = 13, DT_DATE64 = 14,
    DT_TIMESTAMP_S = 15, DT_TIMESTAMP_MS = 16, DT_TIMESTAMP_US = 17, DT_TIMESTAMP_NS = 18,
    DT_LAST = 18,
    DT_BINARY = 20,          // boundary only (schemas): variable-length bytes in the layout of Utf8 (host/core.hpp Field::binary)
    DT_LARGE_UTF8 = 19       // boundary only (schemas): a Utf8 device column whose Arrow form has 64-bit offsets (host/core.hpp Field::large)
};

#if defined(__HIPCC__)
#define BHIP_HD __host__ __device__
#else
#define BHIP_HD
#endif

// how the VM holds a value of each type in a 64-bit slot:
//   signed integers, dates, timestamps  sign-extended int64          (VC_I64)
//   unsigned integers                   zero-extended uint64         (VC_I64; only UInt64 can exceed the int64 range)
//   Float64                             the double's bits            (VC_F64)
//   Float32                             the bits of the double that equals the float: every arithmetic result is rounded to
//                                       float before it is stored back (OP_ROUND_F32), and +,-,*,/ of two floats computed in
//                                       double and rounded once more is the correctly rounded float result (53 >= 2*24+2)
BHIP_HD inline bool dt_is_float(int t) { return t == DT_FLOAT64 || t == DT_FLOAT32; }
BHIP_HD inline bool dt_is_unsigned(int t) { return t == DT_UINT8 || t == DT_UINT16 || t == DT_UINT32 || t == DT_UINT64; }
BHIP_HD inline bool dt_is_temporal(int t) { return t == DT_DATE32 || t == DT_DATE64 || (t >= DT_TIMESTAMP_S && t <= DT_TIMESTAMP_NS); }
BHIP_HD inline bool dt_is_signed(int t) {
    return t == DT_INT8 || t == DT_INT16 || t == DT_INT32 || t == DT_INT64 || dt_is_temporal(t);
}
BHIP_HD inline bool dt_is_integer(int t) { return dt_is_signed(t) || dt_is_unsigned(t); }      // integer-valued slot
// bytes of one value in an Arrow buffer (0: Boolean / Utf8)
BHIP_HD inline int dt_width(int t) {
    switch (t) {
        case DT_INT8: case DT_UINT8: return 1;
        case DT_INT16: case DT_UINT16: return 2;
        case DT_INT32: case DT_UINT32: case DT_DATE32: case DT_FLOAT32: return 4;
        case DT_BOOLEAN: case DT_UTF8: return 0;
        default: return t >= DT_INT32 && t <= DT_LAST ? 8 : 0;
    }
}
// value range of an integer-valued type narrower than 64 bits (the 64-bit ones: see the cast ops)
BHIP_HD inline void dt_int_range(int t, int64_t& lo, int64_t& hi) {
    switch (t) {
        case DT_INT8: lo = -128; hi = 127; break;
        case DT_INT16: lo = -32768; hi = 32767; break;
        case DT_INT32: case DT_DATE32: lo = -2147483648ll; hi = 2147483647ll; break;
        case DT_UINT8: lo = 0; hi = 255; break;
        case DT_UINT16: lo = 0; hi = 65535; break;
        case DT_UINT32: lo = 0; hi = 4294967295ll; break;
        case DT_UINT64: lo = 0; hi = 0x7FFFFFFFFFFFFFFFll; break;          // as a signed source: v >= 0
        default: lo = (int64_t)0x8000000000000000ull; hi = 0x7FFFFFFFFFFFFFFFll; break;
    }
}
// two's-complement wrap of an arithmetic result to the width of the type (arrow's integer arithmetic wraps)
BHIP_HD inline uint64_t dt_wrap(int t, uint64_t v) {
    switch (t) {
        case DT_INT8: return (uint64_t)(int64_t)(int8_t)(uint8_t)v;
        case DT_INT16: return (uint64_t)(int64_t)(int16_t)(uint16_t)v;
        case DT_INT32: case DT_DATE32: return (uint64_t)(int64_t)(int32_t)(uint32_t)v;
        case DT_UINT8: return v & 0xFFull;
        case DT_UINT16: return v & 0xFFFFull;
        case DT_UINT32: return v & 0xFFFFFFFFull;
        default: return v;
    }
}

// slot image -> one value of an Arrow buffer of the type (fixed-width types)
BHIP_HD inline void dt_store(int t, void* data, int64_t i, uint64_t v) {
    switch (dt_width(t)) {
        case 1: static_cast<uint8_t*>(data)[i] = (uint8_t)v; break;
        case 2: static_cast<uint16_t*>(data)[i] = (uint16_t)v; break;
        case 4:
            if (t == DT_FLOAT32) {
                double d;
                __builtin_memcpy(&d, &v, 8);
                static_cast<float*>(data)[i] = (float)d;              // exact: the slot holds a float's value
            } else {
                static_cast<uint32_t*>(data)[i] = (uint32_t)v;
            }
            break;
        default: static_cast<uint64_t*>(data)[i] = v; break;
    }
}
// ... and back (what the VM's column load produces)
BHIP_HD inline uint64_t dt_load(int t, const void* data, int64_t i) {
    switch (t) {
        case DT_INT8: return (uint64_t)(int64_t) static_cast<const int8_t*>(data)[i];
        case DT_INT16: return (uint64_t)(int64_t) static_cast<const int16_t*>(data)[i];
        case DT_INT32: case DT_DATE32: return (uint64_t)(int64_t) static_cast<const int32_t*>(data)[i];
        case DT_UINT8: return static_cast<const uint8_t*>(data)[i];
        case DT_UINT16: return static_cast<const uint16_t*>(data)[i];
        case DT_UINT32: return static_cast<const uint32_t*>(data)[i];
        case DT_FLOAT32: { const double d = (double) static_cast<const float*>(data)[i]; uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }
        default: return static_cast<const uint64_t*>(data)[i];
    }
}

// value class of a VM slot
enum VClass : uint8_t { VC_I64 = 0, VC_F64 = 1, VC_BOOL = 2 };

enum VmOp : uint8_t {
    OP_NOP = 0,
    // 64-bit value ops: dst(V) = a(V|lit) op b(V|lit)
    OP_ADD_F64, OP_SUB_F64, OP_MUL_F64, OP_DIV_F64,
    OP_ADD_I64, OP_SUB_I64, OP_MUL_I64, OP_DIV_I64,   // DIV_I64 flags divide-by-zero
    OP_NEG_F64, OP_NEG_I64,
    // comparisons: dst(B) = a cmp b ; aux = CmpKind
    OP_CMP_F64, OP_CMP_I64, OP_CMP_U64,
    // boolean (Kleene) : dst(B) = a(B) op b(B)
    OP_AND, OP_OR, OP_NOT,
    OP_IS_NULL_V, OP_IS_NULL_B,           // dst(B) = a is null   (aux=1 -> IS NOT NULL)
    // casts
    OP_I64_TO_F64, OP_U64_TO_F64, OP_F64_TO_I64,       // F64_TO_I64: aux = target DType (range -> NULL)
    OP_I64_TO_F32, OP_U64_TO_F32,         // integer -> float, rounded ONCE (not through double)
    OP_ROUND_F32,                         // double -> the nearest float, kept as a double (Float32 arithmetic results, CAST AS Float32)
    OP_I64_NARROW,                        // range check for int -> narrower int, aux = target DType; VF_SRC_U64: the source is a UInt64
    OP_WRAP_I64,                          // two's-complement wrap to the width of aux = DType (Int32 arithmetic)
    OP_B_TO_I64, OP_I64_TO_B,
    OP_SELECT_V, OP_SELECT_B,             // dst = cond(B in `c`) ? a : b   (CASE)
    OP_MOV_V, OP_MOV_B,
    OP_LIT_B,                             // dst(B) = literal bool (aux: bit0 value, bit1 valid)
    // Utf8 column vs literal / column:  dst(B) ; a = column index ; aux = CmpKind | (lit_off<<8)
    OP_STR_CMP_LIT, OP_STR_CMP_COL, OP_STR_LIKE_LIT,
    OP_STR_IS_NULL,
    OP_STR_LEN,                           // dst(V) = byte length of Utf8 column c (octet_length)
    // unary f64 math : aux = MathFn
    OP_MATH_F64,
    OP_COUNT
};

enum CmpKind : uint8_t { CMP_EQ = 0, CMP_NE = 1, CMP_LT = 2, CMP_LE = 3, CMP_GT = 4, CMP_GE = 5 };

enum MathFn : uint8_t {
    FN_SQRT = 0, FN_ABS, FN_FLOOR, FN_CEIL, FN_ROUND, FN_TRUNC, FN_SIGNUM,
    FN_EXP, FN_LN, FN_LOG2, FN_LOG10, FN_SIN, FN_COS, FN_TAN, FN_ASIN, FN_ACOS, FN_ATAN
};

// LIKE pattern shapes the device handles (anything else -> BHIP_ENOTIMPL at compile time)
enum LikeKind : uint8_t { LIKE_EXACT = 0, LIKE_PREFIX = 1, LIKE_SUFFIX = 2, LIKE_CONTAINS = 3, LIKE_GENERAL = 4 };

struct VmInstr {            // 8 bytes, read with scalar loads
    uint8_t op;
    uint8_t dst;
    uint8_t a;
    uint8_t b;
    uint8_t c;              // third operand (SELECT condition) / column index for STR ops
    uint8_t flags;          // bit0: a is literal index, bit1: b is literal index, bit2: negate result
    uint16_t aux;
};
enum : uint8_t { VF_A_LIT = 1, VF_B_LIT = 2, VF_NEGATE = 4, VF_SRC_U64 = 8 };

// hoisted column load: global -> register -> V slot (widened to 64 bit) or B slot
struct VmLoad {
    uint8_t col;            // index into ScanParams::cols
    uint8_t dst;            // slot
    uint8_t dtype;          // DType of the column; DT_UTF8 = short-string pack: [len][bytes...] in one V slot
    uint8_t to_bool;        // 1: Boolean column -> B slot
    uint8_t width;          // DT_UTF8: bytes of the packed image (1 length byte + at most width-1 <= 7 chars)
    uint8_t pad[3];
};

struct ColumnRef {          // one Arrow column, device pointers
    const void* data;       // fixed-width values, or Utf8 bytes
    const int32_t* offsets; // Utf8 only
    const uint64_t* validity; // Arrow validity bitmap (bit i = row i valid) or nullptr
    int32_t dtype;
    int32_t data_bytes;     // Utf8: number of value bytes (bounds the 8-byte short-string loads)
};

constexpr int VM_MAX_COLS = 16;
constexpr int VM_MAX_LOADS = 16;
constexpr int VM_MAX_INSTR = 96;
constexpr int VM_MAX_LITS = 32;
constexpr int VM_STRLIT_BYTES = 192;
constexpr int VM_MAX_VSLOTS = 24;
constexpr int VM_MAX_BSLOTS = 16;
constexpr int VM_MAX_KEYPARTS = 8;
constexpr int VM_MAX_ACC = 16;
constexpr int VM_MAX_OUT = 16;

struct VmProgram {
    int32_t n_loads, n_instr, n_vslots, n_bslots;
    int32_t nullable;       // 1: some input may be NULL -> V-slot validity bytes are live
    int32_t has_utf8_loads; // 1: some load is a DT_UTF8 short-string pack (two dependent load stages)
    int32_t pad[2];
    VmLoad loads[VM_MAX_LOADS];
    VmInstr instr[VM_MAX_INSTR];
    uint64_t lits[VM_MAX_LITS];
    uint8_t strlits[VM_STRLIT_BYTES];
};

// ---- sinks ------------------------------------------------------------------------------

// group-key / join-key / partition-key part: packs into a 16-byte key
enum KeyPartKind : uint8_t { KP_VSLOT = 0, KP_BSLOT = 1, KP_UTF8_COL = 2, KP_VSLOT_F64 = 3 };
struct KeyPart {
    uint8_t kind;
    uint8_t src;            // slot or column index
    uint8_t width;          // bytes this part occupies in the packed key (incl. 1 length/null byte where needed)
    uint8_t nullable;       // reserves one leading byte: 0 = NULL, 1 = valid
};

enum AccKind : uint8_t {
    ACC_SUM_F64 = 0, ACC_SUM_I64 = 1, ACC_COUNT_VALID = 2, ACC_COUNT_ROWS = 3,
    ACC_MIN_F64 = 4, ACC_MAX_F64 = 5, ACC_MIN_I64 = 6, ACC_MAX_I64 = 7,
    ACC_COUNT_VALID_B = 8
};
struct AccSpec {
    uint8_t kind;
    uint8_t slot;           // V slot holding the input value (B slot for ACC_COUNT_VALID_B)
    uint8_t pad[2];
};

struct ScanParams {
    int64_t n_rows;
    int32_t n_cols;
    int32_t pred_slot;      // B slot of the fused predicate, -1 = none
    ColumnRef cols[VM_MAX_COLS];
    VmProgram prog;
    // key (aggregate / join / repartition sinks)
    int32_t n_keyparts;
    int32_t key_bytes;
    KeyPart keyparts[VM_MAX_KEYPARTS];
    // accumulators (aggregate sinks)
    int32_t n_acc;
    int32_t pad0;
    AccSpec acc[VM_MAX_ACC];
    // outputs (projection sink): V/B slot -> output column
    int32_t n_out;
    int32_t pad1;
    uint8_t out_slot[VM_MAX_OUT];
    uint8_t out_dtype[VM_MAX_OUT];
};

// error bits a scan kernel can raise (ScanStatus::flags)
enum : uint32_t {
    SCAN_ERR_DIV_ZERO = 1u,        // integer divide by zero
    SCAN_ERR_KEY_TOO_LONG = 2u,    // Utf8 key part longer than its packed width
    SCAN_OVERFLOW_GROUPS = 4u      // low-cardinality path saw more groups than it holds
};

} // namespace bhip
