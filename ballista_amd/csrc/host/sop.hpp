// sop.hpp — host side of the register-resident aggregate fast path (see sop.h / sop.cpp).
#pragma once
#include <vector>

#include "../sop.h"
#include "expr.hpp"

namespace bhip {

struct SopAccExpr {
    int kind;        // AccKind of accumulator i (GroupRec::acc index order)
    ExprPtr expr;    // its input expression over the scan's source schema
};

struct SopPlan {
    SopProgram prog;
    std::vector<int> col_map;                          // SOP column index -> source schema index
    std::vector<ProgramBuilder::KeyInfo> key_info;     // packed-key layout of THIS path (for the emit kernels)
};

// true when (predicate, keys, accumulators) have the chain-of-products shape
bool build_sop(const Schema& schema, const ExprPtr& predicate, const std::vector<ExprPtr>& keys,
               const std::vector<SopAccExpr>& accs, SopPlan& out);
// false when a referenced column of the batch carries NULLs (then the VM kernel runs)
// range_nulls_ok: the kernel tests the validity bitmap of range (predicate) columns (lean_kernel.h, kernels_range.hip)
bool sop_columns_bindable(const SopPlan& plan, const Batch& b, bool range_nulls_ok = false);
void bind_sop(SopPlan& plan, const Batch& b);

// the wide-load variant (lean_kernel.h): every chain factor a Float64 column, <= 2 key parts of 32 bits
// (Int32 / Date32 / Utf8 of <= 3 bytes, checked by the kernel)
bool lean_eligible(const SopProgram& prog);
// its loads need 16-byte aligned column buffers and slack behind the string bytes (owned buffers have both)
bool lean_bindable(const SopPlan& plan, const Batch& b);

}  // namespace bhip
