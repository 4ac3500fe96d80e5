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
    std::vector<int> col_map;   // SOP column index -> source schema index
};

// true when (predicate, keys, accumulators) have the chain-of-products shape; `key_info` / `key_bytes`
// are the packed-key layout of the VM program so both kernels emit identical GroupRec keys
bool build_sop(const Schema& schema, const ExprPtr& predicate, const std::vector<ExprPtr>& keys,
               const std::vector<ProgramBuilder::KeyInfo>& key_info, int key_bytes,
               const std::vector<SopAccExpr>& accs, SopPlan& out);
// fill in the column pointers of one batch; false when a referenced column carries NULLs
bool bind_sop(SopPlan& plan, const Batch& b);

}  // namespace bhip
