// expr.hpp — host representation of physical expressions and their lowering to VM programs.
//
// The expression kinds are those `compile_expr` can produce in the reference
// (rust/core/src/serde/physical_plan/from_proto.rs:348-364; kinds serialised at
// to_proto.rs:380-511).  `ProgramBuilder` plays the role of DataFusion's
// `create_physical_expr` + per-node arrow kernels: it type-checks against the input schema and
// emits a register program for the device interpreter (vm_isa.h).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "core.hpp"

namespace bhip {

struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;

struct Expr {
    int kind = 0;                 // bhip_expr_kind
    std::string name;             // column / operator / function / Utf8 literal value
    int dtype = 0;                // literal / cast target
    bool is_null = false;         // typed NULL literal
    bool negated = false;         // IN_LIST
    bool has_base = false, has_else = false;   // CASE
    int64_t i64 = 0;
    double f64 = 0;
    std::vector<ExprPtr> args;

    std::string to_string() const;    // Debug-style rendering, also the CSE key
};

ExprPtr parse_expr(const bhip_expr& e);
ExprPtr make_column(const std::string& name);
ExprPtr make_binary(const ExprPtr& l, const std::string& op, const ExprPtr& r);
// replace Column(name) by subst[name] (projection fusion)
ExprPtr substitute(const ExprPtr& e, const std::map<std::string, ExprPtr>& subst);
void collect_columns(const ExprPtr& e, std::vector<std::string>& out);

// the casts DataFusion's physical planner inserts (create_physical_expr behind compile_expr,
// rust/core/src/serde/physical_plan/from_proto.rs:348-364): both sides of a comparison / arithmetic node, the items of an
// IN list and the branches of a CASE are brought to one type, numeric literals are re-typed instead of cast
ExprPtr coerce_expr(const ExprPtr& e, const Schema& schema);
// BHIP_ENOTIMPL unless the device evaluates `name` with `n_args` arguments
void check_scalar_function(const std::string& name, int n_args);

int expr_type(const ExprPtr& e, const Schema& schema);
bool expr_nullable(const ExprPtr& e, const Schema& schema);

struct AggregateDesc {
    int fn;               // bhip_agg_fn
    ExprPtr arg;
    std::string name;
};

struct SortDesc {
    ExprPtr expr;
    bool descending;
    bool nulls_first;
};

// ---- program builder --------------------------------------------------------------------------
struct Operand {
    bool is_lit = false;
    int index = -1;       // virtual register or literal index
    int vclass = VC_I64;  // VC_I64 / VC_F64 / VC_BOOL
    int dtype = 0;
    bool is_utf8_col = false;   // a bare Utf8 column (usable only by string ops / keys / pass-through)
    int col = -1;
};

// a Utf8-valued expression whose Arrow form is LargeUtf8 (Field::large of the columns it comes from)
bool expr_large(const ExprPtr& e, const Schema& schema);
bool expr_binary(const ExprPtr& e, const Schema& schema);         // ... is Binary (a Binary column, sha224 .. sha512)
int sha_fn(const std::string& name);                              // digest bits of sha224 / sha256 / sha384 / sha512, 0 otherwise
// index of a string-valued scalar function (lower, upper, trim, ltrim, rtrim; evaluated as columns, utf8_exprs.cpp), -1 otherwise
int str_fn(const std::string& name);

class ProgramBuilder {
public:
    explicit ProgramBuilder(const Schema& input);
    // compile `e`; the value ends up in a virtual register (or literal)
    Operand compile(const ExprPtr& e);
    // boolean predicate fused into the kernel (AND-ed with any previous one)
    void set_predicate(const ExprPtr& e);
    // key parts (group / join / partition keys)
    // force_not_null: lay the part out without a NULL byte (join keys: NULLs are filtered out first)
    void add_key(const ExprPtr& e, bool force_not_null = false);
    // keys are only hashed, never packed: skip the 16-byte layout check
    void set_hash_only() { hash_only_ = true; }
    // accumulators; returns accumulator index (deduplicated)
    int add_acc(int kind, const Operand& src);
    // projection output
    void add_output(const ExprPtr& e);
    // force a value operand into a register (literals get materialised)
    Operand materialize(const Operand& o);

    // finish: register allocation -> ScanParams template (column pointers unbound)
    void finish(ScanParams& P);
    // per-batch binding
    static void bind(ScanParams& P, const std::vector<int>& col_map, const Batch& b, bool creates_nulls);

    const std::vector<int>& columns() const { return col_map_; }   // VM column index -> input schema index
    // every key part a plain NULL-free integer / date / timestamp column: per part {input schema index, width, byte position in the
    // packed key} (after finish()); false otherwise — such keys are packed by a small streaming kernel instead of the VM
    struct PlainKeyPart { int schema_index, width, pos; };
    bool plain_fixed_keys(std::vector<PlainKeyPart>& parts) const;
    bool creates_nulls() const { return creates_nulls_; }
    // the program can set an error flag in its ScanStatus (integer division; a Utf8 key part longer than its packed width):
    // only then does a caller have to read the status back before it hands the result on
    bool can_raise() const;
    int key_bytes() const { return key_bytes_; }
    struct KeyInfo { int pos, width, nullable, dtype; };
    const std::vector<KeyInfo>& key_info() const { return key_info_; }
    const std::vector<int>& out_dtypes() const { return out_dtypes_; }
    int n_acc() const { return (int)accs_.size(); }
    int acc_kind(int i) const { return accs_[i].kind; }

private:
    struct VInstr { VmInstr ins; bool dst_is_b; int dst_v, a_v, b_v, c_b; };   // operands as virtual regs (-1 none)
    int new_vreg(bool is_b);
    int column_index(int schema_idx);
    int literal_index(uint64_t bits);
    int strlit(const std::string& s);
    Operand load_column(int schema_idx);
    Operand emit(uint8_t op, const Operand* a, const Operand* b, bool dst_b, int vclass, int dtype, uint16_t aux = 0,
                 int c_breg = -1, uint8_t flags = 0);
    Operand compile_uncached(const ExprPtr& e);
    Operand compile_binary(const Expr& e);
    Operand compile_cast(const Operand& x, int to);
    Operand compile_string_cmp(const ExprPtr& l, const ExprPtr& r, const std::string& op);
    Operand to_bool(const Operand& o);

    const Schema& schema_;
    std::vector<int> col_map_;
    std::vector<VmLoad> loads_;            // dst = virtual reg
    std::vector<bool> load_dst_is_b_;
    std::vector<int> load_vregs_;
    std::vector<VInstr> instrs_;
    std::vector<uint64_t> lits_;
    std::string strlits_;
    std::vector<bool> vreg_is_b_;
    std::map<std::string, Operand> cse_;
    int pred_vreg_ = -1;
    struct KeyV { int kind; int src; int width; int nullable; };
    std::vector<KeyV> keys_;
    std::vector<KeyInfo> key_info_;
    int key_bytes_ = 0;
    struct AccV { int kind; int vreg; bool is_b; };
    std::vector<AccV> accs_;
    std::vector<std::pair<int, bool>> outs_;   // vreg, is_b
    std::vector<int> out_dtypes_;
    bool creates_nulls_ = false;
    int div_guard_ = -1;
    bool hash_only_ = false;
};

}  // namespace bhip
