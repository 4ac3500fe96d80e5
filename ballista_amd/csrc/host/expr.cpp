// expr.cpp — expression parsing, typing and lowering to VM programs (see expr.hpp).
#include "expr.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <sstream>

namespace bhip {

// ---- construction ------------------------------------------------------------------------------
static const char* BINARY_OPS[] = {"And", "Or", "Eq", "NotEq", "LtEq", "Lt", "Gt", "GtEq", "Plus", "Minus",
                                   "Multiply", "Divide", "Like", "NotLike"};
static bool known_binary(const std::string& op) {
    for (auto o : BINARY_OPS)
        if (op == o) return true;
    return false;
}
static bool is_compare(const std::string& op) {
    return op == "Eq" || op == "NotEq" || op == "Lt" || op == "LtEq" || op == "Gt" || op == "GtEq";
}
static bool is_arith(const std::string& op) { return op == "Plus" || op == "Minus" || op == "Multiply" || op == "Divide"; }
static int cmp_kind(const std::string& op) {
    if (op == "Eq") return CMP_EQ;
    if (op == "NotEq") return CMP_NE;
    if (op == "Lt") return CMP_LT;
    if (op == "LtEq") return CMP_LE;
    if (op == "Gt") return CMP_GT;
    return CMP_GE;
}
static int flip_cmp(int k) {
    switch (k) {
        case CMP_LT: return CMP_GT;
        case CMP_LE: return CMP_GE;
        case CMP_GT: return CMP_LT;
        case CMP_GE: return CMP_LE;
        default: return k;
    }
}
static const char* MATH_FNS[] = {"sqrt", "abs", "floor", "ceil", "round", "trunc", "signum", "exp", "ln",
                                 "log2", "log10", "sin", "cos", "tan", "asin", "acos", "atan"};
static int math_fn(const std::string& n) {
    for (int i = 0; i < (int)(sizeof(MATH_FNS) / sizeof(MATH_FNS[0])); ++i)
        if (n == MATH_FNS[i]) return i;
    return -1;
}

int str_fn(const std::string& name) {
    static const char* names[] = {"lower", "upper", "trim", "ltrim", "rtrim"};
    for (int i = 0; i < 5; ++i)
        if (name == names[i]) return i;
    return -1;
}

int sha_fn(const std::string& name) {
    return name == "sha224" ? 224 : name == "sha256" ? 256 : name == "sha384" ? 384 : name == "sha512" ? 512 : 0;
}

void check_scalar_function(const std::string& name, int n_args) {
    if (math_fn(name) < 0 && str_fn(name) < 0 && sha_fn(name) == 0 && name != "octet_length")
        fail(BHIP_ENOTIMPL, "scalar function '" + name + "' is not supported");
    if (n_args != 1) fail(BHIP_EINVAL, "scalar function takes one argument");
}

ExprPtr make_column(const std::string& name) {
    auto e = std::make_shared<Expr>();
    e->kind = BHIP_EXPR_COLUMN;
    e->name = name;
    return e;
}

ExprPtr make_binary(const ExprPtr& l, const std::string& op, const ExprPtr& r) {
    auto e = std::make_shared<Expr>();
    e->kind = BHIP_EXPR_BINARY;
    e->name = op;
    e->args = {l, r};
    return e;
}

ExprPtr parse_expr(const bhip_expr& pe) {
    if (!pe.nodes || pe.n_nodes <= 0) fail(BHIP_EINVAL, "empty expression");
    std::vector<ExprPtr> stack;
    auto pop = [&]() {
        if (stack.empty()) fail(BHIP_EINVAL, "malformed postfix expression (stack underflow)");
        ExprPtr x = stack.back();
        stack.pop_back();
        return x;
    };
    for (int i = 0; i < pe.n_nodes; ++i) {
        const bhip_expr_node& n = pe.nodes[i];
        auto e = std::make_shared<Expr>();
        e->kind = n.kind;
        switch (n.kind) {
            case BHIP_EXPR_COLUMN:
                if (!n.name) fail(BHIP_EINVAL, "column node without a name");
                e->name = n.name;
                break;
            case BHIP_EXPR_LITERAL:
                e->dtype = n.dtype;
                e->is_null = (n.flags & 1) != 0;
                e->i64 = n.i64;
                e->f64 = n.f64;
                if (n.dtype == DT_UTF8 && !e->is_null) {
                    if (!n.name) fail(BHIP_EINVAL, "Utf8 literal without a value");
                    e->name = n.name;
                }
                if (n.dtype < DT_INT32 || n.dtype > DT_LAST) fail(BHIP_ENOTIMPL, "literal of unsupported type");
                if (n.dtype == DT_FLOAT32) e->f64 = (double)(float)e->f64;
                break;
            case BHIP_EXPR_BINARY: {
                if (!n.name || !known_binary(n.name))
                    fail(BHIP_EINVAL, std::string("Unsupported binary operator '") + (n.name ? n.name : "") + "'");
                e->name = n.name;
                ExprPtr r = pop(), l = pop();
                e->args = {l, r};
            } break;
            case BHIP_EXPR_CAST:
                e->dtype = n.dtype;
                if (n.dtype < DT_INT32 || n.dtype > DT_LAST) fail(BHIP_ENOTIMPL, "cast to unsupported type");
                e->args = {pop()};
                break;
            case BHIP_EXPR_NOT:
            case BHIP_EXPR_IS_NULL:
            case BHIP_EXPR_IS_NOT_NULL:
            case BHIP_EXPR_NEGATIVE:
                e->args = {pop()};
                break;
            case BHIP_EXPR_IN_LIST: {
                e->negated = (n.flags & 1) != 0;
                if (n.n_args < 1) fail(BHIP_EINVAL, "IN list without items");
                std::vector<ExprPtr> items(n.n_args);
                for (int k = n.n_args - 1; k >= 0; --k) items[k] = pop();
                e->args.push_back(pop());
                for (auto& it : items) e->args.push_back(it);
            } break;
            case BHIP_EXPR_CASE: {
                e->has_base = (n.flags & 1) != 0;
                e->has_else = (n.flags & 2) != 0;
                if (n.n_args < 1) fail(BHIP_EINVAL, "CASE without WHEN");
                const int total = (e->has_base ? 1 : 0) + 2 * n.n_args + (e->has_else ? 1 : 0);
                std::vector<ExprPtr> items(total);
                for (int k = total - 1; k >= 0; --k) items[k] = pop();
                e->args = items;
            } break;
            case BHIP_EXPR_SCALAR_FN: {
                check_scalar_function(n.name ? n.name : "", n.n_args);
                e->name = n.name;
                e->args = {pop()};
            } break;
            default: fail(BHIP_ENOTIMPL, "unsupported expression kind " + std::to_string(n.kind));
        }
        stack.push_back(e);
    }
    if (stack.size() != 1) fail(BHIP_EINVAL, "malformed postfix expression (left-over operands)");
    return stack[0];
}

std::string Expr::to_string() const {
    std::ostringstream o;
    switch (kind) {
        case BHIP_EXPR_COLUMN: o << name; break;
        case BHIP_EXPR_LITERAL:
            if (is_null) o << "NULL:" << dtype_name(dtype);
            else if (dtype == DT_FLOAT64) { char b[40]; snprintf(b, sizeof b, "%.17g", f64); o << b; }
            else if (dtype == DT_FLOAT32) { char b[40]; snprintf(b, sizeof b, "%.9g", f64); o << "Float32(" << b << ")"; }
            else if (dtype == DT_UTF8) o << "'" << name << "'";
            else if (dtype == DT_BOOLEAN) o << (i64 ? "true" : "false");
            else o << dtype_name(dtype) << "(" << i64 << ")";
            break;
        case BHIP_EXPR_BINARY: o << "(" << args[0]->to_string() << " " << name << " " << args[1]->to_string() << ")"; break;
        case BHIP_EXPR_CAST: o << "CAST(" << args[0]->to_string() << " AS " << dtype_name(dtype) << ")"; break;
        case BHIP_EXPR_NOT: o << "NOT " << args[0]->to_string(); break;
        case BHIP_EXPR_IS_NULL: o << args[0]->to_string() << " IS NULL"; break;
        case BHIP_EXPR_IS_NOT_NULL: o << args[0]->to_string() << " IS NOT NULL"; break;
        case BHIP_EXPR_NEGATIVE: o << "(- " << args[0]->to_string() << ")"; break;
        case BHIP_EXPR_IN_LIST:
            o << args[0]->to_string() << (negated ? " NOT IN (" : " IN (");
            for (size_t i = 1; i < args.size(); ++i) o << (i > 1 ? ", " : "") << args[i]->to_string();
            o << ")";
            break;
        case BHIP_EXPR_CASE: {
            o << "CASE ";
            size_t i = 0;
            if (has_base) o << args[i++]->to_string() << " ";
            const size_t end = args.size() - (has_else ? 1 : 0);
            for (; i + 1 < end; i += 2)
                o << "WHEN " << args[i]->to_string() << " THEN " << args[i + 1]->to_string() << " ";
            if (has_else) o << "ELSE " << args.back()->to_string() << " ";
            o << "END";
        } break;
        case BHIP_EXPR_SCALAR_FN: o << name << "(" << args[0]->to_string() << ")"; break;
        default: o << "?";
    }
    return o.str();
}

ExprPtr substitute(const ExprPtr& e, const std::map<std::string, ExprPtr>& subst) {
    if (e->kind == BHIP_EXPR_COLUMN) {
        auto it = subst.find(e->name);
        if (it == subst.end()) fail(BHIP_EINVAL, "No field named '" + e->name + "'");
        return it->second;
    }
    if (e->args.empty()) return e;
    auto c = std::make_shared<Expr>(*e);
    for (auto& a : c->args) a = substitute(a, subst);
    return c;
}

void collect_columns(const ExprPtr& e, std::vector<std::string>& out) {
    if (e->kind == BHIP_EXPR_COLUMN) {
        if (std::find(out.begin(), out.end(), e->name) == out.end()) out.push_back(e->name);
        return;
    }
    for (auto& a : e->args) collect_columns(a, out);
}

static void case_layout(const Expr& e, size_t& first_when, size_t& n_pairs) {
    first_when = e.has_base ? 1 : 0;
    n_pairs = (e.args.size() - first_when - (e.has_else ? 1 : 0)) / 2;
}

int expr_type(const ExprPtr& e, const Schema& schema) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = schema.index_of(e->name);
            if (i < 0) fail(BHIP_EINVAL, "No field named '" + e->name + "'");
            return schema.fields[i].dtype;
        }
        case BHIP_EXPR_LITERAL: return e->dtype;
        case BHIP_EXPR_BINARY: return is_arith(e->name) ? expr_type(e->args[0], schema) : (int)DT_BOOLEAN;
        case BHIP_EXPR_CAST: return e->dtype;
        case BHIP_EXPR_NOT:
        case BHIP_EXPR_IS_NULL:
        case BHIP_EXPR_IS_NOT_NULL:
        case BHIP_EXPR_IN_LIST: return DT_BOOLEAN;
        case BHIP_EXPR_NEGATIVE: return expr_type(e->args[0], schema);
        case BHIP_EXPR_CASE: {
            size_t fw, np;
            case_layout(*e, fw, np);
            return expr_type(e->args[fw + 1], schema);
        }
        case BHIP_EXPR_SCALAR_FN: return (str_fn(e->name) >= 0 || sha_fn(e->name)) ? DT_UTF8 : (e->name == "octet_length" ? DT_INT32 : DT_FLOAT64);
        default: fail(BHIP_ENOTIMPL, "unsupported expression kind");
    }
}

bool expr_large(const ExprPtr& e, const Schema& schema) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = schema.index_of(e->name);
            return i >= 0 && schema.fields[i].large;
        }
        case BHIP_EXPR_SCALAR_FN: return str_fn(e->name) >= 0 && expr_large(e->args[0], schema);
        case BHIP_EXPR_CASE: {
            size_t fw, np;
            case_layout(*e, fw, np);
            for (size_t i = 0; i < np; ++i)
                if (expr_large(e->args[fw + 2 * i + 1], schema)) return true;
            return e->has_else && expr_large(e->args.back(), schema);
        }
        default: return false;
    }
}

bool expr_binary(const ExprPtr& e, const Schema& schema) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = schema.index_of(e->name);
            return i >= 0 && schema.fields[i].binary;
        }
        case BHIP_EXPR_SCALAR_FN: return sha_fn(e->name) != 0;
        case BHIP_EXPR_CASE: {
            size_t fw, np;
            case_layout(*e, fw, np);
            return np > 0 && expr_binary(e->args[fw + 1], schema);
        }
        default: return false;
    }
}

// ---- coercion -------------------------------------------------------------------------------------
// DataFusion's numerical_coercion (datafusion 4.0.0-SNAPSHOT rev 46161d2, physical_plan/expressions/coercion.rs — the crate is not
// vendored in the reference; rust/Cargo.lock:497-500 pins it): equal types stay; otherwise the FIRST of
// Float64, Float32, Int64, Int32, Int16, Int8, UInt64, UInt32, UInt16, UInt8 that either side has.
static int numeric_rank(int t) {
    switch (t) {
        case DT_FLOAT64: return 10;
        case DT_FLOAT32: return 9;
        case DT_INT64: return 8;
        case DT_INT32: return 7;
        case DT_INT16: return 6;
        case DT_INT8: return 5;
        case DT_UINT64: return 4;
        case DT_UINT32: return 3;
        case DT_UINT16: return 2;
        case DT_UINT8: return 1;
        default: return 0;
    }
}

static int common_type(int a, int b) {
    if (a == b) return a;
    const int ra = numeric_rank(a), rb = numeric_rank(b);
    if (ra && rb) return ra >= rb ? a : b;
    // a temporal column against an integer literal of its storage width (the planner's own casts are explicit)
    if (dt_is_temporal(a) && rb && !dt_is_float(b)) return a;
    if (dt_is_temporal(b) && ra && !dt_is_float(a)) return b;
    fail(BHIP_EINVAL, std::string("cannot coerce ") + dtype_name(a) + " and " + dtype_name(b));
}

static ExprPtr cast_to(const ExprPtr& x, int t, const Schema& schema) {
    if (expr_type(x, schema) == t) return x;
    if (x->kind == BHIP_EXPR_LITERAL && !x->is_null && (numeric_rank(t) || dt_is_temporal(t)) && numeric_rank(x->dtype)) {
        // a numeric literal is re-typed, not cast — when the value survives (an out-of-range one stays a CAST: NULL at run time)
        auto l = std::make_shared<Expr>(*x);
        bool fits = true;
        if (dt_is_float(t)) {
            if (!dt_is_float(x->dtype)) l->f64 = x->dtype == DT_UINT64 ? (double)(uint64_t)x->i64 : (double)x->i64;
            if (t == DT_FLOAT32) l->f64 = (double)(float)l->f64;
        } else {
            if (dt_is_float(x->dtype)) {
                fits = x->f64 == (double)(int64_t)x->f64 && x->f64 >= -9.2e18 && x->f64 <= 9.2e18;
                if (fits) l->i64 = (int64_t)x->f64;
            }
            int64_t lo, hi;
            dt_int_range(t, lo, hi);
            if (fits && !(x->dtype == DT_UINT64 && x->i64 < 0 && t != DT_UINT64)) fits = (t == DT_UINT64 && x->dtype == DT_UINT64) || (l->i64 >= lo && l->i64 <= hi);
            else fits = false;
        }
        if (fits) { l->dtype = t; return l; }
    }
    auto c = std::make_shared<Expr>();
    c->kind = BHIP_EXPR_CAST;
    c->dtype = t;
    c->args = {x};
    return c;
}

ExprPtr coerce_expr(const ExprPtr& e, const Schema& schema) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN:
        case BHIP_EXPR_LITERAL: return e;
        case BHIP_EXPR_BINARY: {
            ExprPtr l = coerce_expr(e->args[0], schema), r = coerce_expr(e->args[1], schema);
            if (is_arith(e->name) || is_compare(e->name)) {
                const int t = common_type(expr_type(l, schema), expr_type(r, schema));
                l = cast_to(l, t, schema);
                r = cast_to(r, t, schema);
            }
            return make_binary(l, e->name, r);
        }
        case BHIP_EXPR_IN_LIST: {
            auto c = std::make_shared<Expr>(*e);
            c->args[0] = coerce_expr(e->args[0], schema);
            const int t = expr_type(c->args[0], schema);
            for (size_t i = 1; i < c->args.size(); ++i) c->args[i] = cast_to(coerce_expr(e->args[i], schema), t, schema);
            return c;
        }
        case BHIP_EXPR_CASE: {
            auto c = std::make_shared<Expr>(*e);
            for (auto& a : c->args) a = coerce_expr(a, schema);
            size_t fw, np;
            case_layout(*c, fw, np);
            int t = expr_type(c->args[fw + 1], schema);
            for (size_t i = 1; i < np; ++i) t = common_type(t, expr_type(c->args[fw + 2 * i + 1], schema));
            if (c->has_else) t = common_type(t, expr_type(c->args.back(), schema));
            for (size_t i = 0; i < np; ++i) {
                if (c->has_base) c->args[fw + 2 * i] = cast_to(c->args[fw + 2 * i], expr_type(c->args[0], schema), schema);
                c->args[fw + 2 * i + 1] = cast_to(c->args[fw + 2 * i + 1], t, schema);
            }
            if (c->has_else) c->args.back() = cast_to(c->args.back(), t, schema);
            return c;
        }
        case BHIP_EXPR_SCALAR_FN: {
            auto c = std::make_shared<Expr>(*e);
            const bool math = math_fn(e->name) >= 0;        // Signature::Uniform(1, [Float64, Float32]): the first type the argument coerces to
            for (auto& a : c->args) a = math ? cast_to(coerce_expr(a, schema), DT_FLOAT64, schema) : coerce_expr(a, schema);
            return c;
        }
        default: {
            auto c = std::make_shared<Expr>(*e);
            for (auto& a : c->args) a = coerce_expr(a, schema);
            return c;
        }
    }
}

bool expr_nullable(const ExprPtr& e, const Schema& schema) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = schema.index_of(e->name);
            if (i < 0) fail(BHIP_EINVAL, "No field named '" + e->name + "'");
            return schema.fields[i].nullable;
        }
        case BHIP_EXPR_LITERAL: return e->is_null;
        case BHIP_EXPR_IS_NULL:
        case BHIP_EXPR_IS_NOT_NULL: return false;
        case BHIP_EXPR_CAST: {
            // a cast that can fall outside the target's range yields NULLs (float -> integer, integer -> narrower integer)
            const int from = expr_type(e->args[0], schema);
            if (dt_is_float(from) && !dt_is_float(e->dtype)) return true;
            if (dtype_width(e->dtype) && dtype_width(from) && !dt_is_float(e->dtype) && from != e->dtype) return true;
            return expr_nullable(e->args[0], schema);
        }
        case BHIP_EXPR_CASE:
            if (!e->has_else) return true;
            [[fallthrough]];
        default:
            for (auto& a : e->args)
                if (expr_nullable(a, schema)) return true;
            return false;
    }
}

// =================================================================================================
// ProgramBuilder
// =================================================================================================
ProgramBuilder::ProgramBuilder(const Schema& input) : schema_(input) {}

int ProgramBuilder::new_vreg(bool is_b) {
    vreg_is_b_.push_back(is_b);
    return (int)vreg_is_b_.size() - 1;
}

int ProgramBuilder::column_index(int schema_idx) {
    for (size_t i = 0; i < col_map_.size(); ++i)
        if (col_map_[i] == schema_idx) return (int)i;
    if ((int)col_map_.size() >= VM_MAX_COLS) fail(BHIP_ENOTIMPL, "expression references more than 16 columns");
    col_map_.push_back(schema_idx);
    return (int)col_map_.size() - 1;
}

int ProgramBuilder::literal_index(uint64_t bits) {
    for (size_t i = 0; i < lits_.size(); ++i)
        if (lits_[i] == bits) return (int)i;
    if ((int)lits_.size() >= VM_MAX_LITS) fail(BHIP_ENOTIMPL, "expression has more than 32 distinct literals");
    lits_.push_back(bits);
    return (int)lits_.size() - 1;
}

int ProgramBuilder::strlit(const std::string& s) {
    if (s.size() > 255) fail(BHIP_ENOTIMPL, "Utf8 literal longer than 255 bytes");
    const size_t pos = strlits_.find(s);
    if (pos != std::string::npos && !s.empty()) return (int)pos;
    if (strlits_.size() + s.size() > (size_t)VM_STRLIT_BYTES) fail(BHIP_ENOTIMPL, "Utf8 literals exceed 192 bytes");
    const int off = (int)strlits_.size();
    strlits_ += s;
    return off;
}

static uint64_t f64_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

Operand ProgramBuilder::load_column(int schema_idx) {
    const Field& f = schema_.fields[schema_idx];
    const std::string key = "#col" + std::to_string(schema_idx);
    auto it = cse_.find(key);
    if (it != cse_.end()) return it->second;
    Operand o;
    o.dtype = f.dtype;
    o.col = column_index(schema_idx);
    if (f.dtype == DT_UTF8) {
        o.is_utf8_col = true;
    } else {
        if ((int)loads_.size() >= VM_MAX_LOADS) fail(BHIP_ENOTIMPL, "expression loads more than 16 columns");
        const bool is_b = f.dtype == DT_BOOLEAN;
        o.index = new_vreg(is_b);
        o.vclass = is_b ? VC_BOOL : (dt_is_float(f.dtype) ? VC_F64 : VC_I64);
        VmLoad ld;
        memset(&ld, 0, sizeof(ld));
        ld.col = (uint8_t)o.col;
        ld.dst = 0;
        ld.dtype = (uint8_t)f.dtype;
        ld.to_bool = is_b;
        loads_.push_back(ld);
        load_dst_is_b_.push_back(is_b);
        // remember the vreg in a parallel slot: reuse `dst` after allocation; keep vreg in instr-less table
        load_vregs_.push_back(o.index);
    }
    cse_[key] = o;
    return o;
}

bool ProgramBuilder::plain_fixed_keys(std::vector<PlainKeyPart>& parts) const {
    parts.clear();
    if (keys_.empty() || key_bytes_ > 16) return false;
    for (size_t i = 0; i < keys_.size(); ++i) {
        const KeyV& k = keys_[i];
        if (k.kind != KP_VSLOT || k.nullable) return false;
        int load = -1;
        for (size_t j = 0; j < load_vregs_.size(); ++j)
            if (load_vregs_[j] == k.src) load = (int)j;
        if (load < 0) return false;                                            // a computed key
        const int dt = loads_[(size_t)load].dtype;
        if (dt_is_float(dt) || dt == DT_BOOLEAN || dt == DT_UTF8) return false;
        if (key_info_[i].width != dtype_width(dt) || key_info_[i].nullable) return false;
        parts.push_back(PlainKeyPart{col_map_[loads_[(size_t)load].col], key_info_[i].width, key_info_[i].pos});
    }
    return true;
}

bool ProgramBuilder::can_raise() const {
    for (auto& k : keys_)
        if (k.kind == KP_UTF8_COL) return true;       // SCAN_ERR_KEY_TOO_LONG: only a Utf8 key part can outgrow its packed width
    for (auto& vi : instrs_)
        if (vi.ins.op == OP_DIV_I64) return true;
    return false;
}

Operand ProgramBuilder::emit(uint8_t op, const Operand* a, const Operand* b, bool dst_b, int vclass, int dtype,
                             uint16_t aux, int c_breg, uint8_t flags) {
    if ((int)instrs_.size() >= VM_MAX_INSTR) fail(BHIP_ENOTIMPL, "expression needs more than 96 VM instructions");
    VInstr vi{};
    vi.ins.op = op;
    vi.ins.aux = aux;
    vi.ins.flags = flags;
    vi.ins.c = 0xFF;
    vi.a_v = vi.b_v = vi.c_b = -1;
    if (a) {
        if (a->is_lit) { vi.ins.flags |= VF_A_LIT; vi.ins.a = (uint8_t)a->index; }
        else vi.a_v = a->index;
    }
    if (b) {
        if (b->is_lit) { vi.ins.flags |= VF_B_LIT; vi.ins.b = (uint8_t)b->index; }
        else vi.b_v = b->index;
    }
    vi.c_b = c_breg;
    vi.dst_is_b = dst_b;
    Operand out;
    out.index = new_vreg(dst_b);
    out.vclass = dst_b ? VC_BOOL : vclass;
    out.dtype = dtype;
    vi.dst_v = out.index;
    instrs_.push_back(vi);
    return out;
}

Operand ProgramBuilder::materialize(const Operand& o) {
    if (!o.is_lit) return o;
    return emit(OP_MOV_V, &o, nullptr, false, o.vclass, o.dtype);
}

Operand ProgramBuilder::compile(const ExprPtr& e) {
    const std::string key = e->to_string();
    auto it = cse_.find(key);
    if (it != cse_.end()) return it->second;
    Operand o = compile_uncached(e);
    cse_[key] = o;
    return o;
}

static int class_of(int dtype) { return dt_is_float(dtype) ? VC_F64 : (dtype == DT_BOOLEAN ? VC_BOOL : VC_I64); }

static int64_t parse_date(const std::string& s) {
    int y, m, d;
    if (sscanf(s.c_str(), "%d-%d-%d", &y, &m, &d) != 3) fail(BHIP_EEXEC, "Cannot cast string '" + s + "' to Date32");
    // days from civil (proleptic Gregorian)
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153u * (unsigned)(m + (m > 2 ? -3 : 9)) + 2u) / 5u + (unsigned)d - 1u;
    const unsigned doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

// multiplier between two temporal types with a common epoch (Date32 days, Date64 ms, Timestamp s/ms/us/ns); 0 = no such cast
static int64_t temporal_units_per_day(int t) {
    switch (t) {
        case DT_DATE32: return 1;
        case DT_TIMESTAMP_S: return 86400ll;
        case DT_DATE64:
        case DT_TIMESTAMP_MS: return 86400000ll;
        case DT_TIMESTAMP_US: return 86400000000ll;
        case DT_TIMESTAMP_NS: return 86400000000000ll;
        default: return 0;
    }
}

// CAST (arrow's cast kernel, non-"safe=false": a value the target cannot hold becomes NULL)
Operand ProgramBuilder::compile_cast(const Operand& x, int to) {
    const int from = x.dtype;
    if (from == to) return x;
    if (x.is_utf8_col || from == DT_UTF8 || to == DT_UTF8) fail(BHIP_ENOTIMPL, std::string("cast ") + dtype_name(from) + " -> " + dtype_name(to));
    // constant folding of numeric literals (what DataFusion's planner-inserted casts amount to)
    if (x.is_lit) {
        const uint64_t bits = lits_[x.index];
        Operand o;
        o.is_lit = true;
        o.dtype = to;
        o.vclass = class_of(to);
        if (dt_is_float(to)) {
            double d;
            if (dt_is_float(from)) memcpy(&d, &bits, 8);
            else d = from == DT_UINT64 ? (double)bits : (double)(int64_t)bits;
            if (to == DT_FLOAT32) d = dt_is_float(from) ? (double)(float)d : (from == DT_UINT64 ? (double)(float)bits : (double)(float)(int64_t)bits);
            o.index = literal_index(f64_bits(d));
            return o;
        }
        if (to != DT_BOOLEAN && !dt_is_float(from) && from != DT_BOOLEAN && !(dt_is_temporal(from) && dt_is_temporal(to))) {
            const int64_t v = (int64_t)bits;
            int64_t lo, hi;
            dt_int_range(to, lo, hi);
            const bool ok = from == DT_UINT64 ? (to == DT_UINT64 || (v >= 0 && v <= hi)) : (v >= lo && v <= hi);
            if (ok) { o.index = x.index; return o; }
        }
    }
    if (dt_is_temporal(from) && dt_is_temporal(to)) {
        // Date32 <-> Date64 <-> Timestamp(unit): the same instant in the other unit (arrow's cast multiplies / divides)
        const int64_t uf = temporal_units_per_day(from), ut = temporal_units_per_day(to);
        if (uf == ut) { Operand o = x; o.dtype = to; return o; }
        Operand src = materialize(x), k;
        k.is_lit = true;
        k.vclass = VC_I64;
        k.dtype = DT_INT64;
        if (ut > uf) {
            k.index = literal_index((uint64_t)(ut / uf));
            Operand r = emit(OP_MUL_I64, &src, &k, false, VC_I64, to);
            return to == DT_DATE32 ? emit(OP_WRAP_I64, &r, nullptr, false, VC_I64, to, (uint16_t)to) : r;
        }
        k.index = literal_index((uint64_t)(uf / ut));
        Operand r = emit(OP_DIV_I64, &src, &k, false, VC_I64, to, 0, -1);
        return to == DT_DATE32 ? emit(OP_WRAP_I64, &r, nullptr, false, VC_I64, to, (uint16_t)to) : r;
    }
    if (dt_is_float(to)) {
        if (from == DT_BOOLEAN) {
            Operand i = emit(OP_B_TO_I64, &x, nullptr, false, VC_I64, DT_INT64);
            return emit(to == DT_FLOAT32 ? OP_I64_TO_F32 : OP_I64_TO_F64, &i, nullptr, false, VC_F64, to);
        }
        if (from == DT_FLOAT32) { Operand o = x; o.dtype = to; return o; }                      // Float32 -> Float64: the value is held as a double already
        if (from == DT_FLOAT64) { Operand src = materialize(x); return emit(OP_ROUND_F32, &src, nullptr, false, VC_F64, to); }
        Operand src = materialize(x);
        if (to == DT_FLOAT32) return emit(from == DT_UINT64 ? OP_U64_TO_F32 : OP_I64_TO_F32, &src, nullptr, false, VC_F64, to);
        return emit(from == DT_UINT64 ? OP_U64_TO_F64 : OP_I64_TO_F64, &src, nullptr, false, VC_F64, to);
    }
    if (dt_is_float(from)) {
        if (to == DT_BOOLEAN) fail(BHIP_ENOTIMPL, std::string("cast ") + dtype_name(from) + " -> Boolean");
        creates_nulls_ = true;
        Operand src = materialize(x);
        return emit(OP_F64_TO_I64, &src, nullptr, false, VC_I64, to, (uint16_t)to);
    }
    if (to == DT_BOOLEAN) { Operand src = materialize(x); return emit(OP_I64_TO_B, &src, nullptr, true, VC_BOOL, DT_BOOLEAN); }
    if (from == DT_BOOLEAN) return emit(OP_B_TO_I64, &x, nullptr, false, VC_I64, to);
    // integer-valued -> integer-valued: a check only where the target's range does not cover the source's
    int64_t flo, fhi, tlo, thi;
    dt_int_range(from, flo, fhi);
    dt_int_range(to, tlo, thi);
    bool narrowing = flo < tlo || fhi > thi;
    if (from == DT_UINT64) narrowing = to != DT_UINT64;
    if (!narrowing) { Operand o = x; o.dtype = to; return o; }
    creates_nulls_ = true;
    Operand src = materialize(x);
    return emit(OP_I64_NARROW, &src, nullptr, false, VC_I64, to, (uint16_t)to, -1, from == DT_UINT64 ? VF_SRC_U64 : 0);
}

Operand ProgramBuilder::to_bool(const Operand& o) {
    if (o.vclass != VC_BOOL || o.is_utf8_col) fail(BHIP_EINVAL, "expected a Boolean expression");
    return o;
}

static bool classify_like(const std::string& pat, int& kind, std::string& needle) {
    // fast shapes: literal, lit%, %lit, %lit%; anything with '_' or an inner '%' runs the general matcher on the whole
    // pattern (at most 255 bytes: the instruction carries the length in one byte)
    const bool lead = !pat.empty() && pat.front() == '%';
    const bool trail = pat.size() > (lead ? 1u : 0u) && pat.back() == '%';
    needle = pat.substr(lead ? 1 : 0, pat.size() - (lead ? 1 : 0) - (trail ? 1 : 0));
    if (pat.find('_') != std::string::npos || needle.find('%') != std::string::npos) {
        if (pat.size() > 255) return false;
        kind = LIKE_GENERAL;
        needle = pat;
        return true;
    }
    if (pat == "%") { kind = LIKE_PREFIX; needle = ""; return true; }
    kind = lead && trail ? LIKE_CONTAINS : (lead ? LIKE_SUFFIX : (trail ? LIKE_PREFIX : LIKE_EXACT));
    return true;
}

Operand ProgramBuilder::compile_string_cmp(const ExprPtr& l, const ExprPtr& r, const std::string& op) {
    auto is_str_lit = [](const ExprPtr& e) { return e->kind == BHIP_EXPR_LITERAL && e->dtype == DT_UTF8; };
    auto null_bool = [&]() { creates_nulls_ = true; return emit(OP_LIT_B, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, 0); };
    if (op == "Like" || op == "NotLike") {
        if (!is_str_lit(r)) fail(BHIP_ENOTIMPL, "LIKE pattern must be a Utf8 literal");
        if (r->is_null) return null_bool();
        Operand lc = compile(l);
        if (!lc.is_utf8_col) fail(BHIP_ENOTIMPL, "LIKE is supported on Utf8 columns only");
        int kind;
        std::string needle;
        if (!classify_like(r->name, kind, needle)) fail(BHIP_ENOTIMPL, "LIKE pattern '" + r->name + "' is not supported");
        const int off = strlit(needle);
        Operand o = emit(OP_STR_LIKE_LIT, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, (uint16_t)kind, -1,
                         op == "NotLike" ? VF_NEGATE : 0);
        instrs_.back().ins.a = (uint8_t)off;
        instrs_.back().ins.b = (uint8_t)needle.size();
        instrs_.back().ins.c = (uint8_t)lc.col;
        return o;
    }
    int kind = cmp_kind(op);
    ExprPtr colside = l, litside = r;
    if (is_str_lit(l) && !is_str_lit(r)) { colside = r; litside = l; kind = flip_cmp(kind); }
    if (is_str_lit(litside)) {
        if (is_str_lit(colside)) fail(BHIP_ENOTIMPL, "comparison of two Utf8 literals");
        if (litside->is_null) return null_bool();
        Operand lc = compile(colside);
        if (!lc.is_utf8_col) fail(BHIP_ENOTIMPL, "Utf8 comparison is supported on columns only");
        const int off = strlit(litside->name);
        Operand o = emit(OP_STR_CMP_LIT, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, (uint16_t)kind);
        instrs_.back().ins.a = (uint8_t)off;
        instrs_.back().ins.b = (uint8_t)litside->name.size();
        instrs_.back().ins.c = (uint8_t)lc.col;
        return o;
    }
    Operand a = compile(l), b = compile(r);
    if (!a.is_utf8_col || !b.is_utf8_col) fail(BHIP_ENOTIMPL, "Utf8 comparison is supported on columns and literals only");
    Operand o = emit(OP_STR_CMP_COL, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, (uint16_t)kind);
    instrs_.back().ins.a = (uint8_t)a.col;
    instrs_.back().ins.b = (uint8_t)b.col;
    return o;
}

Operand ProgramBuilder::compile_binary(const Expr& e) {
    const std::string& op = e.name;
    const int lt = expr_type(e.args[0], schema_), rt = expr_type(e.args[1], schema_);
    if (op == "And" || op == "Or") {
        if (lt != DT_BOOLEAN || rt != DT_BOOLEAN)
            fail(BHIP_EINVAL, "Cannot evaluate binary expression " + op + " with types " + dtype_name(lt) + " and " + dtype_name(rt));
        Operand a = to_bool(compile(e.args[0])), b = to_bool(compile(e.args[1]));
        return emit(op == "And" ? OP_AND : OP_OR, &a, &b, true, VC_BOOL, DT_BOOLEAN);
    }
    if (lt != rt)
        fail(BHIP_EINVAL, "Cannot evaluate binary expression " + op + " with types " + dtype_name(lt) + " and " + dtype_name(rt));
    if (lt == DT_UTF8) {
        if (is_arith(op)) fail(BHIP_EINVAL, "Cannot evaluate binary expression " + op + " with types Utf8 and Utf8");
        return compile_string_cmp(e.args[0], e.args[1], op);
    }
    if (op == "Like" || op == "NotLike") fail(BHIP_EINVAL, "LIKE requires Utf8 operands");
    Operand a = compile(e.args[0]), b = compile(e.args[1]);
    if (is_compare(op)) {
        if (lt == DT_BOOLEAN) {
            a = emit(OP_B_TO_I64, &a, nullptr, false, VC_I64, DT_INT64);
            b = emit(OP_B_TO_I64, &b, nullptr, false, VC_I64, DT_INT64);
        }
        if (a.is_lit && b.is_lit) a = materialize(a);
        const uint8_t opc = dt_is_float(lt) ? OP_CMP_F64 : (lt == DT_UINT64 ? OP_CMP_U64 : OP_CMP_I64);
        return emit(opc, &a, &b, true, VC_BOOL, DT_BOOLEAN, (uint16_t)cmp_kind(op));
    }
    // arithmetic
    if (lt == DT_BOOLEAN) fail(BHIP_EINVAL, "Cannot evaluate binary expression " + op + " with types Boolean and Boolean");
    if (a.is_lit && b.is_lit) a = materialize(a);
    const bool f = dt_is_float(lt);
    uint8_t opc;
    if (op == "Plus") opc = f ? OP_ADD_F64 : OP_ADD_I64;
    else if (op == "Minus") opc = f ? OP_SUB_F64 : OP_SUB_I64;
    else if (op == "Multiply") opc = f ? OP_MUL_F64 : OP_MUL_I64;
    else opc = f ? OP_DIV_F64 : OP_DIV_I64;
    Operand r = emit(opc, &a, &b, false, f ? VC_F64 : VC_I64, lt, 0, opc == OP_DIV_I64 ? pred_vreg_ : -1,
                     opc == OP_DIV_I64 && lt == DT_UINT64 ? VF_SRC_U64 : 0);
    if (lt == DT_FLOAT32) r = emit(OP_ROUND_F32, &r, nullptr, false, VC_F64, lt);          // every Float32 node rounds to float
    else if (!f && dt_width(lt) < 8) r = emit(OP_WRAP_I64, &r, nullptr, false, VC_I64, lt, (uint16_t)lt);
    return r;
}

Operand ProgramBuilder::compile_uncached(const ExprPtr& ep) {
    const Expr& e = *ep;
    switch (e.kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = schema_.index_of(e.name);
            if (i < 0) fail(BHIP_EINVAL, "No field named '" + e.name + "'");
            return load_column(i);
        }
        case BHIP_EXPR_LITERAL: {
            if (e.dtype == DT_UTF8) fail(BHIP_ENOTIMPL, "Utf8 literal outside a comparison");
            if (e.dtype == DT_BOOLEAN) {
                if (e.is_null) creates_nulls_ = true;
                return emit(OP_LIT_B, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, e.is_null ? 0 : (uint16_t)((e.i64 ? 1 : 0) | 2));
            }
            Operand o;
            o.dtype = e.dtype;
            o.vclass = class_of(e.dtype);
            if (e.is_null) {
                creates_nulls_ = true;
                Operand z;
                z.is_lit = true;
                z.index = literal_index(0);
                z.vclass = o.vclass;
                z.dtype = e.dtype;
                return emit(OP_MOV_V, &z, nullptr, false, o.vclass, e.dtype, 1);
            }
            o.is_lit = true;
            o.index = literal_index(dt_is_float(e.dtype) ? f64_bits(e.f64) : (uint64_t)e.i64);
            return o;
        }
        case BHIP_EXPR_BINARY: return compile_binary(e);
        case BHIP_EXPR_CAST: {
            const ExprPtr& x = e.args[0];
            if (x->kind == BHIP_EXPR_LITERAL && x->dtype == DT_UTF8 && e.dtype == DT_DATE32 && !x->is_null) {
                Operand o;
                o.is_lit = true;
                o.dtype = DT_DATE32;
                o.vclass = VC_I64;
                o.index = literal_index((uint64_t)parse_date(x->name));
                return o;
            }
            return compile_cast(compile(x), e.dtype);
        }
        case BHIP_EXPR_NOT: {
            if (expr_type(e.args[0], schema_) != DT_BOOLEAN) fail(BHIP_EINVAL, "NOT requires a Boolean operand");
            Operand a = to_bool(compile(e.args[0]));
            return emit(OP_NOT, &a, nullptr, true, VC_BOOL, DT_BOOLEAN);
        }
        case BHIP_EXPR_IS_NULL:
        case BHIP_EXPR_IS_NOT_NULL: {
            const uint16_t want_valid = e.kind == BHIP_EXPR_IS_NOT_NULL ? 1 : 0;
            Operand a = compile(e.args[0]);
            if (a.is_utf8_col) {
                Operand o = emit(OP_STR_IS_NULL, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, want_valid);
                instrs_.back().ins.c = (uint8_t)a.col;
                return o;
            }
            if (a.is_lit) return emit(OP_LIT_B, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, (uint16_t)(want_valid | 2));
            return emit(a.vclass == VC_BOOL ? OP_IS_NULL_B : OP_IS_NULL_V, &a, nullptr, true, VC_BOOL, DT_BOOLEAN, want_valid);
        }
        case BHIP_EXPR_NEGATIVE: {
            const int t = expr_type(e.args[0], schema_);
            if (t == DT_BOOLEAN || t == DT_UTF8) fail(BHIP_EINVAL, std::string("Cannot negate ") + dtype_name(t));
            Operand a = materialize(compile(e.args[0]));
            Operand r = emit(dt_is_float(t) ? OP_NEG_F64 : OP_NEG_I64, &a, nullptr, false, a.vclass, t);
            if (!dt_is_float(t) && dt_width(t) < 8) r = emit(OP_WRAP_I64, &r, nullptr, false, VC_I64, t, (uint16_t)t);
            return r;
        }
        case BHIP_EXPR_IN_LIST: {
            Operand acc;
            bool first = true;
            for (size_t i = 1; i < e.args.size(); ++i) {
                Operand c = compile(make_binary(e.args[0], "Eq", e.args[i]));
                if (first) { acc = c; first = false; }
                else acc = emit(OP_OR, &acc, &c, true, VC_BOOL, DT_BOOLEAN);
            }
            if (e.negated) acc = emit(OP_NOT, &acc, nullptr, true, VC_BOOL, DT_BOOLEAN);
            return acc;
        }
        case BHIP_EXPR_CASE: {
            size_t fw, np;
            case_layout(e, fw, np);
            const int rt = expr_type(e.args[fw + 1], schema_);
            if (rt == DT_UTF8) fail(BHIP_ENOTIMPL, "CASE producing Utf8");
            const bool is_b = rt == DT_BOOLEAN;
            Operand result;
            if (e.has_else) {
                if (expr_type(e.args.back(), schema_) != rt) fail(BHIP_EINVAL, "CASE branches have different types");
                result = compile(e.args.back());
            } else {
                creates_nulls_ = true;
                if (is_b) result = emit(OP_LIT_B, nullptr, nullptr, true, VC_BOOL, DT_BOOLEAN, 0);
                else {
                    Operand z;
                    z.is_lit = true;
                    z.index = literal_index(0);
                    z.vclass = class_of(rt);
                    z.dtype = rt;
                    result = emit(OP_MOV_V, &z, nullptr, false, z.vclass, rt, 1);
                }
            }
            for (size_t k = np; k-- > 0;) {
                const ExprPtr& w = e.args[fw + 2 * k];
                const ExprPtr& t = e.args[fw + 2 * k + 1];
                if (expr_type(t, schema_) != rt) fail(BHIP_EINVAL, "CASE branches have different types");
                Operand cond = e.has_base ? compile(make_binary(e.args[0], "Eq", w)) : compile(w);
                cond = to_bool(cond);
                Operand tv = compile(t);
                if (is_b) result = emit(OP_SELECT_B, &tv, &result, true, VC_BOOL, DT_BOOLEAN, 0, cond.index);
                else result = emit(OP_SELECT_V, &tv, &result, false, class_of(rt), rt, 0, cond.index);
            }
            return result;
        }
        case BHIP_EXPR_SCALAR_FN: {
            if (e.name == "octet_length") {
                if (expr_type(e.args[0], schema_) != DT_UTF8) fail(BHIP_EINVAL, "octet_length requires a Utf8 argument");
                Operand a = compile(e.args[0]);
                if (!a.is_utf8_col) fail(BHIP_ENOTIMPL, "octet_length of a Utf8 expression that is not a column");
                Operand o = emit(OP_STR_LEN, nullptr, nullptr, false, VC_I64, DT_INT32);
                instrs_.back().ins.c = (uint8_t)a.col;
                return o;
            }
            if (str_fn(e.name) >= 0 || sha_fn(e.name)) fail(BHIP_ENOTIMPL, "expression producing Utf8 / Binary: " + e.name + "() (evaluated as a column, host/utf8_exprs.cpp)");
            if (expr_type(e.args[0], schema_) != DT_FLOAT64) fail(BHIP_EINVAL, e.name + " requires a Float64 argument");
            Operand a = materialize(compile(e.args[0]));
            return emit(OP_MATH_F64, &a, nullptr, false, VC_F64, DT_FLOAT64, (uint16_t)math_fn(e.name));
        }
        default: fail(BHIP_ENOTIMPL, "unsupported expression kind");
    }
}

void ProgramBuilder::set_predicate(const ExprPtr& e) {
    if (expr_type(e, schema_) != DT_BOOLEAN) fail(BHIP_EINVAL, "Filter predicate must return boolean values");
    Operand p = to_bool(compile(e));
    if (pred_vreg_ >= 0) {
        Operand prev;
        prev.index = pred_vreg_;
        prev.vclass = VC_BOOL;
        prev.dtype = DT_BOOLEAN;
        p = emit(OP_AND, &prev, &p, true, VC_BOOL, DT_BOOLEAN);
    }
    pred_vreg_ = p.index;
}

void ProgramBuilder::add_key(const ExprPtr& e, bool force_not_null) {
    if ((int)keys_.size() >= VM_MAX_KEYPARTS) fail(BHIP_ENOTIMPL, "more than 8 key columns");
    const int t = expr_type(e, schema_);
    const bool nullable = !force_not_null && expr_nullable(e, schema_);
    Operand o = compile(e);
    KeyV k{};
    k.nullable = nullable;
    if (o.is_utf8_col) { k.kind = KP_UTF8_COL; k.src = o.col; k.width = 0; }
    else {
        o = materialize(o);
        k.src = o.index;
        if (o.vclass == VC_BOOL) { k.kind = KP_BSLOT; k.width = 1; }
        // (a Float32 key part holds the double's bits: 8 bytes, turned back into a float when the group table is emitted)
        else { k.kind = dt_is_float(t) ? KP_VSLOT_F64 : KP_VSLOT; k.width = t == DT_FLOAT32 ? 8 : dtype_width(t); }
        k.width += nullable ? 1 : 0;
    }
    keys_.push_back(k);
    key_info_.push_back(KeyInfo{0, k.width, nullable ? 1 : 0, t});
}

int ProgramBuilder::add_acc(int kind, const Operand& src) {
    Operand s = src;
    bool is_b = false;
    int vreg = -1;
    if (kind != ACC_COUNT_ROWS) {
        if (s.is_utf8_col) fail(BHIP_ENOTIMPL, "aggregate over a Utf8 column");
        s = materialize(s);
        vreg = s.index;
        is_b = s.vclass == VC_BOOL;
        if (is_b && kind != ACC_COUNT_VALID_B) fail(BHIP_ENOTIMPL, "aggregate over a Boolean expression");
    }
    for (size_t i = 0; i < accs_.size(); ++i)
        if (accs_[i].kind == kind && accs_[i].vreg == vreg) return (int)i;
    if ((int)accs_.size() >= VM_MAX_ACC) fail(BHIP_ENOTIMPL, "more than 16 distinct accumulators");
    accs_.push_back(AccV{kind, vreg, is_b});
    return (int)accs_.size() - 1;
}

void ProgramBuilder::add_output(const ExprPtr& e) {
    if ((int)outs_.size() >= VM_MAX_OUT) fail(BHIP_ENOTIMPL, "more than 16 computed output columns");
    const int t = expr_type(e, schema_);
    Operand o = compile(e);
    if (o.is_utf8_col) fail(BHIP_EINVAL, "Utf8 pass-through columns are not VM outputs");
    o = materialize(o);
    outs_.push_back({o.index, o.vclass == VC_BOOL});
    out_dtypes_.push_back(t);
}

// ---- register allocation + emission ------------------------------------------------------------
void ProgramBuilder::finish(ScanParams& P) {
    memset(&P, 0, sizeof(P));
    // key layout: fixed-width parts take their width, Utf8 parts share what is left of 16 bytes
    int fixed = 0, n_utf8 = 0;
    for (auto& k : keys_) {
        if (k.kind == KP_UTF8_COL) ++n_utf8;
        else fixed += k.width;
    }
    if (!hash_only_ && (fixed > 16 || (n_utf8 > 0 && (16 - fixed) / n_utf8 < 2)))
        fail(BHIP_ENOTIMPL, "key columns do not fit the 16-byte packed key");
    const int utf8_width = n_utf8 ? (16 - fixed) / n_utf8 : 0;
    bool has_utf8_loads = false;
    for (auto& k : keys_) {
        if (k.kind != KP_UTF8_COL) continue;
        k.width = utf8_width;
        // short strings (<= 7 chars + length byte) are packed by the hoisted load stage into one V
        // register; wider parts are read by the sink itself (slower: loads are not hoisted)
        if (!hash_only_ && utf8_width <= 8 && (int)loads_.size() < VM_MAX_LOADS) {
            VmLoad ld;
            memset(&ld, 0, sizeof(ld));
            ld.col = (uint8_t)k.src;
            ld.dtype = (uint8_t)DT_UTF8;
            ld.width = (uint8_t)utf8_width;
            const int vreg = new_vreg(false);
            loads_.push_back(ld);
            load_dst_is_b_.push_back(false);
            load_vregs_.push_back(vreg);
            k.kind = KP_VSLOT;
            k.src = vreg;
            has_utf8_loads = true;
        }
    }
    const int n_v = (int)vreg_is_b_.size();
    const int INF = 1 << 30;
    std::vector<int> last_use(n_v, -1);
    for (size_t i = 0; i < instrs_.size(); ++i) {
        const VInstr& vi = instrs_[i];
        if (vi.a_v >= 0) last_use[vi.a_v] = (int)i;
        if (vi.b_v >= 0) last_use[vi.b_v] = (int)i;
        if (vi.c_b >= 0) last_use[vi.c_b] = (int)i;
    }
    if (pred_vreg_ >= 0) last_use[pred_vreg_] = INF;
    for (auto& k : keys_)
        if (k.kind != KP_UTF8_COL) last_use[k.src] = INF;
    for (auto& a : accs_)
        if (a.vreg >= 0) last_use[a.vreg] = INF;
    for (auto& o : outs_) last_use[o.first] = INF;

    std::vector<int> phys(n_v, -1);
    std::vector<bool> v_busy(VM_MAX_VSLOTS, false), b_busy(VM_MAX_BSLOTS, false);
    int v_high = 0, b_high = 0;
    auto take = [&](bool is_b) {
        auto& busy = is_b ? b_busy : v_busy;
        for (size_t s = 0; s < busy.size(); ++s)
            if (!busy[s]) {
                busy[s] = true;
                int& high = is_b ? b_high : v_high;
                if ((int)s + 1 > high) high = (int)s + 1;
                return (int)s;
            }
        fail(BHIP_ENOTIMPL, "expression needs too many VM registers");
    };
    auto release = [&](int vreg) {
        if (vreg < 0 || phys[vreg] < 0) return;
        (vreg_is_b_[vreg] ? b_busy : v_busy)[phys[vreg]] = false;
    };
    // hoisted loads are all live from the start of the tile
    for (size_t i = 0; i < loads_.size(); ++i) phys[load_vregs_[i]] = take(load_dst_is_b_[i]);
    for (size_t i = 0; i < loads_.size(); ++i)
        if (last_use[load_vregs_[i]] < 0) release(load_vregs_[i]);
    for (size_t i = 0; i < instrs_.size(); ++i) {
        VInstr& vi = instrs_[i];
        // sources whose last use is this instruction free their register first: an instruction may
        // write its result over one of its own operands (every element is thread-private)
        const int srcs[3] = {vi.a_v, vi.b_v, vi.c_b};
        int pa = vi.a_v >= 0 ? phys[vi.a_v] : -1, pb = vi.b_v >= 0 ? phys[vi.b_v] : -1, pc = vi.c_b >= 0 ? phys[vi.c_b] : -1;
        for (int s : srcs)
            if (s >= 0 && last_use[s] == (int)i) release(s);
        // SELECT reads a and b after testing c in the same element: writing over c's B register is
        // only safe when dst is a V register (different file) — always true for SELECT_V; for
        // SELECT_B the element is read before it is written, also safe.
        phys[vi.dst_v] = take(vi.dst_is_b);
        if (last_use[vi.dst_v] < 0) release(vi.dst_v);   // dead result
        if (pa >= 0) vi.ins.a = (uint8_t)pa;
        if (pb >= 0) vi.ins.b = (uint8_t)pb;
        if (pc >= 0) vi.ins.c = (uint8_t)pc;
        vi.ins.dst = (uint8_t)phys[vi.dst_v];
    }
    VmProgram& G = P.prog;
    G.n_loads = (int)loads_.size();
    G.has_utf8_loads = has_utf8_loads ? 1 : 0;
    for (size_t i = 0; i < loads_.size(); ++i) {
        G.loads[i] = loads_[i];
        G.loads[i].dst = (uint8_t)phys[load_vregs_[i]];
    }
    G.n_instr = (int)instrs_.size();
    for (size_t i = 0; i < instrs_.size(); ++i) G.instr[i] = instrs_[i].ins;
    G.n_vslots = v_high > 0 ? v_high : 1;
    G.n_bslots = b_high > 0 ? b_high : 1;
    for (size_t i = 0; i < lits_.size(); ++i) G.lits[i] = lits_[i];
    memcpy(G.strlits, strlits_.data(), strlits_.size());
    P.pred_slot = pred_vreg_ >= 0 ? phys[pred_vreg_] : -1;
    P.n_cols = (int)col_map_.size();

    int pos = 0;
    P.n_keyparts = (int)keys_.size();
    for (size_t i = 0; i < keys_.size(); ++i) {
        KeyV& k = keys_[i];
        KeyPart kp;
        kp.kind = (uint8_t)k.kind;
        kp.src = (uint8_t)(k.kind == KP_UTF8_COL ? k.src : phys[k.src]);
        kp.width = (uint8_t)k.width;
        kp.nullable = (uint8_t)k.nullable;
        P.keyparts[i] = kp;
        key_info_[i].pos = pos;
        key_info_[i].width = k.width;
        pos += k.width;
    }
    key_bytes_ = pos;
    P.key_bytes = pos;
    P.n_acc = (int)accs_.size();
    for (size_t i = 0; i < accs_.size(); ++i) {
        P.acc[i].kind = (uint8_t)accs_[i].kind;
        P.acc[i].slot = (uint8_t)(accs_[i].vreg >= 0 ? phys[accs_[i].vreg] : 0);
    }
    P.n_out = (int)outs_.size();
    for (size_t i = 0; i < outs_.size(); ++i) {
        P.out_slot[i] = (uint8_t)phys[outs_[i].first];
        P.out_dtype[i] = (uint8_t)out_dtypes_[i];
    }
}

void ProgramBuilder::bind(ScanParams& P, const std::vector<int>& col_map, const Batch& b, bool creates_nulls) {
    P.n_rows = b.n_rows;
    bool nullable = creates_nulls;
    for (size_t i = 0; i < col_map.size(); ++i) {
        const Column& c = b.cols[col_map[i]];
        P.cols[i] = c.ref();
        if (c.validity) nullable = true;
    }
    P.prog.nullable = nullable ? 1 : 0;
}

}  // namespace bhip
