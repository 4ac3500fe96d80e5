// sop.cpp — recognises aggregate plans of the "chain of products" shape and builds the plan table
// of the register-resident fast path (sop.h, sop_kernel.h).  Anything that does not match keeps the
// general VM program: this is an optimisation of the same operator, never a different result.
#include "sop.hpp"

#include <cmath>
#include <cstring>
#include <limits>

namespace bhip {

namespace {

struct Builder {
    const Schema& schema;
    SopPlan& out;
    Builder(const Schema& s, SopPlan& o) : schema(s), out(o) {}

    int column(int schema_idx) {
        for (size_t i = 0; i < out.col_map.size(); ++i)
            if (out.col_map[i] == schema_idx) return (int)i;
        if ((int)out.col_map.size() >= SOP_NCOL) return -1;
        out.col_map.push_back(schema_idx);
        return (int)out.col_map.size() - 1;
    }

    // column reference -> schema index, or -1.  A nullable field is accepted: whether a batch really carries
    // NULLs (a validity bitmap) is decided per batch by sop_columns_bindable
    int plain_column(const ExprPtr& e) const {
        if (e->kind != BHIP_EXPR_COLUMN) return -1;
        return schema.index_of(e->name);
    }

    // ---- predicate: AND of  column <op> literal  over Float64 / Int32 / Date32 columns, folded into one
    // [lo, hi] range per column.  Strict bounds move to the neighbouring representable value, which is
    // exact: x < L  <=>  x <= pred(L) for doubles (NaN fails both), x <= L - 1 for integers.
    bool predicate(const ExprPtr& e) {
        if (e->kind == BHIP_EXPR_BINARY && e->name == "And") return predicate(e->args[0]) && predicate(e->args[1]);
        if (e->kind != BHIP_EXPR_BINARY) return false;
        static const char* ops[] = {"Eq", "Lt", "LtEq", "Gt", "GtEq"};
        static const int flipped[] = {0, 3, 4, 1, 2};
        int k = -1;
        for (int i = 0; i < 5; ++i)
            if (e->name == ops[i]) k = i;
        if (k < 0) return false;                        // NotEq, Like ...: not a range
        ExprPtr colside = e->args[0], litside = e->args[1];
        if (colside->kind == BHIP_EXPR_LITERAL) { std::swap(colside, litside); k = flipped[k]; }
        const int ci = plain_column(colside);
        if (ci < 0) return false;
        const int t = schema.fields[ci].dtype;
        if (litside->kind != BHIP_EXPR_LITERAL || litside->is_null || litside->dtype != t) return false;
        double lit;
        bool is_int;
        if (t == DT_FLOAT64) { lit = litside->f64; is_int = false; }
        else if (t == DT_INT32 || t == DT_DATE32) { lit = (double)litside->i64; is_int = true; }
        else return false;                              // Int64 / UInt64 do not embed exactly in double
        if (std::isnan(lit)) return false;
        const double inf = std::numeric_limits<double>::infinity();
        double lo = -inf, hi = inf;
        switch (k) {
            case 0: lo = hi = lit; break;
            case 1: hi = is_int ? lit - 1.0 : std::nextafter(lit, -inf); break;     // x <  L
            case 2: hi = lit; break;                                                // x <= L
            case 3: lo = is_int ? lit + 1.0 : std::nextafter(lit, inf); break;      // x >  L
            default: lo = lit; break;                                               // x >= L
        }
        const int c = column(ci);
        if (c < 0) return false;
        for (int i = 0; i < out.prog.n_ranges; ++i)
            if (out.prog.ranges[i].col == c) {           // intersect with the range this column already has
                if (lo > out.prog.ranges[i].lo) out.prog.ranges[i].lo = lo;
                if (hi < out.prog.ranges[i].hi) out.prog.ranges[i].hi = hi;
                return true;
            }
        if (out.prog.n_ranges >= SOP_NRANGE) return false;
        SopRange& r = out.prog.ranges[out.prog.n_ranges++];
        memset(&r, 0, sizeof(r));
        r.col = (uint8_t)c;
        r.is32 = is_int ? 1 : 0;
        r.lo = lo;
        r.hi = hi;
        return true;
    }

    struct Factor { bool has_col; bool is32; int schema_col; double sgn, add; std::string text; };

    // Float64 column, or an Int32 / Date32 column under CAST(... AS Float64)
    bool f64_column(const ExprPtr& e, int& col, bool& is32) const {
        if (e->kind == BHIP_EXPR_CAST && e->dtype == DT_FLOAT64) {
            const int i = plain_column(e->args[0]);
            if (i < 0) return false;
            const int t = schema.fields[i].dtype;
            if (t != DT_INT32 && t != DT_DATE32) return false;
            col = i; is32 = true;
            return true;
        }
        const int i = plain_column(e);
        if (i < 0 || schema.fields[i].dtype != DT_FLOAT64) return false;
        col = i; is32 = false;
        return true;
    }

    static bool f64_literal(const ExprPtr& e, double& v) {
        if (e->kind != BHIP_EXPR_LITERAL || e->dtype != DT_FLOAT64 || e->is_null) return false;
        v = e->f64;
        return true;
    }

    // f = sgn * x + add : one correctly rounded addition of the same operands the reference adds
    bool factor(const ExprPtr& e, Factor& f) const {
        f.text = e->to_string();
        f.has_col = true; f.is32 = false; f.schema_col = -1; f.sgn = 1.0; f.add = -0.0;
        double lit;
        if (f64_literal(e, lit)) { f.has_col = false; f.sgn = 0.0; f.add = lit; return true; }
        if (f64_column(e, f.schema_col, f.is32)) return true;                       // x + (-0.0) == x
        if (e->kind == BHIP_EXPR_BINARY && (e->name == "Plus" || e->name == "Minus")) {
            const bool plus = e->name == "Plus";
            if (f64_literal(e->args[0], lit) && f64_column(e->args[1], f.schema_col, f.is32)) {
                f.sgn = plus ? 1.0 : -1.0;              // lit + x | lit - x == (-x) + lit
                f.add = lit;
                return true;
            }
            if (f64_literal(e->args[1], lit) && f64_column(e->args[0], f.schema_col, f.is32)) {
                f.sgn = 1.0;                            // x + lit | x - lit == x + (-lit)
                f.add = plus ? lit : -lit;
                return true;
            }
        }
        return false;
    }

    // ((f0 * f1) * f2) ... : left-nested products only (keeps the reference's rounding order)
    bool chain(const ExprPtr& e, std::vector<Factor>& fs) const {
        Factor f;
        if (factor(e, f)) { fs.push_back(f); return true; }
        if (e->kind == BHIP_EXPR_BINARY && e->name == "Multiply") {
            if (!chain(e->args[0], fs)) return false;
            if (!factor(e->args[1], f)) return false;
            fs.push_back(f);
            return true;
        }
        return false;
    }
};

}  // namespace

bool build_sop(const Schema& schema, const ExprPtr& predicate, const std::vector<ExprPtr>& keys,
               const std::vector<SopAccExpr>& accs, SopPlan& out) {
    memset(&out.prog, 0, sizeof(out.prog));
    out.col_map.clear();
    out.key_info.clear();
    Builder b(schema, out);
    if (predicate && !b.predicate(predicate)) return false;
    for (int i = 0; i < out.prog.n_ranges; ++i)
        if (out.prog.ranges[i].lo > out.prog.ranges[i].hi) return false;            // empty range: leave it to the VM
    // keys: plain non-nullable columns; part 0 -> word 0, part 1 -> word 1, part 2 -> high half of word 1
    if (keys.size() > (size_t)SOP_NKEY) return false;
    for (size_t i = 0; i < keys.size(); ++i) {
        const int si = b.plain_column(keys[i]);
        if (si < 0) return false;
        const int t = schema.fields[si].dtype;
        int kind, width;
        if (t == DT_INT32 || t == DT_DATE32) { kind = SOP_KEY_I32; width = 4; }
        else if (t == DT_INT64 || t == DT_UINT64) { kind = SOP_KEY_I64; width = 8; }
        else if (t == DT_UTF8) { kind = SOP_KEY_UTF8; width = 8; }
        else return false;
        if (keys.size() == 3 && i >= 1 && kind != SOP_KEY_I32) return false;        // words 1 holds two 32-bit parts
        const int c = b.column(si);
        if (c < 0) return false;
        SopKey& k = out.prog.keys[out.prog.n_keys++];
        k.col = (uint8_t)c;
        k.kind = (uint8_t)kind;
        k.pad[0] = k.pad[1] = 0;
        const int pos = i == 0 ? 0 : (i == 1 ? 8 : 12);
        out.key_info.push_back(ProgramBuilder::KeyInfo{pos, width, 0, t});
    }
    // accumulators: SUM(Float64) over chains; a chain that extends the chain emitted just before it
    // continues from its running product (Q1: price, price*(1-d), price*(1-d)*(1+t))
    std::vector<std::string> last_chain;
    for (size_t a = 0; a < accs.size(); ++a) {
        if (accs[a].kind != ACC_SUM_F64) return false;
        std::vector<Builder::Factor> fs;
        if (!b.chain(accs[a].expr, fs)) return false;
        size_t common = 0;
        if (!last_chain.empty() && fs.size() > last_chain.size()) {
            common = last_chain.size();
            for (size_t i = 0; i < last_chain.size(); ++i)
                if (fs[i].text != last_chain[i]) { common = 0; break; }
        }
        for (size_t i = common; i < fs.size(); ++i) {
            if (out.prog.n_steps >= SOP_NSTEP) return false;
            SopStep& st = out.prog.steps[out.prog.n_steps++];
            memset(&st, 0, sizeof(st));
            st.has_col = fs[i].has_col ? 1 : 0;
            st.is32 = fs[i].is32 ? 1 : 0;
            st.start = i == 0 ? 1 : 0;
            st.acc = 0xFF;
            st.sgn = fs[i].sgn;
            st.add = fs[i].add;
            if (fs[i].has_col) {
                const int c = b.column(fs[i].schema_col);
                if (c < 0) return false;
                st.col = (uint8_t)c;
            }
        }
        out.prog.steps[out.prog.n_steps - 1].acc = (uint8_t)a;
        last_chain.clear();
        for (auto& f : fs) last_chain.push_back(f.text);
    }
    out.prog.n_cols = (int)out.col_map.size();
    return true;
}

bool sop_columns_bindable(const SopPlan& plan, const Batch& b, bool range_nulls_ok) {
    for (size_t i = 0; i < plan.col_map.size(); ++i) {
        if (!b.cols[plan.col_map[i]].validity) continue;
        // NULLs present.  A column the predicate constrains may have them where the kernel tests the bitmap: a NULL
        // fails the range, so the row reaches neither a key nor an accumulator.  Anything else: the VM kernel.
        bool ranged = false;
        for (int r = 0; r < plan.prog.n_ranges; ++r) ranged = ranged || plan.prog.ranges[r].col == (uint8_t)i;
        if (!(range_nulls_ok && ranged)) return false;
    }
    return true;
}

void bind_sop(SopPlan& plan, const Batch& b) {
    plan.prog.n_rows = b.n_rows;
    for (size_t i = 0; i < plan.col_map.size(); ++i) {
        const Column& c = b.cols[plan.col_map[i]];
        SopColumn& sc = plan.prog.cols[i];
        sc.data = c.data ? c.data->ptr() : nullptr;
        sc.offsets = c.offsets ? c.offsets->as<int32_t>() : nullptr;
        sc.validity = c.validity ? c.validity->as<uint64_t>() : nullptr;
        sc.dtype = c.dtype;
        sc.data_bytes = (int32_t)c.data_bytes;
    }
}

bool lean_eligible(const SopProgram& prog) {
    if (prog.n_keys > 2) return false;
    for (int q = 0; q < prog.n_keys; ++q)
        if (prog.keys[q].kind != SOP_KEY_I32 && prog.keys[q].kind != SOP_KEY_UTF8) return false;
    for (int s = 0; s < prog.n_steps; ++s) {
        const SopStep& st = prog.steps[s];
        if (!st.has_col || st.is32) return false;
        if (st.sgn != 1.0 && st.sgn != -1.0) return false;
    }
    return true;
}

bool lean_bindable(const SopPlan& plan, const Batch& b) {
    for (int ci : plan.col_map) {
        const Column& c = b.cols[ci];
        if (c.data && ((uintptr_t)c.data->ptr() & 15)) return false;
        if (c.offsets && ((uintptr_t)c.offsets->ptr() & 15)) return false;
        if (c.dtype == DT_UTF8 && c.data && !c.data->owned()) return false;
    }
    return true;
}

}  // namespace bhip
