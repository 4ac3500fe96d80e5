// sop.cpp — recognises aggregate plans of the "chain of products" shape and builds the plan table
// of the register-resident fast path (sop.h, kernels_sop.hip).  Anything that does not match keeps
// the general VM program: this is an optimisation of the same operator, never a different result.
#include "sop.hpp"

#include <cstring>

namespace bhip {

namespace {

struct Builder {
    const Schema& schema;
    SopPlan& out;
    explicit Builder(const Schema& s, SopPlan& o) : schema(s), out(o) {}

    int column(int schema_idx) {
        for (size_t i = 0; i < out.col_map.size(); ++i)
            if (out.col_map[i] == schema_idx) return (int)i;
        if ((int)out.col_map.size() >= SOP_NCOL) return -1;
        out.col_map.push_back(schema_idx);
        return (int)out.col_map.size() - 1;
    }

    // numeric column reference -> schema index, or -1
    int numeric_column(const ExprPtr& e) const {
        if (e->kind != BHIP_EXPR_COLUMN) return -1;
        const int i = schema.index_of(e->name);
        if (i < 0 || schema.fields[i].nullable) return -1;
        const int t = schema.fields[i].dtype;
        if (t == DT_UTF8 || t == DT_BOOLEAN) return -1;
        return i;
    }

    static bool numeric_literal(const ExprPtr& e, int want_type, uint64_t& bits) {
        if (e->kind != BHIP_EXPR_LITERAL || e->is_null || e->dtype != want_type) return false;
        if (e->dtype == DT_FLOAT64) memcpy(&bits, &e->f64, 8);
        else bits = (uint64_t)e->i64;
        return true;
    }

    // conjunction of  column <op> literal
    bool predicate(const ExprPtr& e) {
        if (e->kind == BHIP_EXPR_BINARY && e->name == "And") return predicate(e->args[0]) && predicate(e->args[1]);
        if (e->kind != BHIP_EXPR_BINARY) return false;
        static const char* ops[] = {"Eq", "NotEq", "Lt", "LtEq", "Gt", "GtEq"};
        static const int kinds[] = {CMP_EQ, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE};
        static const int flipped[] = {CMP_EQ, CMP_NE, CMP_GT, CMP_GE, CMP_LT, CMP_LE};
        int k = -1;
        for (int i = 0; i < 6; ++i)
            if (e->name == ops[i]) k = i;
        if (k < 0) return false;
        ExprPtr colside = e->args[0], litside = e->args[1];
        int kind = kinds[k];
        if (colside->kind == BHIP_EXPR_LITERAL) { std::swap(colside, litside); kind = flipped[k]; }
        const int ci = numeric_column(colside);
        if (ci < 0) return false;
        const int t = schema.fields[ci].dtype;
        uint64_t bits;
        if (!numeric_literal(litside, t, bits)) return false;
        if (out.prog.n_pred >= SOP_NPRED) return false;
        const int c = column(ci);
        if (c < 0) return false;
        SopCmp& p = out.prog.pred[out.prog.n_pred++];
        memset(&p, 0, sizeof(p));
        p.col = (uint8_t)c;
        p.cmp = (uint8_t)kind;
        p.vclass = t == DT_FLOAT64 ? (uint8_t)VC_F64 : (t == DT_UINT64 ? (uint8_t)3 : (uint8_t)VC_I64);
        p.lit = bits;
        return true;
    }

    struct Factor { int mode; int schema_col; double lit; std::string text; };

    // Float64-valued column (or an integer column under CAST ... AS Float64)
    int f64_column(const ExprPtr& e) const {
        if (e->kind == BHIP_EXPR_CAST && e->dtype == DT_FLOAT64) {
            const int i = numeric_column(e->args[0]);
            return i;
        }
        const int i = numeric_column(e);
        if (i < 0 || schema.fields[i].dtype != DT_FLOAT64) return -1;
        return i;
    }

    bool factor(const ExprPtr& e, Factor& f) const {
        f.text = e->to_string();
        f.lit = 0;
        f.schema_col = -1;
        if (e->kind == BHIP_EXPR_LITERAL && e->dtype == DT_FLOAT64 && !e->is_null) { f.mode = SOP_F_LIT; f.lit = e->f64; return true; }
        const int c = f64_column(e);
        if (c >= 0) { f.mode = SOP_F_COL; f.schema_col = c; return true; }
        if (e->kind == BHIP_EXPR_BINARY && (e->name == "Plus" || e->name == "Minus")) {
            const ExprPtr& l = e->args[0];
            const ExprPtr& r = e->args[1];
            const bool l_lit = l->kind == BHIP_EXPR_LITERAL && l->dtype == DT_FLOAT64 && !l->is_null;
            const bool r_lit = r->kind == BHIP_EXPR_LITERAL && r->dtype == DT_FLOAT64 && !r->is_null;
            if (l_lit && !r_lit) {
                const int rc = f64_column(r);
                if (rc < 0) return false;
                f.schema_col = rc; f.lit = l->f64;
                f.mode = e->name == "Plus" ? SOP_F_LIT_PLUS_COL : SOP_F_LIT_MINUS_COL;
                return true;
            }
            if (r_lit && !l_lit) {
                const int lc = f64_column(l);
                if (lc < 0) return false;
                f.schema_col = lc; f.lit = r->f64;
                f.mode = e->name == "Plus" ? SOP_F_COL_PLUS_LIT : SOP_F_COL_MINUS_LIT;
                return true;
            }
        }
        return false;
    }

    struct Link { Factor f; int op; };

    // ((f0 op f1) op f2) ... : left-nested products / quotients only (keeps the rounding order)
    bool chain(const ExprPtr& e, std::vector<Link>& links) const {
        Factor f;
        if (factor(e, f)) { links.push_back(Link{f, SOP_OP_START}); return true; }
        if (e->kind == BHIP_EXPR_BINARY && (e->name == "Multiply" || e->name == "Divide")) {
            if (!chain(e->args[0], links)) return false;
            if (!factor(e->args[1], f)) return false;
            links.push_back(Link{f, e->name == "Multiply" ? SOP_OP_MUL : SOP_OP_DIV});
            return true;
        }
        return false;
    }
};

}  // namespace

bool build_sop(const Schema& schema, const ExprPtr& predicate, const std::vector<ExprPtr>& keys,
               const std::vector<ProgramBuilder::KeyInfo>& key_info, int key_bytes,
               const std::vector<SopAccExpr>& accs, SopPlan& out) {
    memset(&out.prog, 0, sizeof(out.prog));
    out.col_map.clear();
    Builder b(schema, out);
    if (predicate && !b.predicate(predicate)) return false;
    // keys: plain non-nullable columns, fixed width or a short-string pack of at most 8 bytes
    if (keys.size() > (size_t)SOP_NKEY || keys.size() != key_info.size()) return false;
    for (size_t i = 0; i < keys.size(); ++i) {
        if (keys[i]->kind != BHIP_EXPR_COLUMN || key_info[i].nullable) return false;
        const int si = schema.index_of(keys[i]->name);
        if (si < 0 || schema.fields[si].nullable) return false;
        const int t = schema.fields[si].dtype;
        if (t == DT_BOOLEAN || t == DT_FLOAT64) return false;
        if (key_info[i].width > 8) return false;
        const int c = b.column(si);
        if (c < 0) return false;
        SopKey& k = out.prog.keys[out.prog.n_keys++];
        k.col = (uint8_t)c;
        k.width = (uint8_t)key_info[i].width;
        k.pos = (uint8_t)key_info[i].pos;
        k.pad = 0;
    }
    out.prog.key_bytes = key_bytes;
    // accumulators: SUM(Float64) over chains; a chain that extends the chain emitted just before it
    // continues from its running product (Q1: price, price*(1-d), price*(1-d)*(1+t))
    std::vector<std::string> last_chain;     // factor texts of the most recently emitted chain
    for (size_t a = 0; a < accs.size(); ++a) {
        if (accs[a].kind != ACC_SUM_F64) return false;
        std::vector<Builder::Link> links;
        if (!b.chain(accs[a].expr, links)) return false;
        size_t common = 0;
        if (!last_chain.empty() && links.size() > last_chain.size()) {
            common = last_chain.size();
            for (size_t i = 0; i < last_chain.size(); ++i)
                if (links[i].f.text != last_chain[i]) { common = 0; break; }
        }
        for (size_t i = common; i < links.size(); ++i) {
            if (out.prog.n_steps >= SOP_NSTEP) return false;
            SopStep& st = out.prog.steps[out.prog.n_steps++];
            memset(&st, 0, sizeof(st));
            st.mode = (uint8_t)links[i].f.mode;
            st.op = (uint8_t)(i == 0 ? SOP_OP_START : links[i].op);
            st.acc = 0xFF;
            st.lit = links[i].f.lit;
            if (links[i].f.schema_col >= 0) {
                const int c = b.column(links[i].f.schema_col);
                if (c < 0) return false;
                st.col = (uint8_t)c;
            }
        }
        out.prog.steps[out.prog.n_steps - 1].acc = (uint8_t)a;
        last_chain.clear();
        for (auto& l : links) last_chain.push_back(l.f.text);
    }
    out.prog.n_cols = (int)out.col_map.size();
    return true;
}

bool bind_sop(SopPlan& plan, const Batch& b) {
    plan.prog.n_rows = b.n_rows;
    for (size_t i = 0; i < plan.col_map.size(); ++i) {
        const Column& c = b.cols[plan.col_map[i]];
        if (c.validity) return false;          // NULLs present: the VM kernel handles them
        SopColumn& sc = plan.prog.cols[i];
        sc.data = c.data ? c.data->ptr() : nullptr;
        sc.offsets = c.offsets ? c.offsets->as<int32_t>() : nullptr;
        sc.dtype = c.dtype;
        sc.data_bytes = (int32_t)c.data_bytes;
    }
    return true;
}

}  // namespace bhip
