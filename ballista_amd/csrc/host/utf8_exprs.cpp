// utf8_exprs.cpp — expressions that produce Utf8 values, as temporary columns.
//
// Reference: the serde ships lower / upper / trim / ltrim / rtrim (rust/core/src/serde/logical_plan/from_proto.rs:910-918),
// CASE with string branches (ballista.proto LogicalExprNode.case_) and string literals in projections.  DataFusion evaluates
// each as an arrow kernel producing a StringArray; here each such node becomes ONE extra Utf8 column of the input batch
// (kernels_str.hip: lengths -> scan -> bytes), and the expression around it — comparisons, LIKE, group / sort keys, the other
// output columns — goes to the expression VM with the node replaced by a reference to that column.
#include "../str_kernels.h"
#include "../util_kernels.h"
#include "plan.hpp"

namespace bhip {

static bool produces_utf8(const ExprPtr& e, const Schema& schema) {
    if (e->kind == BHIP_EXPR_SCALAR_FN) return str_fn(e->name) >= 0 || sha_fn(e->name) != 0;
    if (e->kind == BHIP_EXPR_CASE) return expr_type(e, schema) == DT_UTF8;
    return false;
}

bool has_utf8_node(const ExprPtr& e, const Schema& schema) {
    if (produces_utf8(e, schema)) return true;
    for (auto& a : e->args)
        if (has_utf8_node(a, schema)) return true;
    return false;
}

Utf8Lowering::Utf8Lowering(const Schema& in) : in_(in) {}

ExprPtr Utf8Lowering::rewrite(const ExprPtr& e, bool output) {
    const bool lit = output && e->kind == BHIP_EXPR_LITERAL && e->dtype == DT_UTF8;
    if (lit || produces_utf8(e, in_)) {
        const std::string text = e->to_string();
        for (size_t i = 0; i < nodes_.size(); ++i)
            if (nodes_[i]->to_string() == text) return make_column(names_[i]);
        nodes_.push_back(e);
        names_.push_back("__utf8_" + std::to_string(nodes_.size() - 1));
        if (in_.index_of(names_.back()) >= 0) fail(BHIP_EINVAL, "column name '" + names_.back() + "' is reserved");
        return make_column(names_.back());
    }
    if (e->args.empty()) return e;
    auto c = std::make_shared<Expr>(*e);
    for (auto& a : c->args) a = rewrite(a, false);
    return c;
}

SchemaPtr Utf8Lowering::schema() const {
    auto s = std::make_shared<Schema>(in_);
    for (size_t i = 0; i < nodes_.size(); ++i) s->fields.push_back(Field{names_[i], DT_UTF8, expr_nullable(nodes_[i], in_)});
    return s;
}

namespace {

Column utf8_from_lengths(const Exec& ex, Temp& tmp, uint32_t* lengths, int64_t n, uint64_t** total_dev) {
    Column out;
    out.dtype = DT_UTF8;
    out.length = n;
    out.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
    uint64_t* total = tmp.get<uint64_t>(1);
    void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n));
    HIP_CHECK(exclusive_scan_u32_i32(ex.stream, lengths, n, out.offsets->as<int32_t>(), true, total, scan_tmp));
    *total_dev = total;
    return out;
}

Column eval_utf8(const Exec& ex, const Batch& in, const ExprPtr& e);

Column eval_literal(const Exec& ex, const Expr& e, int64_t n) {
    Column out;
    out.dtype = DT_UTF8;
    out.length = n;
    out.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
    if (e.is_null) {
        HIP_CHECK(hipMemsetAsync(out.offsets->ptr(), 0, (size_t)(n + 1) * 4, ex.stream));
        out.data = make_buffer(ex, 8);
        out.validity = make_buffer(ex, bitmap_bytes(n) + 8);
        HIP_CHECK(hipMemsetAsync(out.validity->ptr(), 0, bitmap_bytes(n) + 8, ex.stream));
        return out;
    }
    if (e.name.size() > (size_t)STR_LITERAL_MAX) fail(BHIP_ENOTIMPL, "Utf8 literal longer than 240 bytes as a column value");
    if ((uint64_t)n * e.name.size() > 0x7FFFFFFFull) fail(BHIP_EEXEC, "Utf8 column exceeds 2 GiB of value bytes");
    StrLiteral lit;
    lit.len = (int32_t)e.name.size();
    memcpy(lit.bytes, e.name.data(), e.name.size());
    out.data_bytes = n * lit.len;
    out.data = make_buffer(ex, (size_t)out.data_bytes + 8);
    TIMED_LAUNCH_N(ex, "str_broadcast", n, launch_str_broadcast(ex.cfg(), lit, n, out.offsets->as<int32_t>(), out.data->as<uint8_t>()));
    return out;
}

Column eval_transform(const Exec& ex, const Batch& in, const Expr& e) {
    const int kind = str_fn(e.name);
    const Column arg = eval_utf8(ex, in, e.args[0]);
    const int64_t n = in.n_rows;
    Temp tmp(ex);
    uint32_t* lengths = tmp.get<uint32_t>((size_t)n + 1);
    const ColumnRef cr = arg.ref();
    TIMED_LAUNCH_N(ex, "str_transform_lengths", n, launch_str_transform_lengths(ex.cfg(), kind, cr, n, lengths));
    uint64_t* total;
    Column out = utf8_from_lengths(ex, tmp, lengths, n, &total);
    // lower / upper keep the byte count, a trim can only shrink it: the argument's byte count bounds the result
    out.data = make_buffer(ex, (size_t)arg.data_bytes + 8);
    uint32_t* flags = tmp.get<uint32_t>(1);
    HIP_CHECK(hipMemsetAsync(flags, 0, 4, ex.stream));
    TIMED_LAUNCH_N(ex, "str_transform_write", n, launch_str_transform_write(ex.cfg(), kind, cr, n, out.offsets->as<int32_t>(), out.data->as<uint8_t>(), flags));
    struct Back { uint64_t total; uint32_t flags, pad; };
    static_assert(sizeof(Back) == 16, "one read");
    Back* back = tmp.get<Back>(1);
    HIP_CHECK(hipMemcpyAsync(&back->total, total, 8, hipMemcpyDeviceToDevice, ex.stream));
    HIP_CHECK(hipMemcpyAsync(&back->flags, flags, 4, hipMemcpyDeviceToDevice, ex.stream));
    const Back b = read_device(ex, back);
    if (b.flags)
        fail(BHIP_ENOTIMPL, e.name + "() over text with non-ASCII characters (Unicode case mapping is not on the GPU path)");
    out.data_bytes = (int64_t)b.total;
    out.validity = arg.validity;
    return out;
}

// sha224 / sha256 / sha384 / sha512: fixed-length digests, one per row, NULL where the argument is
Column eval_sha(const Exec& ex, const Batch& in, const Expr& e) {
    const int bits = sha_fn(e.name);
    const Column arg = eval_utf8(ex, in, e.args[0]);
    const int64_t n = in.n_rows;
    if ((uint64_t)n * (bits / 8) > 0x7FFFFFFFull) fail(BHIP_EEXEC, "Binary column exceeds 2 GiB of value bytes");
    Column out;
    out.dtype = DT_UTF8;
    out.length = n;
    out.offsets = make_buffer(ex, (size_t)(n + 1) * 4);
    out.data_bytes = n * (bits / 8);
    out.data = make_buffer(ex, (size_t)out.data_bytes + 8);
    if (arg.validity) HIP_CHECK(hipMemsetAsync(out.data->ptr(), 0, (size_t)out.data_bytes + 8, ex.stream));    // a NULL row's bytes stay defined
    TIMED_LAUNCH_N(ex, "sha2", n, launch_sha2(ex.cfg(), bits, arg.ref(), n, out.offsets->as<int32_t>(), out.data->as<uint8_t>()));
    out.validity = arg.validity;
    return out;
}

Column eval_case(const Exec& ex, const Batch& in, const Expr& e) {
    const size_t fw = e.has_base ? 1 : 0;
    const size_t np = (e.args.size() - fw - (e.has_else ? 1 : 0)) / 2;
    if (np > (size_t)STR_SELECT_MAX) fail(BHIP_ENOTIMPL, "CASE producing Utf8 with more than 8 WHEN branches");
    const int64_t n = in.n_rows;
    // the conditions as Boolean columns, in one projection (they may hold string nodes of their own)
    std::vector<std::pair<ExprPtr, std::string>> conds;
    auto cs = std::make_shared<Schema>();
    for (size_t i = 0; i < np; ++i) {
        ExprPtr c = e.args[fw + 2 * i];
        if (e.has_base) c = make_binary(e.args[0], "Eq", c);
        if (expr_type(c, *in.schema) != DT_BOOLEAN) fail(BHIP_EINVAL, "CASE WHEN condition must be Boolean");
        conds.push_back({c, "c" + std::to_string(i)});
        cs->fields.push_back(Field{conds.back().second, DT_BOOLEAN, true});
    }
    BatchPtr cb = project_batch(ex, in, conds, cs);
    std::vector<Column> vals;
    for (size_t i = 0; i < np; ++i) vals.push_back(eval_utf8(ex, in, e.args[fw + 2 * i + 1]));
    if (e.has_else) vals.push_back(eval_utf8(ex, in, e.args.back()));
    StrSelectArgs A;
    memset(&A, 0, sizeof(A));
    A.n_when = (int32_t)np;
    A.has_else = e.has_else ? 1 : 0;
    for (size_t i = 0; i < np; ++i) A.cond[i] = cb->cols[i].ref();
    int64_t bound = 0;
    for (size_t i = 0; i < vals.size(); ++i) { A.val[i] = vals[i].ref(); bound += vals[i].data_bytes; }
    Temp tmp(ex);
    uint32_t* lengths = tmp.get<uint32_t>((size_t)n + 1);
    BufferPtr validity = make_buffer(ex, bitmap_bytes(n) + 8);
    TIMED_LAUNCH_N(ex, "str_select_lengths", n, launch_str_select_lengths(ex.cfg(), A, n, lengths, validity->as<uint64_t>()));
    uint64_t* total;
    Column out = utf8_from_lengths(ex, tmp, lengths, n, &total);
    // a row takes its value from ONE branch: the sum of the branches' byte counts bounds the result when it is small; else read it
    if (bound > (64 << 20)) bound = (int64_t)read_device(ex, total);
    out.data = make_buffer(ex, (size_t)bound + 8);
    TIMED_LAUNCH_N(ex, "str_select_write", n, launch_str_select_write(ex.cfg(), A, n, out.offsets->as<int32_t>(), out.data->as<uint8_t>()));
    out.data_bytes = (int64_t)read_device(ex, total);
    out.validity = validity;
    return out;
}

Column eval_utf8(const Exec& ex, const Batch& in, const ExprPtr& e) {
    switch (e->kind) {
        case BHIP_EXPR_COLUMN: {
            const int i = in.schema->index_of(e->name);
            if (i < 0) fail(BHIP_EINVAL, "No field named '" + e->name + "'");
            if (in.cols[i].dtype != DT_UTF8) fail(BHIP_EINVAL, "expected a Utf8 column: " + e->name);
            return in.cols[i];
        }
        case BHIP_EXPR_LITERAL:
            if (e->dtype != DT_UTF8) fail(BHIP_EINVAL, "expected a Utf8 literal");
            return eval_literal(ex, *e, in.n_rows);
        case BHIP_EXPR_SCALAR_FN:
            if (str_fn(e->name) >= 0) return eval_transform(ex, in, *e);
            if (sha_fn(e->name)) return eval_sha(ex, in, *e);
            break;
        case BHIP_EXPR_CASE: return eval_case(ex, in, *e);
        default: break;
    }
    fail(BHIP_ENOTIMPL, "expression producing Utf8: " + e->to_string());
}

}  // namespace

BatchPtr Utf8Lowering::apply(const Exec& ex, const Batch& in) const {
    auto out = std::make_shared<Batch>(in);
    out->schema = schema();
    for (auto& node : nodes_) {
        if (in.n_rows == 0) {
            Column c;
            c.dtype = DT_UTF8;
            c.offsets = make_buffer(ex, 8);
            HIP_CHECK(hipMemsetAsync(c.offsets->ptr(), 0, 8, ex.stream));
            c.data = make_buffer(ex, 8);
            out->cols.push_back(c);
        } else {
            out->cols.push_back(eval_utf8(ex, in, node));
        }
    }
    return out;
}

// plan-time check: every string node is one this file evaluates (BHIP_ENOTIMPL otherwise, before anything runs)
void Utf8Lowering::validate() const {
    std::function<void(const ExprPtr&)> walk = [&](const ExprPtr& e) {
        switch (e->kind) {
            case BHIP_EXPR_COLUMN: {
                const int i = in_.index_of(e->name);
                if (i < 0) fail(BHIP_EINVAL, "No field named '" + e->name + "'");
                if (in_.fields[i].dtype != DT_UTF8) fail(BHIP_EINVAL, "expected a Utf8 column: " + e->name);
            } break;
            case BHIP_EXPR_LITERAL:
                if (e->dtype != DT_UTF8) fail(BHIP_EINVAL, "expected a Utf8 literal");
                break;
            case BHIP_EXPR_SCALAR_FN:
                if (str_fn(e->name) < 0 && !sha_fn(e->name)) fail(BHIP_ENOTIMPL, "expression producing Utf8: " + e->to_string());
                walk(e->args[0]);
                break;
            case BHIP_EXPR_CASE: {
                const size_t fw = e->has_base ? 1 : 0;
                const size_t np = (e->args.size() - fw - (e->has_else ? 1 : 0)) / 2;
                if (np > (size_t)STR_SELECT_MAX) fail(BHIP_ENOTIMPL, "CASE producing Utf8 with more than 8 WHEN branches");
                for (size_t i = 0; i < np; ++i) walk(e->args[fw + 2 * i + 1]);
                if (e->has_else) walk(e->args.back());
            } break;
            default: fail(BHIP_ENOTIMPL, "expression producing Utf8: " + e->to_string());
        }
    };
    for (auto& n : nodes_) walk(n);
}

}  // namespace bhip
