// ops_agg_wide.cpp — HashAggregateExec over group keys that do not fit the 16-byte packed key (several Utf8 columns,
// long strings: TPC-H Q10 groups by c_name, c_address, c_phone, c_comment ...).  Reference operator:
// rust/core/src/serde/physical_plan/from_proto.rs:173-252 (HashAggregateExec, any group expressions).
//
// Two steps on top of the packed-key machinery:
//   1. wide_key_assign_kernel (kernels_hash.hip) gives every input row a representative row with an equal key
//      (hash table of row ids over 64-bit row hashes; equality on the key columns themselves, NULL == NULL);
//   2. the ordinary aggregate runs with that ONE Int32 column as its key, and the key columns of the result are
//      gathered from the input at the representative rows.
#include "../str_kernels.h"
#include "hash_kernels.h"
#include "plan.hpp"

namespace bhip {

namespace {
const char* const kRepName = "__group_rep";

bool key_width_error(const Error& e) {
    if (e.code != BHIP_ENOTIMPL) return false;
    const std::string m = e.what();
    return m.find("packed key") != std::string::npos || m.find("packed-key") != std::string::npos;
}
}  // namespace

bool HashAggregateExec::run_single_partial(const Exec& ex, std::vector<BatchPtr>& out) const {
    static const bool disabled = [] { const char* v = getenv("BHIP_NO_FINAL_ELISION"); return v && atoi(v) != 0; }();
    if (disabled || mode_ != BHIP_AGG_FINAL) return false;
    const ExecutionPlan* p = input_.get();
    PlanPtr below;
    while (true) {
        if (auto m = dynamic_cast<const MergeExec*>(p)) { below = m->input(); p = below.get(); continue; }
        if (auto c = dynamic_cast<const CoalesceBatchesExec*>(p)) { below = c->input(); p = below.get(); continue; }
        break;
    }
    auto pa = dynamic_cast<const HashAggregateExec*>(p);
    if (!pa || !below || pa->mode_ != BHIP_AGG_PARTIAL || pa->output_partitioning().count != 1 || pa->strings_) return false;
    if (pa->group_.size() != group_.size() || pa->aggr_.size() != aggr_.size()) return false;
    const Schema& ps = *pa->schema();
    // this operator's keys must be the partial's key columns, in order, and the functions must be the same
    for (size_t i = 0; i < group_.size(); ++i)
        if (group_[i].first->kind != BHIP_EXPR_COLUMN || group_[i].first->name != ps.fields[i].name) return false;
    for (size_t i = 0; i < aggr_.size(); ++i)
        if (aggr_[i].fn != pa->aggr_[i].fn) return false;
    // the Partial emits its group table as this operator's columns (AVG = sum / count in the emit kernel): no state batch at all
    if (group_.empty() || !pa->wide_keys_.load()) {
        try {
            out = pa->run_packed(0, ex, this);
            return true;
        } catch (const Error& e) {
            if (group_.empty() || !key_width_error(e)) throw;
            pa->wide_keys_.store(true);
        }
    }
    // wide keys: the Partial's state batch, then AVG = [sum] / [count] as a projection
    std::vector<std::pair<ExprPtr, std::string>> exprs;
    for (size_t i = 0; i < group_.size(); ++i) exprs.push_back({make_column(ps.fields[i].name), group_[i].second});
    size_t pos = group_.size();
    for (auto& a : aggr_) {
        if (a.fn == BHIP_AGG_AVG) {
            auto cast = std::make_shared<Expr>();
            cast->kind = BHIP_EXPR_CAST;
            cast->dtype = DT_FLOAT64;
            cast->args = {make_column(ps.fields[pos].name)};                          // [count]
            exprs.push_back({make_binary(make_column(ps.fields[pos + 1].name), "Divide", ExprPtr(cast)), a.name});
            pos += 2;
        } else {
            exprs.push_back({make_column(ps.fields[pos].name), a.name});
            pos += 1;
        }
    }
    auto s = below->execute(0, ex);
    while (BatchPtr b = s->next()) out.push_back(project_batch(ex, *b, exprs, schema_));
    return true;
}

std::vector<BatchPtr> HashAggregateExec::run(int partition, const Exec& ex) const {
    if (strings_) return run_strings(partition, ex);
    {
        std::vector<BatchPtr> out;
        if (run_single_partial(ex, out)) return out;
    }
    if (!group_.empty() && wide_keys_.load()) return run_wide(partition, ex);
    try {
        return run_packed(partition, ex);
    } catch (const Error& e) {
        if (group_.empty() || !key_width_error(e)) throw;
    }
    wide_keys_.store(true);
    return run_wide(partition, ex);
}

std::vector<BatchPtr> HashAggregateExec::run_wide(int partition, const Exec& ex) const {
    const SchemaPtr in_schema = input_->schema();
    if (in_schema->index_of(kRepName) >= 0) fail(BHIP_EINVAL, std::string("column name '") + kRepName + "' is reserved");
    if (group_.size() > (size_t)VM_MAX_COLS) fail(BHIP_ENOTIMPL, "more than 16 group expressions");
    std::vector<BatchPtr> parts;
    {
        auto s = input_->execute(partition, ex);
        while (BatchPtr b = s->next())
            if (b->n_rows > 0) parts.push_back(b);
    }
    if (parts.empty()) return {};
    const BatchPtr in = parts.size() == 1 ? parts[0] : concat_batches(ex, in_schema, parts);
    parts.clear();
    const int64_t n = in->n_rows;
    if (n > 0x7FFFFFF0ll) fail(BHIP_ENOTIMPL, "wide-key aggregate over more than 2^31 input rows per partition");
    const LaunchCfg cfg = ex.cfg();

    // ---- key columns, their row hashes, the representative of every row -------------------------------------
    std::vector<Column> keys;
    for (auto& g : group_) keys.push_back(evaluate_column(ex, *in, g.first));
    BufferPtr rep = make_buffer(ex, (size_t)n * 4 + 8);
    {
        Temp tmp(ex);
        ProgramBuilder pb(*in_schema);
        pb.set_hash_only();
        for (auto& g : group_) pb.add_key(g.first);
        ScanParams P;
        pb.finish(P);
        ProgramBuilder::bind(P, pb.columns(), *in, pb.creates_nulls());
        uint64_t* hashes = tmp.get<uint64_t>((size_t)n);
        ScanStatus* st = tmp.get<ScanStatus>(1);
        HIP_CHECK(hipMemsetAsync(st, 0, sizeof(ScanStatus), ex.stream));
        TIMED_LAUNCH(ex, "scan_keys", launch_scan_keys(cfg, P, nullptr, hashes, nullptr, st));
        check_scan_status(ex, st);
        uint64_t cap = 1024;
        while (cap < 2ull * (uint64_t)n) cap <<= 1;
        uint32_t* table = tmp.get<uint32_t>(cap);
        HIP_CHECK(hipMemsetAsync(table, 0, cap * 4, ex.stream));
        WideKeyCols K;
        memset(&K, 0, sizeof(K));
        K.n = (int32_t)keys.size();
        for (size_t i = 0; i < keys.size(); ++i) K.col[i] = keys[i].ref();
        TIMED_LAUNCH(ex, "wide_key_assign", launch_wide_key_assign(cfg, K, hashes, table, cap - 1, (uint32_t)n, rep->as<uint32_t>()));
    }

    // ---- the ordinary aggregate, keyed by the representative row ---------------------------------------------
    auto s2 = std::make_shared<Schema>();
    auto aug = std::make_shared<Batch>();
    aug->ctx = ex.ctx;
    aug->n_rows = n;
    s2->fields.push_back(Field{kRepName, DT_INT32, false});
    {
        Column c;
        c.dtype = DT_INT32;
        c.length = n;
        c.data = rep;
        aug->cols.push_back(std::move(c));
    }
    // Partial: the arguments are expressions over the input's names; Final: the state columns follow the key by position
    const size_t first = mode_ == BHIP_AGG_PARTIAL ? 0 : group_.size();
    for (size_t i = first; i < in_schema->fields.size(); ++i) {
        s2->fields.push_back(in_schema->fields[i]);
        aug->cols.push_back(in->cols[i]);
    }
    aug->schema = s2;
    auto src = std::make_shared<MemoryExec>(ex.ctx, s2, std::vector<std::vector<BatchPtr>>{{BatchPtr(aug)}});
    auto inner = std::make_shared<HashAggregateExec>(
        mode_, std::vector<std::pair<ExprPtr, std::string>>{{make_column(kRepName), kRepName}}, aggr_, src);
    if (n >= 4096) inner->path_hint_.store(-1);                     // many rows: straight to the device-wide table
    std::vector<BatchPtr> res = inner->run_packed(0, ex);

    // ---- key columns of the groups: the input's, at the representative rows ----------------------------------
    std::vector<BatchPtr> outv;
    for (auto& r : res) {
        auto out = std::make_shared<Batch>();
        out->schema = schema_;
        out->ctx = ex.ctx;
        out->n_rows = r->n_rows;
        std::vector<const Column*> kp;
        for (auto& k : keys) kp.push_back(&k);
        std::vector<Column> got = take_columns(ex, kp, r->cols[0].data->as<uint32_t>(), r->n_rows, false);
        for (auto& c : got) out->cols.push_back(std::move(c));
        for (size_t i = 1; i < r->cols.size(); ++i) out->cols.push_back(r->cols[i]);
        outv.push_back(out);
    }
    return outv;
}

// ---- string nodes in the keys / arguments, MIN / MAX over Utf8 -------------------------------------------------------
// MIN(s) over strings = the string at the smallest position in the sort order of s: the column is sorted once (the SortExec
// passes), every row gets its rank as an Int64 with the string's validity, the ordinary aggregate takes MIN / MAX of the ranks,
// and the result's strings are gathered through the sort permutation.
std::vector<BatchPtr> HashAggregateExec::run_strings(int partition, const Exec& ex) const {
    const SchemaPtr in_schema = input_->schema();
    std::vector<BatchPtr> parts;
    {
        auto s = input_->execute(partition, ex);
        while (BatchPtr b = s->next())
            if (b->n_rows > 0) parts.push_back(b);
    }
    BatchPtr in;
    if (parts.empty()) {
        auto e = std::make_shared<Batch>();
        e->schema = in_schema;
        e->ctx = ex.ctx;
        for (auto& f : in_schema->fields) {
            Column c;
            c.dtype = f.dtype;
            c.data = make_buffer(ex, 8);
            if (f.dtype == DT_UTF8) { c.offsets = make_buffer(ex, 8); HIP_CHECK(hipMemsetAsync(c.offsets->ptr(), 0, 8, ex.stream)); }
            e->cols.push_back(c);
        }
        in = e;
    } else {
        in = parts.size() == 1 ? parts[0] : concat_batches(ex, in_schema, parts);
    }
    parts.clear();
    const int64_t n = in->n_rows;
    const LaunchCfg cfg = ex.cfg();

    // the string nodes as columns
    Utf8Lowering low(*in_schema);
    std::vector<std::pair<ExprPtr, std::string>> group2;
    for (auto& g : group_) group2.push_back({low.rewrite(g.first, true), g.second});
    std::vector<AggregateDesc> aggr2 = aggr_;
    if (mode_ == BHIP_AGG_PARTIAL)
        for (auto& a : aggr2) a.arg = low.rewrite(a.arg, true);
    const BatchPtr aug = low.any() ? low.apply(ex, *in) : in;
    auto s2 = std::make_shared<Schema>(*aug->schema);
    auto b2 = std::make_shared<Batch>(*aug);

    // MIN / MAX over Utf8 -> over ranks
    struct Ranked { size_t agg; Column strings; BufferPtr perm; };
    std::vector<Ranked> ranked;
    size_t pos = group_.size();
    for (size_t i = 0; i < aggr2.size(); ++i) {
        AggregateDesc& a = aggr2[i];
        const size_t state_pos = pos;
        pos += a.fn == BHIP_AGG_AVG ? 2 : 1;
        if (a.fn != BHIP_AGG_MIN && a.fn != BHIP_AGG_MAX) continue;
        int ci;
        if (mode_ == BHIP_AGG_PARTIAL) {
            if (expr_type(a.arg, *s2) != DT_UTF8) continue;
            if (a.arg->kind != BHIP_EXPR_COLUMN) fail(BHIP_ENOTIMPL, "MIN/MAX over a Utf8 expression: " + a.arg->to_string());
            ci = s2->index_of(a.arg->name);
        } else {
            ci = (int)state_pos;
            if (s2->fields[ci].dtype != DT_UTF8) continue;
        }
        const Column strings = aug->cols[ci];
        Ranked r;
        r.agg = i;
        r.strings = strings;
        Column rank;
        rank.dtype = DT_INT64;
        rank.length = n;
        rank.data = make_buffer(ex, (size_t)n * 8 + 8);
        rank.validity = strings.validity;
        if (n > 0) {
            r.perm = sort_permutation(ex, *aug, {SortDesc{make_column(s2->fields[ci].name), false, false}});
            TIMED_LAUNCH_N(ex, "invert_perm", n, launch_invert_perm(cfg, r.perm->as<uint32_t>(), n, rank.data->as<int64_t>()));
        }
        if (mode_ == BHIP_AGG_PARTIAL) {
            const std::string name = "__rank_" + std::to_string(i);
            s2->fields.push_back(Field{name, DT_INT64, s2->fields[ci].nullable});
            b2->cols.push_back(rank);
            a.arg = make_column(name);
        } else {
            s2->fields[ci].dtype = DT_INT64;                  // Final reads its states by position
            b2->cols[ci] = rank;
        }
        ranked.push_back(std::move(r));
    }
    b2->schema = s2;
    auto src = std::make_shared<MemoryExec>(ex.ctx, s2, std::vector<std::vector<BatchPtr>>{{BatchPtr(b2)}});
    auto inner = std::make_shared<HashAggregateExec>(mode_, group2, aggr2, src);
    if (inner->strings_) fail(BHIP_ENOTIMPL, "aggregate over string expressions: " + describe());
    std::vector<BatchPtr> res = inner->run(0, ex);

    std::vector<BatchPtr> outv;
    for (auto& r : res) {
        auto out = std::make_shared<Batch>(*r);
        out->schema = schema_;
        for (auto& rk : ranked) {
            size_t col = group_.size();
            for (size_t i = 0; i < rk.agg; ++i) col += aggr_[i].fn == BHIP_AGG_AVG && mode_ == BHIP_AGG_PARTIAL ? 2 : 1;
            const Column& ranks = r->cols[col];
            BufferPtr idx = make_buffer(ex, (size_t)r->n_rows * 4 + 8);
            TIMED_LAUNCH_N(ex, "rank_to_row", r->n_rows, launch_rank_to_row(cfg, ranks.data->as<int64_t>(), ranks.validity ? ranks.validity->as<uint64_t>() : nullptr,
                                                                             rk.perm ? rk.perm->as<uint32_t>() : nullptr, r->n_rows, idx->as<uint32_t>()));
            std::vector<const Column*> one{&rk.strings};
            out->cols[col] = take_columns(ex, one, idx->as<uint32_t>(), r->n_rows, /*may_null=*/true, false)[0];
        }
        outv.push_back(out);
    }
    return outv;
}

}  // namespace bhip
