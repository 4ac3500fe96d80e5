// ops_agg.cpp — HashAggregateExec (Partial / Final) over the fused scan kernels.
//
// Reference: built at rust/core/src/serde/physical_plan/from_proto.rs:173-252 (mode :181-184,
// aggregates via create_aggregate_expr :230-236; only Sum/Avg/Count are serialisable,
// to_proto.rs:352-363); the stage split Partial | Final is rust/scheduler/src/planner.rs:149-171.
// State layout: SUM -> [sum]; AVG -> [count: UInt64, sum: Float64];
// COUNT -> [count: UInt64].  Final merges by position: state columns follow the group columns
// in aggregate order.
//
// execute() folds the FilterExec / CoalesceBatchesExec / ProjectionExec chain below the
// aggregate into the aggregate's own kernel (predicate fused, projection expressions
// substituted), i.e. the whole of TPC-H Q1/Q6 stage 1 is one kernel launch per input batch.
#include "../sort_kernels.h"
#include "../util_kernels.h"
#include "hash_kernels.h"
#include "plan.hpp"
#include "sop.hpp"

namespace bhip {

// sum_return_type of DataFusion 4.0 (physical_plan/aggregates.rs): signed -> Int64, unsigned -> UInt64, Float32 -> Float32, Float64 -> Float64
static int sum_type(int t) {
    if (dt_is_float(t)) return t;
    if (dt_is_unsigned(t)) return DT_UINT64;
    return DT_INT64;
}

static const char* agg_name(int fn) {
    switch (fn) {
        case BHIP_AGG_SUM: return "SUM";
        case BHIP_AGG_AVG: return "AVG";
        case BHIP_AGG_COUNT: return "COUNT";
        case BHIP_AGG_MIN: return "MIN";
        default: return "MAX";
    }
}

HashAggregateExec::HashAggregateExec(int mode, std::vector<std::pair<ExprPtr, std::string>> group_exprs,
                                     std::vector<AggregateDesc> aggr, PlanPtr input)
    : mode_(mode), group_(std::move(group_exprs)), aggr_(std::move(aggr)) {
    input_ = std::move(input);
    ctx_ = input_->context();
    if (mode != BHIP_AGG_PARTIAL && mode != BHIP_AGG_FINAL) fail(BHIP_EINVAL, "Unsupported aggregate mode");
    const Schema& in = *input_->schema();
    auto s = std::make_shared<Schema>();
    for (auto& g : group_) {
        const int t = expr_type(g.first, in);
        s->fields.push_back(Field{g.second, t, expr_nullable(g.first, in), t == DT_UTF8 && expr_large(g.first, in), t == DT_UTF8 && expr_binary(g.first, in)});
    }
    size_t state_pos = group_.size();
    for (auto& a : aggr_) {
        if (a.fn < BHIP_AGG_SUM || a.fn > BHIP_AGG_MAX) fail(BHIP_ENOTIMPL, "Unsupported aggregate function");
        if (mode == BHIP_AGG_PARTIAL) {
            const int t = expr_type(a.arg, in);
            if ((a.fn == BHIP_AGG_SUM || a.fn == BHIP_AGG_AVG) && (t == DT_UTF8 || t == DT_BOOLEAN))
                fail(BHIP_EINVAL, std::string(agg_name(a.fn)) + " does not support " + dtype_name(t));
            switch (a.fn) {
                case BHIP_AGG_SUM: s->fields.push_back(Field{a.name + "[sum]", sum_type(t), true}); break;
                case BHIP_AGG_AVG:
                    s->fields.push_back(Field{a.name + "[count]", DT_UINT64, false});
                    s->fields.push_back(Field{a.name + "[sum]", DT_FLOAT64, true});
                    break;
                case BHIP_AGG_COUNT: s->fields.push_back(Field{a.name + "[count]", DT_UINT64, false}); break;
                case BHIP_AGG_MIN: s->fields.push_back(Field{a.name + "[min]", t, true, t == DT_UTF8 && expr_large(a.arg, in)}); break;
                default: s->fields.push_back(Field{a.name + "[max]", t, true, t == DT_UTF8 && expr_large(a.arg, in)}); break;
            }
        } else {
            const size_t need = a.fn == BHIP_AGG_AVG ? 2 : 1;
            if (state_pos + need > in.fields.size()) fail(BHIP_EINVAL, "Final aggregate: input has too few state columns");
            switch (a.fn) {
                case BHIP_AGG_AVG: s->fields.push_back(Field{a.name, DT_FLOAT64, true}); break;
                case BHIP_AGG_COUNT: s->fields.push_back(Field{a.name, DT_UINT64, false}); break;
                default: s->fields.push_back(Field{a.name, in.fields[state_pos].dtype, true, in.fields[state_pos].large}); break;
            }
            state_pos += need;
        }
    }
    schema_ = s;
    // string nodes in the expressions, MIN / MAX over Utf8: run_strings (ops_agg_wide.cpp)
    for (auto& g : group_) strings_ = strings_ || has_utf8_node(g.first, in) || (g.first->kind == BHIP_EXPR_LITERAL && g.first->dtype == DT_UTF8);
    size_t sp = group_.size();
    for (auto& a : aggr_) {
        if (mode == BHIP_AGG_PARTIAL) {
            strings_ = strings_ || has_utf8_node(a.arg, in);
            if ((a.fn == BHIP_AGG_MIN || a.fn == BHIP_AGG_MAX) && expr_type(a.arg, in) == DT_UTF8) strings_ = true;
        } else {
            if ((a.fn == BHIP_AGG_MIN || a.fn == BHIP_AGG_MAX) && in.fields[sp].dtype == DT_UTF8) strings_ = true;
            sp += a.fn == BHIP_AGG_AVG ? 2 : 1;
        }
    }
}

PlanPtr HashAggregateExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 1) fail(BHIP_EINVAL, "HashAggregateExec wrong number of children");
    return std::make_shared<HashAggregateExec>(mode_, group_, aggr_, c[0]);
}

std::string HashAggregateExec::describe() const {
    std::string s = std::string("HashAggregateExec: mode=") + (mode_ == BHIP_AGG_PARTIAL ? "Partial" : "Final") + ", gby=[";
    for (size_t i = 0; i < group_.size(); ++i) s += (i ? ", " : "") + group_[i].first->to_string();
    s += "], aggr=[";
    for (size_t i = 0; i < aggr_.size(); ++i)
        s += (i ? ", " : "") + std::string(agg_name(aggr_[i].fn)) + "(" + aggr_[i].arg->to_string() + ")";
    return s + "]";
}

StreamPtr HashAggregateExec::execute(int partition, const Exec& ex) const {
    check_partition(*this, partition);
    auto self = std::static_pointer_cast<const HashAggregateExec>(shared_from_this());
    return StreamPtr(new LazyStream(schema_, [self, partition, ex]() { return self->run(partition, ex); }));
}

namespace {

struct EmitPlan {
    EmitValueSpec spec;
};

struct FusedInput {
    PlanPtr source;
    ExprPtr predicate;                       // over source schema, may be null
    std::vector<ExprPtr> group;              // over source schema
    std::vector<ExprPtr> args;               // Partial: aggregate arguments; Final: unused
};

FusedInput fuse_below(const PlanPtr& input, std::vector<ExprPtr> group, std::vector<ExprPtr> args) {
    FusedInput f;
    f.source = input;
    f.group = std::move(group);
    f.args = std::move(args);
    for (;;) {
        if (auto* flt = dynamic_cast<const FilterExec*>(f.source.get())) {
            if (has_utf8_node(flt->predicate(), *flt->input()->schema())) break;      // string nodes: the filter runs on its own (utf8_exprs.cpp)
            f.predicate = f.predicate ? make_binary(flt->predicate(), "And", f.predicate) : flt->predicate();
            f.source = flt->input();
        } else if (auto* co = dynamic_cast<const CoalesceBatchesExec*>(f.source.get())) {
            f.source = co->input();
        } else if (auto* pr = dynamic_cast<const ProjectionExec*>(f.source.get())) {
            bool strings = false;
            for (auto& en : pr->exprs()) strings = strings || has_utf8_node(en.first, *pr->input()->schema()) || (en.first->kind == BHIP_EXPR_LITERAL && en.first->dtype == DT_UTF8);
            if (strings) break;
            std::map<std::string, ExprPtr> subst;
            for (auto& en : pr->exprs()) subst[en.second] = en.first;
            for (auto& g : f.group) g = substitute(g, subst);
            for (auto& a : f.args) a = substitute(a, subst);
            if (f.predicate) f.predicate = substitute(f.predicate, subst);
            f.source = pr->input();
        } else {
            break;
        }
    }
    return f;
}

struct TimedLaunches {
    const Exec& ex;
    bool on;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    explicit TimedLaunches(const Exec& e) : ex(e), on(e.ctx->timing_enabled()) {}
    void begin() {
        if (!on) return;
        hipEvent_t a, b;
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, ex.stream));
        ev.push_back({a, b});
    }
    void end() {
        if (on) HIP_CHECK(hipEventRecord(ev.back().second, ex.stream));
    }
    const char* kernel = nullptr;
    void collect() {   // call after the stream was synchronised
        if (!on) return;
        double ms = 0;
        for (auto& p : ev) {
            float t = 0;
            if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        ex.ctx->add_kernel_time(ms, ev.size(), kernel);
        ev.clear();
    }
};

// high-cardinality path (device-wide hash table); filled in by ops_agg_hash.cpp
GroupRec* hash_aggregate(const Exec& ex, Temp& tmp, const ScanParams& P0, const ProgramBuilder& pb,
                         const std::vector<BatchPtr>& inputs, bool nullable, int64_t* n_groups, ScanStatus* status,
                         TimedLaunches& /*timer: only the register-path scan kernel is the timed (dominant) kernel*/,
                         std::atomic<int>* clustered_hint, SlotSource* slots_out) {
    const LaunchCfg cfg = ex.cfg();
    int64_t total_rows = 0;
    for (auto& b : inputs) total_rows += b->n_rows;
    // slots are 32-bit indices into a table of >= 2 x rows entries
    if (total_rows > 0x7FFFFFF0ll) fail(BHIP_ENOTIMPL, "hash aggregate over more than 2^31 input rows per partition");
    const int n_acc = P0.n_acc > 0 ? P0.n_acc : 1;
    HashAggTable T;
    memset(&T, 0, sizeof(T));
    T.n_acc = P0.n_acc;
    uint64_t* keys = tmp.get<uint64_t>(2 * (size_t)total_rows);
    T.keys128 = keys;
    MergeAccKinds kinds;
    for (int i = 0; i < VM_MAX_ACC; ++i) kinds.kind[i] = i < P0.n_acc ? P0.acc[i].kind : (uint8_t)ACC_COUNT_ROWS;
    uint64_t* tail = tmp.get<uint64_t>(2);          // [0] the group count, [1] the spill list's entry count | "lists too long" << 32: read in one piece
    {
        FillMany fm;
        static_assert(sizeof(ScanStatus) % 4 == 0, "cleared word-wise");
        fm.add(status, sizeof(ScanStatus));
        fm.add(tail, 16);
        TIMED_LAUNCH(ex, "fill_many", launch_fill_many(cfg, fm));
    }
    // SUM(Float64) accumulators are summed in row order after the scan (kernels_dagg.hip); BHIP_AGG_ATOMIC=1: atomic adds
    // (order of addition left to the scheduler: the same sums to ~1e-16 relative, not bit for bit)
    static const bool atomic_sums = [] { const char* v = getenv("BHIP_AGG_ATOMIC"); return v && atoi(v) != 0; }();
    memset(T.fsum_of_acc, 0xFF, sizeof(T.fsum_of_acc));
    DetSum D;
    memset(&D, 0, sizeof(D));
    if (!atomic_sums && total_rows > 0) {
        for (int a = 0; a < P0.n_acc; ++a)
            if (P0.acc[a].kind == ACC_SUM_F64) { D.acc_of_fsum[T.n_fsum] = (uint8_t)a; T.fsum_of_acc[a] = (uint8_t)T.n_fsum++; }
    }
    T.total_rows = (uint64_t)total_rows;
    T.rowslot = tmp.get<uint32_t>((size_t)total_rows + 1);
    if (T.n_fsum) T.fvals = tmp.get<double>((size_t)total_rows * T.n_fsum);

    // ---- the packed key of every row --------------------------------------------------------------------------------------
    {
        // plain NULL-free integer / date key columns: a streaming pack (kernels_util.hip) instead of a launch of the expression VM
        static const bool no_fixed_pack = [] { const char* v = getenv("BHIP_NO_FIXED_KEY_PACK"); return v && atoi(v) != 0; }();
        std::vector<ProgramBuilder::PlainKeyPart> parts;
        const bool plain = !no_fixed_pack && pb.plain_fixed_keys(parts) && parts.size() <= (size_t)FIXED_KEY_PARTS_MAX;
        uint32_t row_base = 0;
        for (auto& b : inputs) {
            bool packed = false;
            if (plain) {
                FixedKeyParts K;
                memset(&K, 0, sizeof(K));
                K.n = (int32_t)parts.size();
                packed = true;
                for (size_t p = 0; p < parts.size(); ++p) {
                    const Column& c = b->cols[(size_t)parts[p].schema_index];
                    if (c.validity || c.is_view() || !c.data) { packed = false; break; }
                    K.src[p] = c.data->ptr(); K.width[p] = (uint8_t)parts[p].width; K.pos[p] = (uint8_t)parts[p].pos;
                }
                if (packed) TIMED_LAUNCH_N(ex, "pack_fixed_keys", b->n_rows, launch_pack_fixed_keys(cfg, K, b->n_rows, keys + 2ull * row_base));
            }
            if (!packed) {
                ScanParams P = P0;
                ProgramBuilder::bind(P, pb.columns(), *b, nullable);
                TIMED_LAUNCH_N(ex, "scan_keys", b->n_rows, launch_scan_keys(cfg, P, keys + 2ull * row_base, nullptr, nullptr, status));
            }
            row_base += (uint32_t)b->n_rows;
        }
    }

    // ---- rows of a group mostly consecutive?  the table is consulted per RUN of equal keys, slots = runs (kernels_hash.hip) ------
    // Decided on the leading rows: at most half as many runs as rows.  Not with a fused predicate (a filtered-out row would
    // have to leave its run).  The operator remembers what it found.
    static const bool no_runs = [] { const char* v = getenv("BHIP_NO_RUN_AGG"); return v && atoi(v) != 0; }();
    // clustered input: 0 = look every run up in the run table; 1 = the runs are distinct groups (first key part ascending);
    // 2 = two ascending stretches (one place where it does not ascend): the second stretch is matched against the first by binary search
    bool runs = false;
    int distinct_runs = 0;
    uint32_t* run_head = nullptr;
    uint64_t n_runs_host = 0;
    if (!no_runs && P0.pred_slot < 0 && total_rows >= 4096 && clustered_hint->load() >= 0) {
        uint32_t* flags = tmp.get<uint32_t>((size_t)total_rows + 1);
        uint32_t* before = tmp.get<uint32_t>((size_t)total_rows + 1);
        uint64_t* n_runs_dev = tmp.get<uint64_t>(3);               // [0] runs, [1] places where the first key part does not ascend, [2] the first of them
        void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(total_rows));
        // the first key part's bytes in the packed key (parts are laid out from byte 0: ProgramBuilder::finish)
        const int w0 = P0.n_keyparts > 0 ? P0.keyparts[0].width : 0;
        const uint64_t first_mask = w0 >= 8 ? ~0ull : w0 > 0 ? ((1ull << (8 * w0)) - 1ull) : 0ull;
        static const bool no_distinct = [] { const char* v = getenv("BHIP_NO_DISTINCT_RUNS"); return v && atoi(v) != 0; }();      // A/B: always the run table
        const int64_t sample = std::min<int64_t>(total_rows, 1 << 20);
        const uint64_t info0[3] = {0, 0, ~0ull};
        HIP_CHECK(hipMemcpyAsync(n_runs_dev, info0, sizeof(info0), hipMemcpyHostToDevice, ex.stream));
        TIMED_LAUNCH_N(ex, "run_heads", sample, launch_run_heads(cfg, keys, (uint32_t)sample, flags, first_mask, n_runs_dev + 1));
        HIP_CHECK(exclusive_scan_u32_u32(ex.stream, flags, sample, before, false, n_runs_dev, scan_tmp));
        struct RunInfo { uint64_t n_runs, breaks, first_break; };
        RunInfo ri = read_device(ex, reinterpret_cast<const RunInfo*>(n_runs_dev));
        runs = clustered_hint->load() == 1 || 2 * ri.n_runs <= (uint64_t)sample;
        clustered_hint->store(runs ? 1 : -1);
        if (runs) {
            if (sample < total_rows) {
                HIP_CHECK(hipMemcpyAsync(n_runs_dev, info0, sizeof(info0), hipMemcpyHostToDevice, ex.stream));
                TIMED_LAUNCH_N(ex, "run_heads", total_rows, launch_run_heads(cfg, keys, (uint32_t)total_rows, flags, first_mask, n_runs_dev + 1));
                HIP_CHECK(exclusive_scan_u32_u32(ex.stream, flags, total_rows, before, false, n_runs_dev, scan_tmp));
                // one more (short) wait: with the run count on the host every table below is sized by the runs, not by the rows, and
                // when the runs turn out distinct the run table, the slot flags, their scan and the slot compaction are not run at all
                if (ri.breaks <= 1 && !no_distinct) ri = read_device(ex, reinterpret_cast<const RunInfo*>(n_runs_dev));
                else ri.breaks = 2;
            }
            distinct_runs = (no_distinct || first_mask == 0 || ri.breaks > 1) ? 0 : ri.breaks == 0 ? 1 : 2;
            n_runs_host = ri.n_runs;
            uint32_t* head = tmp.get<uint32_t>((size_t)total_rows + 1);
            run_head = head;
            TIMED_LAUNCH_N(ex, "run_slots", total_rows, launch_run_slots(cfg, flags, before, (uint32_t)total_rows, T.rowslot, head));
            if (distinct_runs == 2) {
                // the runs from the break on: matched against the first stretch, new groups numbered behind it
                const uint32_t split_row = (uint32_t)ri.first_break;
                uint32_t* head2 = tmp.get<uint32_t>((size_t)n_runs_host + 1);
                uint32_t* match = tmp.get<uint32_t>((size_t)n_runs_host + 1);
                uint32_t* fresh = tmp.get<uint32_t>((size_t)n_runs_host + 1);
                uint32_t* fresh_before = tmp.get<uint32_t>((size_t)n_runs_host + 2);
                // (the second stretch's length is only known on the device: sized by the runs; entries past it are never read)
                HIP_CHECK(hipMemsetAsync(fresh, 0, ((size_t)n_runs_host + 1) * 4, ex.stream));
                TIMED_LAUNCH_N(ex, "run_tail_resolve", total_rows, launch_run_tail_resolve(cfg, keys, head, n_runs_dev, T.rowslot, (uint32_t)total_rows, split_row, first_mask,
                                                                                          head2, match, fresh));
                // the scan runs over all `n_runs` entries of `fresh` (zeros past the second stretch): fresh_before[t] for t < tail length,
                // and fresh_before[tail length] = the number of new groups, whatever the tail length is
                HIP_CHECK(exclusive_scan_u32_u32(ex.stream, fresh, (int64_t)n_runs_host + 1, fresh_before, false, nullptr, scan_tmp));
                TIMED_LAUNCH_N(ex, "run_tail_remap", total_rows, launch_run_tail_remap(cfg, head, n_runs_dev, (uint32_t)total_rows, split_row, match, fresh_before, T.rowslot,
                                                                                      head2, tail));
                run_head = head2;
            } else if (!distinct_runs) {
                uint64_t tcap = 1024;
                while (tcap < 2ull * (uint64_t)total_rows) tcap <<= 1;
                uint32_t* table = tmp.get<uint32_t>(tcap);
                uint32_t* min_head = tmp.get<uint32_t>(tcap);
                uint32_t* slot_of_run = tmp.get<uint32_t>((size_t)total_rows + 1);
                uint32_t* winner = tmp.get<uint32_t>((size_t)total_rows + 1);
                T.owner = tmp.get<uint32_t>((size_t)total_rows + 1);
                HIP_CHECK(hipMemsetAsync(table, 0, tcap * 4, ex.stream));
                HIP_CHECK(hipMemsetAsync(min_head, 0xFF, tcap * 4, ex.stream));
                HIP_CHECK(hipMemsetAsync(T.owner, 0, ((size_t)total_rows + 1) * 4, ex.stream));
                TIMED_LAUNCH_N(ex, "run_groups", total_rows, launch_run_groups(cfg, keys, (uint32_t)total_rows, head, n_runs_dev, table, tcap - 1, min_head, slot_of_run,
                                                                                winner, T.owner, T.rowslot));
            }
        }
    }

    uint64_t cap;
    bool table_owner = false;
    if (runs) {
        // the slot space is the space of runs: at most one per row (unused ones stay empty); exactly the runs when they are distinct
        cap = distinct_runs ? std::max<uint64_t>(n_runs_host, 1) : (uint64_t)total_rows;
        T.slots_given = 1;
        T.mask = cap - 1;
    } else {
        // capacity: power of two >= 2 x rows (every row could be its own group)
        cap = 1024;
        while (cap < 2ull * (uint64_t)total_rows) cap <<= 1;
        T.mask = cap - 1;
        T.owner = tmp.get<uint32_t>(cap);
        table_owner = true;
    }
    T.acc = tmp.get<uint64_t>(cap * n_acc);
    T.rows = tmp.get<uint64_t>(cap);
    if (nullable) T.nvalid = tmp.get<uint64_t>(cap * n_acc);
    if (T.n_fsum) {
        D.runs = tmp.get<uint32_t>(cap);
        D.spill_head = tmp.get<uint32_t>(cap);
    }
    {
        FillMany fm;                                 // everything the scan and the ordered sums expect cleared, one launch
        if (table_owner) fm.add(T.owner, cap * 4);       // (the run paths filled theirs)
        fm.add(T.rows, cap * 8);
        if (nullable) fm.add(T.nvalid, cap * n_acc * 8);
        if (T.n_fsum) {
            fm.add(D.runs, cap * 4);
            fm.add(D.spill_head, cap * 4, 0xFFFFFFFFu);
        }
        TIMED_LAUNCH(ex, "fill_many", launch_fill_many(cfg, fm));
    }
    if (P0.n_acc > 0) TIMED_LAUNCH(ex, "hash_agg_init", launch_hash_agg_init(cfg, T, kinds));

    uint32_t row_base = 0;
    for (auto& b : inputs) {
        ScanParams P = P0;
        ProgramBuilder::bind(P, pb.columns(), *b, nullable);
        TIMED_LAUNCH_N(ex, "scan_agg_hash", b->n_rows, launch_scan_agg_hash(cfg, P, T, row_base, status));
        row_base += (uint32_t)b->n_rows;
    }
    if (T.n_fsum) {
        const size_t n_tiles = ((size_t)total_rows + 1023) / 1024, stage_n = n_tiles * 1024;
        D.rowslot = T.rowslot;
        D.fvals = T.fvals;
        D.total_rows = T.total_rows;
        D.n_fsum = T.n_fsum;
        D.n_acc = T.n_acc;
        D.seg_slot = tmp.get<uint32_t>(stage_n);
        D.seg_first = tmp.get<uint32_t>(stage_n);
        D.seg_sum = tmp.get<double>(stage_n * T.n_fsum);
        D.tile_nseg = tmp.get<uint32_t>(n_tiles);
        D.acc = T.acc;
        D.rows = T.rows;
        D.spill_key = tmp.get<uint64_t>(stage_n);
        D.spill_seg = tmp.get<uint32_t>(stage_n);
        D.spill_count = reinterpret_cast<uint32_t*>(tail + 1);
        D.spill_next = tmp.get<uint32_t>(stage_n);
        // (D.runs, D.spill_head and the spill count — the second word of `tail` — were cleared above)
        TIMED_LAUNCH_N(ex, "det_segments", total_rows, launch_det_segments(cfg, D));
        TIMED_LAUNCH_N(ex, "det_apply", total_rows, launch_det_apply(cfg, D));
        // groups with several runs: combined through their lists right here; the entry count and the "lists too long" flag are
        // read together with the group count below (ONE host wait for the whole tail of the aggregate)
        TIMED_LAUNCH_N(ex, "det_spill_lists", total_rows, launch_det_spill_lists(cfg, D));
    }
    // used slots -> dense records (slot order: deterministic for a given input); distinct runs ARE the dense records
    uint64_t* dense = nullptr;
    if (!distinct_runs) {
        uint32_t* flags = tmp.get<uint32_t>(cap);
        dense = tmp.get<uint64_t>(cap + 1);
        void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes((int64_t)cap));
        TIMED_LAUNCH_N(ex, "hash_agg_flags", cap, launch_hash_agg_flags(cfg, T, flags));
        HIP_CHECK(exclusive_scan_u32_u64(ex.stream, flags, (int64_t)cap, dense, false, tail, scan_tmp));
    }
    struct Tail { uint64_t n_groups; uint32_t n_spill, lists_too_long; };
    const Tail tl = read_device(ex, reinterpret_cast<const Tail*>(tail));
    const uint64_t ng = distinct_runs == 1 ? n_runs_host : tl.n_groups;           // (two stretches: written by run_tail_remap)
    if (T.n_fsum && tl.lists_too_long) {
        // some group has many runs (unclustered input): the whole list ordered by (slot, first row), then added up left to right
        const uint32_t n_spill = tl.n_spill;
        BufferPtr kb = std::make_shared<Buffer>(ex.ctx, D.spill_key, (size_t)n_spill * 8), pb2 = std::make_shared<Buffer>(ex.ctx, D.spill_seg, (size_t)n_spill * 4);
        radix_sort_pairs(ex, kb, pb2, (int64_t)n_spill);
        TIMED_LAUNCH_N(ex, "det_spill_combine", n_spill, launch_det_spill_combine(cfg, D, kb->as<uint64_t>(), pb2->as<uint32_t>(), n_spill));
    }
    if (pb.can_raise()) check_scan_status(ex, status);   // (after the one wait above: immediate; fixed-width keys and no integer division raise nothing)
    *n_groups = (int64_t)ng;
    static const bool no_slot_emit = [] { const char* v = getenv("BHIP_NO_SLOT_EMIT"); return v && atoi(v) != 0; }();
    if (ng && distinct_runs && slots_out && !no_slot_emit) {
        // distinct runs: slot g IS group g — the caller emits its columns straight from the slot arrays (no GroupRec table in between)
        slots_out->keys128 = T.keys128;
        slots_out->head = run_head;
        slots_out->acc = T.acc;
        slots_out->nvalid = nullable ? T.nvalid : nullptr;
        slots_out->rows = T.rows;
        slots_out->n_acc = T.n_acc;
        slots_out->valid = 1;
        return nullptr;
    }
    GroupRec* table = tmp.get<GroupRec>(ng ? ng : 1);
    if (ng && distinct_runs) TIMED_LAUNCH_N(ex, "run_compact", ng, launch_run_compact(cfg, T, run_head, (uint32_t)ng, nullable, table));
    else if (ng) TIMED_LAUNCH_N(ex, "hash_agg_compact", cap, launch_hash_agg_compact(cfg, T, dense, nullable, table));
    *n_groups = (int64_t)ng;
    return table;
}

}  // namespace

std::vector<BatchPtr> HashAggregateExec::run_packed(int partition, const Exec& ex, const HashAggregateExec* as_final) const {
    const Schema& in_schema = *input_->schema();
    const SchemaPtr out_schema = as_final ? as_final->schema_ : schema_;
    // ---- expressions over the (fused) source ---------------------------------------------------
    std::vector<ExprPtr> group, args;
    for (auto& g : group_) group.push_back(g.first);
    if (mode_ == BHIP_AGG_PARTIAL) {
        for (auto& a : aggr_) args.push_back(a.arg);
    } else {
        size_t pos = group_.size();
        for (auto& a : aggr_) {
            args.push_back(make_column(in_schema.fields[pos].name));
            if (a.fn == BHIP_AGG_AVG) args.push_back(make_column(in_schema.fields[pos + 1].name));
            pos += a.fn == BHIP_AGG_AVG ? 2 : 1;
        }
    }
    FusedInput f = fuse_below(input_, group, args);
    const Schema& src_schema = *f.source->schema();

    ProgramBuilder pb(src_schema);
    if (f.predicate) pb.set_predicate(f.predicate);
    for (auto& g : f.group) pb.add_key(g);

    std::vector<SopAccExpr> acc_exprs;  // accumulator index -> (kind, input expression)
    auto add_acc = [&](int kind, const ExprPtr& e) {
        const int idx = pb.add_acc(kind, pb.compile(e));
        if (idx == (int)acc_exprs.size()) acc_exprs.push_back(SopAccExpr{kind, e});
        return idx;
    };
    std::vector<EmitValueSpec> emits;   // one per output state/value column
    auto emit = [&](int kind, int a, int b, int dtype) { emits.push_back(EmitValueSpec{kind, a, b, 0, dtype}); };
    size_t ai = 0;
    for (auto& a : aggr_) {
        if (mode_ == BHIP_AGG_PARTIAL) {
            const ExprPtr& arg = f.args[ai++];
            const int t = expr_type(arg, src_schema);
            const bool lit_nonnull = arg->kind == BHIP_EXPR_LITERAL && !arg->is_null;
            switch (a.fn) {
                case BHIP_AGG_SUM: {
                    // (SUM(Float32) adds doubles and rounds the total to float once: closer to the exact sum than the reference's running float)
                    const int acc = add_acc(dt_is_float(t) ? ACC_SUM_F64 : ACC_SUM_I64, arg);
                    emit(EMIT_VALUE, acc, 0, sum_type(t));
                } break;
                case BHIP_AGG_AVG: {
                    ExprPtr farg = arg;
                    if (t != DT_FLOAT64) {
                        auto c = std::make_shared<Expr>();
                        c->kind = BHIP_EXPR_CAST;
                        c->dtype = DT_FLOAT64;
                        c->args = {arg};
                        farg = c;
                    }
                    const int acc = add_acc(ACC_SUM_F64, farg);
                    if (as_final) {
                        // [count], [sum] of the one partition -> sum / count, NULL without a non-NULL input: what the Final makes of them
                        emit(EMIT_AVG, acc, 0, DT_FLOAT64);
                        break;
                    }
                    emit(EMIT_COUNT, acc, 0, DT_UINT64);
                    emit(EMIT_VALUE, acc, 0, DT_FLOAT64);
                } break;
                case BHIP_AGG_COUNT: {
                    if (lit_nonnull || !expr_nullable(arg, src_schema)) { emit(EMIT_ROWS, 0, 0, DT_UINT64); break; }
                    ExprPtr carg = arg;
                    if (t == DT_UTF8) {
                        // COUNT(s) counts the non-NULL strings: count CASE WHEN s IS NOT NULL THEN 1 END instead,
                        // an Int64 value with the same validity (the VM keeps no per-row Utf8 values)
                        auto nn = std::make_shared<Expr>();
                        nn->kind = BHIP_EXPR_IS_NOT_NULL;
                        nn->args = {arg};
                        auto one = std::make_shared<Expr>();
                        one->kind = BHIP_EXPR_LITERAL;
                        one->dtype = DT_INT64;
                        one->i64 = 1;
                        auto cs = std::make_shared<Expr>();
                        cs->kind = BHIP_EXPR_CASE;
                        cs->args = {ExprPtr(nn), ExprPtr(one)};
                        carg = cs;
                    }
                    Operand x = pb.compile(carg);
                    if (x.is_utf8_col) fail(BHIP_ENOTIMPL, "COUNT over a Utf8-valued expression");
                    const int acc = add_acc(x.vclass == VC_BOOL ? ACC_COUNT_VALID_B : ACC_COUNT_VALID, carg);
                    emit(EMIT_RAW, acc, 0, DT_UINT64);
                } break;
                default: {
                    if (t == DT_UTF8) fail(BHIP_ENOTIMPL, "MIN/MAX over Utf8 on the packed-key path (run_strings handles it)");
                    if (t == DT_BOOLEAN) fail(BHIP_ENOTIMPL, "MIN/MAX over Boolean");
                    const bool is_min = a.fn == BHIP_AGG_MIN;
                    const int kind = dt_is_float(t) ? (is_min ? ACC_MIN_F64 : ACC_MAX_F64) : (is_min ? ACC_MIN_I64 : ACC_MAX_I64);
                    emit(EMIT_VALUE, add_acc(kind, arg), 0, t);
                } break;
            }
        } else {
            const ExprPtr& st0 = f.args[ai++];
            const int t = expr_type(st0, src_schema);
            switch (a.fn) {
                case BHIP_AGG_SUM: {
                    const int acc = add_acc(dt_is_float(t) ? ACC_SUM_F64 : ACC_SUM_I64, st0);
                    emit(EMIT_VALUE, acc, 0, t);
                } break;
                case BHIP_AGG_AVG: {
                    const ExprPtr& st1 = f.args[ai++];
                    const int acc_c = add_acc(ACC_SUM_I64, st0);
                    const int acc_s = add_acc(ACC_SUM_F64, st1);
                    emit(EMIT_AVG_ACC, acc_s, acc_c, DT_FLOAT64);
                } break;
                case BHIP_AGG_COUNT: emit(EMIT_RAW, add_acc(ACC_SUM_I64, st0), 0, DT_UINT64); break;
                default: {
                    const bool is_min = a.fn == BHIP_AGG_MIN;
                    const int kind = dt_is_float(t) ? (is_min ? ACC_MIN_F64 : ACC_MAX_F64) : (is_min ? ACC_MIN_I64 : ACC_MAX_I64);
                    emit(EMIT_VALUE, add_acc(kind, st0), 0, t);
                } break;
            }
        }
    }
    ScanParams P0;
    pb.finish(P0);
    const int n_acc = P0.n_acc;
    // register-resident fast path when the plan has the chain-of-products shape (kernels_sop.hip)
    SopPlan sop;
    static const bool sop_disabled = [] { const char* v = getenv("BHIP_NO_SOP"); return v && atoi(v) != 0; }();
    bool use_sop = !sop_disabled && !pb.creates_nulls() && (int)acc_exprs.size() == n_acc && n_acc <= SOP_NSTEP &&
                   build_sop(src_schema, f.predicate, f.group, acc_exprs, sop);

    // ---- input ------------------------------------------------------------------------------------
    std::vector<BatchPtr> inputs;
    {
        StreamPtr s;
        if (auto hj = dynamic_cast<const HashJoinExec*>(f.source.get())) {
            // aggregate over a join: the join gathers only the columns the aggregate's program reads
            std::vector<bool> needed(src_schema.fields.size(), false);
            for (int ci : pb.columns()) needed[ci] = true;
            s = hj->execute_needed(partition, ex, needed);
        } else {
            s = f.source->execute(partition, ex);
        }
        while (BatchPtr b = s->next())
            if (b->n_rows > 0) inputs.push_back(b);
    }
    trace_point("aggregate: inputs ready");
    bool nullable = pb.creates_nulls();
    for (auto& b : inputs)
        for (int ci : pb.columns())
            if (b->cols[ci].validity) nullable = true;

    // the fast path packs keys its own way, so it serves either every batch of the run or none.
    // Its wide-load variant (lean_kernel.h) also takes NULLs in the columns the predicate constrains.
    static const bool lean_disabled = [] { const char* v = getenv("BHIP_NO_LEAN"); return v && atoi(v) != 0; }();
    bool use_lean = use_sop && !lean_disabled && lean_eligible(sop.prog);
    for (auto& b : inputs) {
        if (use_lean && !(sop_columns_bindable(sop, *b, true) && lean_bindable(sop, *b))) use_lean = false;
        if (use_sop && !sop_columns_bindable(sop, *b)) use_sop = false;
    }
    // A Utf8 key longer than the fast paths hold (3 bytes wide-load, 7 bytes register kernel) sends them back empty-handed after a
    // whole pass over the input each.  Short codes (Q1's flags: 1 byte on average) are taken on trust; a key column that averages
    // more than 2 bytes per value has the longest string of its leading 64 Ki rows measured first (one small launch).
    if ((use_lean || use_sop) && !inputs.empty() && inputs[0]->n_rows >= (1 << 17)) {
        const Batch& b0 = *inputs[0];
        uint32_t longest = 0;
        for (auto& g : f.group) {
            if (g->kind != BHIP_EXPR_COLUMN) continue;
            const int ci = src_schema.index_of(g->name);
            if (ci < 0 || b0.cols[ci].dtype != DT_UTF8 || b0.cols[ci].data_bytes <= 2 * b0.n_rows) continue;
            Temp t2(ex);
            uint32_t* dev = t2.get<uint32_t>(1);
            TIMED_LAUNCH(ex, "utf8_max_len", launch_utf8_max_len(ex.cfg(), b0.cols[ci].offsets->as<int32_t>(), std::min<int64_t>(b0.n_rows, 65536), dev));
            longest = std::max(longest, read_device(ex, dev));
        }
        if (longest > 3) use_lean = false;
        if (longest > 7) use_sop = false;
    }
    bool sop_layout = false;             // the pass that produced `table` packed keys the fast path's way

    Temp tmp(ex);
    const LaunchCfg cfg = ex.cfg();
    // status of the scan + merge and the byte totals of the Utf8 key columns, side by side: ONE read brings both back
    TailInfo* info = tmp.get<TailInfo>(1);
    ScanStatus* status = &info->st;
    uint64_t* totals = group_.size() <= (size_t)TAIL_TOTALS ? info->totals : tmp.get<uint64_t>(group_.size() + 1);
    GroupRec* table = nullptr;
    int64_t n_groups = 0;
    TimedLaunches timer(ex);

    // ---- group table -> output batch ------------------------------------------------------------
    // `n_alloc` rows are allocated; dev_n != nullptr: the kernels read the count themselves (<= n_alloc), the host learns it later.
    // Utf8 key columns: the value bytes have the packed key's bound (width - 1 per group), so the bytes are written before their
    // total is known; the totals come back in one read after everything is queued.
    std::vector<size_t> utf8_cols;
    auto emit_table = [&](const GroupRec* tab, int64_t n_alloc, const ScanStatus* dev_n) {
        auto out = std::make_shared<Batch>();
        out->schema = out_schema;
        out->ctx = ex.ctx;
        out->n_rows = n_alloc;
        const auto& kinfo = sop_layout ? sop.key_info : pb.key_info();
        utf8_cols.clear();
        if (dev_n && n_alloc <= EMIT_ALL_MAX_GROUPS && group_.size() <= (size_t)EMIT_ALL_MAX_KEYS && emits.size() <= (size_t)EMIT_ALL_MAX_VALUES) {
            // the small table of the register path: every column in ONE launch, count read on the device
            EmitAllArgs A;
            memset(&A, 0, sizeof(A));
            A.table = tab;
            A.status = dev_n;
            A.n_keys = (int32_t)group_.size();
            A.n_values = (int32_t)emits.size();
            for (size_t gi = 0; gi < group_.size(); ++gi) {
                Column c;
                c.dtype = out_schema->fields[gi].dtype;
                c.length = n_alloc;
                A.key[gi] = EmitKeySpec{kinfo[gi].pos, kinfo[gi].width, kinfo[gi].nullable, c.dtype};
                if (kinfo[gi].nullable) { c.validity = make_buffer(ex, bitmap_bytes(n_alloc) + 8); A.key_validity[gi] = c.validity->as<uint64_t>(); }
                if (c.dtype == DT_UTF8) {
                    c.offsets = make_buffer(ex, (size_t)(n_alloc + 1) * 4);
                    c.data = make_buffer(ex, (size_t)n_alloc * (size_t)kinfo[gi].width + 8);
                    A.key_offsets[gi] = c.offsets->as<int32_t>();
                    A.key_total[gi] = totals + gi;
                    utf8_cols.push_back(gi);
                } else {
                    c.data = make_buffer(ex, (c.dtype == DT_BOOLEAN ? bitmap_bytes(n_alloc) : (size_t)n_alloc * dtype_width(c.dtype)) + 8);
                }
                A.key_data[gi] = c.data->ptr();
                out->cols.push_back(std::move(c));
            }
            for (size_t k = 0; k < emits.size(); ++k) {
                const Field& fld = out_schema->fields[group_.size() + k];
                Column c;
                c.dtype = fld.dtype;
                c.length = n_alloc;
                c.data = make_buffer(ex, (size_t)n_alloc * dtype_width(c.dtype) + 8);
                if (fld.nullable) { c.validity = make_buffer(ex, bitmap_bytes(n_alloc) + 8); A.value_validity[k] = c.validity->as<uint64_t>(); }
                A.value[k] = emits[k];
                A.value[k].count_is_rows = nullable ? 0 : 1;
                A.value_data[k] = c.data->ptr();
                out->cols.push_back(std::move(c));
            }
            TIMED_LAUNCH(ex, "emit_all", launch_emit_all(cfg, A));
            return out;
        }
        for (size_t gi = 0; gi < group_.size(); ++gi) {
            Column c;
            c.dtype = out_schema->fields[gi].dtype;
            c.length = n_alloc;
            EmitKeySpec ks{kinfo[gi].pos, kinfo[gi].width, kinfo[gi].nullable, c.dtype};
            if (kinfo[gi].nullable) c.validity = make_buffer(ex, bitmap_bytes(n_alloc) + 8);
            uint64_t* vptr = c.validity ? c.validity->as<uint64_t>() : nullptr;
            if (c.dtype == DT_UTF8) {
                c.offsets = make_buffer(ex, (size_t)(n_alloc + 1) * 4);
                c.data = make_buffer(ex, (size_t)n_alloc * (size_t)kinfo[gi].width + 8);
                if (n_alloc > 0 && n_alloc <= EMIT_UTF8_SMALL_MAX) {
                    TIMED_LAUNCH(ex, "emit_group_utf8_small", launch_emit_group_utf8_small(cfg, tab, n_alloc, ks, vptr, c.offsets->as<int32_t>(),
                                                           c.data->as<uint8_t>(), totals + gi, dev_n));
                } else {
                    uint32_t* lengths = tmp.get<uint32_t>((size_t)n_alloc + 1);
                    void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_alloc));
                    if (n_alloc) TIMED_LAUNCH(ex, "emit_group_key", launch_emit_group_key(cfg, tab, n_alloc, ks, nullptr, vptr, lengths));
                    HIP_CHECK(exclusive_scan_u32_i32(ex.stream, lengths, n_alloc, c.offsets->as<int32_t>(), true, totals + gi, scan_tmp));
                    if (n_alloc) TIMED_LAUNCH(ex, "emit_group_utf8", launch_emit_group_utf8(cfg, tab, n_alloc, ks, c.offsets->as<int32_t>(), c.data->as<uint8_t>()));
                }
                utf8_cols.push_back(gi);
            } else {
                const size_t bytes = c.dtype == DT_BOOLEAN ? bitmap_bytes(n_alloc) : (size_t)n_alloc * dtype_width(c.dtype);
                c.data = make_buffer(ex, bytes + 8);
                if (n_alloc) TIMED_LAUNCH(ex, "emit_group_key", launch_emit_group_key(cfg, tab, n_alloc, ks, c.data->ptr(), vptr, nullptr, dev_n));
            }
            out->cols.push_back(std::move(c));
        }
        // value columns: one launch per EMIT_BATCH_MAX columns
        EmitValueBatch vb;
        vb.n = 0;
        for (size_t k = 0; k < emits.size(); ++k) {
            EmitValueSpec sp = emits[k];
            sp.count_is_rows = nullable ? 0 : 1;
            const Field& fld = out_schema->fields[group_.size() + k];
            Column c;
            c.dtype = fld.dtype;
            c.length = n_alloc;
            c.data = make_buffer(ex, (size_t)n_alloc * dtype_width(c.dtype) + 8);
            if (fld.nullable) c.validity = make_buffer(ex, bitmap_bytes(n_alloc) + 8);
            vb.spec[vb.n] = sp;
            vb.data[vb.n] = c.data->ptr();
            vb.validity[vb.n] = c.validity ? c.validity->as<uint64_t>() : nullptr;
            if (++vb.n == EMIT_BATCH_MAX || k + 1 == emits.size()) {
                TIMED_LAUNCH(ex, "emit_group_values", launch_emit_group_values(cfg, tab, n_alloc, vb, dev_n));
                vb.n = 0;
            }
            out->cols.push_back(std::move(c));
        }
        return out;
    };
    // ... and from the run slots of a clustered hash aggregate (kernels_util.hip: emit_slots_kernel), every column in one launch
    SlotSource slots;
    memset(&slots, 0, sizeof(slots));
    auto emit_slots = [&](int64_t n) {
        auto out = std::make_shared<Batch>();
        out->schema = out_schema;
        out->ctx = ex.ctx;
        out->n_rows = n;
        const auto& kinfo = pb.key_info();               // (the hash path packs keys with the VM's layout)
        utf8_cols.clear();
        EmitAllArgs A;
        memset(&A, 0, sizeof(A));
        A.n_keys = (int32_t)group_.size();
        A.n_values = (int32_t)emits.size();
        for (size_t gi = 0; gi < group_.size(); ++gi) {
            Column c;
            c.dtype = out_schema->fields[gi].dtype;
            c.length = n;
            A.key[gi] = EmitKeySpec{kinfo[gi].pos, kinfo[gi].width, kinfo[gi].nullable, c.dtype};
            if (kinfo[gi].nullable) { c.validity = make_buffer(ex, bitmap_bytes(n) + 8); A.key_validity[gi] = c.validity->as<uint64_t>(); }
            c.data = make_buffer(ex, (size_t)n * dtype_width(c.dtype) + 8);
            A.key_data[gi] = c.data->ptr();
            out->cols.push_back(std::move(c));
        }
        for (size_t k = 0; k < emits.size(); ++k) {
            const Field& fld = out_schema->fields[group_.size() + k];
            Column c;
            c.dtype = fld.dtype;
            c.length = n;
            c.data = make_buffer(ex, (size_t)n * dtype_width(c.dtype) + 8);
            if (fld.nullable) { c.validity = make_buffer(ex, bitmap_bytes(n) + 8); A.value_validity[k] = c.validity->as<uint64_t>(); }
            A.value[k] = emits[k];
            A.value[k].count_is_rows = nullable ? 0 : 1;
            A.value_data[k] = c.data->ptr();
            out->cols.push_back(std::move(c));
        }
        TIMED_LAUNCH_N(ex, "emit_slots", n, launch_emit_slots(cfg, slots, n, A));
        return out;
    };
    // the row count (and the Utf8 byte totals) once the host knows them
    auto finish_table = [&](const std::shared_ptr<Batch>& out, int64_t n, const uint64_t* host_totals) {
        out->n_rows = n;
        for (auto& c : out->cols) c.length = n;
        for (size_t gi : utf8_cols) out->cols[gi].data_bytes = (int64_t)host_totals[gi];
    };

    int gmax = group_.empty() ? 1 : 4;
    const int hint = path_hint_.load();
    if (hint == 8 || hint == -1) gmax = hint;
    int64_t rows_in = 0;
    for (auto& b : inputs) rows_in += b->n_rows;
    // more than 8 accumulators: the hash path — except over a tiny input (the Final aggregate over a few partial-state rows
    // per rank), where the 16-accumulator variant of the register kernel is ONE launch instead of a dozen
    const bool wide_acc = n_acc > AGG_NACC;
    if (wide_acc && (rows_in > 65536 || hint == -1)) gmax = -1;

    // A plan that has not run yet does not know how many groups there are.  With a large input the ladder below (4 groups per
    // workgroup -> 8 -> hash table) is first walked on the leading 32 Ki rows only: an aggregate with many groups (Q3: one per
    // order) finds out for the price of two tiny launches instead of two passes over the whole input.
    std::vector<BatchPtr> sample;
    int64_t total_in = 0;
    for (auto& b : inputs) total_in += b->n_rows;
    // (not for the wide-load path: its plans have at most two 32-bit key parts — flags, short codes — and its launches are the
    // ones the bench's roofline line and the rocprofv3 averages are about)
    // (a mid-sized input — up to 1 Mi rows: one of these launches is ~0.05 ms whatever it reads — is not sampled; with at most four
    // accumulators it starts at 8 groups, whose kernel then holds as many accumulator registers as the 4-group one with eight)
    const bool mid_sized = total_in < (1 << 20);
    if (hint == 0 && gmax == 4 && mid_sized && !wide_acc && n_acc <= 4 && !use_lean) gmax = 8;
    if (hint == 0 && gmax > 0 && !group_.empty() && !use_lean && (!mid_sized || (gmax == 4 && total_in >= (1 << 17)))) {
        auto head = std::make_shared<Batch>(*inputs[0]);
        head->n_rows = std::min<int64_t>(head->n_rows, 32768);
        sample.push_back(head);
    }
    // the head is read at the wider of the two register widths straight away: its group count then picks the width for the whole
    // input (<= 4: the 4-group kernel, which prefetches; <= 8: this one; more: the hash table) — one small launch, not two
    const bool probe8 = !sample.empty() && gmax == 4 && !wide_acc;
    if (probe8) gmax = 8;
    std::shared_ptr<Batch> early;
    TailInfo tail;
    memset(&tail, 0, sizeof(tail));
    static const bool no_early_emit = [] { const char* v = getenv("BHIP_NO_EARLY_EMIT"); return v && atoi(v) != 0; }();
    while (!inputs.empty()) {
        const bool sampling = !sample.empty();
        const std::vector<BatchPtr>& cur = sampling ? sample : inputs;
        if (gmax == -1) {
            // ---- hash path: one device-wide table, atomics ----------------------------------------
            early.reset();
            sop_layout = false;              // the hash path packs keys with the VM's layout
            // (fixed-width, non-Boolean keys: a clustered input's groups can be emitted straight from their run slots)
            bool slot_keys = group_.size() <= (size_t)EMIT_ALL_MAX_KEYS && emits.size() <= (size_t)EMIT_ALL_MAX_VALUES;
            for (size_t gi = 0; gi < group_.size(); ++gi)
                slot_keys = slot_keys && out_schema->fields[gi].dtype != DT_UTF8 && out_schema->fields[gi].dtype != DT_BOOLEAN;
            table = hash_aggregate(ex, tmp, P0, pb, inputs, nullable, &n_groups, status, timer, &clustered_hint_, slot_keys ? &slots : nullptr);
            break;
        }
        // ---- register path ----------------------------------------------------------------------
        const int max_grid = scan_agg_lowcard_max_grid(cfg);
        const size_t max_parts = cur.size() * (size_t)max_grid;
        GroupRec* partials = tmp.get<GroupRec>(max_parts * gmax);
        uint32_t* partial_ng = tmp.get<uint32_t>(max_parts);
        HIP_CHECK(hipMemsetAsync(status, 0, sizeof(ScanStatus), ex.stream));
        int n_part = 0;
        const bool lean_now = use_lean && (gmax == 1 || gmax == 4);
        const bool sop_now = !lean_now && use_sop;
        sop_layout = lean_now || sop_now;
        // several SMALL input batches on the VM kernel (the partial states of N ranks under a Final aggregate: one 4-row batch per
        // rank): one launch for all of them — the kernel's fixed cost is ~0.1 ms per launch with 16 accumulators, which a rank of an
        // 8-GPU Q1 would pay eight times per step
        int64_t cur_rows = 0;
        for (auto& b : cur) cur_rows += b->n_rows;
        const bool together = !lean_now && !sop_now && cur.size() > 1 && cur.size() <= 4096 && cur_rows <= (1 << 20);
        std::vector<ScanParams> Ps;                    // (lives until this round's host wait below: the copy to the device may read it late)
        if (together) {
            Ps.assign(cur.size(), P0);
            for (size_t i = 0; i < cur.size(); ++i) ProgramBuilder::bind(Ps[i], pb.columns(), *cur[i], nullable);
            int grid = 0;
            HIP_CHECK(launch_scan_agg_lowcard(cfg, Ps[0], tmp.get<ScanParams>(cur.size()), gmax, partials, partial_ng, max_grid, status, &grid, (int)cur.size()));
            n_part = grid;
        }
        for (auto& b : cur) {
            if (together) break;
            ScanParams P = P0;
            ProgramBuilder::bind(P, pb.columns(), *b, nullable);
            int grid = 0;
            const bool timed = b->n_rows >= (1 << 16);   // the bench hook times the dominant (large) launches only
            if (timed) { timer.begin(); timer.kernel = lean_now ? "scan_agg_lean_kernel" : sop_now ? "scan_agg_sop_kernel" : "scan_agg_lowcard_kernel"; }
            if (lean_now) {
                bind_sop(sop, *b);
                HIP_CHECK(launch_scan_agg_lean(cfg, sop.prog, tmp.get<SopProgram>(1), gmax, partials + (size_t)n_part * gmax,
                                               partial_ng + n_part, max_grid, status, &grid));
            } else if (sop_now) {
                bind_sop(sop, *b);
                HIP_CHECK(launch_scan_agg_sop(cfg, sop.prog, tmp.get<SopProgram>(1), gmax, partials + (size_t)n_part * gmax,
                                              partial_ng + n_part, max_grid, status, &grid));
            } else
                HIP_CHECK(launch_scan_agg_lowcard(cfg, P, tmp.get<ScanParams>(1), gmax, partials + (size_t)n_part * gmax,
                                                  partial_ng + n_part, max_grid, status, &grid));
            if (timed) timer.end();
            n_part += grid;
        }
        const int cap = 1024;
        table = tmp.get<GroupRec>(cap);
        uint32_t* entry_group = tmp.get<uint32_t>((size_t)n_part * gmax);
        AccSpec specs[VM_MAX_ACC];
        for (int i = 0; i < n_acc; ++i) specs[i] = P0.acc[i];
        TIMED_LAUNCH(ex, "merge_partials", launch_merge_partials(cfg, partials, partial_ng, n_part, gmax, specs, n_acc, table, cap, entry_group, status));
        // the table has at most `cap` groups: its columns are emitted for that bound straight away, behind the merge and with the
        // count read on the device, so the host waits ONCE per aggregate (the result is dropped if the ladder has to go on)
        early.reset();
        if (!sampling && !no_early_emit && group_.size() <= (size_t)TAIL_TOTALS) early = emit_table(table, cap, status);
        trace_point("aggregate: scan + merge + emit queued");
        tail = read_device(ex, info);
        const ScanStatus st = tail.st;
        timer.collect();
        if (lean_now && (st.flags & SCAN_ERR_KEY_TOO_LONG)) {
            use_lean = false;                // a string key longer than 3 bytes: the 7-byte variant next
            continue;
        }
        if (sop_now && (st.flags & SCAN_ERR_KEY_TOO_LONG)) {
            use_sop = false;                 // a string key longer than the fast path's 7 bytes: the VM packs up to 15
            continue;
        }
        check_scan_flags(st);
        if (st.flags & SCAN_OVERFLOW_GROUPS) {
            gmax = (gmax == 4 && !wide_acc) ? 8 : -1;       // more groups than the register path holds: widen, then hash
            path_hint_.store(gmax);
            if (gmax == -1) sample.clear();
            continue;
        }
        if (sampling) {                                   // the head fits this width: now the whole input
            if (probe8 && st.n_groups <= 4) gmax = 4;
            sample.clear();
            continue;
        }
        n_groups = st.n_groups;
        break;
    }

    if (n_groups == 0 && group_.empty()) {
        early.reset();
        // no GROUP BY: exactly one output row even for empty input (SUM = NULL, COUNT = 0)
        GroupRec id;
        memset(&id, 0, sizeof(id));
        for (int i = 0; i < n_acc; ++i) {
            switch (P0.acc[i].kind) {
                case ACC_MIN_F64: { double v = __builtin_huge_val(); memcpy(&id.acc[i], &v, 8); } break;
                case ACC_MAX_F64: { double v = -__builtin_huge_val(); memcpy(&id.acc[i], &v, 8); } break;
                case ACC_MIN_I64: id.acc[i] = (uint64_t)INT64_MAX; break;
                case ACC_MAX_I64: id.acc[i] = (uint64_t)INT64_MIN; break;
                default: break;
            }
        }
        table = tmp.get<GroupRec>(1);
        HIP_CHECK(hipMemcpyAsync(table, &id, sizeof(id), hipMemcpyHostToDevice, ex.stream));
        stream_wait(ex);
        n_groups = 1;
    }

    trace_point("aggregate: done");
    if (early) {
        // the emit was queued behind the merge; the single read above brought the count and the Utf8 totals
        finish_table(early, n_groups, tail.totals);
        return {early};
    }
    auto out = slots.valid ? emit_slots(n_groups) : emit_table(table, n_groups, nullptr);
    if (!utf8_cols.empty()) {
        std::vector<uint64_t> host(group_.size() + 1);
        if (group_.size() <= (size_t)TAIL_TOTALS) {
            const TailInfo ti = read_device(ex, info);                      // the totals sit in the info block: one pinned-slot read
            for (size_t i = 0; i < group_.size(); ++i) host[i] = ti.totals[i];
        } else {
            HIP_CHECK(hipMemcpyAsync(host.data(), totals, group_.size() * 8, hipMemcpyDeviceToHost, ex.stream));
            stream_wait(ex);
        }
        finish_table(out, n_groups, host.data());
    }
    // (no wait otherwise: everything downstream is queued on the same stream, and scratch is released in stream order)
    return {out};
}

}  // namespace bhip
