// ops_join.cpp — HashJoinExec (Inner / Left / Right), left = build side.
//
// Reference: HashJoinExec::try_new(left, right, on: &[(String, String)], join_type) built at
// rust/core/src/serde/physical_plan/from_proto.rs:253-276 (join types :268-272, key pairs by
// column NAME :256-260); the 4-argument constructor of that DataFusion revision is the
// collect-left mode: every task drains the whole left child and probes it with one right
// partition.  Output schema = left fields then right fields, a right key column
// dropped when it has the same name as its left partner (Appendix A).  Row order unspecified.
#include <mutex>

#include "../util_kernels.h"
#include "hash_kernels.h"
#include "plan.hpp"
#include "sop.hpp"
#include "../sort_kernels.h"

namespace bhip {

struct JoinBuildSide {
    BatchPtr batch;                 // all left rows
    BufferPtr keys, sel, owner, head, next, dup;
    JoinTable table;
    bool has_sel = false;
    bool unique = false;            // no two build rows share a key: probe rows have at most one partner
    bool narrow = false;            // ONE integer key, unique: NarrowJoinTable instead of JoinTable
    int narrow_width = 0;           // its key bytes (4: Int32 / Date32, 8: Int64 / UInt64)
    BufferPtr slots, present, rpack, rbits, rperm;
    NarrowJoinTable ntable;
    Column key_holder;              // two-column join: the key column built for it (packed pair, or the first key with both validities)
    bool resid = false;             // two-column join by the first key; the second is compared on every match (ntable.resid_build)
    // BHIP_JOIN_RADIX=1: the build side in partition order for the LDS join (kernels_radix_join.hip), the A/B partner
    BufferPtr rj_keys, rj_rows, rj_first;
    int rj_log2p = -1;
};

namespace {
void radix_partition_side(const Exec& ex, const uint32_t* keys, int64_t n, int log2p, BufferPtr& skeys, BufferPtr& srows, BufferPtr& first);
}

static const char* join_name(int t) { return t == BHIP_JOIN_INNER ? "Inner" : (t == BHIP_JOIN_LEFT ? "Left" : "Right"); }

HashJoinExec::HashJoinExec(PlanPtr left, PlanPtr right, std::vector<std::pair<std::string, std::string>> on, int join_type)
    : left_(std::move(left)), right_(std::move(right)), on_(std::move(on)), join_type_(join_type) {
    ctx_ = left_->context();
    if (join_type < BHIP_JOIN_INNER || join_type > BHIP_JOIN_RIGHT) fail(BHIP_ENOTIMPL, "Unsupported join type");
    const Schema& ls = *left_->schema();
    const Schema& rs = *right_->schema();
    for (auto& p : on_) {
        const int li = ls.index_of(p.first), ri = rs.index_of(p.second);
        if (li < 0) fail(BHIP_EINVAL, "The left side of the join does not have column '" + p.first + "'");
        if (ri < 0) fail(BHIP_EINVAL, "The right side of the join does not have column '" + p.second + "'");
        if (ls.fields[li].dtype != rs.fields[ri].dtype)
            fail(BHIP_EINVAL, "join keys " + p.first + " / " + p.second + " have different types (" +
                                  dtype_name(ls.fields[li].dtype) + " vs " + dtype_name(rs.fields[ri].dtype) + ")");
    }
    auto s = std::make_shared<Schema>();
    for (auto f : ls.fields) {
        if (join_type == BHIP_JOIN_RIGHT) f.nullable = true;
        s->fields.push_back(f);
    }
    for (size_t i = 0; i < rs.fields.size(); ++i) {
        bool drop = false;
        for (auto& p : on_)
            if (p.second == rs.fields[i].name && p.first == p.second) drop = true;
        if (drop) continue;
        Field f = rs.fields[i];
        if (join_type == BHIP_JOIN_LEFT) f.nullable = true;
        if (s->index_of(f.name) >= 0) fail(BHIP_EINVAL, "join output would have two columns named '" + f.name + "'");
        s->fields.push_back(f);
        right_cols_.push_back((int)i);
    }
    schema_ = s;
    // key layout must be valid (surfaces BHIP_ENOTIMPL at plan time)
    ProgramBuilder pb(ls);
    for (auto& p : on_) pb.add_key(make_column(p.first), true);
    ScanParams P;
    pb.finish(P);
    cache_ = std::make_shared<BuildCache>();
}

PlanPtr HashJoinExec::with_new_children(const std::vector<PlanPtr>& c) const {
    if (c.size() != 2) fail(BHIP_EINVAL, "HashJoinExec wrong number of children");
    return std::make_shared<HashJoinExec>(c[0], c[1], on_, join_type_);
}

std::string HashJoinExec::describe() const {
    std::string s = std::string("HashJoinExec: mode=CollectLeft, join_type=") + join_name(join_type_) + ", on=[";
    for (size_t i = 0; i < on_.size(); ++i) s += (i ? ", " : "") + std::string("(") + on_[i].first + ", " + on_[i].second + ")";
    return s + "]";
}

// packed keys (+ "no NULL key" selection) of one side
static void side_keys(const Exec& ex, const Batch& b, const std::vector<std::string>& cols, BufferPtr& keys, BufferPtr& sel,
                      bool& has_sel) {
    // one NULL-free integer key column (every TPC-H join): the image is a widening copy, no expression program
    if (cols.size() == 1) {
        const int ci = b.schema->index_of(cols[0]);
        const Column& c = b.cols[ci];
        const int w = (c.dtype == DT_INT32 || c.dtype == DT_DATE32) ? 4 : (c.dtype == DT_INT64 || c.dtype == DT_UINT64) ? 8 : 0;
        if (w && !c.validity) {
            keys = make_buffer(ex, (size_t)b.n_rows * 16 + 16);
            has_sel = false;
            TIMED_LAUNCH_N(ex, "widen_key", b.n_rows, launch_widen_key(ex.cfg(), c.data->ptr(), w, b.n_rows, keys->as<uint64_t>()));
            return;
        }
    }
    ProgramBuilder pb(*b.schema);
    ExprPtr pred;
    for (auto& c : cols) {
        const int ci = b.schema->index_of(c);
        if (b.schema->fields[ci].nullable || b.cols[ci].validity) {
            auto e = std::make_shared<Expr>();
            e->kind = BHIP_EXPR_IS_NOT_NULL;
            e->args = {make_column(c)};
            pred = pred ? make_binary(pred, "And", e) : ExprPtr(e);
        }
    }
    if (pred) pb.set_predicate(pred);
    for (auto& c : cols) pb.add_key(make_column(c), true);
    ScanParams P;
    pb.finish(P);
    ProgramBuilder::bind(P, pb.columns(), b, pb.creates_nulls());
    keys = make_buffer(ex, (size_t)b.n_rows * 16 + 16);
    has_sel = (bool)pred;
    if (has_sel) sel = make_buffer(ex, bitmap_bytes(b.n_rows) + 8);
    if (b.n_rows == 0) return;
    Temp tmp(ex);
    ScanStatus* st = tmp.get<ScanStatus>(1);
    HIP_CHECK(hipMemsetAsync(st, 0, sizeof(ScanStatus), ex.stream));
    TIMED_LAUNCH_N(ex, "scan_keys", b.n_rows, launch_scan_keys(ex.cfg(), P, keys->as<uint64_t>(), nullptr, has_sel ? sel->as<uint64_t>() : nullptr, st));
    check_scan_status(ex, st);
}

// ONE key pair of integer columns of the same width on both sides: 4 (Int32 / Date32), 8 (Int64 / UInt64), else 0
bool HashJoinExec::pair_keys() const {
    if (on_.size() != 2) return false;
    const Schema &ls = *left_->schema(), &rs = *right_->schema();
    auto four = [](int t) { return t == DT_INT32 || t == DT_DATE32 || t == DT_UINT32; };
    for (auto& p : on_) {
        const int lt = ls.fields[ls.index_of(p.first)].dtype, rt = rs.fields[rs.index_of(p.second)].dtype;
        if (!four(lt) || lt != rt) return false;
    }
    return true;
}

// the key column the single-key path works on: the column itself, or the two 4-byte key columns packed into one Int64 column
static Column pack_key_pair(const Exec& ex, const Column& a, const Column& b, int64_t n) {
    Column k;
    k.dtype = DT_INT64;
    k.length = n;
    k.data = make_buffer(ex, (size_t)n * 8 + 8);
    if (a.validity || b.validity) k.validity = make_buffer(ex, bitmap_bytes(n) + 8);
    TIMED_LAUNCH_N(ex, "pack_key_pair", n, launch_pack_key_pair(ex.cfg(), a.data->ptr(), b.data->ptr(), a.validity ? a.validity->as<uint64_t>() : nullptr,
                                                                 b.validity ? b.validity->as<uint64_t>() : nullptr, n, k.data->as<uint64_t>(),
                                                                 k.validity ? k.validity->as<uint64_t>() : nullptr));
    return k;
}

int HashJoinExec::narrow_key_width() const {
    if (pair_keys()) return 8;
    if (on_.size() != 1) return 0;
    const Schema &ls = *left_->schema(), &rs = *right_->schema();
    const int lt = ls.fields[ls.index_of(on_[0].first)].dtype, rt = rs.fields[rs.index_of(on_[0].second)].dtype;
    auto four = [](int t) { return t == DT_INT32 || t == DT_DATE32; };
    if (four(lt) && four(rt)) return 4;
    if (lt == rt && (lt == DT_INT64 || lt == DT_UINT64)) return 8;
    return 0;
}

// A child of a join, executed so that the columns this join only passes on (everything but its keys) may arrive as views: the
// child is a HashJoinExec, or a projection of plain columns over one (the shape of TPC-H's join chains).  Anything else: execute().
static bool join_views_disabled() {
    static const bool no_views = [] { const char* v = getenv("BHIP_NO_JOIN_VIEWS"); return v && atoi(v) != 0; }();
    return no_views;
}
static StreamPtr open_join_child(const PlanPtr& child, int partition, const Exec& ex, const std::vector<std::string>& key_names) {
    const bool no_views = join_views_disabled();
    auto is_key = [&](const std::string& n) { return std::find(key_names.begin(), key_names.end(), n) != key_names.end(); };
    if (no_views) return child->execute(partition, ex);
    if (auto hj = dynamic_cast<const HashJoinExec*>(child.get())) {
        const Schema& js = *hj->schema();
        std::vector<bool> needed(js.fields.size(), true), defer(js.fields.size(), false);
        for (size_t i = 0; i < js.fields.size(); ++i) defer[i] = !is_key(js.fields[i].name);
        return hj->execute_needed(partition, ex, needed, defer);
    }
    auto pr = dynamic_cast<const ProjectionExec*>(child.get());
    const HashJoinExec* hj = pr ? dynamic_cast<const HashJoinExec*>(pr->input().get()) : nullptr;
    if (!hj) return child->execute(partition, ex);
    const Schema& js = *hj->schema();
    std::vector<int> src;
    for (auto& en : pr->exprs()) {
        if (en.first->kind != BHIP_EXPR_COLUMN) return child->execute(partition, ex);
        const int i = js.index_of(en.first->name);
        if (i < 0) return child->execute(partition, ex);
        src.push_back(i);
    }
    std::vector<bool> needed(js.fields.size(), false), defer(js.fields.size(), true);
    for (size_t k = 0; k < src.size(); ++k) {
        needed[src[k]] = true;
        if (is_key(pr->exprs()[k].second)) defer[src[k]] = false;          // a key of the parent join (possibly under another output name too)
    }
    std::shared_ptr<RecordBatchStream> inner(hj->execute_needed(partition, ex, needed, defer).release());
    const SchemaPtr sch = pr->schema();
    return StreamPtr(new LazyStream(sch, [inner, sch, src]() {
        std::vector<BatchPtr> out;
        while (BatchPtr b = inner->next()) {
            auto nb = std::make_shared<Batch>();
            nb->schema = sch;
            nb->ctx = b->ctx;
            nb->n_rows = b->n_rows;
            for (int i : src) nb->cols.push_back(b->cols[i]);
            out.push_back(nb);
        }
        return out;
    }));
}

std::shared_ptr<const JoinBuildSide> HashJoinExec::build_side(const Exec& ex) const {
    std::lock_guard<std::mutex> g(cache_->mu);
    if (cache_->built) return cache_->built;
    auto bs = std::make_shared<JoinBuildSide>();
    std::vector<BatchPtr> parts;
    const int np = left_->output_partitioning().count;
    std::vector<std::string> left_keys;
    for (auto& p : on_) left_keys.push_back(p.first);
    for (int p = 0; p < np; ++p) {
        auto s = open_join_child(left_, p, ex, left_keys);
        while (BatchPtr b = s->next())
            if (b->n_rows > 0) parts.push_back(b);
    }
    // concat works on ordinary columns; and a small build side (Q5: the five nations of a region) is gathered here, once, so that
    // all its columns travel on as views over ONE index vector (a view in, a view out: one more index vector to compose per join above)
    if (parts.size() > 1 || (parts.size() == 1 && parts[0]->n_rows <= 65536))
        for (auto& b : parts) b = materialize_batch(ex, b);
    if (parts.empty()) {
        auto e = std::make_shared<Batch>();
        e->schema = left_->schema();
        e->ctx = ex.ctx;
        for (auto& f : e->schema->fields) {
            Column c;
            c.dtype = f.dtype;
            c.data = make_buffer(ex, 8);
            if (f.dtype == DT_UTF8) { c.offsets = make_buffer(ex, 8); HIP_CHECK(hipMemsetAsync(c.offsets->ptr(), 0, 8, ex.stream)); }
            e->cols.push_back(c);
        }
        bs->batch = e;
    } else {
        bs->batch = concat_batches(ex, left_->schema(), parts);
    }
    const int64_t n = bs->batch->n_rows;
    // build rows are addressed by 32-bit slots / ranks in every table form
    if (n > 0x7FFFFFF0ll) fail(BHIP_ENOTIMPL, "hash join build side of more than 2^31 rows per partition");
    std::vector<std::string> lcols;
    for (auto& p : on_) lcols.push_back(p.first);
    uint64_t cap = 1024;
    while (cap < 2ull * (uint64_t)n) cap <<= 1;
    static const bool narrow_disabled = [] { const char* v = getenv("BHIP_NO_NARROW_JOIN"); return v && atoi(v) != 0; }();
    // BHIP_JOIN_TABLE=1: always the CAS table (+ key-set bitmap), the round-1 design — the A/B partner of the rank map
    static const bool force_table = [] { const char* v = getenv("BHIP_JOIN_TABLE"); return v && atoi(v) != 0; }();
    // the single-key structures over key column `kc` of width `nkw`: rank map, else CAS table; false: the keys are not unique
    auto try_narrow = [&](const Column& kc, int nkw) -> bool {
        // optimistic: the build side of a key join is almost always unique
        const uint64_t* ksel = kc.validity ? kc.validity->as<uint64_t>() : nullptr;
        memset(&bs->ntable, 0, sizeof(bs->ntable));
        bs->narrow_width = nkw;
        bs->dup = make_buffer(ex, 8);
        Temp tmp(ex);
        struct Stats3 { uint64_t v[3]; };
        // a dimension table's keys (<= 1024 rows spanning <= 2^16 values: Q5's nation and region): statistics, key set, packed map and
        // permutation in ONE launch and ONE host read (kernels_join.hip: tiny_rank_build_kernel) instead of four launches and two or
        // three reads; anything else — a wider span, duplicate keys — carries on below as if nothing had happened
        static const bool no_tiny = [] { const char* v = getenv("BHIP_NO_TINY_BUILD"); return v && atoi(v) != 0; }();
        static const bool radix_join_ab = [] { const char* v = getenv("BHIP_JOIN_RADIX"); return v && atoi(v) != 0; }();
        if (!no_tiny && !force_table && !radix_join_ab && n >= 1 && n <= tiny_rank_build_max_rows()) {
            BufferPtr rp = make_buffer(ex, tiny_rank_build_map_words() * 8 + 16), pm = make_buffer(ex, (size_t)n * 4 + 8);
            uint64_t* out = tmp.get<uint64_t>(3);
            TIMED_LAUNCH(ex, "tiny_rank_build", launch_tiny_rank_build(ex.cfg(), kc.data->ptr(), nkw, ksel, (uint32_t)n, rp->as<uint64_t>(), pm->as<uint32_t>(), out));
            const Stats3 got = read_device(ex, reinterpret_cast<const Stats3*>(out));    // (the kernel has finished: the map is complete before it is published)
            const bool unsorted = got.v[2] & 1, dup = got.v[2] & 2, built = got.v[2] & 4;
            if (built && !dup) {
                const uint64_t bias = nkw == 4 ? 0x80000000ull : (1ull << 63);
                const uint64_t range = got.v[1] - got.v[0], kmin = got.v[0] ^ bias;
                bs->rpack = rp;
                if (unsorted) bs->rperm = pm;
                bs->ntable.kmin64 = kmin;
                bs->ntable.kmin = (uint32_t)kmin;
                bs->ntable.rpack = rp->as<uint64_t>();
                bs->ntable.rbits = nullptr;
                static const bool no_scalar = [] { const char* v = getenv("BHIP_PROBE_NO_SCALAR_MAP"); return v && atoi(v) != 0; }();
                bs->ntable.scalar_map = no_scalar ? 0u : 1u;
                bs->ntable.rzero = 2u * ((uint32_t)(range >> 6) + 1u);
                bs->ntable.rperm = unsorted ? pm->as<uint32_t>() : nullptr;
                bs->ntable.krange64 = range;
                bs->narrow = bs->unique = true;
                return true;
            }
        }
        // one pass: min / max / "strictly increasing"
        uint64_t* stats = tmp.get<uint64_t>(3);
        {
            FillMany fm;                                         // the duplicate flag and the seed {~0, 0, 0} of the statistics: one launch
            fm.add(bs->dup->ptr(), 8);
            fm.add(stats, 8, 0xFFFFFFFFu);
            fm.add(stats + 1, 16);
            TIMED_LAUNCH(ex, "fill_many", launch_fill_many(ex.cfg(), fm));
        }
        TIMED_LAUNCH_N(ex, "join_key_stats", n, launch_join_key_stats(ex.cfg(), kc.data->ptr(), nkw, ksel, (uint32_t)n, stats));
        const Stats3 back = read_device(ex, reinterpret_cast<const Stats3*>(stats));        // one pinned-slot read, no staged copy
        const uint64_t* host_stats = back.v;
        const uint64_t bias = nkw == 4 ? 0x80000000ull : (1ull << 63);
        const bool any_key = host_stats[0] <= host_stats[1];
        const uint64_t range = any_key ? host_stats[1] - host_stats[0] : 0;
        const uint64_t kmin = (host_stats[0] ^ bias);                        // raw key bits of the smallest key
        const bool sorted = host_stats[2] == 0;
        bs->ntable.kmin64 = kmin;
        bs->ntable.kmin = (uint32_t)kmin;
        // a window of at most 2^36 values (granule indices and `rzero` are 32-bit; SF1000 order keys span 1.5 - 6 x 10^9 and hash
        // partitioning does not narrow a rank's window) that is not absurdly sparse (<= 1 KiB of map per build row).
        // BHIP_RANK_WINDOW_LOG2 lowers the bound (30 = the round-2 limit: the A/B partner, profiles/r03_rank_window_ab.txt)
        static const int window_log2 = [] { const char* v = getenv("BHIP_RANK_WINDOW_LOG2"); const int b = v ? atoi(v) : 36; return b < 10 ? 10 : (b > 36 ? 36 : b); }();
        const bool window_ok = any_key && range <= (1ull << window_log2) && range / 4096 <= (uint64_t)n + 256;
        if (window_ok && !force_table) {
            // ---- rank map ----------------------------------------------------------------------------------------------
            const int64_t n_words = (int64_t)(range >> 6) + 1, n_gran = 2 * n_words;
            // the key-set words outlive the build when they are small (<= 256 MiB: windows up to 2^31 values): semi-joins probe them
            const bool keep_bits = (size_t)n_words * 8 <= ((size_t)256 << 20);
            if (keep_bits) bs->rbits = make_buffer(ex, ((size_t)n_words + 2) * 8);
            uint64_t* bits = keep_bits ? bs->rbits->as<uint64_t>() : tmp.get<uint64_t>((size_t)n_words + 2);
            bs->rpack = make_buffer(ex, (size_t)n_gran * 8 + 16);
            {
                FillMany fm;
                fm.add(bs->rpack->as<uint64_t>() + n_gran, 8);         // NarrowJoinTable::rzero
                fm.add(bits, ((size_t)n_words + 2) * 8);               // the key set, and the granules at and behind `rzero`
                TIMED_LAUNCH(ex, "fill_many", launch_fill_many(ex.cfg(), fm));
            }
            TIMED_LAUNCH_N(ex, sorted ? "rank_bits_sorted" : "rank_bits_any", n,
                           launch_rank_bits(ex.cfg(), kc.data->ptr(), nkw, ksel, (uint32_t)n, kmin, sorted, bits, bs->dup->as<uint32_t>()));
            void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_gran));
            uint64_t* total = tmp.get<uint64_t>(1);
            TIMED_LAUNCH_N(ex, "rank_pack", n_gran, launch_rank_pack(ex.stream, reinterpret_cast<const uint32_t*>(bits), n_gran, bs->rpack->as<uint64_t>(), total, scan_tmp));
            bool dup = false;
            if (!sorted) {
                dup = read_device(ex, bs->dup->as<uint32_t>()) != 0;           // duplicates would collide in perm[]
                if (!dup) {
                    bs->rperm = make_buffer(ex, (size_t)n * 4 + 8);
                    TIMED_LAUNCH_N(ex, "rank_perm", n, launch_rank_perm(ex.cfg(), kc.data->ptr(), nkw, ksel, (uint32_t)n, kmin, bs->rpack->as<uint64_t>(),
                                                                       bs->rperm->as<uint32_t>()));
                }
            }
            if (!dup) {
                stream_wait(ex);                 // other tasks (other streams) read the map: complete before it is published
                bs->ntable.rpack = bs->rpack->as<uint64_t>();
                bs->ntable.rbits = bs->rbits ? bs->rbits->as<uint32_t>() : nullptr;
                static const bool no_scalar_map = [] { const char* v = getenv("BHIP_PROBE_NO_SCALAR_MAP"); return v && atoi(v) != 0; }();
                bs->ntable.scalar_map = no_scalar_map ? 0u : 1u;
                bs->ntable.rzero = (uint32_t)n_gran;
                bs->ntable.rperm = bs->rperm ? bs->rperm->as<uint32_t>() : nullptr;
                bs->ntable.krange64 = range;
                bs->narrow = bs->unique = true;
                static const bool radix_ab = [] { const char* v = getenv("BHIP_JOIN_RADIX"); return v && atoi(v) != 0; }();
                if (radix_ab && nkw == 4 && !ksel && n < (1ll << 31)) {
                    int lg = 0;
                    while ((n >> lg) > 1024 && lg < 16) ++lg;
                    bs->rj_log2p = lg;
                    radix_partition_side(ex, kc.data->as<uint32_t>(), n, lg, bs->rj_keys, bs->rj_rows, bs->rj_first);
                    stream_wait(ex);
                }
                return true;
            }
            bs->rpack.reset();
            bs->rbits.reset();
        } else {
            // ---- CAS table (sparse keys), with the key set as a bitmap in front of it when the window allows -------------
            const size_t slot_bytes = nkw == 4 ? 8 : 16;
            bs->slots = make_buffer(ex, cap * slot_bytes);
            HIP_CHECK(hipMemsetAsync(bs->slots->ptr(), 0, cap * slot_bytes, ex.stream));
            bs->ntable.slots = bs->slots->as<uint64_t>();
            bs->ntable.mask = cap - 1;
            bs->ntable.dup_flag = bs->dup->as<uint32_t>();
            TIMED_LAUNCH_N(ex, "join_build_narrow", n, launch_join_build_narrow(ex.cfg(), bs->ntable, kc.data->ptr(), nkw, ksel, (uint32_t)n));
            if (any_key && range <= (1ull << 30) && n >= (1 << 18)) {
                const size_t words = (size_t)range / 32 + 2;
                bs->present = make_buffer(ex, words * 4);
                HIP_CHECK(hipMemsetAsync(bs->present->ptr(), 0, words * 4, ex.stream));
                if (nkw == 4)
                    TIMED_LAUNCH_N(ex, "join_key_present", n, launch_join_key_present(ex.cfg(), kc.data->as<uint32_t>(), ksel, (uint32_t)n, (uint32_t)kmin,
                                                                                     bs->present->as<uint32_t>()));
                else
                    TIMED_LAUNCH_N(ex, "join_key_present64", n, launch_join_key_present64(ex.cfg(), kc.data->as<uint64_t>(), ksel, (uint32_t)n, kmin,
                                                                                         bs->present->as<uint32_t>()));
            }
            if (read_device(ex, bs->dup->as<uint32_t>()) == 0) {
                bs->ntable.dup_flag = nullptr;
                if (bs->present) {
                    bs->ntable.present = bs->present->as<uint32_t>();
                    bs->ntable.krange64 = range;
                }
                bs->narrow = bs->unique = true;
                return true;
            }
            bs->slots.reset();
            bs->present.reset();
        }
        memset(&bs->ntable, 0, sizeof(bs->ntable));
            return false;
    };
    const int nkw = narrow_disabled ? 0 : narrow_key_width();
    if (nkw && n > 0) {
        const Schema& lsch = *bs->batch->schema;
        const Column& c0 = bs->batch->cols[lsch.index_of(lcols[0])];
        if (pair_keys()) {
            const Column& c1 = bs->batch->cols[lsch.index_of(lcols[1])];
            // ON (a, b) = (c, d), 4-byte integers.  First choice: the build side unique on `a` alone (a key and an attribute it determines:
            // TPC-H Q5's s_suppkey, s_nationkey) — the join goes by `a` (rank map where the keys are dense) and a match stands only if
            // the second columns agree (join_filter_probe_kernel<RESID>).  Second: both columns packed into one 8-byte key.
            Column first = c0;
            if (c1.validity) {                         // a build row with a NULL in either part matches nothing
                first.validity = make_buffer(ex, bitmap_bytes(n) + 8);
                HIP_CHECK(launch_and_bitmaps(ex.cfg(), c0.validity ? c0.validity->as<uint64_t>() : nullptr, c1.validity->as<uint64_t>(), n, first.validity->as<uint64_t>()));
            }
            if (try_narrow(first, 4)) {
                bs->key_holder = first;
                bs->ntable.resid_build = c1.data->as<uint32_t>();
                bs->resid = true;
                cache_->built = bs;
                return bs;
            }
            const Column packed = pack_key_pair(ex, c0, c1, n);
            if (try_narrow(packed, 8)) { bs->key_holder = packed; cache_->built = bs; return bs; }
        } else if (try_narrow(c0, nkw)) {
            cache_->built = bs;
            return bs;
        }
    }
    side_keys(ex, *bs->batch, lcols, bs->keys, bs->sel, bs->has_sel);
    bs->owner = make_buffer(ex, cap * 8);
    bs->head = make_buffer(ex, cap * 4);
    bs->next = make_buffer(ex, (size_t)(n + 1) * 4);
    HIP_CHECK(hipMemsetAsync(bs->owner->ptr(), 0, cap * 8, ex.stream));
    HIP_CHECK(hipMemsetAsync(bs->head->ptr(), 0, cap * 4, ex.stream));
    bs->table.owner = bs->owner->as<uint64_t>();
    bs->table.head = bs->head->as<uint32_t>();
    bs->table.next = bs->next->as<uint32_t>();
    bs->table.mask = cap - 1;
    bs->table.keys128 = bs->keys->as<uint64_t>();
    bs->dup = make_buffer(ex, 8);
    HIP_CHECK(hipMemsetAsync(bs->dup->ptr(), 0, 8, ex.stream));
    bs->table.dup_flag = bs->dup->as<uint32_t>();
    TIMED_LAUNCH_N(ex, "join_build", n, launch_join_build(ex.cfg(), bs->table, bs->has_sel ? bs->sel->as<uint64_t>() : nullptr, (uint32_t)n));
    // other tasks (other HIP streams) will read the table: it must be complete before it is published
    bs->unique = read_device(ex, bs->dup->as<uint32_t>()) == 0;
    bs->table.dup_flag = nullptr;
    cache_->built = bs;
    return bs;
}

// the integer image of a double range [lo, hi] over Int32 values (empty: lo > hi) — the SOP plan table keeps ranges as doubles
static void int_bounds(double lo, double hi, int32_t* lo_i, int32_t* hi_i) {
    *lo_i = 1; *hi_i = 0;
    if (lo != lo || hi != hi || lo > 2147483647.0 || hi < -2147483648.0) return;
    const double l = __builtin_ceil(lo), h = __builtin_floor(hi);
    *lo_i = l <= -2147483648.0 ? (int32_t)(-2147483647 - 1) : (int32_t)l;
    *hi_i = h >= 2147483647.0 ? (int32_t)2147483647 : (int32_t)h;
}

namespace {

// one side of the radix join in partition order: sort keys (partition id << 32 | key), row ids, partition bounds
void radix_partition_side(const Exec& ex, const uint32_t* keys, int64_t n, int log2p, BufferPtr& skeys, BufferPtr& srows, BufferPtr& first) {
    const LaunchCfg cfg = ex.cfg();
    skeys = make_buffer(ex, (size_t)n * 8 + 8);
    srows = make_buffer(ex, (size_t)n * 4 + 8);
    TIMED_LAUNCH_N(ex, "radix_join_keys", n, launch_radix_join_keys(cfg, keys, (uint32_t)n, log2p, skeys->as<uint64_t>(), srows->as<uint32_t>()));
    if (log2p > 0 && n > 1) {
        Temp tmp(ex);
        BufferPtr k2 = make_buffer(ex, (size_t)n * 8 + 8), r2 = make_buffer(ex, (size_t)n * 4 + 8);
        void* pass_tmp = tmp.get<uint8_t>(radix_sort_temp_bytes(n));
        for (int byte = 4; byte < 4 + (log2p + 7) / 8; ++byte) {
            KernelTimer kt(ex, "radix_join_partition_pass", n, (uint64_t)n * 32);
            HIP_CHECK(radix_pass(cfg, skeys->as<uint64_t>(), srows->as<uint32_t>(), n, byte, k2->as<uint64_t>(), r2->as<uint32_t>(), pass_tmp));
            kt.stop();
            std::swap(skeys, k2);
            std::swap(srows, r2);
        }
        stream_wait(ex);
    }
    first = make_buffer(ex, ((size_t)(1u << log2p) + 2) * 4);
    TIMED_LAUNCH(ex, "radix_join_bounds", launch_radix_join_bounds(cfg, skeys->as<uint64_t>(), (uint32_t)n, 1u << log2p, first->as<uint32_t>()));
}

// where the probe rows come from: right_ = [ProjectionExec(plain columns)] over [CoalesceBatchesExec]* over [FilterExec] over src.
// The probe then runs on src's UNFILTERED batches and only the rows that join are ever gathered (late materialisation).
struct ProbeChain {
    bool ok = false;
    PlanPtr src;
    ExprPtr pred;                  // may be null
    std::vector<int> src_of;       // column i of right_'s schema -> column of src's schema
};

ProbeChain probe_chain(const PlanPtr& right) {
    ProbeChain c;
    const ExecutionPlan* p = right.get();
    PlanPtr cur = right;
    const ProjectionExec* proj = dynamic_cast<const ProjectionExec*>(p);
    if (proj) {
        for (auto& en : proj->exprs())
            if (en.first->kind != BHIP_EXPR_COLUMN) return c;          // computed columns: the operator runs as it stands
        cur = proj->input();
    }
    while (auto co = dynamic_cast<const CoalesceBatchesExec*>(cur.get())) cur = co->input();
    if (auto flt = dynamic_cast<const FilterExec*>(cur.get())) {
        c.pred = flt->predicate();
        cur = flt->input();
    }
    c.src = cur;
    const Schema& ss = *c.src->schema();
    if (proj) {
        for (auto& en : proj->exprs()) {
            const int j = ss.index_of(en.first->name);
            if (j < 0) return c;
            c.src_of.push_back(j);
        }
    } else {
        for (size_t i = 0; i < right->schema()->fields.size(); ++i) c.src_of.push_back((int)i);
    }
    c.ok = true;
    return c;
}

// `pred` as an AND of integer ranges over NULL-free Int32 / Date32 columns of `b` (the fused probe's filter form)
bool int_ranges_of(const ExprPtr& pred, const Batch& b, ProbeFilter& F) {
    memset(&F, 0, sizeof(F));
    if (!pred) return true;
    SopPlan rp;
    if (!build_sop(*b.schema, pred, {}, {}, rp) || rp.prog.n_ranges < 1 || rp.prog.n_ranges > JOIN_FILTER_MAX) return false;
    for (int i = 0; i < rp.prog.n_ranges; ++i) {
        const SopRange& r = rp.prog.ranges[i];
        if (!r.is32) return false;
        const Column& c = b.cols[rp.col_map[r.col]];
        if (c.validity || (c.dtype != DT_INT32 && c.dtype != DT_DATE32)) return false;
        F.col[i] = c.data->as<int32_t>();
        int_bounds(r.lo, r.hi, &F.lo[i], &F.hi[i]);
    }
    F.n = rp.prog.n_ranges;
    return true;
}

}  // namespace

StreamPtr HashJoinExec::execute(int partition, const Exec& ex) const {
    return execute_needed(partition, ex, std::vector<bool>(schema_->fields.size(), true));
}

StreamPtr HashJoinExec::execute_needed(int partition, const Exec& ex, const std::vector<bool>& needed_in, const std::vector<bool>& deferrable_in) const {
    check_partition(*this, partition);
    auto self = std::static_pointer_cast<const HashJoinExec>(shared_from_this());
    std::vector<bool> needed = needed_in, deferrable = deferrable_in;
    needed.resize(schema_->fields.size(), true);
    deferrable.resize(schema_->fields.size(), false);
    return StreamPtr(new LazyStream(schema_, [self, partition, ex, needed, deferrable]() {
        std::vector<BatchPtr> out;
        auto bs = self->build_side(ex);
        const Batch& L = *bs->batch;
        const int64_t n_left = L.n_rows;
        const size_t n_lcols = L.cols.size();
        const bool right_outer = self->join_type_ == BHIP_JOIN_RIGHT;
        const bool left_outer = self->join_type_ == BHIP_JOIN_LEFT;
        const LaunchCfg cfg = ex.cfg();
        bool need_left = false;
        for (size_t i = 0; i < n_lcols; ++i) need_left = need_left || needed[i];
        BufferPtr matched;
        if (left_outer) {
            matched = make_buffer(ex, (size_t)(n_left / 32 + 2) * 4);
            HIP_CHECK(hipMemsetAsync(matched->ptr(), 0, (size_t)(n_left / 32 + 2) * 4, ex.stream));
        }
        std::vector<std::string> rcols;
        for (auto& p : self->on_) rcols.push_back(p.second);

        // output batch from index pairs; R: the batch the right columns are gathered from, rmap[i] = its column for right
        // output column i (nullptr: self->right_cols_).  Columns no parent reads stay placeholders.
        // (lbuf / rbuf: the buffers that own lidx / ridx, when the caller has them — view columns keep them instead of a copy)
        auto emit = [&](const Batch* R, const std::vector<int>* rmap, const uint32_t* lidx, const uint32_t* ridx, int64_t n_out,
                        const BufferPtr& lbuf = nullptr, const BufferPtr& rbuf = nullptr) {
            auto b = std::make_shared<Batch>();
            b->schema = self->schema_;
            b->ctx = ex.ctx;
            b->n_rows = n_out;
            b->cols.resize(self->schema_->fields.size());
            for (size_t i = 0; i < b->cols.size(); ++i) { b->cols[i].dtype = self->schema_->fields[i].dtype; b->cols[i].length = n_out; }
            // [0]: gathered now; [1]: handed on as views (the parent asked for them that way; a column that arrives as a view is
            // composed with this join's indices either way)
            std::vector<const Column*> lc[2], rc[2];
            std::vector<size_t> lpos[2], rpos[2];
            for (size_t i = 0; i < n_lcols; ++i)
                if (needed[i]) { lc[deferrable[i] ? 1 : 0].push_back(&L.cols[i]); lpos[deferrable[i] ? 1 : 0].push_back(i); }
            for (int v = 0; v < 2; ++v) {
                if (lc[v].empty()) continue;
                if (lidx) {
                    auto got = take_columns(ex, lc[v], lidx, n_out, right_outer, false, v == 1, lbuf);
                    for (size_t k = 0; k < got.size(); ++k) b->cols[lpos[v][k]] = std::move(got[k]);
                } else {
                    for (size_t k = 0; k < lc[v].size(); ++k) b->cols[lpos[v][k]] = null_column(ex, lc[v][k]->dtype, n_out);
                }
            }
            for (size_t k = 0; k < self->right_cols_.size(); ++k) {
                const size_t oi = n_lcols + k;
                if (!needed[oi]) continue;
                if (!R) { b->cols[oi] = null_column(ex, self->schema_->fields[oi].dtype, n_out); continue; }
                rc[deferrable[oi] ? 1 : 0].push_back(&R->cols[rmap ? (*rmap)[k] : self->right_cols_[k]]);
                rpos[deferrable[oi] ? 1 : 0].push_back(oi);
            }
            for (int v = 0; v < 2; ++v) {
                if (rc[v].empty()) continue;
                auto got = take_columns(ex, rc[v], ridx, n_out, left_outer, false, v == 1, rbuf);
                for (size_t k = 0; k < got.size(); ++k) b->cols[rpos[v][k]] = std::move(got[k]);
            }
            out.push_back(b);
        };

        // ---- probe of a materialised batch (general table, or a probe side that is not a filter chain) ---------------------
        // `probe` holds the key columns of the probe rows; output columns are gathered from `outsrc`, whose row of probe row i
        // is remap[i] (nullptr: the same row)
        auto process = [&](const Batch& probe, const Batch* outsrc, const std::vector<int>* rmap, const uint32_t* remap) {
            const int64_t n_right = probe.n_rows;
            if (n_right == 0) return;
            BufferPtr rkeys, rsel;
            bool has_rsel = false;
            side_keys(ex, probe, rcols, rkeys, rsel, has_rsel);
            Temp tmp(ex);
            const uint64_t* rselp = has_rsel ? rsel->as<uint64_t>() : nullptr;
            uint64_t* total = tmp.get<uint64_t>(1);
            uint32_t *lidx = nullptr, *ridx = nullptr;
            uint64_t n_out = 0;
            if (bs->unique) {
                // one probe per row -> selection bitmap -> indices (the index pass of FilterExec)
                const int64_t n_tiles = (n_right + SEL_TILE - 1) / SEL_TILE;
                uint32_t* partner = tmp.get<uint32_t>((size_t)n_right + 1);
                uint64_t* bitmap = tmp.get<uint64_t>((size_t)(n_right + 63) / 64 + 1);
                uint32_t* tile_counts = tmp.get<uint32_t>((size_t)n_tiles + 1);
                uint64_t* tile_off = tmp.get<uint64_t>((size_t)n_tiles + 1);
                void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_tiles));
                TIMED_LAUNCH_N(ex, "join_probe_match", n_right,
                               launch_join_probe_match(cfg, bs->table, rkeys->as<uint64_t>(), rselp, (uint32_t)n_right, right_outer, partner,
                                                       bitmap, tile_counts, left_outer ? matched->as<uint32_t>() : nullptr));
                HIP_CHECK(exclusive_scan_u32_u64(ex.stream, tile_counts, n_tiles, tile_off, false, total, scan_tmp));
                n_out = read_device(ex, total);
                if (n_out == 0) return;
                lidx = tmp.get<uint32_t>((size_t)n_out);
                ridx = tmp.get<uint32_t>((size_t)n_out);
                TIMED_LAUNCH_N(ex, "select_indices", n_right, launch_select_indices(cfg, bitmap, tile_off, n_right, ridx));
                if (need_left) TIMED_LAUNCH_N(ex, "take_fixed", n_out, launch_take_fixed(cfg, partner, 4, ridx, (int64_t)n_out, lidx));
            } else {
                uint32_t* counts = tmp.get<uint32_t>((size_t)n_right + 1);
                uint64_t* offsets = tmp.get<uint64_t>((size_t)n_right + 1);
                void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_right));
                TIMED_LAUNCH_N(ex, "join_probe_count", n_right,
                               launch_join_probe_count(cfg, bs->table, rkeys->as<uint64_t>(), rselp, (uint32_t)n_right, right_outer, counts));
                HIP_CHECK(exclusive_scan_u32_u64(ex.stream, counts, n_right, offsets, false, total, scan_tmp));
                n_out = read_device(ex, total);
                if (n_out > 0xFFFFFFF0ull) fail(BHIP_EEXEC, "join output of one probe batch exceeds 2^32 rows");
                if (n_out == 0) return;
                lidx = tmp.get<uint32_t>((size_t)n_out);
                ridx = tmp.get<uint32_t>((size_t)n_out);
                TIMED_LAUNCH_N(ex, "join_probe_emit", n_right,
                               launch_join_probe_emit(cfg, bs->table, rkeys->as<uint64_t>(), rselp, (uint32_t)n_right, right_outer, offsets,
                                                      lidx, ridx, left_outer ? matched->as<uint32_t>() : nullptr));
            }
            if (remap) {
                uint32_t* orig = tmp.get<uint32_t>((size_t)n_out);
                TIMED_LAUNCH_N(ex, "take_fixed", n_out, launch_take_fixed(cfg, remap, 4, ridx, (int64_t)n_out, orig));
                ridx = orig;
            }
            emit(outsrc, rmap, lidx, ridx, (int64_t)n_out);
            // (no wait: the index scratch is released in stream order — host/core.cpp Context::alloc)
        };

        // ---- narrow build side: one pass over the probe rows: ranges -> key-set bit -> rank map / table (kernels_join.hip) -----
        // n probe rows whose keys are `kc`; output columns come from `outsrc` (row remap[i] of it for probe row i; nullptr: row i)
        auto process_fused = [&](int64_t n, const ProbeFilter& F, const Column& kc, const Batch* outsrc, const std::vector<int>* rmap,
                                 const uint32_t* remap, const Column* resid = nullptr) {
            if (n == 0) return;
            const int64_t n_tiles = (n + SEL_TILE - 1) / SEL_TILE;
            Temp tmp(ex);
            uint64_t* bitmap = tmp.get<uint64_t>((size_t)n_tiles * (SEL_TILE / 64) + 1);       // whole tiles: the kernel writes every word of a tile
            uint32_t* tile_counts = tmp.get<uint32_t>((size_t)n_tiles + 1);
            uint64_t* tile_off = tmp.get<uint64_t>((size_t)n_tiles + 1);
            uint64_t* total = tmp.get<uint64_t>(1);
            void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_tiles));
            uint32_t* staging = need_left ? tmp.get<uint32_t>((size_t)n_tiles * SEL_TILE) : nullptr;
            // the emitted rows are staged per tile like their partners: the compaction below yields both index vectors, no pass over
            // the selection bitmap (a semi-join — no build column read — has nothing else to compact and takes the bitmap route)
            uint32_t* staging_rows = need_left ? tmp.get<uint32_t>((size_t)n_tiles * SEL_TILE) : nullptr;
            // algorithmic bytes: the predicate columns and the key column once, one selection bit per row
            // (named after the kernel the launcher picks — kernels_join.hip launch_join_filter_probe: rank map, no NULL probe keys,
            // an inner join -> join_rank_probe_kernel — so that bench.py's roofline line and the rocprofv3 summaries agree)
            const bool direct = bs->ntable.rpack != nullptr && !kc.validity && !left_outer && !right_outer;
            TIMED_LAUNCH_B(ex, direct ? "join_rank_probe" : "join_filter_probe", n, (uint64_t)n * (uint64_t)(bs->narrow_width + 4 * F.n) + (uint64_t)n / 8,
                           launch_join_filter_probe(cfg, bs->ntable, F, kc.data->ptr(), bs->narrow_width,
                                                    kc.validity ? kc.validity->as<uint64_t>() : nullptr, (uint32_t)n, right_outer, bitmap,
                                                    tile_counts, staging, left_outer ? matched->as<uint32_t>() : nullptr,
                                                    resid ? resid->data->as<uint32_t>() : nullptr, staging_rows));
            HIP_CHECK(exclusive_scan_u32_u64(ex.stream, tile_counts, n_tiles, tile_off, false, total, scan_tmp));
            const uint64_t n_out = read_device(ex, total);
            if (n_out == 0) return;
            BufferPtr rbuf = make_buffer(ex, (size_t)n_out * 4 + 8), lbuf = need_left ? make_buffer(ex, (size_t)n_out * 4 + 8) : nullptr;
            uint32_t* ridx = rbuf->as<uint32_t>();
            uint32_t* lidx = need_left ? lbuf->as<uint32_t>() : nullptr;
            if (need_left) TIMED_LAUNCH_N(ex, "join_compact_staged", n, launch_join_compact_staged(cfg, staging, tile_off, n_out, n_tiles, lidx, staging_rows, ridx));
            else TIMED_LAUNCH_N(ex, "select_indices", n, launch_select_indices(cfg, bitmap, tile_off, n, ridx));
            if (remap) {
                BufferPtr orig = make_buffer(ex, (size_t)n_out * 4 + 8);
                TIMED_LAUNCH_N(ex, "take_fixed", n_out, launch_take_fixed(cfg, remap, 4, ridx, (int64_t)n_out, orig->as<uint32_t>()));
                rbuf = orig;
                ridx = orig->as<uint32_t>();
            }
            emit(outsrc, rmap, lidx, ridx, (int64_t)n_out, lbuf, rbuf);
        };
        ProbeFilter no_filter;
        memset(&no_filter, 0, sizeof(no_filter));

        // ---- A/B partner: radix-partitioned probe side against LDS-resident tables (kernels_radix_join.hip).  false: fall back ----
        auto process_radix = [&](int64_t n, const Column& kc, const Batch* outsrc, const std::vector<int>* rmap, const uint32_t* remap) -> bool {
            if (n == 0) return true;
            if (kc.validity || n >= (1ll << 31)) return false;
            BufferPtr pk, pr, pf;
            radix_partition_side(ex, kc.data->as<uint32_t>(), n, bs->rj_log2p, pk, pr, pf);
            const int64_t n_tiles = (n + SEL_TILE - 1) / SEL_TILE;
            Temp tmp(ex);
            uint32_t* partner = tmp.get<uint32_t>((size_t)n + 1);
            uint64_t* bitmap = tmp.get<uint64_t>((size_t)(n + 63) / 64 + 1);
            uint32_t* tile_counts = tmp.get<uint32_t>((size_t)n_tiles + 1);
            uint64_t* tile_off = tmp.get<uint64_t>((size_t)n_tiles + 1);
            uint64_t* total = tmp.get<uint64_t>(1);
            uint32_t* flags = tmp.get<uint32_t>(2);
            void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_tiles));
            HIP_CHECK(hipMemsetAsync(bitmap, 0, ((size_t)(n + 63) / 64 + 1) * 8, ex.stream));
            HIP_CHECK(hipMemsetAsync(tile_counts, 0, ((size_t)n_tiles + 1) * 4, ex.stream));
            HIP_CHECK(hipMemsetAsync(flags, 0, 8, ex.stream));
            TIMED_LAUNCH_N(ex, "radix_join_lds", n,
                           launch_radix_join_lds(cfg, bs->rj_keys->as<uint64_t>(), bs->rj_rows->as<uint32_t>(), bs->rj_first->as<uint32_t>(), pk->as<uint64_t>(),
                                                 pr->as<uint32_t>(), pf->as<uint32_t>(), 1u << bs->rj_log2p, partner, bitmap, tile_counts, flags));
            HIP_CHECK(exclusive_scan_u32_u64(ex.stream, tile_counts, n_tiles, tile_off, false, total, scan_tmp));
            if (read_device(ex, flags) != 0) return false;               // a build partition outgrew the LDS table
            const uint64_t n_out = read_device(ex, total);
            if (n_out == 0) return true;
            uint32_t* ridx = tmp.get<uint32_t>((size_t)n_out);
            uint32_t* lidx = tmp.get<uint32_t>((size_t)n_out);
            TIMED_LAUNCH_N(ex, "select_indices", n, launch_select_indices(cfg, bitmap, tile_off, n, ridx));
            TIMED_LAUNCH_N(ex, "take_fixed", n_out, launch_take_fixed(cfg, partner, 4, ridx, (int64_t)n_out, lidx));
            if (remap) {
                uint32_t* orig = tmp.get<uint32_t>((size_t)n_out);
                TIMED_LAUNCH_N(ex, "take_fixed", n_out, launch_take_fixed(cfg, remap, 4, ridx, (int64_t)n_out, orig));
                ridx = orig;
            }
            emit(outsrc, rmap, lidx, ridx, (int64_t)n_out);
            return true;
        };
        const bool radix_mode = bs->narrow && bs->rj_log2p >= 0 && !bs->resid && !right_outer && !left_outer;     // (the LDS join knows one key only)

        const bool pair = self->pair_keys();
        // the probe side's key for a two-column join: the first key (+ the second as the residual) when the build side went that way,
        // else both packed into one 8-byte key; a NULL in either part never matches
        struct ProbeKey { Column key, second; bool resid; };
        auto probe_key = [&](const Column& a, const Column& b2, int64_t rows) {
            ProbeKey pk;
            pk.resid = bs->resid;
            if (!bs->resid) { pk.key = pack_key_pair(ex, a, b2, rows); return pk; }
            pk.key = a;
            pk.second = b2;
            if (b2.validity) {
                pk.key.validity = make_buffer(ex, bitmap_bytes(rows) + 8);
                HIP_CHECK(launch_and_bitmaps(cfg, a.validity ? a.validity->as<uint64_t>() : nullptr, b2.validity->as<uint64_t>(), rows, pk.key.validity->as<uint64_t>()));
            }
            return pk;
        };
        const ProbeChain chain = probe_chain(self->right_);
        static const bool fused_disabled = [] { const char* v = getenv("BHIP_NO_FUSED_PROBE"); return v && atoi(v) != 0; }();
        if (chain.ok) {
            const SchemaPtr out_schema = self->right_->schema();
            std::vector<int> rmap;                             // right OUTPUT column k -> source column
            for (int ci : self->right_cols_) rmap.push_back(chain.src_of[ci]);
            std::vector<int> key_src;
            auto key_schema = std::make_shared<Schema>();
            for (auto& rc : rcols) {
                const int j = out_schema->index_of(rc);
                key_schema->fields.push_back(out_schema->fields[j]);
                key_src.push_back(chain.src_of[j]);
            }
            // a join below (no filter in between): only the columns read here, and everything but the keys may arrive as views
            StreamPtr ss;
            auto src_hj = dynamic_cast<const HashJoinExec*>(chain.src.get());
            if (src_hj && !chain.pred && !join_views_disabled()) {
                const size_t n_src = chain.src->schema()->fields.size();
                std::vector<bool> need(n_src, false), defer(n_src, true);
                for (size_t k = 0; k < self->right_cols_.size(); ++k)
                    if (needed[n_lcols + k]) need[rmap[k]] = true;
                for (int ci : key_src) { need[ci] = true; defer[ci] = false; }
                ss = src_hj->execute_needed(partition, ex, need, defer);
            } else {
                ss = chain.src->execute(partition, ex);
            }
            while (BatchPtr b = ss->next()) {
                if (b->n_rows == 0) continue;
                ProbeFilter F;
                if (radix_mode) {
                    // the filtered key column first (FilterExec as it stands), then the partitioned join
                    const uint32_t* remap = nullptr;
                    BufferPtr sel;
                    Column keys = b->cols[key_src[0]];
                    int64_t n_probe = b->n_rows;
                    if (chain.pred) {
                        n_probe = filter_indices(ex, *b, chain.pred, sel);
                        if (n_probe == 0) continue;
                        remap = sel->as<uint32_t>();
                        keys = take_batch_column(ex, b->cols[key_src[0]], remap, n_probe);
                    }
                    if (process_radix(n_probe, keys, b.get(), &rmap, remap)) continue;
                }
                if (bs->narrow && !fused_disabled && int_ranges_of(chain.pred, *b, F)) {
                    if (pair) {
                        const ProbeKey pk = probe_key(b->cols[key_src[0]], b->cols[key_src[1]], b->n_rows);
                        process_fused(b->n_rows, F, pk.key, b.get(), &rmap, nullptr, pk.resid ? &pk.second : nullptr);
                    } else {
                        process_fused(b->n_rows, F, b->cols[key_src[0]], b.get(), &rmap, nullptr);
                    }
                    continue;
                }
                const uint32_t* remap = nullptr;
                BufferPtr sel;
                auto kb = std::make_shared<Batch>();           // the key columns under their probe-side names ...
                kb->schema = key_schema;
                kb->ctx = b->ctx;
                kb->n_rows = b->n_rows;
                if (chain.pred) {                              // ... of the surviving rows only
                    const int64_t n_sel = filter_indices(ex, *b, chain.pred, sel);
                    if (n_sel == 0) continue;
                    kb->n_rows = n_sel;
                    remap = sel->as<uint32_t>();
                    for (int ci : key_src) kb->cols.push_back(take_batch_column(ex, b->cols[ci], remap, n_sel));
                } else {
                    for (int ci : key_src) kb->cols.push_back(b->cols[ci]);
                }
                if (bs->narrow && pair) {
                    const ProbeKey pk = probe_key(kb->cols[0], kb->cols[1], kb->n_rows);
                    process_fused(kb->n_rows, no_filter, pk.key, b.get(), &rmap, remap, pk.resid ? &pk.second : nullptr);
                } else if (bs->narrow) process_fused(kb->n_rows, no_filter, kb->cols[0], b.get(), &rmap, remap);
                else process(*kb, b.get(), &rmap, remap);
            }
        } else {
            auto rs = open_join_child(self->right_, partition, ex, rcols);          // payload columns of a join below may arrive as views
            while (BatchPtr rb = rs->next()) {
                if (bs->narrow && pair) {
                    const ProbeKey pk = probe_key(rb->cols[rb->schema->index_of(rcols[0])], rb->cols[rb->schema->index_of(rcols[1])], rb->n_rows);
                    process_fused(rb->n_rows, no_filter, pk.key, rb.get(), nullptr, nullptr, pk.resid ? &pk.second : nullptr);
                } else if (bs->narrow)
                    process_fused(rb->n_rows, no_filter, rb->cols[rb->schema->index_of(rcols[0])], rb.get(), nullptr, nullptr);
                else process(*rb, rb.get(), nullptr, nullptr);
            }
        }
        if (left_outer && n_left > 0) {
            // left rows no probe row matched: right columns NULL
            Temp tmp(ex);
            uint32_t* flags = tmp.get<uint32_t>((size_t)n_left + 1);
            uint64_t* offsets = tmp.get<uint64_t>((size_t)n_left + 1);
            uint64_t* total = tmp.get<uint64_t>(1);
            void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_left));
            TIMED_LAUNCH(ex, "join_unmatched_flags", launch_join_unmatched_flags(cfg, matched->as<uint32_t>(), (uint32_t)n_left, flags));
            HIP_CHECK(exclusive_scan_u32_u64(ex.stream, flags, n_left, offsets, false, total, scan_tmp));
            const uint64_t n_un = read_device(ex, total);
            if (n_un) {
                uint32_t* lidx = tmp.get<uint32_t>((size_t)n_un);
                TIMED_LAUNCH(ex, "compact_flags", launch_compact_flags(cfg, flags, offsets, (uint32_t)n_left, lidx));
                emit(nullptr, nullptr, lidx, nullptr, (int64_t)n_un);
            }
        }
        return out;
    }));
}

}  // namespace bhip
