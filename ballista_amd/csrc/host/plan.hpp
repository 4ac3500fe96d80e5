// plan.hpp — the operator tree: a C++ mirror of DataFusion's `ExecutionPlan` /
// `RecordBatchStream` traits as the reference uses them.
//
//   trait ExecutionPlan { as_any, schema, output_partitioning, children, with_new_children,
//                         async execute(partition) -> stream }
//       rust/core/src/execution_plans/query_stage.rs:49-85 (in-tree implementation showing every method)
//   trait RecordBatchStream = Stream<Item = ArrowResult<RecordBatch>> + schema()
//       rust/core/src/memory_stream.rs:57-92
//
// One class per operator the physical-plan serde can build
// (rust/core/src/serde/physical_plan/from_proto.rs:58-346).  `execute` fuses the
// Filter / Projection / CoalesceBatches chain under a pipeline breaker into the breaker's scan
// kernel, so those operators only run stand-alone when they are at the top of a stage.
#pragma once
#include <functional>
#include <mutex>

#include "core.hpp"
#include "expr.hpp"

namespace bhip {

struct Partitioning {
    int scheme = BHIP_PART_UNKNOWN;
    int count = 1;
    std::vector<ExprPtr> exprs;
};

class RecordBatchStream {
public:
    virtual ~RecordBatchStream() = default;
    virtual SchemaPtr schema() const = 0;
    virtual BatchPtr next() = 0;   // nullptr = end of stream
};
using StreamPtr = std::unique_ptr<RecordBatchStream>;

class ExecutionPlan;
using PlanPtr = std::shared_ptr<const ExecutionPlan>;

class ExecutionPlan : public std::enable_shared_from_this<ExecutionPlan> {
public:
    virtual ~ExecutionPlan() = default;
    virtual const char* name() const = 0;                                   // as_any()
    virtual SchemaPtr schema() const = 0;
    virtual Partitioning output_partitioning() const = 0;
    virtual std::vector<PlanPtr> children() const = 0;
    virtual PlanPtr with_new_children(const std::vector<PlanPtr>& children) const = 0;
    virtual StreamPtr execute(int partition, const Exec& ex) const = 0;
    virtual std::string describe() const { return name(); }                 // fmt::Debug one-liner
    ContextPtr context() const { return ctx_; }
protected:
    ContextPtr ctx_;
};

std::string display_plan(const PlanPtr& p);
void check_partition(const ExecutionPlan& p, int partition);
std::vector<BatchPtr> drain(RecordBatchStream& s);

// ---- streams -----------------------------------------------------------------------------------
class VecStream : public RecordBatchStream {   // MemoryStream (memory_stream.rs:29-55)
public:
    VecStream(SchemaPtr s, std::vector<BatchPtr> b) : schema_(std::move(s)), batches_(std::move(b)) {}
    SchemaPtr schema() const override { return schema_; }
    BatchPtr next() override { return pos_ < batches_.size() ? batches_[pos_++] : nullptr; }
private:
    SchemaPtr schema_;
    std::vector<BatchPtr> batches_;
    size_t pos_ = 0;
};

// computes its batches on first pull
class LazyStream : public RecordBatchStream {
public:
    LazyStream(SchemaPtr s, std::function<std::vector<BatchPtr>()> f) : schema_(std::move(s)), fn_(std::move(f)) {}
    SchemaPtr schema() const override { return schema_; }
    BatchPtr next() override {
        if (!ran_) { batches_ = fn_(); ran_ = true; }
        return pos_ < batches_.size() ? batches_[pos_++] : nullptr;
    }
private:
    SchemaPtr schema_;
    std::function<std::vector<BatchPtr>()> fn_;
    std::vector<BatchPtr> batches_;
    bool ran_ = false;
    size_t pos_ = 0;
};

// ---- operators ---------------------------------------------------------------------------------
class MemoryExec : public ExecutionPlan {
public:
    MemoryExec(ContextPtr ctx, SchemaPtr schema, std::vector<std::vector<BatchPtr>> partitions);
    const char* name() const override { return "MemoryExec"; }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, (int)parts_.size(), {}}; }
    std::vector<PlanPtr> children() const override { return {}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    const std::vector<std::vector<BatchPtr>>& partitions() const { return parts_; }
private:
    SchemaPtr schema_;
    std::vector<std::vector<BatchPtr>> parts_;
};

class EmptyExec : public ExecutionPlan {
public:
    EmptyExec(ContextPtr ctx, SchemaPtr schema, bool produce_one_row);
    const char* name() const override { return "EmptyExec"; }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, 1, {}}; }
    std::vector<PlanPtr> children() const override { return {}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
private:
    SchemaPtr schema_;
    bool one_row_;
};

class UnaryExec : public ExecutionPlan {
public:
    std::vector<PlanPtr> children() const override { return {input_}; }
    Partitioning output_partitioning() const override { return input_->output_partitioning(); }
    const PlanPtr& input() const { return input_; }
protected:
    PlanPtr input_;
};

class FilterExec : public UnaryExec {
public:
    FilterExec(ExprPtr predicate, PlanPtr input);
    const char* name() const override { return "FilterExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override { return "FilterExec: " + predicate_->to_string(); }
    const ExprPtr& predicate() const { return predicate_; }
private:
    ExprPtr predicate_;
};

class ProjectionExec : public UnaryExec {
public:
    ProjectionExec(std::vector<std::pair<ExprPtr, std::string>> exprs, PlanPtr input);
    const char* name() const override { return "ProjectionExec"; }
    SchemaPtr schema() const override { return schema_; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override;
    const std::vector<std::pair<ExprPtr, std::string>>& exprs() const { return exprs_; }
private:
    std::vector<std::pair<ExprPtr, std::string>> exprs_;
    SchemaPtr schema_;
};

class CoalesceBatchesExec : public UnaryExec {
public:
    CoalesceBatchesExec(PlanPtr input, int64_t target);
    const char* name() const override { return "CoalesceBatchesExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override { return "CoalesceBatchesExec: target_batch_size=" + std::to_string(target_); }
private:
    int64_t target_;
};

class MergeExec : public UnaryExec {
public:
    explicit MergeExec(PlanPtr input);
    const char* name() const override { return "MergeExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, 1, {}}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
};

class LimitExec : public UnaryExec {     // GlobalLimitExec / LocalLimitExec
public:
    LimitExec(PlanPtr input, int64_t limit, bool global);
    const char* name() const override { return global_ ? "GlobalLimitExec" : "LocalLimitExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    Partitioning output_partitioning() const override;
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override { return std::string(name()) + ": limit=" + std::to_string(limit_); }
private:
    int64_t limit_;
    bool global_;
};

class HashAggregateExec : public UnaryExec {
public:
    HashAggregateExec(int mode, std::vector<std::pair<ExprPtr, std::string>> group_exprs,
                      std::vector<AggregateDesc> aggr, PlanPtr input);
    const char* name() const override { return "HashAggregateExec"; }
    SchemaPtr schema() const override { return schema_; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override;
private:
    std::vector<BatchPtr> run(int partition, const Exec& ex) const;
    // keys packed into 16 bytes (every fast path).  as_final != nullptr (a Partial whose only consumer is that Final aggregate over
    // this single partition, run_single_partial): the group table is emitted straight into the Final's output columns
    std::vector<BatchPtr> run_packed(int partition, const Exec& ex, const HashAggregateExec* as_final = nullptr) const;
    std::vector<BatchPtr> run_wide(int partition, const Exec& ex) const;     // keys of any width (ops_agg_wide.cpp)
    // Final over the Merge of ONE partition of a Partial aggregate with the same keys: every group arrives exactly once, the
    // merge is the identity and the operator is a projection of the state columns (AVG = sum / count)
    bool run_single_partial(const Exec& ex, std::vector<BatchPtr>& out) const;
    // group / argument expressions with string nodes (lower(s), CASE ... THEN 'a'), MIN / MAX over Utf8: the strings become
    // columns, MIN / MAX(Utf8) become MIN / MAX over sort ranks, and the ordinary aggregate runs on that (ops_agg_wide.cpp)
    std::vector<BatchPtr> run_strings(int partition, const Exec& ex) const;
    bool strings_ = false;
    int mode_;
    std::vector<std::pair<ExprPtr, std::string>> group_;
    std::vector<AggregateDesc> aggr_;
    SchemaPtr schema_;
    mutable std::atomic<int> path_hint_{0};    // 0 = unknown, 4/8 = register path with that many groups, -1 = hash path
    mutable std::atomic<bool> wide_keys_{false};   // a run found key values the packed key cannot hold
    mutable std::atomic<int> clustered_hint_{0};   // hash path: 0 = unknown, 1 = the input came clustered by group key last time, -1 = it did not
};

struct JoinBuildSide;
class HashJoinExec : public ExecutionPlan {
public:
    HashJoinExec(PlanPtr left, PlanPtr right, std::vector<std::pair<std::string, std::string>> on, int join_type);
    const char* name() const override { return "HashJoinExec"; }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return right_->output_partitioning(); }
    std::vector<PlanPtr> children() const override { return {left_, right_}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    // the same stream with only the output columns a parent reads materialised (needed[i]: column i of schema()); the other
    // columns of the batches are placeholders without buffers.  Called by ProjectionExec / HashAggregateExec above a join
    // (the reference's join copies every column of both sides and the parent then drops most of them).
    // needed[i]: a parent reads output column i (others stay placeholders).  deferrable[i]: the parent only hands column i on to
    // take_columns (another HashJoinExec passing a payload column through): it may come back as a VIEW (core.hpp Column::view_base)
    StreamPtr execute_needed(int partition, const Exec& ex, const std::vector<bool>& needed, const std::vector<bool>& deferrable = {}) const;
    std::string describe() const override;
private:
    std::shared_ptr<const JoinBuildSide> build_side(const Exec& ex) const;
    PlanPtr left_, right_;
    std::vector<std::pair<std::string, std::string>> on_;
    int join_type_;
    SchemaPtr schema_;
    std::vector<int> right_cols_;   // right columns kept in the output
    int narrow_key_width() const;
    bool pair_keys() const;          // TWO 4-byte integer key pairs: packed into one 8-byte key per side, then the single-key machinery
    // the build side (hash table over the whole left child) is built once and shared by every
    // partition's task, like DataFusion's collect-left build future
    struct BuildCache { std::mutex mu; std::shared_ptr<const JoinBuildSide> built; };
    std::shared_ptr<BuildCache> cache_;
};

class SortExec : public UnaryExec {
public:
    SortExec(std::vector<SortDesc> exprs, PlanPtr input);
    const char* name() const override { return "SortExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, 1, {}}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override;
private:
    std::vector<SortDesc> exprs_;
};

class RepartitionExec : public UnaryExec {
public:
    RepartitionExec(PlanPtr input, Partitioning part);
    const char* name() const override { return "RepartitionExec"; }
    SchemaPtr schema() const override { return input_->schema(); }
    Partitioning output_partitioning() const override { return part_; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override;
    StreamPtr execute(int partition, const Exec& ex) const override;
    std::string describe() const override;
private:
    Partitioning part_;
    // all output partitions are produced by the first execute() and handed out from here
    struct SplitCache { std::mutex mu; bool done = false; std::vector<std::vector<BatchPtr>> parts; };
    std::shared_ptr<SplitCache> cache_;
};

// ParquetExec (parquet.cpp): one partition per chunk of files, one batch per row group
PlanPtr make_parquet_exec(const ContextPtr& ctx, const std::vector<std::string>& files, const std::vector<uint32_t>& projection, bool has_projection,
                          int num_partitions);
// the wire plan (proto.cpp): protobuf PhysicalPlanNode -> operator tree; ctx may be null (inspection only)
PlanPtr plan_from_proto(const ContextPtr& ctx, const void* bytes, size_t len, bhip_leaf_resolver resolver, void* user);
ExprPtr expr_from_proto(const void* bytes, size_t len);

// shared helpers (ops_*.cpp)
// evaluate `predicate` over `in` and return the surviving rows as ascending indices
int64_t filter_indices(const Exec& ex, const Batch& in, const ExprPtr& predicate, BufferPtr& indices_out);

// expressions that produce Utf8 values (lower / upper / trim / ltrim / rtrim, CASE with string branches, string literals as
// output columns) are evaluated as extra Utf8 columns of the input batch; the rest of the expression goes to the VM with those
// nodes replaced by column references (utf8_exprs.cpp)
bool has_utf8_node(const ExprPtr& e, const Schema& schema);
class Utf8Lowering {
public:
    explicit Utf8Lowering(const Schema& in);
    ExprPtr rewrite(const ExprPtr& e, bool output = false);       // output: `e` is a whole output column (a bare string literal counts)
    bool any() const { return !nodes_.empty(); }
    SchemaPtr schema() const;                                     // the input's fields + one Utf8 field per node
    void validate() const;                                        // BHIP_ENOTIMPL for a node this layer cannot evaluate (plan time)
    BatchPtr apply(const Exec& ex, const Batch& in) const;        // the input's columns + the evaluated nodes
private:
    const Schema& in_;
    std::vector<ExprPtr> nodes_;
    std::vector<std::string> names_;
};
// split a batch by hash(exprs) % n, keeping input order inside each part
std::vector<BatchPtr> hash_partition_batch(const Exec& ex, const BatchPtr& in, const std::vector<ExprPtr>& exprs, int n);
void check_scan_status(const Exec& ex, const ScanStatus* dev_status, ScanStatus* host_out = nullptr);
void check_scan_flags(const ScanStatus& host_status);   // the same checks on a status already read back
// value of `e` over `in` as a column (a plain Column reference shares the input buffers)
Column evaluate_column(const Exec& ex, const Batch& in, const ExprPtr& e);
struct SortDesc;
// SortExec's order as a row permutation (ops_sort.cpp): stable, perm[i] = the input row that goes to position i
BufferPtr sort_permutation(const Exec& ex, const Batch& in, const std::vector<SortDesc>& exprs);
// ProjectionExec over one batch: plain columns share their buffers, everything else is one VM launch
BatchPtr project_batch(const Exec& ex, const Batch& in, const std::vector<std::pair<ExprPtr, std::string>>& exprs, const SchemaPtr& schema);
// gather of one column incl. its validity bitmap
Column take_batch_column(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n);
// columns gathered by one index vector; may_null: the indices may hold 0xFFFFFFFF (outer joins) -> validity always built
std::vector<Column> take_columns(const Exec& ex, const std::vector<const Column*>& cols, const uint32_t* idx, int64_t n,
                                 bool may_null, bool permutation = false, bool keep_views = false, const BufferPtr& idx_owner = nullptr);
// view columns (core.hpp Column::view_base) as ordinary columns
Column materialize_column(const Exec& ex, const Column& c);
BatchPtr materialize_batch(const Exec& ex, const BatchPtr& b);
// n NULLs of the given type
Column null_column(const Exec& ex, int dtype, int64_t n);
// gather where indices may hold 0xFFFFFFFF (= NULL row): always carries a validity bitmap
Column take_column_nullable(const Exec& ex, const Column& c, const uint32_t* idx, int64_t n);
// stable sort of (keys, perm) by the u64 keys; returns the buffers holding the sorted result
void radix_sort_pairs(const Exec& ex, BufferPtr& keys, BufferPtr& perm, int64_t n);

}  // namespace bhip

struct bhip_plan { bhip::PlanPtr p; std::atomic<int> rc{1}; };
struct bhip_stream {
    bhip::StreamPtr s;
    bhip::Exec ex;
    std::vector<std::string> names;
    ~bhip_stream() { if (ex.ctx && ex.stream) ex.ctx->release_stream(ex.stream); }
};
