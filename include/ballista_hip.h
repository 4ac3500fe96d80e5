/*
 * ballista_hip.h — C ABI of libballista_hip.so, the MI355X-native physical execution layer
 * that drops in underneath a Ballista executor.
 *
 * What it replaces.  A Ballista executor runs one call per task,
 *     partition.plan.execute(part)          rust/executor/src/flight_service.rs:117-121
 * on a tree of DataFusion operators rebuilt from the wire plan
 *     rust/core/src/serde/physical_plan/from_proto.rs:58-346  (operators)
 *     rust/core/src/serde/physical_plan/from_proto.rs:348-364 (compile_expr)
 * and drains the resulting RecordBatchStream (rust/core/src/utils.rs:49-84).  The trait it
 * calls through is DataFusion's `ExecutionPlan` as visible in the reference's own
 * implementations (rust/core/src/execution_plans/query_stage.rs:49-85):
 *     schema(), output_partitioning(), children(), with_new_children(), execute(partition)
 * and `RecordBatchStream` (rust/core/src/memory_stream.rs:57-92).  Every entry point below
 * names the reference interface it stands in for.  A Rust `GpuExec: ExecutionPlan` shim binds
 * these with `extern "C"` (INTEGRATION.md shows the binding).
 *
 * Conventions:
 *   - every function returns a bhip_status; 0 = OK.  Nothing aborts or throws across the ABI;
 *     the message of the last failure on the calling thread is bhip_last_error().
 *   - BHIP_ENOTIMPL means "this plan/expression/type is outside the GPU path": the caller keeps
 *     its CPU operator for that subtree (there is NO CPU fallback inside this library).
 *   - handles are reference counted and immutable after creation; plans may be shared between
 *     threads and executed concurrently (the executor runs `concurrent_tasks` tasks at once,
 *     rust/executor/executor_config_spec.toml:57-62); a stream is single-consumer.
 *   - batches cross the boundary as Arrow C Data Interface structs (host memory) or stay
 *     device-resident (bhip_batch) between chained GPU operators.
 *   - strings are UTF-8, NUL terminated, and are copied; arrays are copied.
 */
#ifndef BALLISTA_HIP_H
#define BALLISTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data / C Stream interface (https://arrow.apache.org/docs/format/CDataInterface.html) */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
    int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
    int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
    const char* (*get_last_error)(struct ArrowArrayStream*);
    void (*release)(struct ArrowArrayStream*);
    void* private_data;
};
#endif

/* ---- status ---------------------------------------------------------------------------
 * mirrors the error kinds the executor maps to tonic::Status::internal
 * (rust/executor/src/flight_service.rs:344-354; rust/core/src/error.rs:30-163) */
typedef int32_t bhip_status;
enum {
    BHIP_OK = 0,
    BHIP_EINVAL = 1,     /* DataFusionError::Plan / Internal: malformed plan, unknown column, type mismatch */
    BHIP_ENOTIMPL = 2,   /* DataFusionError::NotImplemented: keep the CPU operator */
    BHIP_EEXEC = 3,      /* DataFusionError::Execution / ArrowError: e.g. "Divide by zero error" */
    BHIP_EHIP = 4,       /* a HIP runtime call failed (message carries hipGetErrorString) */
    BHIP_EOOM = 5        /* device allocation failed */
};
const char* bhip_last_error(void);
const char* bhip_version(void);

/* ---- types (Arrow types of the TPC-H schemas, rust/benchmarks/tpch/src/main.rs:267-360) */
typedef enum {
    BHIP_INT32 = 1, BHIP_INT64 = 2, BHIP_UINT8 = 3, BHIP_UINT64 = 4, BHIP_FLOAT64 = 5,
    BHIP_DATE32 = 6, BHIP_BOOLEAN = 7, BHIP_UTF8 = 8,
    /* the other primitive types of the serde (rust/core/proto/ballista.proto:755-790) */
    BHIP_INT8 = 9, BHIP_INT16 = 10, BHIP_UINT16 = 11, BHIP_UINT32 = 12, BHIP_FLOAT32 = 13, BHIP_DATE64 = 14,
    BHIP_TIMESTAMP_S = 15, BHIP_TIMESTAMP_MS = 16, BHIP_TIMESTAMP_US = 17, BHIP_TIMESTAMP_NS = 18,
    /* schemas only: a Utf8 column whose Arrow / IPC form is LargeUtf8 (64-bit offsets); on the device it is BHIP_UTF8 */
    BHIP_LARGE_UTF8 = 19,
    /* schemas only: variable-length bytes (the digests of sha224 .. sha512) in the buffer layout of BHIP_UTF8 */
    BHIP_BINARY = 20
} bhip_dtype;

typedef struct bhip_ctx bhip_ctx;        /* one GPU: allocator, stream pool */
typedef struct bhip_batch bhip_batch;    /* device-resident RecordBatch */
typedef struct bhip_plan bhip_plan;      /* Arc<dyn ExecutionPlan> */
typedef struct bhip_stream bhip_stream;  /* Pin<Box<dyn RecordBatchStream>> */

/* ---- context ------------------------------------------------------------------------------ */
bhip_status bhip_ctx_create(int device, bhip_ctx** out);
void bhip_ctx_release(bhip_ctx* ctx);
bhip_status bhip_ctx_synchronize(bhip_ctx* ctx);
/* bytes currently held / high-water mark of the context's device allocator */
bhip_status bhip_ctx_memory(bhip_ctx* ctx, uint64_t* in_use, uint64_t* peak);

/* ---- batches ------------------------------------------------------------------------------
 * RecordBatch (columns are immutable, shared buffers).  Host constructors copy to the device;
 * bhip_batch_from_device adopts device pointers without copying (caller keeps them alive and
 * unchanged until the batch is released).  Layout per column = Arrow: fixed-width values, or
 * Utf8 int32 offsets (length+1) + bytes, Boolean = bitmap; validity bitmap optional (NULL =
 * no nulls); offset must be 0 for device columns. */
typedef struct {
    const char* name;
    int32_t dtype;            /* bhip_dtype */
    int32_t nullable;         /* schema nullability */
    const void* data;         /* values | Utf8 bytes | Boolean bitmap */
    const int32_t* offsets;   /* Utf8 only */
    const uint8_t* validity;  /* Arrow validity bitmap or NULL */
    int64_t data_bytes;       /* Utf8: number of value bytes; else ignored */
} bhip_column_desc;

bhip_status bhip_batch_from_host(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* cols, int64_t n_rows,
                                 bhip_batch** out);
bhip_status bhip_batch_from_device(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* cols, int64_t n_rows,
                                   bhip_batch** out);
/* Scan leaf for TPC-H `.tbl` text — where the reference has CsvExec(delimiter '|', no header, explicit schema):
 * rust/benchmarks/tpch/src/main.rs:129-150, rust/core/src/serde/physical_plan/from_proto.rs:93-110.
 * `text` (host memory, < 4 GiB, whole lines) is copied to the device once; lines, fields and values are found
 * there.  fields[i].name / .dtype / .nullable describe the file's fields in order (data pointers unused);
 * `projection` = indices of the fields to materialise, in output order (NULL: all).  Int32, Int64, Float64
 * ([-]digits[.digits], converted exactly), Date32 (YYYY-MM-DD) and Utf8 columns.  Malformed text -> BHIP_EEXEC;
 * decimals beyond 15 significant digits or other column types -> BHIP_ENOTIMPL (keep the CPU reader). */
bhip_status bhip_batch_from_tbl(bhip_ctx* ctx, const void* text, int64_t n_bytes, int32_t n_fields,
                                const bhip_column_desc* fields, int32_t n_projection, const int32_t* projection,
                                bhip_batch** out);
/* Arrow C Data Interface: `array` is a struct array (one child per column) as produced by
 * RecordBatch export; it is consumed (released) on success. */
bhip_status bhip_batch_import_arrow(bhip_ctx* ctx, struct ArrowArray* array, struct ArrowSchema* schema,
                                    bhip_batch** out);
/* copies the batch to host memory and exports it; the caller releases out_array/out_schema. */
bhip_status bhip_batch_export_arrow(bhip_batch* batch, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
void bhip_batch_retain(bhip_batch* batch);
void bhip_batch_release(bhip_batch* batch);
int64_t bhip_batch_num_rows(const bhip_batch* batch);
int32_t bhip_batch_num_columns(const bhip_batch* batch);
/* column metadata; `name` stays valid while the batch lives */
bhip_status bhip_batch_column_info(const bhip_batch* batch, int32_t i, const char** name, int32_t* dtype,
                                   int32_t* nullable, int64_t* data_bytes, int32_t* has_validity);
/* raw device pointers of column i (for zero-copy consumers, e.g. an RCCL exchange) */
bhip_status bhip_batch_column_device(const bhip_batch* batch, int32_t i, const void** data, const int32_t** offsets,
                                     const uint8_t** validity);
/* copy column i to caller-provided host buffers (any of the three may be NULL) */
bhip_status bhip_batch_column_to_host(const bhip_batch* batch, int32_t i, void* data, int32_t* offsets, uint8_t* validity);
/* ArrayRef::get_array_memory_size summed over columns — the num_bytes of PartitionStats
 * (rust/core/src/utils.rs:60-83) */
int64_t bhip_batch_memory_size(const bhip_batch* batch);

/* ---- physical expressions ---------------------------------------------------------------
 * A flat POSTFIX program of nodes mirroring the expression kinds the physical-plan serde can
 * ship (rust/core/src/serde/physical_plan/to_proto.rs:380-511; LogicalExprNode,
 * rust/core/proto/ballista.proto:14-45).  Binary operators are named by the wire strings of
 * rust/core/src/serde/logical_plan/from_proto.rs:937-957 ("And" "Or" "Eq" "NotEq" "LtEq" "Lt"
 * "Gt" "GtEq" "Plus" "Minus" "Multiply" "Divide" "Like" "NotLike"). */
typedef enum {
    BHIP_EXPR_COLUMN = 1,      /* name                                  pushes 1            */
    BHIP_EXPR_LITERAL = 2,     /* dtype + value (is_null: typed NULL)   pushes 1            */
    BHIP_EXPR_BINARY = 3,      /* op ; pops right, left                                     */
    BHIP_EXPR_CAST = 4,        /* dtype ; pops 1                                            */
    BHIP_EXPR_NOT = 5,
    BHIP_EXPR_IS_NULL = 6,
    BHIP_EXPR_IS_NOT_NULL = 7,
    BHIP_EXPR_NEGATIVE = 8,
    BHIP_EXPR_IN_LIST = 9,     /* n_args list items + 1 ; negated       pops n_args+1       */
    BHIP_EXPR_CASE = 10,       /* n_args = #when/then pairs; flags bit0: has base expr,
                                  bit1: has else.  Stack order: [base] w1 t1 ... wn tn [else] */
    BHIP_EXPR_SCALAR_FN = 11   /* name = function ("sqrt", "abs", ...) ; n_args             */
} bhip_expr_kind;

typedef struct {
    int32_t kind;            /* bhip_expr_kind */
    int32_t dtype;           /* LITERAL / CAST */
    int32_t n_args;
    int32_t flags;           /* LITERAL: bit0 = is_null ; IN_LIST: bit0 = negated ; CASE: see above */
    const char* name;        /* COLUMN name / BINARY operator / SCALAR_FN name / Utf8 literal */
    int64_t i64;             /* integer, Date32, Boolean literal */
    double f64;              /* Float64 literal */
} bhip_expr_node;

typedef struct {
    const bhip_expr_node* nodes;
    int32_t n_nodes;
} bhip_expr;

/* AggregateExpr (to_proto.rs:348-378: Sum / Avg / Count are what the serde ships; Min / Max are
 * accepted too) */
typedef enum { BHIP_AGG_SUM = 1, BHIP_AGG_AVG = 2, BHIP_AGG_COUNT = 3, BHIP_AGG_MIN = 4, BHIP_AGG_MAX = 5 } bhip_agg_fn;
typedef struct {
    int32_t fn;              /* bhip_agg_fn */
    bhip_expr arg;
    const char* name;        /* output field name */
} bhip_aggregate;

/* PhysicalSortExpr (ballista.proto PhysicalSortExprNode) */
typedef struct {
    bhip_expr expr;
    int32_t descending;
    int32_t nulls_first;
} bhip_sort_expr;

typedef enum { BHIP_AGG_PARTIAL = 0, BHIP_AGG_FINAL = 1 } bhip_agg_mode;       /* from_proto.rs:181-184 */
typedef enum { BHIP_JOIN_INNER = 0, BHIP_JOIN_LEFT = 1, BHIP_JOIN_RIGHT = 2 } bhip_join_type; /* :268-272 */
typedef enum {                                                                /* from_proto.rs:143-158 */
    BHIP_PART_UNKNOWN = 0, BHIP_PART_ROUND_ROBIN = 1, BHIP_PART_HASH = 2
} bhip_partitioning;

/* ---- plans: one constructor per operator the serde can build (from_proto.rs line cited) ----
 * Constructors take a reference on their inputs; the caller still releases its own. */
/* leaf: in-memory partitions (datafusion MemoryExec; stands in for CsvExec/ParquetExec
 * :93-121 and ShuffleReaderExec :277-286 whose decoding stays on the host side).
 * batches[offsets[p] .. offsets[p+1]) are the batches of partition p. */
bhip_status bhip_plan_memory(bhip_ctx* ctx, int32_t n_partitions, const int32_t* offsets, bhip_batch* const* batches,
                             bhip_plan** out);
/* EmptyExec :287-290 — schema given as column descs with NULL data */
/* leaf over a host-side Arrow C stream — the C image of a CPU child operator's RecordBatchStream
 * (rust/core/src/memory_stream.rs:57-92): the stream is MOVED into the plan, drained on the first execute (each batch
 * imported to the device) and replayed on later executes.  One output partition. */
bhip_status bhip_plan_arrow_stream(bhip_ctx* ctx, struct ArrowArrayStream* stream, bhip_plan** out);
/* the same with one stream per output partition of the CPU child (all moved into the plan): partition p of the leaf is
 * streams[p]; a join above drains every partition of its build side, as the reference's collect-left join does */
bhip_status bhip_plan_arrow_streams(bhip_ctx* ctx, int32_t n_partitions, struct ArrowArrayStream* const* streams, bhip_plan** out);
/* ParquetExec::try_from_files(filenames, projection, None, batch_size, num_partitions)  :111-121 — the files are dealt out to
 * `num_partitions` output partitions in chunks; one batch per row group.  projection NULL: every column.  Pages are located
 * and decompressed (Snappy) on the host, values are decoded on the device (dictionary runs, NULL re-insertion, gathers).
 * Flat schemas of INT32 / INT64 / DOUBLE / BOOLEAN / BYTE_ARRAY (+ DATE), PLAIN and RLE_DICTIONARY, data pages V1 / V2,
 * UNCOMPRESSED / SNAPPY; anything else is BHIP_ENOTIMPL. */
bhip_status bhip_plan_parquet(bhip_ctx* ctx, int32_t n_files, const char* const* paths, int32_t n_projection, const uint32_t* projection,
                              int32_t num_partitions, bhip_plan** out);
bhip_status bhip_plan_empty(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* schema, int32_t produce_one_row,
                            bhip_plan** out);
bhip_status bhip_plan_filter(bhip_plan* input, const bhip_expr* predicate, bhip_plan** out);          /* :81-92  */
bhip_status bhip_plan_projection(bhip_plan* input, int32_t n, const bhip_expr* exprs, const char* const* names,
                                 bhip_plan** out);                                                   /* :69-80  */
bhip_status bhip_plan_hash_aggregate(bhip_plan* input, int32_t mode, int32_t n_group, const bhip_expr* group_exprs,
                                     const char* const* group_names, int32_t n_aggr, const bhip_aggregate* aggr,
                                     bhip_plan** out);                                               /* :173-252 */
bhip_status bhip_plan_hash_join(bhip_plan* left, bhip_plan* right, int32_t n_on, const char* const* left_keys,
                                const char* const* right_keys, int32_t join_type, bhip_plan** out);  /* :253-276 */
bhip_status bhip_plan_sort(bhip_plan* input, int32_t n, const bhip_sort_expr* exprs, bhip_plan** out); /* :291-331 */
bhip_status bhip_plan_repartition(bhip_plan* input, int32_t scheme, int32_t n_exprs, const bhip_expr* hash_exprs,
                                  int32_t partition_count, bhip_plan** out);                         /* :133-164 */
bhip_status bhip_plan_coalesce_batches(bhip_plan* input, int64_t target_batch_size, bhip_plan** out);  /* :122-128 */
bhip_status bhip_plan_merge(bhip_plan* input, bhip_plan** out);                                       /* :129-132 */
bhip_status bhip_plan_global_limit(bhip_plan* input, int64_t limit, bhip_plan** out);                 /* :165-168 */
bhip_status bhip_plan_local_limit(bhip_plan* input, int64_t limit, bhip_plan** out);                  /* :169-172 */

/* ---- the wire plan ------------------------------------------------------------------------------
 * bhip_plan_from_proto: `impl TryInto<Arc<dyn ExecutionPlan>> for &protobuf::PhysicalPlanNode`
 * (rust/core/src/serde/physical_plan/from_proto.rs:58-346): `bytes` is the protobuf encoding of a PhysicalPlanNode
 * (rust/core/proto/ballista.proto:294-312) exactly as a task carries it (TaskDefinition.plan :531-534,
 * ExecutePartition.plan :451-458).  Expressions are planned against their input schema with DataFusion's coercion
 * rules (compile_expr, :348-364).  Every leaf (CsvScan / ParquetScan / ShuffleReader / UnresolvedShuffle) is offered
 * to `resolve` (may be NULL): it returns BHIP_OK with *out = a plan that produces the leaf's rows (a bhip_plan_memory, a
 * bhip_plan_arrow_stream over a CPU reader ...; ownership of that handle passes to the library), or BHIP_OK with *out = NULL to leave the leaf to the library:
 * a CsvScan over '|'-separated header-less local files becomes the device `.tbl` scan (bhip_batch_from_tbl), any
 * other leaf an operator that describes itself and fails on execute with BHIP_EEXEC
 * (UnresolvedShuffleExec::execute, rust/core/src/execution_plans/unresolved_shuffle.rs:83-90).
 * `ctx` may be NULL when every leaf is left unresolved: the plan can then be inspected (bhip_plan_display, _schema,
 * _children) but not executed — what the CPU-only tests use. */
typedef enum {
    BHIP_LEAF_CSV_SCAN = 1, BHIP_LEAF_PARQUET_SCAN = 2, BHIP_LEAF_SHUFFLE_READER = 3, BHIP_LEAF_UNRESOLVED_SHUFFLE = 4
} bhip_leaf_kind;
typedef struct {                 /* PartitionLocation, ballista.proto:461-465 */
    const char* job_id;
    uint32_t stage_id, partition_id;
    const char* executor_id;
    const char* host;
    uint32_t port;
    int64_t num_rows, num_batches, num_bytes;       /* PartitionStats; -1 = not sent */
} bhip_partition_location;
typedef struct {
    int32_t kind;                                   /* bhip_leaf_kind */
    const char* path;                               /* CsvScan */
    int32_t n_filenames;
    const char* const* filenames;                   /* CsvScan partition files / ParquetScan files */
    int32_t has_projection, n_projection;
    const uint32_t* projection;                     /* indices into `fields` (CsvScan) / the file schema (ParquetScan) */
    int32_t n_fields;
    const bhip_column_desc* fields;                 /* CsvScan: the FILE's fields; shuffle leaves: the output schema */
    int32_t has_header;
    const char* delimiter;
    const char* file_extension;
    uint32_t batch_size, num_partitions;
    int32_t n_locations;
    const bhip_partition_location* locations;       /* ShuffleReader */
    int32_t n_stage_ids;
    const uint32_t* stage_ids;                      /* UnresolvedShuffle */
    uint32_t partition_count;
} bhip_leaf_desc;
typedef bhip_status (*bhip_leaf_resolver)(void* user, const bhip_leaf_desc* leaf, bhip_plan** out);
bhip_status bhip_plan_from_proto(bhip_ctx* ctx, const void* bytes, size_t len, bhip_leaf_resolver resolve, void* user,
                                 bhip_plan** out);
/* one LogicalExprNode (ballista.proto:14-45) rendered the way plan displays render expressions; for tools and tests */
bhip_status bhip_expr_from_proto_display(const void* bytes, size_t len, char* buf, size_t cap);

void bhip_plan_retain(bhip_plan* plan);
void bhip_plan_release(bhip_plan* plan);

/* ---- ExecutionPlan trait (query_stage.rs:49-85) ---------------------------------------------- */
/* as_any(): the operator's type name, e.g. "HashAggregateExec" */
const char* bhip_plan_name(const bhip_plan* plan);
/* schema(): fills up to `cap` entries, *n_cols = total number */
bhip_status bhip_plan_schema(const bhip_plan* plan, int32_t cap, const char** names, int32_t* dtypes,
                             int32_t* nullable, int32_t* n_cols);
/* output_partitioning() */
bhip_status bhip_plan_output_partitioning(const bhip_plan* plan, int32_t* scheme, int32_t* partition_count);
/* children(): NEW handles (the caller releases each) */
bhip_status bhip_plan_children(const bhip_plan* plan, int32_t cap, bhip_plan** children, int32_t* n_children);
/* with_new_children() */
bhip_status bhip_plan_with_new_children(const bhip_plan* plan, int32_t n, bhip_plan* const* children, bhip_plan** out);
/* async execute(partition) -> stream.  Blocking (wrap in spawn_blocking); re-entrant. */
bhip_status bhip_plan_execute(bhip_plan* plan, int32_t partition, bhip_stream** out);
/* Debug-style one-line-per-operator rendering (utils.rs:96-188 pretty printer) */
bhip_status bhip_plan_display(const bhip_plan* plan, char* buf, size_t cap);

/* ---- RecordBatchStream (memory_stream.rs:57-92) ------------------------------------------- */
/* stream.next(): *out = NULL at end of stream */
bhip_status bhip_stream_next(bhip_stream* stream, bhip_batch** out);
bhip_status bhip_stream_schema(const bhip_stream* stream, int32_t cap, const char** names, int32_t* dtypes,
                               int32_t* nullable, int32_t* n_cols);
void bhip_stream_release(bhip_stream* stream);
/* datafusion::physical_plan::collect(plan): executes every output partition in turn and returns all batches, in partition order
 * (the client-side `collect` of rust/core/src/execution_plans/... / SURVEY.md §8 a12).  At most `cap` batches are returned
 * (BHIP_EINVAL if the plan yields more); each one is released with bhip_batch_release. */
bhip_status bhip_plan_collect(bhip_plan* plan, int32_t cap, bhip_batch** out, int32_t* n_out);

/* hands the stream to an Arrow C Stream consumer (batches are copied to host as they are pulled);
 * the bhip_stream is consumed. */
bhip_status bhip_stream_export_arrow(bhip_stream* stream, struct ArrowArrayStream* out);
/* drains a stream like utils::write_stream_to_disk (rust/core/src/utils.rs:49-84) and reports
 * PartitionStats{num_rows,num_batches,num_bytes}; batches are handed to `sink` (may be NULL). */
typedef bhip_status (*bhip_batch_sink)(void* user, bhip_batch* batch);
bhip_status bhip_stream_drain(bhip_stream* stream, bhip_batch_sink sink, void* user, uint64_t* num_rows,
                              uint64_t* num_batches, uint64_t* num_bytes);

/* ---- Arrow IPC files: the two sides of a stage boundary ------------------------------------------------------
 * bhip_stream_write_ipc = utils::write_stream_to_disk (rust/core/src/utils.rs:49-84): drains the stream into an Arrow IPC
 * FILE at `path` (what the executor serves to the next stage, rust/executor/src/flight_service.rs:104-150) and reports
 * PartitionStats.  The stream is CONSUMED on every path, success or error (a NULL `stream` aside): the caller must NOT call
 * bhip_stream_release on it afterwards.
 * bhip_plan_ipc_files: a leaf over such files, one output partition per file — the local half of ShuffleReaderExec
 * (rust/core/src/execution_plans/shuffle_reader.rs:77-99).  Files written by arrow-rs / pyarrow are read as well
 * (metadata V4 or V5, no compression, no dictionaries; other files -> BHIP_ENOTIMPL).
 * bhip_ipc_write_file / bhip_ipc_open_file are the same on host-side Arrow C streams (no GPU involved). */
bhip_status bhip_stream_write_ipc(bhip_stream* stream, const char* path, uint64_t* num_rows, uint64_t* num_batches,
                                  uint64_t* num_bytes);
bhip_status bhip_plan_ipc_files(bhip_ctx* ctx, int32_t n_files, const char* const* paths, bhip_plan** out);
bhip_status bhip_ipc_write_file(struct ArrowArrayStream* stream, const char* path, uint64_t* num_rows, uint64_t* num_batches,
                                uint64_t* num_bytes);
bhip_status bhip_ipc_open_file(const char* path, struct ArrowArrayStream* out);

/* ---- hash repartition exchange support (RepartitionExec(Hash) across GPUs) -----------------
 * The per-row partition id is  bhip_row_hash(key columns) % n  (DESIGN.md "Row hash").  These
 * two calls split a device batch into n device batches (partition p = rows whose id is p, input
 * order kept) so the caller can exchange them (RCCL all-to-all) — used by the multi-GPU path. */
bhip_status bhip_batch_hash_partition(bhip_batch* batch, int32_t n_exprs, const bhip_expr* hash_exprs, int32_t n,
                                      bhip_batch** out /* n handles */);
bhip_status bhip_batch_concat(bhip_ctx* ctx, int32_t n, bhip_batch* const* batches, bhip_batch** out);

/* ---- exchange between the GPUs of one node (RCCL over xGMI, inside the library) -----------------------------
 * What a stage boundary is in the reference — partitions written as IPC files (rust/core/src/utils.rs:49-84) and pulled by
 * ShuffleReaderExec over Flight (rust/core/src/execution_plans/shuffle_reader.rs:77-99) — between processes that each
 * own one GPU of the node: batches move device to device.  One communicator per process and GPU; the 128-byte id made
 * by rank 0 (bhip_comm_unique_id) reaches the other ranks by whatever channel they share.  Collective calls: every rank
 * of the communicator must make the same call, with batches of ONE schema.  librccl is loaded at the first call. */
#define BHIP_COMM_ID_BYTES 128
typedef struct bhip_comm bhip_comm;
bhip_status bhip_comm_unique_id(uint8_t* id /* BHIP_COMM_ID_BYTES */);
bhip_status bhip_comm_create(bhip_ctx* ctx, const uint8_t* id, int32_t world, int32_t rank, bhip_comm** out);
/* Everything above the byte movers (block layout, header / count matrices, who sends what to whom, the streaming shuffle)
 * is one piece of code over a two-call transport; RCCL is one of three:
 * bhip_comm_create_loopback: `world` communicators inside ONE process on one device (any 128 bytes as the id; an id serves
 * one world once), each driven by its own thread; regions move with device-to-device copies after a host rendezvous.  It runs
 * the N-rank code paths on a single-GPU box (tests).
 * bhip_comm_create_host: the caller moves HOST bytes through two callbacks (0 = ok): an executor that only has its
 * Flight / TCP channel between processes, or a rehearsal over gloo.  Regions of one peer pair are matched in order. */
bhip_status bhip_comm_create_loopback(bhip_ctx* ctx, const uint8_t* id, int32_t world, int32_t rank, bhip_comm** out);
typedef struct bhip_comm_region { void* ptr; uint64_t bytes; int32_t peer; } bhip_comm_region;
typedef struct bhip_comm_host_transport {
    void* user;
    /* recv[r * bytes, (r + 1) * bytes) <- rank r's send[0, bytes) */
    int32_t (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes);
    /* one grouped exchange: every send reaches the matching receive of its peer */
    int32_t (*exchange)(void* user, int32_t n_sends, const bhip_comm_region* sends, int32_t n_recvs, const bhip_comm_region* recvs);
} bhip_comm_host_transport;
bhip_status bhip_comm_create_host(bhip_ctx* ctx, const bhip_comm_host_transport* transport, int32_t world, int32_t rank, bhip_comm** out);
void bhip_comm_release(bhip_comm* comm);
/* world, rank and the transport's name ("rccl" | "loopback" | "host"; valid while the communicator lives) */
bhip_status bhip_comm_info(bhip_comm* comm, int32_t* world, int32_t* rank, const char** transport);
/* wall seconds inside collective calls, bytes this rank sent to other ranks, calls — since the last reset */
bhip_status bhip_comm_stats(bhip_comm* comm, int32_t reset, double* seconds, uint64_t* bytes_out, uint64_t* calls);
/* the MergeExec side of a stage boundary: out[r] = rank r's batch, r = 0 .. world-1 (out[rank] is `mine`); the caller
 * releases every handle.  Small batches (partial aggregate states) travel in ONE ncclAllGather. */
bhip_status bhip_comm_all_gather(bhip_comm* comm, bhip_batch* mine, bhip_batch** out /* world handles */);
/* the shuffle of RepartitionExec(Hash(keys), world): parts[d] (bhip_batch_hash_partition) goes to rank d;
 * out[s] = what rank s held for this rank.  One grouped ncclSend / ncclRecv per peer pair. */
bhip_status bhip_comm_all_to_all(bhip_comm* comm, bhip_batch* const* parts /* world */, bhip_batch** out /* world handles */);
/* RepartitionExec(Hash([key_column], world)) (rust/core/src/serde/physical_plan/from_proto.rs:133-147) + the shuffle read
 * (rust/core/src/execution_plans/shuffle_reader.rs:77-99) in one collective call: *out = the rows of every rank's `batch`
 * whose key hashes to this rank, in source-rank order, input order within a source.  Fixed-width NULL-free columns with a
 * NULL-free integer key STREAM in chunks of `chunk_rows` rows (<= 0: 64 Mi): a count pass sizes the result exactly, a peer's
 * rows land at their final position, device memory beyond input and result is two chunks.  Anything else goes through
 * bhip_batch_hash_partition + bhip_comm_all_to_all + concat inside the call.  stats may be NULL. */
#define BHIP_SHUFFLE_MAX_PEERS 64
typedef struct bhip_shuffle_stats {
    uint64_t rows_in, rows_out, chunks, streamed;       /* streamed: 1 = the chunked path ran */
    uint64_t bytes_sent_remote, bytes_kept_local, staging_bytes;
    uint64_t rows_to[BHIP_SHUFFLE_MAX_PEERS];           /* rows of `batch` that went to each rank (the xGMI numerator) */
    double ms_count, ms_total;                          /* the count pass; the whole call (host wall clock) */
} bhip_shuffle_stats;
bhip_status bhip_comm_shuffle(bhip_comm* comm, bhip_batch* batch, const char* key_column, int64_t chunk_rows, bhip_batch** out,
                              bhip_shuffle_stats* stats);
/* The exchange as plan nodes, so that a rank's whole distributed query is ONE operator tree (one bhip_plan_collect per step).
 * AllGatherExec: executes every partition of `input`, all_gathers the result; `world` output partitions, partition r = rank
 * r's batch (what MergeExec over the previous stage's partitions reads, rust/scheduler/src/planner.rs:136-171).
 * ShuffleExchangeExec: executes every partition of `input`, bhip_comm_shuffle by `key_column`; one output partition.
 * The plan keeps the communicator alive.  Every rank must execute the same exchange nodes in the same order. */
bhip_status bhip_plan_all_gather(bhip_comm* comm, bhip_plan* input, bhip_plan** out);
bhip_status bhip_plan_shuffle(bhip_comm* comm, bhip_plan* input, const char* key_column, int64_t chunk_rows, bhip_plan** out);
/* The block form the exchange moves, for transports other than RCCL: header = 2 + 3 * columns int64 words (rows, block
 * bytes, then per column: data bytes, has offsets, has validity); the block holds every buffer, 64-byte aligned.
 * host_block NULL: only the header and *block_bytes are produced.  bhip_batch_unpack is the inverse (one copy to the
 * device; the columns are slices of it). */
bhip_status bhip_batch_pack(bhip_batch* batch, int64_t* header, int32_t header_cap, void* host_block, int64_t block_cap,
                            int64_t* block_bytes);
bhip_status bhip_batch_unpack(bhip_ctx* ctx, int32_t n_cols, const bhip_column_desc* schema, const int64_t* header,
                              const void* host_block, bhip_batch** out);

/* ---- synthetic TPC-H data (bench / tests) -------------------------------------------------- */
/* lineitem rows [row0,row0+n) at scale factor `sf` generated on the device; key64: Int64 order keys
 * (SF1000).  Columns: l_orderkey l_suppkey l_quantity l_extendedprice l_discount l_tax
 * l_returnflag l_linestatus l_shipdate [l_commitdate l_receiptdate when with_dates]. */
bhip_status bhip_tpch_lineitem(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, int32_t key64,
                               int32_t with_dates, bhip_batch** out);
bhip_status bhip_tpch_orders(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, int32_t key64,
                             bhip_batch** out);
/* The same with options.  sparse_keys: dbgen's order-key layout (the low 3 bits of the order number kept, the rest shifted up
 * by two: 8 of every 32 key values used — SF1000 keys reach 6 x 10^9 and need Int64); key_base is added to every order key
 * (small tables with keys beyond 2^32); columns: only these (NULL = all), in the table's column order. */
typedef struct bhip_tpch_opts {
    int32_t key64, with_dates, sparse_keys, n_columns;
    int64_t key_base;
    const char* const* columns;
} bhip_tpch_opts;
bhip_status bhip_tpch_lineitem_opts(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, const bhip_tpch_opts* opts,
                                    bhip_batch** out);
bhip_status bhip_tpch_orders_opts(bhip_ctx* ctx, double sf, uint64_t seed, uint64_t row0, uint64_t n, const bhip_tpch_opts* opts,
                                  bhip_batch** out);

/* ---- measurement hooks --------------------------------------------------------------------- */
/* time (ms, HIP events on the stream the kernels ran on) and launch count of the dominant scan
 * kernel accumulated on this context since the last reset */
bhip_status bhip_ctx_kernel_time(bhip_ctx* ctx, int32_t reset, double* ms, uint64_t* launches);
/* every kernel timed since the last reset (BHIP_KERNEL_TIMING=1), one line per kernel: "name\tms\tlaunches\talgorithmic bytes\n" (bytes: 0 where the call site does not state them)
 * (NUL terminated; BHIP_EINVAL when `cap` is too small).  Waits for the timed launches to finish. */
bhip_status bhip_ctx_kernel_stats(bhip_ctx* ctx, int32_t reset, char* buf, size_t cap);
/* name of the kernel those launches ran ("" before the first timed launch); valid until the next call */
const char* bhip_ctx_kernel_name(bhip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* BALLISTA_HIP_H */
