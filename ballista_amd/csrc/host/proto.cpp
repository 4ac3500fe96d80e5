// proto.cpp — the executor's wire plan (protobuf `PhysicalPlanNode`) -> operator tree.
//
// Reference: `impl TryInto<Arc<dyn ExecutionPlan>> for &protobuf::PhysicalPlanNode`
// (rust/core/src/serde/physical_plan/from_proto.rs:58-346) and `compile_expr` (:348-364), which turns each
// `LogicalExprNode` into a logical `Expr` (rust/core/src/serde/logical_plan/from_proto.rs:777-957) and then plans it
// against the input schema with DataFusion's `create_physical_expr` — which is where the numeric coercion casts of
// `coerce_expr` below come from.  Messages and field numbers: rust/core/proto/ballista.proto
// (`PhysicalPlanNode` :294-422, `LogicalExprNode` :14-161, `ScalarValue` :685-709, `Schema`/`Field`/`ArrowType` :611-800).
//
// The reader is a hand-rolled proto3 wire parser (varint / 64-bit / length-delimited / 32-bit); unknown fields are skipped as
// proto3 requires.  Leaves (scans, shuffle readers) have no in-library data source by themselves: a resolver callback maps
// each to a plan (an Arrow C stream of a CPU reader, a MemoryExec ...); without a resolver a CsvScan over '|'-separated files
// becomes the library's own device `.tbl` scan and the others stay `UnresolvedLeafExec`s, which describe themselves and fail
// on execute the way `UnresolvedShuffleExec::execute` does (rust/core/src/execution_plans/unresolved_shuffle.rs:83-90).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "plan.hpp"

namespace bhip {

// ---- proto3 wire reader ----------------------------------------------------------------------------------------------
namespace {

struct Pb {
    const uint8_t* p;
    const uint8_t* end;
    Pb(const void* b, size_t n) : p((const uint8_t*)b), end((const uint8_t*)b + n) {}
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p >= end) fail(BHIP_EINVAL, "protobuf: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << (shift < 64 ? shift : 63);
            if (!(b & 0x80)) return v;
        }
        fail(BHIP_EINVAL, "protobuf: varint longer than 10 bytes");
    }
    // next field header; false at end of message
    bool next(uint32_t& field, uint32_t& wt) {
        if (p >= end) return false;
        const uint64_t t = varint();
        field = (uint32_t)(t >> 3);
        wt = (uint32_t)(t & 7);
        if (field == 0) fail(BHIP_EINVAL, "protobuf: field number 0");
        return true;
    }
    Pb sub() {
        const uint64_t n = varint();
        if (n > (uint64_t)(end - p)) fail(BHIP_EINVAL, "protobuf: length-delimited field runs past the end of the message");
        Pb s(p, (size_t)n);
        p += n;
        return s;
    }
    std::string str() {
        Pb s = sub();
        return std::string((const char*)s.p, (size_t)(s.end - s.p));
    }
    uint64_t fixed64() {
        if (end - p < 8) fail(BHIP_EINVAL, "protobuf: truncated 64-bit field");
        uint64_t v;
        memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    uint32_t fixed32() {
        if (end - p < 4) fail(BHIP_EINVAL, "protobuf: truncated 32-bit field");
        uint32_t v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    void skip(uint32_t wt) {
        switch (wt) {
            case 0: varint(); break;
            case 1: fixed64(); break;
            case 2: sub(); break;
            case 5: fixed32(); break;
            default: fail(BHIP_EINVAL, "protobuf: unsupported wire type " + std::to_string(wt));
        }
    }
    void expect(uint32_t wt, uint32_t want, const char* what) {
        if (wt != want) fail(BHIP_EINVAL, std::string("protobuf: wrong wire type for ") + what);
    }
    // repeated uint32, packed or not
    void repeated_u32(uint32_t wt, std::vector<uint32_t>& out) {
        if (wt == 2) {
            Pb s = sub();
            while (!s.done()) out.push_back((uint32_t)s.varint());
        } else {
            expect(wt, 0, "repeated uint32");
            out.push_back((uint32_t)varint());
        }
    }
};

// ---- Arrow types ---------------------------------------------------------------------------------------------------------
// ArrowType oneof (ballista.proto:755-790) -> bhip_dtype; 0 = a type outside the GPU path (reported with its name)
int decode_arrow_type(Pb r, std::string* name_out) {
    uint32_t f, wt;
    int dt = 0;
    std::string name = "NONE";
    while (r.next(f, wt)) {
        switch (f) {
            case 2: dt = DT_BOOLEAN; name = "Boolean"; break;
            case 3: dt = DT_UINT8; name = "UInt8"; break;
            case 4: dt = DT_INT8; name = "Int8"; break;
            case 5: dt = DT_UINT16; name = "UInt16"; break;
            case 6: dt = DT_INT16; name = "Int16"; break;
            case 7: dt = DT_UINT32; name = "UInt32"; break;
            case 8: dt = DT_INT32; name = "Int32"; break;
            case 9: dt = DT_UINT64; name = "UInt64"; break;
            case 10: dt = DT_INT64; name = "Int64"; break;
            case 11: name = "Float16"; break;
            case 12: dt = DT_FLOAT32; name = "Float32"; break;
            case 13: dt = DT_FLOAT64; name = "Float64"; break;
            case 14: dt = DT_UTF8; name = "Utf8"; break;
            case 32: dt = DT_LARGE_UTF8; name = "LargeUtf8"; break;      // schemas only: decode_field turns it into Utf8 + Field::large
            case 15: dt = DT_BINARY; name = "Binary"; break;               // schemas only, as LargeUtf8
            case 17: dt = DT_DATE32; name = "Date32"; break;
            case 18: dt = DT_DATE64; name = "Date64"; break;
            case 20: {                                 // Timestamp{time_unit = 1, timezone = 2}
                Pb t = r.sub();
                uint64_t unit = 0;
                std::string tz;
                uint32_t tf, twt;
                while (t.next(tf, twt)) {
                    if (tf == 1 && twt == 0) unit = t.varint();
                    else if (tf == 2 && twt == 2) tz = t.str();
                    else t.skip(twt);
                }
                name = "Timestamp";
                if (unit <= 3 && tz.empty()) dt = DT_TIMESTAMP_S + (int)unit;
                else name = "Timestamp with a time zone";
                continue;
            }
            default: name = "ArrowType#" + std::to_string(f); break;
        }
        r.skip(wt);
    }
    if (name_out) *name_out = name;
    return dt;
}

Field decode_field(Pb r) {
    Field fld{"", 0, false};
    uint32_t f, wt;
    std::string tname = "NONE";
    while (r.next(f, wt)) {
        if (f == 1 && wt == 2) fld.name = r.str();
        else if (f == 2 && wt == 2) fld.dtype = decode_arrow_type(r.sub(), &tname);
        else if (f == 3 && wt == 0) fld.nullable = r.varint() != 0;
        else r.skip(wt);
    }
    if (!fld.dtype) fail(BHIP_ENOTIMPL, "field '" + fld.name + "' has type " + tname + ", which the GPU path does not carry");
    if (fld.dtype == DT_LARGE_UTF8) { fld.dtype = DT_UTF8; fld.large = true; }
    if (fld.dtype == DT_BINARY) { fld.dtype = DT_UTF8; fld.binary = true; }
    return fld;
}

SchemaPtr decode_schema(Pb r) {
    auto s = std::make_shared<Schema>();
    uint32_t f, wt;
    while (r.next(f, wt)) {
        if (f == 1 && wt == 2) s->fields.push_back(decode_field(r.sub()));
        else r.skip(wt);
    }
    return s;
}

// ---- expressions -----------------------------------------------------------------------------------------------------------
ExprPtr decode_expr(Pb r);

[[maybe_unused]] ExprPtr new_literal(int dtype) {
    auto e = std::make_shared<Expr>();
    e->kind = BHIP_EXPR_LITERAL;
    e->dtype = dtype;
    return e;
}

// PrimitiveScalarType (ballista.proto:712-733) of a typed NULL
int primitive_scalar_type(uint64_t v) {
    switch (v) {
        case 0: return DT_BOOLEAN;
        case 1: return DT_UINT8;
        case 2: return DT_INT8;
        case 3: return DT_UINT16;
        case 4: return DT_INT16;
        case 5: return DT_UINT32;
        case 6: return DT_INT32;
        case 7: return DT_UINT64;
        case 8: return DT_INT64;
        case 9: return DT_FLOAT32;
        case 10: return DT_FLOAT64;
        case 11: return DT_UTF8;
        case 13: return DT_DATE32;
        default: fail(BHIP_ENOTIMPL, "NULL literal of scalar type #" + std::to_string(v));
    }
}

ExprPtr decode_scalar_value(Pb r) {
    uint32_t f, wt;
    std::shared_ptr<Expr> out;
    while (r.next(f, wt)) {
        auto lit = [&](int dt) { auto e = std::make_shared<Expr>(); e->kind = BHIP_EXPR_LITERAL; e->dtype = dt; out = e; return e; };
        switch (f) {
            case 1: lit(DT_BOOLEAN)->i64 = r.varint() != 0; break;
            case 2: lit(DT_UTF8)->name = r.str(); break;
            case 3: fail(BHIP_ENOTIMPL, "LargeUtf8 literal");
            case 4: lit(DT_INT8)->i64 = (int64_t)(int32_t)r.varint(); break;
            case 5: lit(DT_INT16)->i64 = (int64_t)(int32_t)r.varint(); break;
            case 6: lit(DT_INT32)->i64 = (int64_t)(int32_t)r.varint(); break;
            case 7: lit(DT_INT64)->i64 = (int64_t)r.varint(); break;
            case 8: lit(DT_UINT8)->i64 = (int64_t)r.varint(); break;
            case 9: lit(DT_UINT16)->i64 = (int64_t)r.varint(); break;
            case 10: lit(DT_UINT32)->i64 = (int64_t)r.varint(); break;
            case 11: lit(DT_UINT64)->i64 = (int64_t)r.varint(); break;
            case 12: { const uint32_t b = r.fixed32(); float v; memcpy(&v, &b, 4); auto e = lit(DT_FLOAT32); e->f64 = v; } break;
            case 13: { const uint64_t b = r.fixed64(); auto e = lit(DT_FLOAT64); memcpy(&e->f64, &b, 8); } break;
            case 14: lit(DT_DATE32)->i64 = (int64_t)(int32_t)r.varint(); break;
            case 19: { auto e = lit(primitive_scalar_type(r.varint())); e->is_null = true; } break;
            case 15: case 16: case 17: case 18:
                fail(BHIP_ENOTIMPL, "time / list ScalarValue (field " + std::to_string(f) + ")");
            default: r.skip(wt);
        }
    }
    if (!out) fail(BHIP_EINVAL, "protobuf: ScalarValue without a value");
    return out;
}

// ScalarFunction enum (ballista.proto:82-116) -> DataFusion's function names
const char* scalar_function_name(uint64_t v) {
    static const char* names[] = {"sqrt", "sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "log2", "log10", "floor", "ceil",
                                  "round", "trunc", "abs", "signum", "octet_length", "concat", "lower", "upper", "trim", "ltrim",
                                  "rtrim", "to_timestamp", "array", "nullif", "date_trunc", "md5", "sha224", "sha256", "sha384", "sha512"};
    if (v >= sizeof(names) / sizeof(names[0])) fail(BHIP_EINVAL, "protobuf: unknown ScalarFunction " + std::to_string(v));
    return names[v];
}

ExprPtr unary(int kind, Pb r) {      // message { LogicalExprNode expr = 1; }
    auto e = std::make_shared<Expr>();
    e->kind = kind;
    uint32_t f, wt;
    while (r.next(f, wt)) {
        if (f == 1 && wt == 2) e->args = {decode_expr(r.sub())};
        else r.skip(wt);
    }
    if (e->args.empty()) fail(BHIP_EINVAL, "protobuf: expression node without its operand");
    return e;
}

// nesting of one wire plan: expressions inside expressions, plan nodes inside plan nodes.  A message nested deeper than any
// planner writes (TPC-H's deepest expression is ~8 levels, a plan ~12) would otherwise walk the C++ stack down with it.
constexpr int MAX_NESTING = 128;
struct NestGuard {
    static int& depth() { static thread_local int d = 0; return d; }
    NestGuard() { if (++depth() > MAX_NESTING) { --depth(); fail(BHIP_EINVAL, "protobuf: message nested deeper than 128 levels"); } }
    ~NestGuard() { --depth(); }
};

ExprPtr decode_expr(Pb r) {
    NestGuard nest;
    uint32_t f, wt;
    ExprPtr out;
    while (r.next(f, wt)) {
        switch (f) {
            case 1: r.expect(wt, 2, "column_name"); out = make_column(r.str()); break;
            case 2: {                                   // AliasNode: physical expressions carry no alias, the name lives beside it
                Pb a = r.sub();
                uint32_t af, awt;
                while (a.next(af, awt)) {
                    if (af == 1 && awt == 2) out = decode_expr(a.sub());
                    else a.skip(awt);
                }
            } break;
            case 3: out = decode_scalar_value(r.sub()); break;
            case 4: {
                Pb b = r.sub();
                ExprPtr l, rr;
                std::string op;
                uint32_t bf, bwt;
                while (b.next(bf, bwt)) {
                    if (bf == 1 && bwt == 2) l = decode_expr(b.sub());
                    else if (bf == 2 && bwt == 2) rr = decode_expr(b.sub());
                    else if (bf == 3 && bwt == 2) op = b.str();
                    else b.skip(bwt);
                }
                if (!l || !rr) fail(BHIP_EINVAL, "protobuf: BinaryExprNode without both operands");
                // the operator travels as its Debug name (logical_plan/from_proto.rs:937-957)
                static const char* ops[] = {"And", "Or", "Eq", "NotEq", "LtEq", "Lt", "Gt", "GtEq", "Plus", "Minus", "Multiply",
                                            "Divide", "Like", "NotLike"};
                bool known = false;
                for (auto o : ops) known = known || op == o;
                if (!known) fail(op == "Modulus" ? BHIP_ENOTIMPL : BHIP_EINVAL, "Unsupported binary operator '" + op + "'");
                out = make_binary(l, op, rr);
            } break;
            case 5: fail(BHIP_EINVAL, "aggregate expression outside a HashAggregateExecNode");
            case 6: out = unary(BHIP_EXPR_IS_NULL, r.sub()); break;
            case 7: out = unary(BHIP_EXPR_IS_NOT_NULL, r.sub()); break;
            case 8: out = unary(BHIP_EXPR_NOT, r.sub()); break;
            case 9: {                                   // BetweenNode: x >= low AND x <= high (NOT of that when negated)
                Pb b = r.sub();
                ExprPtr x, lo, hi;
                bool negated = false;
                uint32_t bf, bwt;
                while (b.next(bf, bwt)) {
                    if (bf == 1 && bwt == 2) x = decode_expr(b.sub());
                    else if (bf == 2 && bwt == 0) negated = b.varint() != 0;
                    else if (bf == 3 && bwt == 2) lo = decode_expr(b.sub());
                    else if (bf == 4 && bwt == 2) hi = decode_expr(b.sub());
                    else b.skip(bwt);
                }
                if (!x || !lo || !hi) fail(BHIP_EINVAL, "protobuf: BetweenNode is incomplete");
                ExprPtr both = make_binary(make_binary(x, "GtEq", lo), "And", make_binary(x, "LtEq", hi));
                if (negated) {
                    auto n = std::make_shared<Expr>();
                    n->kind = BHIP_EXPR_NOT;
                    n->args = {both};
                    both = n;
                }
                out = both;
            } break;
            case 10: {
                Pb c = r.sub();
                auto e = std::make_shared<Expr>();
                e->kind = BHIP_EXPR_CASE;
                ExprPtr base, els;
                std::vector<ExprPtr> wt_pairs;
                uint32_t cf, cwt;
                while (c.next(cf, cwt)) {
                    if (cf == 1 && cwt == 2) base = decode_expr(c.sub());
                    else if (cf == 2 && cwt == 2) {
                        Pb w = c.sub();
                        ExprPtr when, then;
                        uint32_t wf, wwt;
                        while (w.next(wf, wwt)) {
                            if (wf == 1 && wwt == 2) when = decode_expr(w.sub());
                            else if (wf == 2 && wwt == 2) then = decode_expr(w.sub());
                            else w.skip(wwt);
                        }
                        if (!when || !then) fail(BHIP_EINVAL, "protobuf: WhenThen is incomplete");
                        wt_pairs.push_back(when);
                        wt_pairs.push_back(then);
                    } else if (cf == 3 && cwt == 2) els = decode_expr(c.sub());
                    else c.skip(cwt);
                }
                if (wt_pairs.empty()) fail(BHIP_EINVAL, "CASE without WHEN");
                e->has_base = (bool)base;
                e->has_else = (bool)els;
                if (base) e->args.push_back(base);
                for (auto& x : wt_pairs) e->args.push_back(x);
                if (els) e->args.push_back(els);
                out = e;
            } break;
            case 11: {
                Pb c = r.sub();
                auto e = std::make_shared<Expr>();
                e->kind = BHIP_EXPR_CAST;
                std::string tname;
                uint32_t cf, cwt;
                while (c.next(cf, cwt)) {
                    if (cf == 1 && cwt == 2) e->args = {decode_expr(c.sub())};
                    else if (cf == 2 && cwt == 2) { e->dtype = decode_arrow_type(c.sub(), &tname); if (e->dtype == DT_LARGE_UTF8 || e->dtype == DT_BINARY) e->dtype = 0; }
                    else c.skip(cwt);
                }
                if (e->args.empty()) fail(BHIP_EINVAL, "protobuf: CastNode without an operand");
                if (!e->dtype) fail(BHIP_ENOTIMPL, "cast to " + tname);
                out = e;
            } break;
            case 12: fail(BHIP_EINVAL, "sort expression outside a SortExecNode");
            case 13: out = unary(BHIP_EXPR_NEGATIVE, r.sub()); break;
            case 14: {
                Pb c = r.sub();
                auto e = std::make_shared<Expr>();
                e->kind = BHIP_EXPR_IN_LIST;
                ExprPtr x;
                std::vector<ExprPtr> items;
                uint32_t cf, cwt;
                while (c.next(cf, cwt)) {
                    if (cf == 1 && cwt == 2) x = decode_expr(c.sub());
                    else if (cf == 2 && cwt == 2) items.push_back(decode_expr(c.sub()));
                    else if (cf == 3 && cwt == 0) e->negated = c.varint() != 0;
                    else c.skip(cwt);
                }
                if (!x || items.empty()) fail(BHIP_EINVAL, "IN list without items");
                e->args.push_back(x);
                for (auto& it : items) e->args.push_back(it);
                out = e;
            } break;
            case 15: fail(BHIP_EINVAL, "wildcard in a physical expression");
            case 16: {
                Pb c = r.sub();
                auto e = std::make_shared<Expr>();
                e->kind = BHIP_EXPR_SCALAR_FN;
                e->name = "sqrt";                       // enum value 0 is not written on the wire
                uint32_t cf, cwt;
                while (c.next(cf, cwt)) {
                    if (cf == 1 && cwt == 0) e->name = scalar_function_name(c.varint());
                    else if (cf == 2 && cwt == 2) e->args.push_back(decode_expr(c.sub()));
                    else c.skip(cwt);
                }
                if (e->args.empty()) fail(BHIP_EINVAL, "scalar function without arguments");
                check_scalar_function(e->name, (int)e->args.size());
                out = e;
            } break;
            default: r.skip(wt);
        }
    }
    if (!out) fail(BHIP_EINVAL, "protobuf: LogicalExprNode without an expression");
    return out;
}

// AggregateExprNode inside a LogicalExprNode (field 5); AggregateFunction enum: MIN MAX SUM AVG COUNT (ballista.proto:120-126)
void decode_aggregate(Pb r, int& fn, ExprPtr& arg) {
    uint32_t f, wt;
    bool found = false;
    while (r.next(f, wt)) {
        if (f == 5 && wt == 2) {
            Pb a = r.sub();
            uint64_t v = 0;
            uint32_t af, awt;
            while (a.next(af, awt)) {
                if (af == 1 && awt == 0) v = a.varint();
                else if (af == 2 && awt == 2) arg = decode_expr(a.sub());
                else a.skip(awt);
            }
            static const int map[] = {BHIP_AGG_MIN, BHIP_AGG_MAX, BHIP_AGG_SUM, BHIP_AGG_AVG, BHIP_AGG_COUNT};
            if (v > 4) fail(BHIP_EINVAL, "protobuf: unknown AggregateFunction " + std::to_string(v));
            fn = map[v];
            found = true;
        } else
            r.skip(wt);
    }
    if (!found || !arg) fail(BHIP_EINVAL, "Invalid expression for HashAggregateExec");      // from_proto.rs:238-242
}

// SortExprNode inside a LogicalExprNode (field 12)
SortDesc decode_sort_expr(Pb r) {
    uint32_t f, wt;
    SortDesc d{nullptr, true, false};      // proto3 defaults: asc = false, nulls_first = false
    bool found = false;
    while (r.next(f, wt)) {
        if (f == 12 && wt == 2) {
            Pb s = r.sub();
            uint32_t sf, swt;
            bool asc = false;
            while (s.next(sf, swt)) {
                if (sf == 1 && swt == 2) d.expr = decode_expr(s.sub());
                else if (sf == 2 && swt == 0) asc = s.varint() != 0;
                else if (sf == 3 && swt == 0) d.nulls_first = s.varint() != 0;
                else s.skip(swt);
            }
            d.descending = !asc;
            found = true;
        } else
            r.skip(wt);
    }
    if (!found || !d.expr) fail(BHIP_EINVAL, "physical_plan::from_proto() Unexpected sort expr");
    return d;
}

// ---- leaves -------------------------------------------------------------------------------------------------------------------
struct LeafInfo {
    int kind = 0;
    std::string path, delimiter = ",", file_extension;
    std::vector<std::string> filenames;
    std::vector<uint32_t> projection, stage_ids;
    SchemaPtr file_schema;            // CsvScan: the file's fields; shuffle leaves: the output schema
    bool has_header = false, has_projection = false;
    uint32_t batch_size = 0, num_partitions = 0, partition_count = 0;
    struct Loc { std::string job_id, executor_id, host; uint32_t stage_id = 0, partition_id = 0, port = 0; int64_t rows = -1, batches = -1, bytes = -1; };
    std::vector<Loc> locations;
};

SchemaPtr project_schema(const SchemaPtr& s, const std::vector<uint32_t>& proj, bool has_proj) {
    if (!has_proj) return s;
    auto o = std::make_shared<Schema>();
    for (uint32_t i : proj) {
        if (i >= s->fields.size()) fail(BHIP_EINVAL, "scan projection index " + std::to_string(i) + " is out of range");
        o->fields.push_back(s->fields[i]);
    }
    return o;
}

}  // namespace

// a leaf nobody resolved: it knows its schema and describes itself; execute() is an error, not a crash
class UnresolvedLeafExec : public ExecutionPlan {
public:
    UnresolvedLeafExec(std::string op, std::string text, SchemaPtr schema, int partitions)
        : op_(std::move(op)), text_(std::move(text)), schema_(std::move(schema)), partitions_(partitions < 1 ? 1 : partitions) {}
    const char* name() const override { return op_.c_str(); }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, partitions_, {}}; }
    std::vector<PlanPtr> children() const override { return {}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override {
        if (!c.empty()) fail(BHIP_EINVAL, op_ + " has no children");
        return shared_from_this();
    }
    StreamPtr execute(int, const Exec&) const override {
        fail(BHIP_EEXEC, "Ballista Error: " + op_ + " was not resolved to a data source (pass a leaf resolver to bhip_plan_from_proto)");
    }
    std::string describe() const override { return text_; }
private:
    std::string op_, text_;
    SchemaPtr schema_;
    int partitions_;
};

// CsvExec over TPC-H `.tbl` text, parsed on the device: one partition per file (CsvExec::try_new(path, options, projection,
// batch_size), from_proto.rs:93-110; the files of a directory in name order, as DataFusion lists them)
class TblScanExec : public ExecutionPlan {
public:
    TblScanExec(ContextPtr ctx, std::string path, std::vector<std::string> files, SchemaPtr file_schema, std::vector<uint32_t> proj,
                bool has_proj)
        : path_(std::move(path)), files_(std::move(files)), file_schema_(std::move(file_schema)), proj_(std::move(proj)), has_proj_(has_proj) {
        ctx_ = std::move(ctx);
        schema_ = project_schema(file_schema_, proj_, has_proj_);
    }
    const char* name() const override { return "CsvExec"; }
    SchemaPtr schema() const override { return schema_; }
    Partitioning output_partitioning() const override { return Partitioning{BHIP_PART_UNKNOWN, (int)std::max<size_t>(1, files_.size()), {}}; }
    std::vector<PlanPtr> children() const override { return {}; }
    PlanPtr with_new_children(const std::vector<PlanPtr>& c) const override {
        if (!c.empty()) fail(BHIP_EINVAL, "CsvExec has no children");
        return shared_from_this();
    }
    std::string describe() const override {
        std::string s = "CsvExec: path=" + path_ + ", delimiter='|', device scan, projection=[";
        for (size_t i = 0; i < schema_->fields.size(); ++i) s += (i ? ", " : "") + schema_->fields[i].name;
        return s + "], files=" + std::to_string(files_.size());
    }
    StreamPtr execute(int partition, const Exec& ex) const override {
        check_partition(*this, partition);
        auto self = std::static_pointer_cast<const TblScanExec>(shared_from_this());
        return StreamPtr(new LazyStream(schema_, [self, partition, ex]() -> std::vector<BatchPtr> {
            if (self->files_.empty()) return {};
            const std::string& fn = self->files_[partition];
            std::ifstream in(fn, std::ios::binary | std::ios::ate);
            if (!in) fail(BHIP_EEXEC, "Ballista Error: cannot open " + fn);
            const std::streamsize n = in.tellg();
            if (n >= (std::streamsize)0xFFFF0000ll) fail(BHIP_ENOTIMPL, fn + ": files of 4 GiB and more need a chunked reader");
            std::string text((size_t)n, '\0');
            in.seekg(0);
            if (n && !in.read(&text[0], n)) fail(BHIP_EEXEC, "Ballista Error: cannot read " + fn);
            std::vector<bhip_column_desc> fields(self->file_schema_->fields.size());
            for (size_t i = 0; i < fields.size(); ++i) {
                memset(&fields[i], 0, sizeof(fields[i]));
                fields[i].name = self->file_schema_->fields[i].name.c_str();
                fields[i].dtype = self->file_schema_->fields[i].dtype;
                fields[i].nullable = self->file_schema_->fields[i].nullable;
            }
            std::vector<int32_t> proj(self->proj_.begin(), self->proj_.end());
            BatchPtr b = batch_from_tbl(ex.ctx, text.data(), (int64_t)text.size(), (int)fields.size(), fields.data(),
                                        self->has_proj_ ? (int)proj.size() : 0, self->has_proj_ ? proj.data() : nullptr);
            return {b};
        }));
    }
private:
    std::string path_;
    std::vector<std::string> files_;
    SchemaPtr file_schema_, schema_;
    std::vector<uint32_t> proj_;
    bool has_proj_;
};

namespace {

std::vector<std::string> list_files(const std::string& path, const std::string& ext) {
    struct stat st;
    std::vector<std::string> out;
    if (stat(path.c_str(), &st) != 0) return out;
    if (!S_ISDIR(st.st_mode)) return {path};
    if (DIR* d = opendir(path.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n == "." || n == "..") continue;
            if (!ext.empty() && (n.size() < ext.size() || n.compare(n.size() - ext.size(), ext.size(), ext) != 0)) continue;
            out.push_back(path + (path.back() == '/' ? "" : "/") + n);
        }
        closedir(d);
    }
    std::sort(out.begin(), out.end());
    return out;
}

std::string leaf_text(const LeafInfo& L, const SchemaPtr& out_schema) {
    std::string s;
    auto cols = [&]() {
        std::string c = "[";
        for (size_t i = 0; i < out_schema->fields.size(); ++i) c += (i ? ", " : "") + out_schema->fields[i].name;
        return c + "]";
    };
    switch (L.kind) {
        case BHIP_LEAF_CSV_SCAN:
            return "CsvExec: path=" + L.path + ", delimiter='" + L.delimiter + "', has_header=" + (L.has_header ? "true" : "false") +
                   ", projection=" + cols();
        case BHIP_LEAF_PARQUET_SCAN: {
            s = "ParquetExec: files=[";
            for (size_t i = 0; i < L.filenames.size(); ++i) s += (i ? ", " : "") + L.filenames[i];
            s += "], projection=[";
            for (size_t i = 0; i < L.projection.size(); ++i) s += (i ? ", " : "") + std::to_string(L.projection[i]);
            return s + "], partitions=" + std::to_string(L.num_partitions);
        }
        case BHIP_LEAF_SHUFFLE_READER: {
            s = "ShuffleReaderExec: partition_locations=[";
            for (size_t i = 0; i < L.locations.size(); ++i)
                s += (i ? ", " : "") + L.locations[i].job_id + "/" + std::to_string(L.locations[i].stage_id) + "/" +
                     std::to_string(L.locations[i].partition_id) + "@" + L.locations[i].host + ":" + std::to_string(L.locations[i].port);
            return s + "], schema=" + cols();
        }
        default: {
            s = "UnresolvedShuffleExec: query_stage_ids=[";
            for (size_t i = 0; i < L.stage_ids.size(); ++i) s += (i ? ", " : "") + std::to_string(L.stage_ids[i]);
            return s + "], partition_count=" + std::to_string(L.partition_count) + ", schema=" + cols();
        }
    }
}

struct Decoder {
    ContextPtr ctx;                 // may be null: plans are then only describable
    bhip_leaf_resolver resolver;
    void* user;

    PlanPtr leaf(const LeafInfo& L) {
        SchemaPtr out_schema = L.kind == BHIP_LEAF_CSV_SCAN ? project_schema(L.file_schema, L.projection, L.has_projection)
                                                            : (L.file_schema ? L.file_schema : std::make_shared<Schema>());
        if (resolver) {
            // the C image of the leaf; strings and arrays live until the callback returns
            std::vector<const char*> files;
            for (auto& f : L.filenames) files.push_back(f.c_str());
            std::vector<bhip_column_desc> fields(L.file_schema ? L.file_schema->fields.size() : 0);
            for (size_t i = 0; i < fields.size(); ++i) {
                memset(&fields[i], 0, sizeof(fields[i]));
                fields[i].name = L.file_schema->fields[i].name.c_str();
                fields[i].dtype = L.file_schema->fields[i].dtype;
                fields[i].nullable = L.file_schema->fields[i].nullable;
            }
            std::vector<bhip_partition_location> locs(L.locations.size());
            for (size_t i = 0; i < locs.size(); ++i) {
                const auto& l = L.locations[i];
                locs[i] = bhip_partition_location{l.job_id.c_str(), l.stage_id, l.partition_id, l.executor_id.c_str(), l.host.c_str(), l.port,
                                                  l.rows, l.batches, l.bytes};
            }
            bhip_leaf_desc d;
            memset(&d, 0, sizeof(d));
            d.kind = L.kind;
            d.path = L.path.c_str();
            d.n_filenames = (int32_t)files.size();
            d.filenames = files.data();
            d.has_projection = L.has_projection;
            d.n_projection = (int32_t)L.projection.size();
            d.projection = L.projection.data();
            d.n_fields = (int32_t)fields.size();
            d.fields = fields.data();
            d.has_header = L.has_header;
            d.delimiter = L.delimiter.c_str();
            d.file_extension = L.file_extension.c_str();
            d.batch_size = L.batch_size;
            d.num_partitions = L.num_partitions;
            d.n_locations = (int32_t)locs.size();
            d.locations = locs.data();
            d.n_stage_ids = (int32_t)L.stage_ids.size();
            d.stage_ids = L.stage_ids.data();
            d.partition_count = L.partition_count;
            bhip_plan* got = nullptr;
            const bhip_status st = resolver(user, &d, &got);
            if (st != BHIP_OK) fail(st, std::string("leaf resolver failed: ") + get_last_error());
            if (got) {
                PlanPtr p = got->p;
                bhip_plan_release(got);
                // the resolved leaf must produce what the wire plan says it produces
                const Schema& have = *p->schema();
                if (L.kind != BHIP_LEAF_PARQUET_SCAN) {
                    if (have.fields.size() != out_schema->fields.size()) fail(BHIP_EINVAL, "resolved leaf has a different number of columns than the wire plan");
                    for (size_t i = 0; i < have.fields.size(); ++i)
                        if (have.fields[i].name != out_schema->fields[i].name || have.fields[i].dtype != out_schema->fields[i].dtype)
                            fail(BHIP_EINVAL, "resolved leaf column " + std::to_string(i) + " is " + have.fields[i].name + ": " +
                                                  dtype_name(have.fields[i].dtype) + ", the wire plan says " + out_schema->fields[i].name + ": " +
                                                  dtype_name(out_schema->fields[i].dtype));
                }
                return p;
            }
        }
        if (L.kind == BHIP_LEAF_CSV_SCAN && ctx && L.delimiter == "|" && !L.has_header) {
            std::vector<std::string> files = L.filenames.empty() ? list_files(L.path, L.file_extension) : L.filenames;
            if (!files.empty() || L.path.compare(0, 6, "mem://") != 0)
                return std::make_shared<TblScanExec>(ctx, L.path, files, L.file_schema, L.projection, L.has_projection);
        }
        if (L.kind == BHIP_LEAF_PARQUET_SCAN) {
            // the file schema is not part of the wire plan: it is read from the files' footers, so this leaf needs the files —
            // and a device context to put their rows on
            if (!ctx) fail(BHIP_ENOTIMPL, "ParquetExec: without a device context the leaf cannot be built (its schema lives in the files); "
                                          "resolve it (bhip_leaf_resolver) or pass a context");
            return make_parquet_exec(ctx, L.filenames, L.projection, L.has_projection, (int)L.num_partitions);
        }
        static const char* ops[] = {"", "CsvExec", "ParquetExec", "ShuffleReaderExec", "UnresolvedShuffleExec"};
        int parts = 1;
        // ShuffleReaderExec reports one output partition per location (rust/core/src/execution_plans/shuffle_reader.rs:61)
        if (L.kind == BHIP_LEAF_SHUFFLE_READER) parts = std::max<int>(1, (int)L.locations.size());
        if (L.kind == BHIP_LEAF_UNRESOLVED_SHUFFLE) parts = (int)L.partition_count;
        return std::make_shared<UnresolvedLeafExec>(ops[L.kind], leaf_text(L, out_schema), out_schema, parts);
    }

    PlanPtr input_of(Pb& m, uint32_t wt, const char* what) {
        m.expect(wt, 2, what);
        return plan(m.sub());
    }

    static void need_input(const PlanPtr& p, const char* node) {
        if (!p) fail(BHIP_EINVAL, std::string("protobuf: ") + node + " without an input");       // convert_box_required!
    }

    PlanPtr plan(Pb r) {
        NestGuard nest;
        uint32_t f, wt;
        PlanPtr out;
        while (r.next(f, wt)) {
            if (wt != 2) { r.skip(wt); continue; }
            Pb m = r.sub();
            uint32_t mf, mwt;
            switch (f) {
                case 1: {                                                  // ParquetScanExecNode
                    LeafInfo L;
                    L.kind = BHIP_LEAF_PARQUET_SCAN;
                    L.has_projection = true;
                    while (m.next(mf, mwt)) {
                        if (mf == 1 && mwt == 2) L.filenames.push_back(m.str());
                        else if (mf == 2) m.repeated_u32(mwt, L.projection);
                        else if (mf == 3 && mwt == 0) L.num_partitions = (uint32_t)m.varint();
                        else if (mf == 4 && mwt == 0) L.batch_size = (uint32_t)m.varint();
                        else m.skip(mwt);
                    }
                    out = leaf(L);
                } break;
                case 2: {                                                  // CsvScanExecNode
                    LeafInfo L;
                    L.kind = BHIP_LEAF_CSV_SCAN;
                    L.has_projection = true;                               // Some(projection), from_proto.rs:103
                    while (m.next(mf, mwt)) {
                        if (mf == 1 && mwt == 2) L.path = m.str();
                        else if (mf == 2) m.repeated_u32(mwt, L.projection);
                        else if (mf == 3 && mwt == 2) L.file_schema = decode_schema(m.sub());
                        else if (mf == 4 && mwt == 2) L.file_extension = m.str();
                        else if (mf == 5 && mwt == 0) L.has_header = m.varint() != 0;
                        else if (mf == 6 && mwt == 0) L.batch_size = (uint32_t)m.varint();
                        else if (mf == 7 && mwt == 2) L.delimiter = m.str();
                        else if (mf == 8 && mwt == 2) L.filenames.push_back(m.str());
                        else m.skip(mwt);
                    }
                    if (!L.file_schema) fail(BHIP_EINVAL, "protobuf: CsvScanExecNode without a schema");      // convert_required!
                    if (L.delimiter.empty()) fail(BHIP_EINVAL, "protobuf: CsvScanExecNode without a delimiter");
                    out = leaf(L);
                } break;
                case 3: {                                                  // EmptyExecNode
                    bool one = false;
                    SchemaPtr s;
                    while (m.next(mf, mwt)) {
                        if (mf == 1 && mwt == 0) one = m.varint() != 0;
                        else if (mf == 2 && mwt == 2) s = decode_schema(m.sub());
                        else m.skip(mwt);
                    }
                    if (!s) fail(BHIP_EINVAL, "protobuf: EmptyExecNode without a schema");
                    out = std::make_shared<EmptyExec>(ctx, s, one);
                } break;
                case 4: {                                                  // ProjectionExecNode
                    PlanPtr in;
                    std::vector<ExprPtr> exprs;
                    std::vector<std::string> names;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "ProjectionExecNode.input");
                        else if (mf == 2 && mwt == 2) exprs.push_back(decode_expr(m.sub()));
                        else if (mf == 3 && mwt == 2) names.push_back(m.str());
                        else m.skip(mwt);
                    }
                    need_input(in, "ProjectionExecNode");
                    std::vector<std::pair<ExprPtr, std::string>> en;
                    for (size_t i = 0; i < exprs.size() && i < names.size(); ++i)            // zip(), from_proto.rs:72-78
                        en.push_back({coerce_expr(exprs[i], *in->schema()), names[i]});
                    out = std::make_shared<ProjectionExec>(en, in);
                } break;
                case 6:
                case 7: {                                                  // Global / LocalLimitExecNode
                    PlanPtr in;
                    uint32_t limit = 0;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "LimitExecNode.input");
                        else if (mf == 2 && mwt == 0) limit = (uint32_t)m.varint();
                        else m.skip(mwt);
                    }
                    need_input(in, "LimitExecNode");
                    out = std::make_shared<LimitExec>(in, (int64_t)limit, f == 6);
                } break;
                case 8: {                                                  // HashAggregateExecNode
                    PlanPtr in;
                    std::vector<ExprPtr> gexprs;
                    std::vector<std::pair<int, ExprPtr>> aggs;
                    std::vector<std::string> gnames, anames;
                    SchemaPtr input_schema;
                    uint64_t mode = 0;
                    while (m.next(mf, mwt)) {
                        if (mf == 1 && mwt == 2) gexprs.push_back(decode_expr(m.sub()));
                        else if (mf == 2 && mwt == 2) { int fn = 0; ExprPtr arg; decode_aggregate(m.sub(), fn, arg); aggs.push_back({fn, arg}); }
                        else if (mf == 3 && mwt == 0) mode = m.varint();
                        else if (mf == 4) in = input_of(m, mwt, "HashAggregateExecNode.input");
                        else if (mf == 5 && mwt == 2) gnames.push_back(m.str());
                        else if (mf == 6 && mwt == 2) anames.push_back(m.str());
                        else if (mf == 7 && mwt == 2) input_schema = decode_schema(m.sub());
                        else m.skip(mwt);
                    }
                    need_input(in, "HashAggregateExecNode");
                    if (mode > 1) fail(BHIP_EINVAL, "Received a HashAggregateNode message with unknown AggregateMode " + std::to_string(mode));
                    if (!input_schema) fail(BHIP_EINVAL, "input_schema in HashAggregateNode is missing.");
                    std::vector<std::pair<ExprPtr, std::string>> g;
                    for (size_t i = 0; i < gexprs.size() && i < gnames.size(); ++i) g.push_back({coerce_expr(gexprs[i], *in->schema()), gnames[i]});
                    // aggregate arguments are planned against input_schema — the schema of the PARTIAL aggregate's input, which
                    // a Final aggregate receives too (from_proto.rs:213-236); Final reads its state columns by position
                    std::vector<AggregateDesc> a;
                    for (size_t i = 0; i < aggs.size() && i < anames.size(); ++i) {
                        ExprPtr arg = aggs[i].second;
                        if (mode == 0) arg = coerce_expr(arg, *in->schema());
                        else (void)expr_type(coerce_expr(arg, *input_schema), *input_schema);      // type-checks as the reference does
                        a.push_back(AggregateDesc{aggs[i].first, arg, anames[i]});
                    }
                    out = std::make_shared<HashAggregateExec>(mode == 0 ? BHIP_AGG_PARTIAL : BHIP_AGG_FINAL, g, a, in);
                } break;
                case 9: {                                                  // HashJoinExecNode
                    PlanPtr left, right;
                    std::vector<std::pair<std::string, std::string>> on;
                    uint64_t jt = 0;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) left = input_of(m, mwt, "HashJoinExecNode.left");
                        else if (mf == 2) right = input_of(m, mwt, "HashJoinExecNode.right");
                        else if (mf == 3 && mwt == 2) {
                            Pb o = m.sub();
                            std::pair<std::string, std::string> p;
                            uint32_t of, owt;
                            while (o.next(of, owt)) {
                                if (of == 1 && owt == 2) p.first = o.str();
                                else if (of == 2 && owt == 2) p.second = o.str();
                                else o.skip(owt);
                            }
                            on.push_back(p);
                        } else if (mf == 4 && mwt == 0) jt = m.varint();
                        else m.skip(mwt);
                    }
                    need_input(left, "HashJoinExecNode");
                    need_input(right, "HashJoinExecNode");
                    if (jt > 2) fail(BHIP_EINVAL, "Received a HashJoinNode message with unknown JoinType " + std::to_string(jt));
                    if (on.empty()) fail(BHIP_EINVAL, "HashJoinExec needs at least one key pair");
                    out = std::make_shared<HashJoinExec>(left, right, on, (int)jt);
                } break;
                case 10: {                                                 // ShuffleReaderExecNode
                    LeafInfo L;
                    L.kind = BHIP_LEAF_SHUFFLE_READER;
                    while (m.next(mf, mwt)) {
                        if (mf == 1 && mwt == 2) {
                            Pb pl = m.sub();
                            LeafInfo::Loc loc;
                            uint32_t pf, pwt;
                            while (pl.next(pf, pwt)) {
                                if (pf == 1 && pwt == 2) {                 // PartitionId
                                    Pb id = pl.sub();
                                    uint32_t i_f, iwt;
                                    while (id.next(i_f, iwt)) {
                                        if (i_f == 1 && iwt == 2) loc.job_id = id.str();
                                        else if (i_f == 2 && iwt == 0) loc.stage_id = (uint32_t)id.varint();
                                        else if (i_f == 4 && iwt == 0) loc.partition_id = (uint32_t)id.varint();
                                        else id.skip(iwt);
                                    }
                                } else if (pf == 2 && pwt == 2) {          // ExecutorMetadata
                                    Pb em = pl.sub();
                                    uint32_t ef, ewt;
                                    while (em.next(ef, ewt)) {
                                        if (ef == 1 && ewt == 2) loc.executor_id = em.str();
                                        else if (ef == 2 && ewt == 2) loc.host = em.str();
                                        else if (ef == 3 && ewt == 0) loc.port = (uint32_t)em.varint();
                                        else em.skip(ewt);
                                    }
                                } else if (pf == 3 && pwt == 2) {          // PartitionStats
                                    Pb ps = pl.sub();
                                    uint32_t sf, swt;
                                    while (ps.next(sf, swt)) {
                                        if (sf == 1 && swt == 0) loc.rows = (int64_t)ps.varint();
                                        else if (sf == 2 && swt == 0) loc.batches = (int64_t)ps.varint();
                                        else if (sf == 3 && swt == 0) loc.bytes = (int64_t)ps.varint();
                                        else ps.skip(swt);
                                    }
                                } else
                                    pl.skip(pwt);
                            }
                            L.locations.push_back(loc);
                        } else if (mf == 2 && mwt == 2) L.file_schema = decode_schema(m.sub());
                        else m.skip(mwt);
                    }
                    if (!L.file_schema) fail(BHIP_EINVAL, "protobuf: ShuffleReaderExecNode without a schema");
                    out = leaf(L);
                } break;
                case 11: {                                                 // SortExecNode
                    PlanPtr in;
                    std::vector<SortDesc> v;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "SortExecNode.input");
                        else if (mf == 2 && mwt == 2) v.push_back(decode_sort_expr(m.sub()));
                        else m.skip(mwt);
                    }
                    need_input(in, "SortExecNode");
                    if (v.empty()) fail(BHIP_EINVAL, "SortExec needs at least one sort expression");
                    for (auto& d : v) d.expr = coerce_expr(d.expr, *in->schema());
                    out = std::make_shared<SortExec>(v, in);
                } break;
                case 12: {                                                 // CoalesceBatchesExecNode
                    PlanPtr in;
                    uint32_t target = 0;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "CoalesceBatchesExecNode.input");
                        else if (mf == 2 && mwt == 0) target = (uint32_t)m.varint();
                        else m.skip(mwt);
                    }
                    need_input(in, "CoalesceBatchesExecNode");
                    out = std::make_shared<CoalesceBatchesExec>(in, (int64_t)target);
                } break;
                case 13: {                                                 // FilterExecNode
                    PlanPtr in;
                    ExprPtr pred;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "FilterExecNode.input");
                        else if (mf == 2 && mwt == 2) pred = decode_expr(m.sub());
                        else m.skip(mwt);
                    }
                    need_input(in, "FilterExecNode");
                    if (!pred) fail(BHIP_EINVAL, "filter (FilterExecNode) in PhysicalPlanNode is missing.");      // from_proto.rs:84-88
                    out = std::make_shared<FilterExec>(coerce_expr(pred, *in->schema()), in);
                } break;
                case 14: {                                                 // MergeExecNode
                    PlanPtr in;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "MergeExecNode.input");
                        else m.skip(mwt);
                    }
                    need_input(in, "MergeExecNode");
                    out = std::make_shared<MergeExec>(in);
                } break;
                case 15: {                                                 // UnresolvedShuffleExecNode
                    LeafInfo L;
                    L.kind = BHIP_LEAF_UNRESOLVED_SHUFFLE;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) m.repeated_u32(mwt, L.stage_ids);
                        else if (mf == 2 && mwt == 2) L.file_schema = decode_schema(m.sub());
                        else if (mf == 3 && mwt == 0) L.partition_count = (uint32_t)m.varint();
                        else m.skip(mwt);
                    }
                    if (!L.file_schema) fail(BHIP_EINVAL, "protobuf: UnresolvedShuffleExecNode without a schema");
                    out = leaf(L);
                } break;
                case 16: {                                                 // RepartitionExecNode
                    PlanPtr in;
                    Partitioning part;
                    part.scheme = -1;
                    std::vector<ExprPtr> hexprs;
                    while (m.next(mf, mwt)) {
                        if (mf == 1) in = input_of(m, mwt, "RepartitionExecNode.input");
                        else if (mf == 2 && mwt == 0) { part.scheme = BHIP_PART_ROUND_ROBIN; part.count = (int)m.varint(); }
                        else if (mf == 3 && mwt == 2) {
                            Pb h = m.sub();
                            part.scheme = BHIP_PART_HASH;
                            part.count = 0;
                            uint32_t hf, hwt;
                            while (h.next(hf, hwt)) {
                                if (hf == 1 && hwt == 2) hexprs.push_back(decode_expr(h.sub()));
                                else if (hf == 2 && hwt == 0) part.count = (int)h.varint();
                                else h.skip(hwt);
                            }
                        } else if (mf == 4 && mwt == 0) { part.scheme = BHIP_PART_UNKNOWN; part.count = (int)m.varint(); }
                        else m.skip(mwt);
                    }
                    need_input(in, "RepartitionExecNode");
                    if (part.scheme < 0) fail(BHIP_EINVAL, "Invalid partitioning scheme");                 // from_proto.rs:159-162
                    if (part.count < 1) fail(BHIP_EINVAL, "partition count must be positive");
                    for (auto& e : hexprs) part.exprs.push_back(coerce_expr(e, *in->schema()));
                    if (part.scheme == BHIP_PART_HASH && part.exprs.empty()) fail(BHIP_EINVAL, "hash repartition needs key expressions");
                    out = std::make_shared<RepartitionExec>(in, part);
                } break;
                default: break;                                            // unknown oneof member: skipped
            }
        }
        if (!out) fail(BHIP_EINVAL, "physical_plan::from_proto() Unsupported physical plan");            // from_proto.rs:62-67
        return out;
    }
};

}  // namespace

PlanPtr plan_from_proto(const ContextPtr& ctx, const void* bytes, size_t len, bhip_leaf_resolver resolver, void* user) {
    Decoder d{ctx, resolver, user};
    return d.plan(Pb(bytes, len));
}

ExprPtr expr_from_proto(const void* bytes, size_t len) { return decode_expr(Pb(bytes, len)); }

}  // namespace bhip
