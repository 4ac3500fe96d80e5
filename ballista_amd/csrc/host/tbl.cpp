// tbl.cpp — host side of the `.tbl` scan (kernels_tbl.hip): the leaf where the reference has
// CsvExec(delimiter '|', no header, explicit schema) — rust/benchmarks/tpch/src/main.rs:129-150,
// rust/core/src/serde/physical_plan/from_proto.rs:93-110.  The text crosses PCIe once; lines, fields and values are
// found on the device.
#include <cstring>

#include "../tbl_kernels.h"
#include "../util_kernels.h"
#include "core.hpp"

namespace bhip {

BatchPtr batch_from_tbl(const ContextPtr& ctx, const void* text_host, int64_t n_bytes, int n_fields, const bhip_column_desc* fields,
                        int n_proj, const int32_t* projection) {
    if (n_bytes < 0 || n_bytes > 0xFFFFFFF0ll) fail(BHIP_EINVAL, "tbl text must be < 4 GiB per call (split the file on line boundaries)");
    if (n_fields < 1 || n_fields > TBL_MAX_FIELDS) fail(BHIP_EINVAL, "tbl schema must have 1.." + std::to_string(TBL_MAX_FIELDS) + " fields");
    if (n_bytes > 0 && !text_host) fail(BHIP_EINVAL, "tbl text is null");
    ctx->set_device();
    Exec ex{ctx, nullptr};
    const LaunchCfg cfg = ex.cfg();

    // which fields to materialise, in which order
    std::vector<int> proj;
    if (projection) {
        for (int i = 0; i < n_proj; ++i) {
            if (projection[i] < 0 || projection[i] >= n_fields) fail(BHIP_EINVAL, "tbl projection index out of range");
            proj.push_back(projection[i]);
        }
    } else {
        for (int i = 0; i < n_fields; ++i) proj.push_back(i);
    }
    TblPlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.n_fields = n_fields;
    int last_needed = -1;
    for (int f = 0; f < n_fields; ++f) {
        if (!fields[f].name) fail(BHIP_EINVAL, "tbl field without a name");
        const int dt = fields[f].dtype;
        plan.dtype[f] = dt;
        plan.out[f] = -1;
    }
    auto schema = std::make_shared<Schema>();
    for (size_t s = 0; s < proj.size(); ++s) {
        const int f = proj[s];
        const int dt = fields[f].dtype;
        if (dt != DT_INT32 && dt != DT_INT64 && dt != DT_FLOAT64 && dt != DT_DATE32 && dt != DT_UTF8)
            fail(BHIP_ENOTIMPL, std::string("tbl scan of a ") + dtype_name(dt) + " column: " + fields[f].name);
        if (plan.out[f] >= 0) fail(BHIP_EINVAL, std::string("tbl projection names a field twice: ") + fields[f].name);
        plan.out[f] = (int)s;
        schema->fields.push_back(Field{fields[f].name, dt, fields[f].nullable != 0});
        if (f > last_needed) last_needed = f;
    }
    plan.n_fields = last_needed + 1;                     // fields behind the last projected one are never walked

    auto batch = std::make_shared<Batch>();
    batch->ctx = ctx;
    batch->schema = schema;

    Temp tmp(ex);
    uint8_t* text = tmp.get<uint8_t>((size_t)n_bytes + 64);
    if (n_bytes) HIP_CHECK(hipMemcpyAsync(text, text_host, (size_t)n_bytes, hipMemcpyHostToDevice, ex.stream));

    // ---- lines
    const int64_t n_chunks = (n_bytes + TBL_CHUNK - 1) / TBL_CHUNK;
    uint32_t* chunk_lines = tmp.get<uint32_t>((size_t)n_chunks + 1);
    uint64_t* chunk_base = tmp.get<uint64_t>((size_t)n_chunks + 1);
    uint64_t* total = tmp.get<uint64_t>(1);
    void* scan_tmp = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_chunks > 0 ? n_chunks : 1));
    int64_t n_newlines = 0;
    if (n_chunks) {
        HIP_CHECK(launch_tbl_count(cfg, text, n_bytes, chunk_lines));
        HIP_CHECK(exclusive_scan_u32_u64(ex.stream, chunk_lines, n_chunks, chunk_base, false, total, scan_tmp));
        n_newlines = (int64_t)read_device(ex, total);
    }
    const bool unterminated = n_bytes > 0 && static_cast<const uint8_t*>(text_host)[n_bytes - 1] != '\n';
    const int64_t n_lines = n_newlines + (unterminated ? 1 : 0);
    if (n_lines > 0xFFFFFFF0ll) fail(BHIP_EINVAL, "tbl text holds more than 2^32-16 lines");
    batch->n_rows = n_lines;

    uint64_t* starts = tmp.get<uint64_t>((size_t)n_lines + 2);
    if (n_lines) {
        const uint64_t zero = 0, end = (uint64_t)n_bytes + 1;          // an unterminated last line "ends" one past the text
        HIP_CHECK(hipMemcpyAsync(starts, &zero, 8, hipMemcpyHostToDevice, ex.stream));
        HIP_CHECK(launch_tbl_starts(cfg, text, n_bytes, chunk_base, starts));
        if (unterminated) HIP_CHECK(hipMemcpyAsync(starts + n_lines, &end, 8, hipMemcpyHostToDevice, ex.stream));
    }

    // ---- values
    uint32_t* flags = tmp.get<uint32_t>(2);
    HIP_CHECK(hipMemsetAsync(flags, 0, 8, ex.stream));
    std::vector<uint32_t*> lens(proj.size(), nullptr);
    for (size_t s = 0; s < proj.size(); ++s) {
        const int dt = fields[proj[s]].dtype;
        Column c;
        c.dtype = dt;
        c.length = n_lines;
        if (dt == DT_UTF8) {
            plan.str_start[s] = tmp.get<uint32_t>((size_t)n_lines + 1);
            plan.str_len[s] = lens[s] = tmp.get<uint32_t>((size_t)n_lines + 1);
            c.offsets = make_buffer(ex, (size_t)(n_lines + 1) * 4);
        } else {
            c.data = make_buffer(ex, (size_t)n_lines * dtype_width(dt) + 8);
            plan.data[s] = c.data->ptr();
        }
        batch->cols.push_back(std::move(c));
    }
    HIP_CHECK(launch_tbl_parse(cfg, text, starts, n_lines, n_bytes, plan, flags));

    // ---- strings: lengths -> offsets -> bytes; all totals in one read-back
    uint64_t* totals = tmp.get<uint64_t>(proj.size() + 1);
    std::vector<size_t> utf8;
    for (size_t s = 0; s < proj.size(); ++s)
        if (lens[s]) {
            void* st = tmp.get<uint8_t>(exclusive_scan_temp_bytes(n_lines > 0 ? n_lines : 1));
            HIP_CHECK(exclusive_scan_u32_i32(ex.stream, lens[s], n_lines, batch->cols[s].offsets->as<int32_t>(), true, totals + s, st));
            utf8.push_back(s);
        }
    std::vector<uint64_t> host_totals(proj.size() + 1, 0);
    uint32_t host_flags = 0;
    if (!utf8.empty()) HIP_CHECK(hipMemcpyAsync(host_totals.data(), totals, proj.size() * 8, hipMemcpyDeviceToHost, ex.stream));
    HIP_CHECK(hipMemcpyAsync(&host_flags, flags, 4, hipMemcpyDeviceToHost, ex.stream));
    HIP_CHECK(hipStreamSynchronize(ex.stream));
    if (host_flags & TBL_ERR_MISSING_FIELD) fail(BHIP_EEXEC, "tbl: a line has fewer fields than the schema");
    if (host_flags & TBL_ERR_BLANK_LINE) fail(BHIP_EEXEC, "tbl: blank line");
    if (host_flags & TBL_ERR_BAD_VALUE) fail(BHIP_EEXEC, "tbl: a field is not a value of its column's type");
    if (host_flags & TBL_ERR_PRECISION) fail(BHIP_ENOTIMPL, "tbl: a decimal with more than 15 significant digits");
    for (size_t s : utf8) {
        if (host_totals[s] > 0x7FFFFFFFull) fail(BHIP_EEXEC, "Utf8 column exceeds 2 GiB of value bytes");
        Column& c = batch->cols[s];
        c.data_bytes = (int64_t)host_totals[s];
        c.data = make_buffer(ex, (size_t)c.data_bytes + 8);
        HIP_CHECK(launch_tbl_copy_strings(cfg, text, plan.str_start[s], lens[s], c.offsets->as<int32_t>(), n_lines, c.data->as<uint8_t>()));
    }
    HIP_CHECK(hipStreamSynchronize(ex.stream));
    return batch;
}

}  // namespace bhip
