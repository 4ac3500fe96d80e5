//! `extern "C"` image of include/ballista_hip.h — only the entry points the shim calls.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};

pub type bhip_status = i32;
pub const BHIP_OK: bhip_status = 0;
pub const BHIP_ENOTIMPL: bhip_status = 2;

#[repr(C)]
pub struct bhip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bhip_plan {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bhip_stream {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bhip_batch {
    _private: [u8; 0],
}

/// Arrow C Stream Interface.  arrow-rs at rev 46161d2 ships the C Data Interface (`arrow::ffi`) but not yet the stream
/// half, so the struct is declared here and filled by `crate::cstream`.
#[repr(C)]
pub struct ArrowArrayStream {
    pub get_schema: Option<unsafe extern "C" fn(*mut ArrowArrayStream, *mut FFI_ArrowSchema) -> c_int>,
    pub get_next: Option<unsafe extern "C" fn(*mut ArrowArrayStream, *mut FFI_ArrowArray) -> c_int>,
    pub get_last_error: Option<unsafe extern "C" fn(*mut ArrowArrayStream) -> *const c_char>,
    pub release: Option<unsafe extern "C" fn(*mut ArrowArrayStream)>,
    pub private_data: *mut c_void,
}

#[repr(C)]
pub struct bhip_column_desc {
    pub name: *const c_char,
    pub dtype: i32,
    pub nullable: i32,
    pub data: *const c_void,
    pub offsets: *const i32,
    pub validity: *const u8,
    pub data_bytes: i64,
}

#[repr(C)]
pub struct bhip_partition_location {
    pub job_id: *const c_char,
    pub stage_id: u32,
    pub partition_id: u32,
    pub executor_id: *const c_char,
    pub host: *const c_char,
    pub port: u32,
    pub num_rows: i64,
    pub num_batches: i64,
    pub num_bytes: i64,
}

#[repr(C)]
pub struct bhip_leaf_desc {
    pub kind: i32,
    pub path: *const c_char,
    pub n_filenames: i32,
    pub filenames: *const *const c_char,
    pub has_projection: i32,
    pub n_projection: i32,
    pub projection: *const u32,
    pub n_fields: i32,
    pub fields: *const bhip_column_desc,
    pub has_header: i32,
    pub delimiter: *const c_char,
    pub file_extension: *const c_char,
    pub batch_size: u32,
    pub num_partitions: u32,
    pub n_locations: i32,
    pub locations: *const bhip_partition_location,
    pub n_stage_ids: i32,
    pub stage_ids: *const u32,
    pub partition_count: u32,
}

pub type bhip_leaf_resolver =
    Option<unsafe extern "C" fn(user: *mut c_void, leaf: *const bhip_leaf_desc, out: *mut *mut bhip_plan) -> bhip_status>;

extern "C" {
    pub fn bhip_last_error() -> *const c_char;
    pub fn bhip_ctx_create(device: c_int, out: *mut *mut bhip_ctx) -> bhip_status;
    pub fn bhip_ctx_release(ctx: *mut bhip_ctx);

    pub fn bhip_plan_from_proto(
        ctx: *mut bhip_ctx,
        bytes: *const c_void,
        len: usize,
        resolve: bhip_leaf_resolver,
        user: *mut c_void,
        out: *mut *mut bhip_plan,
    ) -> bhip_status;
    pub fn bhip_plan_arrow_streams(
        ctx: *mut bhip_ctx,
        n_partitions: i32,
        streams: *const *mut ArrowArrayStream,
        out: *mut *mut bhip_plan,
    ) -> bhip_status;
    pub fn bhip_plan_output_partitioning(plan: *const bhip_plan, scheme: *mut i32, partition_count: *mut i32) -> bhip_status;
    pub fn bhip_plan_display(plan: *const bhip_plan, buf: *mut c_char, cap: usize) -> bhip_status;
    pub fn bhip_plan_execute(plan: *mut bhip_plan, partition: i32, out: *mut *mut bhip_stream) -> bhip_status;
    /// datafusion::physical_plan::collect: every output partition in turn, all batches (at most `cap`) in partition order
    pub fn bhip_plan_collect(plan: *mut bhip_plan, cap: i32, out: *mut *mut bhip_batch, n_out: *mut i32) -> bhip_status;
    pub fn bhip_plan_release(plan: *mut bhip_plan);

    /// utils::write_stream_to_disk (rust/core/src/utils.rs:49-84) on the device side: drains `stream` into an Arrow IPC file and
    /// reports PartitionStats; the stream is consumed
    pub fn bhip_stream_write_ipc(
        stream: *mut bhip_stream,
        path: *const c_char,
        num_rows: *mut u64,
        num_batches: *mut u64,
        num_bytes: *mut u64,
    ) -> bhip_status;
    pub fn bhip_stream_next(stream: *mut bhip_stream, out: *mut *mut bhip_batch) -> bhip_status;
    pub fn bhip_stream_release(stream: *mut bhip_stream);
    pub fn bhip_batch_export_arrow(batch: *mut bhip_batch, out_array: *mut FFI_ArrowArray, out_schema: *mut FFI_ArrowSchema) -> bhip_status;
    pub fn bhip_batch_release(batch: *mut bhip_batch);
}
