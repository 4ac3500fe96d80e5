//! A CPU `ExecutionPlan` partition exported as an Arrow C stream (`ArrowArrayStream`): what `bhip_plan_arrow_streams` consumes
//! for the leaves of an offloaded subtree (scans, shuffle readers) — their decoding stays on the host, every batch they yield
//! crosses into the library through the C Data Interface.

use std::ffi::CString;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::Arc;

use arrow::array::{Array, StructArray};
use arrow::datatypes::SchemaRef;
use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use arrow::record_batch::RecordBatch;
use datafusion::physical_plan::ExecutionPlan;
use futures::StreamExt;
use tokio::runtime::Handle;

use crate::ffi::ArrowArrayStream;

struct Private {
    plan: Arc<dyn ExecutionPlan>,
    partition: usize,
    schema: SchemaRef,
    runtime: Handle,
    /// created by the first `get_next`
    stream: Option<std::pin::Pin<Box<dyn datafusion::physical_plan::RecordBatchStream + Send + Sync>>>,
    last_error: CString,
}

/// Export partition `partition` of `plan`.  `get_next` blocks the calling (blocking-pool) thread on the runtime handle; the
/// library only calls it from inside `bhip_plan_execute` / `bhip_stream_next`, which the shim runs under `spawn_blocking`.
pub fn export(plan: Arc<dyn ExecutionPlan>, partition: usize, runtime: Handle) -> ArrowArrayStream {
    let schema = plan.schema();
    let private = Box::new(Private { plan, partition, schema, runtime, stream: None, last_error: CString::default() });
    ArrowArrayStream {
        get_schema: Some(get_schema),
        get_next: Some(get_next),
        get_last_error: Some(get_last_error),
        release: Some(release),
        private_data: Box::into_raw(private) as *mut c_void,
    }
}

unsafe fn private<'a>(s: *mut ArrowArrayStream) -> &'a mut Private {
    &mut *((*s).private_data as *mut Private)
}

fn set_error(p: &mut Private, msg: String) -> c_int {
    p.last_error = CString::new(msg.replace('\0', " ")).unwrap_or_default();
    5 // EIO
}

/// RecordBatch -> struct array -> C Data Interface (one child per column)
fn export_batch(batch: &RecordBatch, out_array: *mut FFI_ArrowArray, out_schema: Option<*mut FFI_ArrowSchema>) -> arrow::error::Result<()> {
    let sa: StructArray = batch.clone().into();
    let (a, s) = sa.to_raw()?;
    unsafe {
        // `to_raw` hands out Arc-allocated structs: move their contents into the caller's structs
        std::ptr::copy_nonoverlapping(a, out_array, 1);
        if let Some(os) = out_schema {
            std::ptr::copy_nonoverlapping(s, os, 1);
        } else {
            drop(Arc::from_raw(s));
        }
        std::mem::forget(Arc::from_raw(a));
    }
    Ok(())
}

unsafe extern "C" fn get_schema(s: *mut ArrowArrayStream, out: *mut FFI_ArrowSchema) -> c_int {
    let p = private(s);
    let empty = RecordBatch::new_empty(p.schema.clone());
    let mut scratch = std::mem::zeroed::<FFI_ArrowArray>();
    match export_batch(&empty, &mut scratch, Some(out)) {
        Ok(()) => {
            drop(scratch); // the empty array itself is not wanted
            0
        }
        Err(e) => set_error(p, format!("{:?}", e)),
    }
}

unsafe extern "C" fn get_next(s: *mut ArrowArrayStream, out: *mut FFI_ArrowArray) -> c_int {
    let p = private(s);
    if p.stream.is_none() {
        let plan = p.plan.clone();
        let part = p.partition;
        match p.runtime.block_on(async move { plan.execute(part).await }) {
            Ok(st) => p.stream = Some(st),
            Err(e) => return set_error(p, format!("{:?}", e)),
        }
    }
    let next = {
        let st = p.stream.as_mut().unwrap();
        p.runtime.block_on(st.next())
    };
    match next {
        None => {
            std::ptr::write_bytes(out, 0, 1); // released array = end of stream
            0
        }
        Some(Ok(batch)) => match export_batch(&batch, out, None) {
            Ok(()) => 0,
            Err(e) => set_error(p, format!("{:?}", e)),
        },
        Some(Err(e)) => set_error(p, format!("{:?}", e)),
    }
}

unsafe extern "C" fn get_last_error(s: *mut ArrowArrayStream) -> *const c_char {
    private(s).last_error.as_ptr()
}

unsafe extern "C" fn release(s: *mut ArrowArrayStream) {
    if (*s).private_data.is_null() {
        return;
    }
    drop(Box::from_raw((*s).private_data as *mut Private));
    (*s).private_data = std::ptr::null_mut();
    (*s).release = None;
}
