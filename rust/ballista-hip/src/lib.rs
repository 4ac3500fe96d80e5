//! `GpuExec`: a DataFusion `ExecutionPlan` whose `execute(partition)` runs a whole query-stage subtree on an MI355X through
//! libballista_hip.so, and `offload`, the rule an executor applies to a task's plan before calling `execute`.
//!
//! Where it plugs in — the ONE call a Ballista executor makes per task,
//! `partition.plan.execute(part)` (rust/executor/src/flight_service.rs:117-121), becomes
//!
//! ```ignore
//! let plan = ballista_hip::offload(partition.plan.clone(), &gpu)?;     // falls back to the CPU plan, node by node
//! let mut stream = plan.execute(part).await?;
//! ```
//!
//! and everything above it (write_stream_to_disk, the `{path, stats}` reply, Flight) is unchanged.  The pattern of an in-tree
//! `ExecutionPlan` is rust/core/src/execution_plans/query_stage.rs:49-85; the stream side is
//! rust/core/src/memory_stream.rs:57-92.
//!
//! How a subtree travels: Ballista already knows how to serialise a physical plan
//! (`TryInto<protobuf::PhysicalPlanNode> for Arc<dyn ExecutionPlan>`, rust/core/src/serde/physical_plan/to_proto.rs:60-346).
//! `GpuExec` keeps those bytes; `bhip_plan_from_proto` rebuilds the operators inside the library.  The leaves of the subtree
//! (CsvExec, ParquetExec, ShuffleReaderExec) stay CPU operators: the library's leaf resolver callback receives each leaf in
//! depth-first order and gets back a `bhip_plan_arrow_streams` leaf over Arrow C streams of that CPU operator's partitions
//! (`cstream::export`).  A subtree the library declines (`BHIP_ENOTIMPL`: a type, expression or operator outside the GPU path)
//! is left as it is and the rule recurses into its children.
//!
//! SOURCE ONLY: the build image has no Rust toolchain; this crate is written against the reference's pinned dependency
//! versions and has not been compiled.

pub mod cstream;
pub mod ffi;

use std::any::Any;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::pin::Pin;
use std::sync::{Arc, Mutex};
use std::task::{Context, Poll};

use arrow::array::{make_array_from_raw, Array, StructArray};
use arrow::datatypes::SchemaRef;
use arrow::error::{ArrowError, Result as ArrowResult};
use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use arrow::record_batch::RecordBatch;
use async_trait::async_trait;
use ballista_core::serde::protobuf;
use datafusion::error::{DataFusionError, Result};
use datafusion::physical_plan::{ExecutionPlan, Partitioning, RecordBatchStream};
use futures::Stream;
use prost::Message;
use std::convert::TryInto;

/// One GPU of the node (`bhip_ctx`): allocator + stream pool.  Shared by every task of the executor process
/// (`concurrent_tasks`, rust/executor/executor_config_spec.toml:57-62); the library is re-entrant.
pub struct GpuContext {
    raw: *mut ffi::bhip_ctx,
}
unsafe impl Send for GpuContext {}
unsafe impl Sync for GpuContext {}

impl GpuContext {
    pub fn try_new(device: i32) -> Result<Arc<Self>> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::bhip_ctx_create(device, &mut raw) })?;
        Ok(Arc::new(Self { raw }))
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { ffi::bhip_ctx_release(self.raw) }
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::bhip_last_error()) }.to_string_lossy().into_owned()
}

/// status -> the error kinds the executor already maps to tonic::Status::internal (flight_service.rs:344-354)
fn check(status: ffi::bhip_status) -> Result<()> {
    match status {
        ffi::BHIP_OK => Ok(()),
        ffi::BHIP_ENOTIMPL => Err(DataFusionError::NotImplemented(last_error())),
        1 => Err(DataFusionError::Plan(last_error())),
        _ => Err(DataFusionError::Execution(last_error())),
    }
}

struct PlanHandle(*mut ffi::bhip_plan);
unsafe impl Send for PlanHandle {}
unsafe impl Sync for PlanHandle {}
impl Drop for PlanHandle {
    fn drop(&mut self) {
        unsafe { ffi::bhip_plan_release(self.0) }
    }
}

/// A query-stage subtree that runs on the GPU.
pub struct GpuExec {
    /// the CPU plan this node stands for: schema / partitioning / children are answered from it, and it is what
    /// `with_new_children` rebuilds from
    cpu: Arc<dyn ExecutionPlan>,
    /// scan / shuffle leaves of `cpu` in depth-first order
    leaves: Vec<Arc<dyn ExecutionPlan>>,
    gpu: Arc<GpuContext>,
    /// built by the first `execute` (it needs a tokio runtime handle for the leaf streams) and shared by every partition's
    /// task afterwards, so a join's build side is built once (the reference's collect-left build future)
    plan: Mutex<Option<Arc<PlanHandle>>>,
    bytes: Vec<u8>,
}

impl std::fmt::Debug for GpuExec {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "GpuExec {{ {:?} }}", self.cpu)
    }
}

fn collect_leaves(plan: &Arc<dyn ExecutionPlan>, out: &mut Vec<Arc<dyn ExecutionPlan>>) {
    let children = plan.children();
    if children.is_empty() {
        out.push(plan.clone());
    }
    for c in &children {
        collect_leaves(c, out);
    }
}

struct ResolverState {
    leaves: Vec<Arc<dyn ExecutionPlan>>,
    next: usize,
    gpu: Arc<GpuContext>,
    runtime: tokio::runtime::Handle,
    error: Option<String>,
}

/// `bhip_leaf_resolver`: leaf number `next` of the wire plan is leaf number `next` of the CPU plan (both are walked depth
/// first, left to right) -> an Arrow-stream leaf with one stream per output partition of the CPU operator.
unsafe extern "C" fn resolve_leaf(user: *mut c_void, _leaf: *const ffi::bhip_leaf_desc, out: *mut *mut ffi::bhip_plan) -> ffi::bhip_status {
    let st = &mut *(user as *mut ResolverState);
    let leaf = match st.leaves.get(st.next) {
        Some(l) => l.clone(),
        None => {
            st.error = Some("the wire plan has more leaves than the CPU plan".to_owned());
            return 1;
        }
    };
    st.next += 1;
    let n = leaf.output_partitioning().partition_count();
    let mut streams: Vec<ffi::ArrowArrayStream> = (0..n).map(|p| cstream::export(leaf.clone(), p, st.runtime.clone())).collect();
    let ptrs: Vec<*mut ffi::ArrowArrayStream> = streams.iter_mut().map(|s| s as *mut _).collect();
    let status = ffi::bhip_plan_arrow_streams(st.gpu.raw, n as i32, ptrs.as_ptr(), out);
    // streams the library did not take over (an error) are still ours
    for s in streams.iter_mut() {
        if let Some(rel) = s.release {
            rel(s);
        }
    }
    status
}

impl GpuExec {
    /// `Err(NotImplemented)` when the subtree cannot be serialised or the library declines it: keep the CPU plan.
    pub fn try_new(cpu: Arc<dyn ExecutionPlan>, gpu: Arc<GpuContext>) -> Result<Self> {
        let node: protobuf::PhysicalPlanNode = cpu
            .clone()
            .try_into()
            .map_err(|e| DataFusionError::NotImplemented(format!("{:?}", e)))?;
        let mut bytes = Vec::with_capacity(node.encoded_len());
        node.encode(&mut bytes).map_err(|e| DataFusionError::Internal(format!("{:?}", e)))?;
        // dry run without a device context: type / expression / operator coverage is decided at plan time
        let mut probe = std::ptr::null_mut();
        check(unsafe { ffi::bhip_plan_from_proto(std::ptr::null_mut(), bytes.as_ptr() as *const c_void, bytes.len(), None, std::ptr::null_mut(), &mut probe) })?;
        unsafe { ffi::bhip_plan_release(probe) };
        let mut leaves = vec![];
        collect_leaves(&cpu, &mut leaves);
        Ok(Self { cpu, leaves, gpu, plan: Mutex::new(None), bytes })
    }

    fn gpu_plan(&self) -> Result<Arc<PlanHandle>> {
        let mut guard = self.plan.lock().unwrap();
        if let Some(p) = guard.as_ref() {
            return Ok(p.clone());
        }
        let mut state = ResolverState {
            leaves: self.leaves.clone(),
            next: 0,
            gpu: self.gpu.clone(),
            runtime: tokio::runtime::Handle::current(),
            error: None,
        };
        let mut raw = std::ptr::null_mut();
        let status = unsafe {
            ffi::bhip_plan_from_proto(
                self.gpu.raw,
                self.bytes.as_ptr() as *const c_void,
                self.bytes.len(),
                Some(resolve_leaf),
                &mut state as *mut ResolverState as *mut c_void,
                &mut raw,
            )
        };
        if let Some(e) = state.error.take() {
            return Err(DataFusionError::Internal(e));
        }
        check(status)?;
        let p = Arc::new(PlanHandle(raw));
        *guard = Some(p.clone());
        Ok(p)
    }
}

/// What `write_stream_to_disk` reports for a stage partition (rust/core/src/utils.rs:49-84, `PartitionStats`).
#[derive(Debug, Clone, Copy, Default)]
pub struct StageFileStats {
    pub num_rows: u64,
    pub num_batches: u64,
    pub num_bytes: u64,
}

impl GpuExec {
    /// One stage partition straight into its Arrow IPC file: the executor's
    /// `plan.execute(partition)` + `utils::write_stream_to_disk(&mut stream, path)` (rust/executor/src/flight_service.rs:104-150)
    /// as one library call — the batches go from HBM to the file writer without becoming host `RecordBatch`es in between.
    pub async fn execute_to_file(&self, partition: usize, path: &str) -> Result<StageFileStats> {
        let plan = self.gpu_plan()?;
        let cpath = std::ffi::CString::new(path).map_err(|e| DataFusionError::Execution(format!("{:?}", e)))?;
        tokio::task::spawn_blocking(move || -> Result<StageFileStats> {
            let mut raw = std::ptr::null_mut();
            check(unsafe { ffi::bhip_plan_execute(plan.0, partition as i32, &mut raw) })?;
            let mut st = StageFileStats::default();
            // bhip_stream_write_ipc CONSUMES the stream on every path (include/ballista_hip.h): no bhip_stream_release here —
            // a second release would free it twice (tests/c/shim_sequence.c pins this ownership rule under the host ASan build)
            let status = unsafe { ffi::bhip_stream_write_ipc(raw, cpath.as_ptr(), &mut st.num_rows, &mut st.num_batches, &mut st.num_bytes) };
            check(status)?;
            Ok(st)
        })
        .await
        .map_err(|e| DataFusionError::Execution(format!("{:?}", e)))?
    }
}

#[async_trait]
impl ExecutionPlan for GpuExec {
    fn as_any(&self) -> &dyn Any {
        self
    }

    fn schema(&self) -> SchemaRef {
        self.cpu.schema()
    }

    fn output_partitioning(&self) -> Partitioning {
        self.cpu.output_partitioning()
    }

    /// the CPU leaves: what a parent rule may still want to rewrite (e.g. resolve shuffle readers)
    fn children(&self) -> Vec<Arc<dyn ExecutionPlan>> {
        self.cpu.children()
    }

    fn with_new_children(&self, children: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        let cpu = self.cpu.with_new_children(children)?;
        Ok(Arc::new(GpuExec::try_new(cpu, self.gpu.clone())?))
    }

    async fn execute(&self, partition: usize) -> Result<Pin<Box<dyn RecordBatchStream + Send + Sync>>> {
        let plan = self.gpu_plan()?;
        let schema = self.schema();
        // bhip_plan_execute is blocking (it may drain CPU leaves and build join tables): off the async workers
        let stream = tokio::task::spawn_blocking(move || -> Result<GpuStream> {
            let mut raw = std::ptr::null_mut();
            check(unsafe { ffi::bhip_plan_execute(plan.0, partition as i32, &mut raw) })?;
            Ok(GpuStream { raw, schema, _plan: plan })
        })
        .await
        .map_err(|e| DataFusionError::Execution(format!("{:?}", e)))??;
        Ok(Box::pin(stream))
    }
}

/// `bhip_stream` as a `RecordBatchStream`: every `poll_next` pulls one device batch, copies it to host memory and imports it
/// through the C Data Interface.
pub struct GpuStream {
    raw: *mut ffi::bhip_stream,
    schema: SchemaRef,
    _plan: Arc<PlanHandle>,
}
unsafe impl Send for GpuStream {}
unsafe impl Sync for GpuStream {}

impl Drop for GpuStream {
    fn drop(&mut self) {
        unsafe { ffi::bhip_stream_release(self.raw) }
    }
}

impl GpuStream {
    fn next_batch(&mut self) -> Option<ArrowResult<RecordBatch>> {
        let mut batch = std::ptr::null_mut();
        if unsafe { ffi::bhip_stream_next(self.raw, &mut batch) } != ffi::BHIP_OK {
            return Some(Err(ArrowError::ExternalError(Box::new(DataFusionError::Execution(last_error())))));
        }
        if batch.is_null() {
            return None;
        }
        // Arc-allocated so that arrow's importer can take them over
        let array = Arc::into_raw(Arc::new(FFI_ArrowArray::empty())) as *mut FFI_ArrowArray;
        let schema = Arc::into_raw(Arc::new(FFI_ArrowSchema::empty())) as *mut FFI_ArrowSchema;
        let status = unsafe { ffi::bhip_batch_export_arrow(batch, array, schema) };
        unsafe { ffi::bhip_batch_release(batch) };
        if status != ffi::BHIP_OK {
            unsafe {
                drop(Arc::from_raw(array));
                drop(Arc::from_raw(schema));
            }
            return Some(Err(ArrowError::ExternalError(Box::new(DataFusionError::Execution(last_error())))));
        }
        Some(unsafe { make_array_from_raw(array, schema) }.and_then(|a| {
            let sa = a
                .as_any()
                .downcast_ref::<StructArray>()
                .ok_or_else(|| ArrowError::CDataInterface("expected a struct array".to_owned()))?;
            // the library's field names / nullability are the plan's; reuse the plan's schema object
            RecordBatch::try_new(self.schema.clone(), sa.columns().iter().map(|c| (*c).clone()).collect())
        }))
    }
}

impl Stream for GpuStream {
    type Item = ArrowResult<RecordBatch>;

    fn poll_next(mut self: Pin<&mut Self>, _: &mut Context<'_>) -> Poll<Option<Self::Item>> {
        // blocking inside poll: the executor drains this stream from `write_stream_to_disk` on a task of its own
        // (rust/core/src/utils.rs:49-84); `tokio::task::block_in_place` tells the runtime so
        Poll::Ready(tokio::task::block_in_place(|| self.next_batch()))
    }
}

impl RecordBatchStream for GpuStream {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
}

/// The offload rule: the largest subtrees the library accepts become `GpuExec`s; everything else stays as DataFusion built it.
/// Applied by the executor to `partition.plan` right before `execute` (rust/executor/src/flight_service.rs:117-121).
pub fn offload(plan: Arc<dyn ExecutionPlan>, gpu: &Arc<GpuContext>) -> Result<Arc<dyn ExecutionPlan>> {
    // a bare leaf gains nothing from a round trip through the device
    if plan.children().is_empty() {
        return Ok(plan);
    }
    match GpuExec::try_new(plan.clone(), gpu.clone()) {
        Ok(exec) => Ok(Arc::new(exec)),
        // NotImplemented = the library declines the subtree; Plan = its dry run refused something the CPU operators may still
        // accept (a coercion or operator it does not model).  Either way the CPU plan stays and reports its own errors.
        Err(DataFusionError::NotImplemented(why)) | Err(DataFusionError::Plan(why)) => {
            log::debug!("not offloaded ({}): {:?}", why, plan);
            let children = plan.children().into_iter().map(|c| offload(c, gpu)).collect::<Result<Vec<_>>>()?;
            plan.with_new_children(children)
        }
        Err(e) => Err(e),
    }
}
