"""ballista_amd — MI355X-native physical execution layer for Ballista's executor hot path.

`ballista_amd.plan` mirrors the DataFusion `ExecutionPlan` operator interface the reference's
executor calls (rust/executor/src/flight_service.rs:117-121); all compute runs in
`lib/libballista_hip.so` (hand-written HIP kernels for gfx950) behind the C ABI of
`include/ballista_hip.h`.  There is no CPU fallback: without the built library, or without a
GPU, calls fail with the library's error.
"""
import os as _os

# kernel arguments in device memory (csrc/host/core.cpp sets the same default when the library loads; here for a process that
# initialises HIP — e.g. through torch — before the library is opened)
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import expr, plan, tpch  # noqa: F401,E402
from ._lib import (BallistaError, ExecutionError, HipError, NotImplementedOnGpu, PlanError, LIB_PATH)  # noqa: F401,E402
from .plan import (Context, RecordBatch, RecordBatchStream, ExecutionPlan, Partitioning, MemoryExec, FilterExec,  # noqa: F401,E402
                   ProjectionExec, HashAggregateExec, HashJoinExec, SortExec, RepartitionExec, CoalesceBatchesExec,
                   MergeExec, GlobalLimitExec, LocalLimitExec, ArrowStreamExec, ParquetExec, IpcFileExec)

__version__ = "0.1.0"
