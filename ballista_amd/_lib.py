"""ctypes binding of libballista_hip.so (include/ballista_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this
module raises, and every call that needs a GPU fails with the library's own error.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BHIP_LIB_PATH") or os.path.join(_HERE, "lib", "libballista_hip.so")

OK, EINVAL, ENOTIMPL, EEXEC, EHIP, EOOM = 0, 1, 2, 3, 4, 5


class BallistaError(Exception):
    """Mirror of the executor-visible error kinds (rust/core/src/error.rs:30-163)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class NotImplementedOnGpu(BallistaError):
    """BHIP_ENOTIMPL — DataFusionError::NotImplemented: keep the CPU operator for this subtree."""


class PlanError(BallistaError):
    """BHIP_EINVAL — DataFusionError::Plan / Internal."""


class ExecutionError(BallistaError):
    """BHIP_EEXEC — DataFusionError::Execution / ArrowError."""


class HipError(BallistaError):
    """BHIP_EHIP / BHIP_EOOM."""


class ColumnDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("nullable", C.c_int32), ("data", C.c_void_p),
                ("offsets", C.c_void_p), ("validity", C.c_void_p), ("data_bytes", C.c_int64)]


class ExprNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("n_args", C.c_int32), ("flags", C.c_int32),
                ("name", C.c_char_p), ("i64", C.c_int64), ("f64", C.c_double)]


class Expr(C.Structure):
    _fields_ = [("nodes", C.POINTER(ExprNode)), ("n_nodes", C.c_int32)]


class Aggregate(C.Structure):
    _fields_ = [("fn", C.c_int32), ("arg", Expr), ("name", C.c_char_p)]


class PartitionLocation(C.Structure):
    _fields_ = [("job_id", C.c_char_p), ("stage_id", C.c_uint32), ("partition_id", C.c_uint32), ("executor_id", C.c_char_p),
                ("host", C.c_char_p), ("port", C.c_uint32), ("num_rows", C.c_int64), ("num_batches", C.c_int64), ("num_bytes", C.c_int64)]


class LeafDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("path", C.c_char_p), ("n_filenames", C.c_int32), ("filenames", C.POINTER(C.c_char_p)),
                ("has_projection", C.c_int32), ("n_projection", C.c_int32), ("projection", C.POINTER(C.c_uint32)),
                ("n_fields", C.c_int32), ("fields", C.POINTER(ColumnDesc)), ("has_header", C.c_int32), ("delimiter", C.c_char_p),
                ("file_extension", C.c_char_p), ("batch_size", C.c_uint32), ("num_partitions", C.c_uint32),
                ("n_locations", C.c_int32), ("locations", C.POINTER(PartitionLocation)), ("n_stage_ids", C.c_int32),
                ("stage_ids", C.POINTER(C.c_uint32)), ("partition_count", C.c_uint32)]


class CommRegion(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_uint64), ("peer", C.c_int32)]


HOST_ALL_GATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
HOST_EXCHANGE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(CommRegion), C.c_int32, C.POINTER(CommRegion))


class CommHostTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", HOST_ALL_GATHER), ("exchange", HOST_EXCHANGE)]


SHUFFLE_MAX_PEERS = 64


class ShuffleStats(C.Structure):
    _fields_ = [("rows_in", C.c_uint64), ("rows_out", C.c_uint64), ("chunks", C.c_uint64), ("streamed", C.c_uint64),
                ("bytes_sent_remote", C.c_uint64), ("bytes_kept_local", C.c_uint64), ("staging_bytes", C.c_uint64),
                ("rows_to", C.c_uint64 * SHUFFLE_MAX_PEERS), ("ms_count", C.c_double), ("ms_total", C.c_double)]


class TpchOpts(C.Structure):
    _fields_ = [("key64", C.c_int32), ("with_dates", C.c_int32), ("sparse_keys", C.c_int32), ("n_columns", C.c_int32),
                ("key_base", C.c_int64), ("columns", C.POINTER(C.c_char_p))]


LEAF_RESOLVER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(LeafDesc), C.POINTER(C.c_void_p))


class SortExprC(C.Structure):
    _fields_ = [("expr", Expr), ("descending", C.c_int32), ("nulls_first", C.c_int32)]


# every symbol include/ballista_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "bhip_last_error": (C.c_char_p, []),
    "bhip_version": (C.c_char_p, []),
    "bhip_ctx_create": (C.c_int32, [C.c_int, _PP]),
    "bhip_ctx_release": (None, [_P]),
    "bhip_ctx_synchronize": (C.c_int32, [_P]),
    "bhip_ctx_memory": (C.c_int32, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bhip_ctx_kernel_time": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "bhip_ctx_kernel_name": (C.c_char_p, [_P]),
    "bhip_ctx_kernel_stats": (C.c_int32, [_P, C.c_int32, C.c_char_p, C.c_size_t]),
    "bhip_batch_from_host": (C.c_int32, [_P, C.c_int32, C.POINTER(ColumnDesc), C.c_int64, _PP]),
    "bhip_batch_from_device": (C.c_int32, [_P, C.c_int32, C.POINTER(ColumnDesc), C.c_int64, _PP]),
    "bhip_batch_from_tbl": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, C.POINTER(ColumnDesc), C.c_int32, C.POINTER(C.c_int32), _PP]),
    "bhip_batch_import_arrow": (C.c_int32, [_P, _P, _P, _PP]),
    "bhip_batch_export_arrow": (C.c_int32, [_P, _P, _P]),
    "bhip_batch_retain": (None, [_P]),
    "bhip_batch_release": (None, [_P]),
    "bhip_batch_num_rows": (C.c_int64, [_P]),
    "bhip_batch_num_columns": (C.c_int32, [_P]),
    "bhip_batch_column_info": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "bhip_batch_column_device": (C.c_int32, [_P, C.c_int32, _PP, _PP, _PP]),
    "bhip_batch_column_to_host": (C.c_int32, [_P, C.c_int32, _P, _P, _P]),
    "bhip_batch_memory_size": (C.c_int64, [_P]),
    "bhip_plan_memory": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_int32), _PP, _PP]),
    "bhip_plan_arrow_stream": (C.c_int32, [_P, _P, _PP]),
    "bhip_plan_arrow_streams": (C.c_int32, [_P, C.c_int32, _PP, _PP]),
    "bhip_plan_parquet": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_uint32), C.c_int32, _PP]),
    "bhip_plan_empty": (C.c_int32, [_P, C.c_int32, C.POINTER(ColumnDesc), C.c_int32, _PP]),
    "bhip_plan_filter": (C.c_int32, [_P, C.POINTER(Expr), _PP]),
    "bhip_plan_projection": (C.c_int32, [_P, C.c_int32, C.POINTER(Expr), C.POINTER(C.c_char_p), _PP]),
    "bhip_plan_hash_aggregate": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(Expr), C.POINTER(C.c_char_p),
                                             C.c_int32, C.POINTER(Aggregate), _PP]),
    "bhip_plan_hash_join": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, _PP]),
    "bhip_plan_sort": (C.c_int32, [_P, C.c_int32, C.POINTER(SortExprC), _PP]),
    "bhip_plan_repartition": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(Expr), C.c_int32, _PP]),
    "bhip_plan_coalesce_batches": (C.c_int32, [_P, C.c_int64, _PP]),
    "bhip_plan_merge": (C.c_int32, [_P, _PP]),
    "bhip_plan_global_limit": (C.c_int32, [_P, C.c_int64, _PP]),
    "bhip_plan_local_limit": (C.c_int32, [_P, C.c_int64, _PP]),
    "bhip_plan_from_proto": (C.c_int32, [_P, C.c_char_p, C.c_size_t, _P, _P, _PP]),
    "bhip_expr_from_proto_display": (C.c_int32, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "bhip_plan_retain": (None, [_P]),
    "bhip_plan_release": (None, [_P]),
    "bhip_plan_name": (C.c_char_p, [_P]),
    "bhip_plan_schema": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]),
    "bhip_plan_output_partitioning": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bhip_plan_children": (C.c_int32, [_P, C.c_int32, _PP, C.POINTER(C.c_int32)]),
    "bhip_plan_with_new_children": (C.c_int32, [_P, C.c_int32, _PP, _PP]),
    "bhip_plan_execute": (C.c_int32, [_P, C.c_int32, _PP]),
    "bhip_plan_collect": (C.c_int32, [_P, C.c_int32, _PP, C.POINTER(C.c_int32)]),
    "bhip_plan_display": (C.c_int32, [_P, C.c_char_p, C.c_size_t]),
    "bhip_stream_next": (C.c_int32, [_P, _PP]),
    "bhip_stream_schema": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]),
    "bhip_stream_release": (None, [_P]),
    "bhip_stream_export_arrow": (C.c_int32, [_P, _P]),
    "bhip_stream_drain": (C.c_int32, [_P, _P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bhip_batch_hash_partition": (C.c_int32, [_P, C.c_int32, C.POINTER(Expr), C.c_int32, _PP]),
    "bhip_batch_concat": (C.c_int32, [_P, C.c_int32, _PP, _PP]),
    "bhip_stream_write_ipc": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bhip_plan_ipc_files": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), _PP]),
    "bhip_ipc_write_file": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bhip_ipc_open_file": (C.c_int32, [C.c_char_p, _P]),
    "bhip_comm_unique_id": (C.c_int32, [C.c_char_p]),
    "bhip_comm_create": (C.c_int32, [_P, C.c_char_p, C.c_int32, C.c_int32, _PP]),
    "bhip_comm_create_loopback": (C.c_int32, [_P, C.c_char_p, C.c_int32, C.c_int32, _PP]),
    "bhip_comm_create_host": (C.c_int32, [_P, C.POINTER(CommHostTransport), C.c_int32, C.c_int32, _PP]),
    "bhip_comm_release": (None, [_P]),
    "bhip_comm_info": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]),
    "bhip_comm_stats": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bhip_comm_shuffle": (C.c_int32, [_P, _P, C.c_char_p, C.c_int64, _PP, C.POINTER(ShuffleStats)]),
    "bhip_plan_all_gather": (C.c_int32, [_P, _P, _PP]),
    "bhip_plan_shuffle": (C.c_int32, [_P, _P, C.c_char_p, C.c_int64, _PP]),
    "bhip_comm_all_gather": (C.c_int32, [_P, _P, _PP]),
    "bhip_comm_all_to_all": (C.c_int32, [_P, _PP, _PP]),
    "bhip_batch_pack": (C.c_int32, [_P, C.POINTER(C.c_int64), C.c_int32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "bhip_batch_unpack": (C.c_int32, [_P, C.c_int32, C.POINTER(ColumnDesc), C.POINTER(C.c_int64), _P, _PP]),
    "bhip_tpch_lineitem": (C.c_int32, [_P, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _PP]),
    "bhip_tpch_orders": (C.c_int32, [_P, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, _PP]),
    "bhip_tpch_lineitem_opts": (C.c_int32, [_P, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(TpchOpts), _PP]),
    "bhip_tpch_orders_opts": (C.c_int32, [_P, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(TpchOpts), _PP]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it was not built — never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(_lib, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(status):
    if status == OK:
        return
    msg = lib().bhip_last_error().decode("utf-8", "replace")
    cls = {EINVAL: PlanError, ENOTIMPL: NotImplementedOnGpu, EEXEC: ExecutionError}.get(status, HipError)
    raise cls(status, msg)
