"""Host-side mirror of the operator interface a Ballista executor drives.

`ExecutionPlan` here has the same five methods as DataFusion's trait as the reference uses it
(rust/core/src/execution_plans/query_stage.rs:49-85): `schema()`, `output_partitioning()`,
`children()`, `with_new_children()`, `execute(partition)` -> `RecordBatchStream`
(rust/core/src/memory_stream.rs:57-92).  Operator constructors take their arguments in the
order the physical-plan serde passes them (rust/core/src/serde/physical_plan/from_proto.rs:
58-346).  All compute happens in libballista_hip.so through the C ABI; these classes only
describe plans and move batches across the boundary.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from . import expr as E

DTYPE_ID = {E.INT32: 1, E.INT64: 2, E.UINT8: 3, E.UINT64: 4, E.FLOAT64: 5, E.DATE32: 6, E.BOOLEAN: 7, E.UTF8: 8,
            "Int8": 9, "Int16": 10, "UInt16": 11, "UInt32": 12, "Float32": 13, "Date64": 14, "Timestamp(Second)": 15,
            "Timestamp(Millisecond)": 16, "Timestamp(Microsecond)": 17, "Timestamp(Nanosecond)": 18,
            # schemas only: a Utf8 column whose Arrow / IPC form has 64-bit offsets (on the device it is Utf8)
            "LargeUtf8": 19, "Binary": 20}
DTYPE_NAME = {v: k for k, v in DTYPE_ID.items()}
NP_DTYPE = {E.INT32: np.int32, E.INT64: np.int64, E.UINT8: np.uint8, E.UINT64: np.uint64,
            E.FLOAT64: np.float64, E.DATE32: np.int32, E.INT8: np.int8, E.INT16: np.int16, E.UINT16: np.uint16,
            E.UINT32: np.uint32, E.FLOAT32: np.float32, E.DATE64: np.int64, E.TIMESTAMP_S: np.int64,
            E.TIMESTAMP_MS: np.int64, E.TIMESTAMP_US: np.int64, E.TIMESTAMP_NS: np.int64}

PARTIAL, FINAL = "Partial", "Final"
INNER, LEFT, RIGHT = "Inner", "Left", "Right"


class Partitioning:
    """datafusion Partitioning::{UnknownPartitioning(n), RoundRobinBatch(n), Hash(exprs, n)}
    (from_proto.rs:143-158)."""
    UNKNOWN, ROUND_ROBIN, HASH = 0, 1, 2

    def __init__(self, scheme, count, exprs=()):
        self.scheme, self.count, self.exprs = scheme, count, list(exprs)

    @staticmethod
    def Hash(exprs, n): return Partitioning(Partitioning.HASH, n, exprs)
    @staticmethod
    def RoundRobinBatch(n): return Partitioning(Partitioning.ROUND_ROBIN, n)
    @staticmethod
    def UnknownPartitioning(n): return Partitioning(Partitioning.UNKNOWN, n)

    def partition_count(self): return self.count

    def __repr__(self):
        return {0: "UnknownPartitioning", 1: "RoundRobinBatch", 2: "Hash"}[self.scheme] + f"({self.count})"


class Context:
    """One GPU (bhip_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        L.check(L.lib().bhip_ctx_create(device, C.byref(h)))
        self._h = h

    def synchronize(self):
        L.check(L.lib().bhip_ctx_synchronize(self._h))

    def memory(self):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(L.lib().bhip_ctx_memory(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        L.check(L.lib().bhip_ctx_kernel_time(self._h, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_name(self):
        return L.lib().bhip_ctx_kernel_name(self._h).decode()

    def kernel_stats(self, reset=False):
        """{kernel name: (total ms, launches, algorithmic bytes)} of every kernel timed since the last reset (BHIP_KERNEL_TIMING=1)"""
        buf = C.create_string_buffer(1 << 16)
        L.check(L.lib().bhip_ctx_kernel_stats(self._h, 1 if reset else 0, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, n, nbytes = line.split("\t")
            out[name] = (float(ms), int(n), int(nbytes))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().bhip_ctx_release(self._h)
                self._h = None
        except Exception:
            pass


# ---- expression lowering: tree -> postfix bhip_expr_node[] -------------------------------------

_EXPR_KIND = dict(COLUMN=1, LITERAL=2, BINARY=3, CAST=4, NOT=5, IS_NULL=6, IS_NOT_NULL=7, NEGATIVE=8,
                  IN_LIST=9, CASE=10, SCALAR_FN=11)


class _Lowered:
    """owns the ctypes storage of lowered expressions until the C call returned"""

    def __init__(self):
        self.keep = []

    def expr(self, e) -> L.Expr:
        nodes = []
        self._walk(e, nodes)
        arr = (L.ExprNode * len(nodes))(*nodes)
        self.keep.append(arr)
        return L.Expr(C.cast(arr, C.POINTER(L.ExprNode)), len(nodes))

    def exprs(self, es):
        arr = (L.Expr * max(1, len(es)))(*[self.expr(e) for e in es])
        self.keep.append(arr)
        return arr

    def strings(self, ss):
        arr = (C.c_char_p * max(1, len(ss)))(*[s.encode() for s in ss])
        self.keep.append(arr)
        return arr

    def _node(self, kind, dtype=0, n_args=0, flags=0, name=None, i64=0, f64=0.0):
        b = name.encode() if name is not None else None
        self.keep.append(b)
        return L.ExprNode(_EXPR_KIND[kind], dtype, n_args, flags, b, i64, f64)

    def _walk(self, e, out):
        if isinstance(e, E.Column):
            out.append(self._node("COLUMN", name=e.name))
        elif isinstance(e, E.Literal):
            dt = DTYPE_ID[e.dtype]
            if e.value is None:
                out.append(self._node("LITERAL", dtype=dt, flags=1))
            elif e.dtype == E.UTF8:
                out.append(self._node("LITERAL", dtype=dt, name=e.value))
            elif e.dtype in (E.FLOAT64, E.FLOAT32):
                out.append(self._node("LITERAL", dtype=dt, f64=float(e.value)))
            elif e.dtype == E.UINT64:
                v = int(e.value)
                out.append(self._node("LITERAL", dtype=dt, i64=v - (1 << 64) if v >= (1 << 63) else v))
            else:
                out.append(self._node("LITERAL", dtype=dt, i64=int(e.value)))
        elif isinstance(e, E.BinaryExpr):
            self._walk(e.left, out)
            self._walk(e.right, out)
            out.append(self._node("BINARY", name=e.op))
        elif isinstance(e, E.CastExpr):
            self._walk(e.expr, out)
            out.append(self._node("CAST", dtype=DTYPE_ID[e.dtype]))
        elif isinstance(e, E.NotExpr):
            self._walk(e.expr, out)
            out.append(self._node("NOT"))
        elif isinstance(e, E.IsNullExpr):
            self._walk(e.expr, out)
            out.append(self._node("IS_NULL"))
        elif isinstance(e, E.IsNotNullExpr):
            self._walk(e.expr, out)
            out.append(self._node("IS_NOT_NULL"))
        elif isinstance(e, E.NegativeExpr):
            self._walk(e.expr, out)
            out.append(self._node("NEGATIVE"))
        elif isinstance(e, E.InListExpr):
            self._walk(e.expr, out)
            for item in e.list:
                self._walk(item, out)
            out.append(self._node("IN_LIST", n_args=len(e.list), flags=1 if e.negated else 0))
        elif isinstance(e, E.CaseExpr):
            flags = 0
            if e.expr is not None:
                self._walk(e.expr, out)
                flags |= 1
            for w, t in e.when_then:
                self._walk(w, out)
                self._walk(t, out)
            if e.else_expr is not None:
                self._walk(e.else_expr, out)
                flags |= 2
            out.append(self._node("CASE", n_args=len(e.when_then), flags=flags))
        elif isinstance(e, E.ScalarFunctionExpr):
            for a in e.args:
                self._walk(a, out)
            out.append(self._node("SCALAR_FN", n_args=len(e.args), name=e.fun))
        else:
            raise TypeError(f"not a PhysicalExpr: {e!r}")


# ---- Arrow C Data / C Stream interface structs (layout of include/ballista_hip.h) --------------------

class _ArrowSchema(C.Structure):
    pass


_ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                         ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                         ("release", C.CFUNCTYPE(None, C.POINTER(_ArrowSchema))), ("private_data", C.c_void_p)]


class _ArrowArray(C.Structure):
    pass


_ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                        ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p),
                        ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.POINTER(_ArrowArray))),
                        ("private_data", C.c_void_p)]


class _ArrowArrayStream(C.Structure):
    _fields_ = [("get_schema", C.c_void_p), ("get_next", C.c_void_p), ("get_last_error", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


# ---- record batches ---------------------------------------------------------------------------------

class RecordBatch:
    """A device-resident Arrow RecordBatch (bhip_batch)."""

    def __init__(self, handle, ctx: Context):
        self._h = handle
        self.ctx = ctx

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().bhip_batch_release(self._h)
                self._h = None
        except Exception:
            pass

    # -- constructors
    @staticmethod
    def from_columns(ctx: Context, columns) -> "RecordBatch":
        """columns: list of (name, dtype, values, validity) — values: numpy array, or for Utf8 a list
        of str / None-free str list; Boolean: bool array; validity: bool array or None."""
        descs, keep, n_rows = [], [], None
        for name, dtype, values, validity in columns:
            d = L.ColumnDesc()
            nb = name.encode()
            keep.append(nb)
            d.name = nb
            d.dtype = DTYPE_ID[dtype]
            d.nullable = 1 if validity is not None else 0
            n = len(values)
            n_rows = n if n_rows is None else n_rows
            if n != n_rows:
                raise ValueError("columns of different lengths")
            if dtype == E.UTF8:
                enc = [("" if s is None else s).encode() for s in values]
                off = np.zeros(n + 1, np.int32)
                if n:
                    off[1:] = np.cumsum([len(b) for b in enc])
                data = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
                keep += [off, data]
                d.offsets = off.ctypes.data
                d.data = data.ctypes.data
                d.data_bytes = int(off[-1])
            elif dtype == E.BOOLEAN:
                bits = np.packbits(np.asarray(values, np.bool_), bitorder="little")
                bits = np.concatenate([bits, np.zeros(8, np.uint8)])
                keep.append(bits)
                d.data = bits.ctypes.data
            else:
                arr = np.ascontiguousarray(values, NP_DTYPE[dtype])
                if arr.size == 0:
                    arr = np.zeros(1, NP_DTYPE[dtype])
                keep.append(arr)
                d.data = arr.ctypes.data
            if validity is not None:
                vb = np.packbits(np.asarray(validity, np.bool_), bitorder="little")
                vb = np.concatenate([vb, np.zeros(8, np.uint8)])
                keep.append(vb)
                d.validity = vb.ctypes.data
            descs.append(d)
        arr = (L.ColumnDesc * max(1, len(descs)))(*descs)
        h = C.c_void_p()
        L.check(L.lib().bhip_batch_from_host(ctx._h, len(descs), arr, n_rows or 0, C.byref(h)))
        return RecordBatch(h, ctx)

    @staticmethod
    def from_device_columns(ctx: Context, owner: "RecordBatch") -> "RecordBatch":
        """bhip_batch_from_device: a batch that BORROWS the device buffers of `owner` (zero copy; the
        result keeps `owner` alive).  Any caller-owned device memory laid out as Arrow works the same."""
        descs, keep = [], []
        for i in range(owner.num_columns):
            name, dtype, nullable, nbytes, _ = owner.column_info(i)
            data, offsets, validity = owner.column_device(i)
            d = L.ColumnDesc()
            nb = name.encode()
            keep.append(nb)
            d.name, d.dtype, d.nullable = nb, DTYPE_ID[dtype], 1 if nullable else 0
            d.data, d.offsets, d.validity, d.data_bytes = data, offsets, validity, nbytes
            descs.append(d)
        arr = (L.ColumnDesc * max(1, len(descs)))(*descs)
        h = C.c_void_p()
        L.check(L.lib().bhip_batch_from_device(ctx._h, len(descs), arr, owner.num_rows, C.byref(h)))
        rb = RecordBatch(h, ctx)
        rb._owner = owner
        return rb

    @staticmethod
    def from_tbl(ctx: Context, text, schema, columns=None) -> "RecordBatch":
        """bhip_batch_from_tbl: TPC-H `.tbl` text ('|'-separated, bytes or a path) parsed on the device — the scan
        leaf the reference builds as CsvExec(delimiter '|', no header, schema)
        (rust/benchmarks/tpch/src/main.rs:129-150).  schema: [(name, dtype)] of the file's fields in order;
        columns: names to materialise, in output order (None: all)."""
        if isinstance(text, str):
            with open(text, "rb") as f:
                text = f.read()
        text = bytes(text)
        descs, keep = [], []
        for name, dtype in schema:
            d = L.ColumnDesc()
            nb = name.encode()
            keep.append(nb)
            d.name, d.dtype, d.nullable = nb, DTYPE_ID[dtype], 0
            descs.append(d)
        arr = (L.ColumnDesc * max(1, len(descs)))(*descs)
        proj, n_proj = None, 0
        if columns is not None:
            names = [n for n, _ in schema]
            idx = [names.index(c) for c in columns]
            proj = (C.c_int32 * max(1, len(idx)))(*idx)
            n_proj = len(idx)
        h = C.c_void_p()
        # the bytes object itself is the text buffer (no copy on the Python side)
        L.check(L.lib().bhip_batch_from_tbl(ctx._h, C.c_char_p(text) if text else None, len(text), len(descs), arr, n_proj, proj,
                                            C.byref(h)))
        return RecordBatch(h, ctx)

    @staticmethod
    def from_device_pointers(ctx: Context, columns, n_rows: int, keep=None) -> "RecordBatch":
        """bhip_batch_from_device over caller-owned device memory: columns = [(name, dtype, data_ptr)], fixed-width
        NULL-free columns (e.g. buffers an RCCL collective has just filled).  `keep`: whatever owns the memory."""
        descs, names = [], []
        for name, dtype, ptr in columns:
            if dtype in (E.UTF8, E.BOOLEAN):
                raise ValueError("from_device_pointers takes fixed-width columns")
            d = L.ColumnDesc()
            nb = name.encode()
            names.append(nb)
            d.name, d.dtype, d.nullable = nb, DTYPE_ID[dtype], 0
            d.data, d.offsets, d.validity, d.data_bytes = ptr, None, None, 0
            descs.append(d)
        arr = (L.ColumnDesc * max(1, len(descs)))(*descs)
        h = C.c_void_p()
        L.check(L.lib().bhip_batch_from_device(ctx._h, len(descs), arr, n_rows, C.byref(h)))
        rb = RecordBatch(h, ctx)
        rb._owner = keep
        return rb

    @staticmethod
    def from_pyarrow(ctx: Context, batch) -> "RecordBatch":
        """through the Arrow C Data Interface (bhip_batch_import_arrow)"""
        import pyarrow as pa
        if isinstance(batch, pa.Table):
            batch = batch.combine_chunks().to_batches()[0] if batch.num_rows else pa.RecordBatch.from_pylist([], schema=batch.schema)
        c_arr, c_sch = _ArrowArray(), _ArrowSchema()
        batch._export_to_c(C.addressof(c_arr), C.addressof(c_sch))
        h = C.c_void_p()
        try:
            L.check(L.lib().bhip_batch_import_arrow(ctx._h, C.addressof(c_arr), C.addressof(c_sch), C.byref(h)))
        finally:
            # the array is consumed on success; whatever is still live is released here
            if c_arr.release:
                c_arr.release(C.byref(c_arr))
            if c_sch.release:
                c_sch.release(C.byref(c_sch))
        return RecordBatch(h, ctx)

    # -- accessors
    @property
    def num_rows(self):
        return L.lib().bhip_batch_num_rows(self._h)

    @property
    def num_columns(self):
        return L.lib().bhip_batch_num_columns(self._h)

    def memory_size(self):
        return L.lib().bhip_batch_memory_size(self._h)

    def column_info(self, i):
        name, dt, nul, nbytes, hv = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        L.check(L.lib().bhip_batch_column_info(self._h, i, C.byref(name), C.byref(dt), C.byref(nul), C.byref(nbytes), C.byref(hv)))
        return name.value.decode(), DTYPE_NAME[dt.value], bool(nul.value), nbytes.value, bool(hv.value)

    def schema(self):
        return [(self.column_info(i)[0], self.column_info(i)[1]) for i in range(self.num_columns)]

    def schema3(self):
        """[(name, dtype, nullable)]"""
        return [self.column_info(i)[:3] for i in range(self.num_columns)]

    def column_device(self, i):
        d, o, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().bhip_batch_column_device(self._h, i, C.byref(d), C.byref(o), C.byref(v)))
        return d.value, o.value, v.value

    def column(self, i):
        """-> (dtype, values, validity): numpy values (Utf8: list of str), validity bool array or None"""
        name, dtype, _, nbytes, has_valid = self.column_info(i)
        n = self.num_rows
        valid = None
        vbuf = np.zeros((n + 7) // 8 + 8, np.uint8) if has_valid else None
        if dtype == E.UTF8:
            off = np.zeros(n + 1, np.int32)
            data = np.zeros(max(1, nbytes), np.uint8)
            L.check(L.lib().bhip_batch_column_to_host(self._h, i, data.ctypes.data, off.ctypes.data,
                                                      vbuf.ctypes.data if has_valid else None))
            raw = data.tobytes()
            values = [raw[off[k]:off[k + 1]].decode() for k in range(n)]
        elif dtype == E.BOOLEAN:
            data = np.zeros((n + 7) // 8 + 8, np.uint8)
            L.check(L.lib().bhip_batch_column_to_host(self._h, i, data.ctypes.data, None, vbuf.ctypes.data if has_valid else None))
            values = np.unpackbits(data, bitorder="little")[:n].astype(np.bool_)
        else:
            values = np.zeros(max(1, n), NP_DTYPE[dtype])
            L.check(L.lib().bhip_batch_column_to_host(self._h, i, values.ctypes.data, None, vbuf.ctypes.data if has_valid else None))
            values = values[:n]
        if has_valid:
            valid = np.unpackbits(vbuf, bitorder="little")[:n].astype(np.bool_)
        return dtype, values, valid

    def to_pydict(self):
        out = {}
        for i in range(self.num_columns):
            name = self.column_info(i)[0]
            _, values, valid = self.column(i)
            vals = [v.item() if hasattr(v, "item") else v for v in values]
            if valid is not None:
                vals = [v if ok else None for v, ok in zip(vals, valid)]
            out[name] = vals
        return out

    def to_pyarrow(self):
        """through the Arrow C Data Interface (bhip_batch_export_arrow)"""
        import pyarrow as pa
        c_arr, c_sch = _ArrowArray(), _ArrowSchema()
        L.check(L.lib().bhip_batch_export_arrow(self._h, C.addressof(c_arr), C.addressof(c_sch)))
        return pa.RecordBatch._import_from_c(C.addressof(c_arr), C.addressof(c_sch))


class RecordBatchStream:
    """bhip_stream: single-consumer iterator of RecordBatch."""

    def __init__(self, handle, ctx):
        self._h = handle
        self.ctx = ctx

    def schema(self):
        return _read_schema(L.lib().bhip_stream_schema, self._h)

    def __iter__(self):
        return self

    def __next__(self) -> RecordBatch:
        if not self._h:
            raise StopIteration
        b = C.c_void_p()
        L.check(L.lib().bhip_stream_next(self._h, C.byref(b)))
        if not b.value:
            raise StopIteration
        return RecordBatch(b, self.ctx)

    def drain(self):
        """utils::write_stream_to_disk's accounting (rust/core/src/utils.rs:49-84) -> PartitionStats"""
        r, n, by = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().bhip_stream_drain(self._h, None, None, C.byref(r), C.byref(n), C.byref(by)))
        return dict(num_rows=r.value, num_batches=n.value, num_bytes=by.value)

    def write_ipc(self, path: str):
        """bhip_stream_write_ipc = utils::write_stream_to_disk (rust/core/src/utils.rs:49-84): drain into an Arrow IPC file,
        return PartitionStats.  Consumes the stream."""
        r, n, by = C.c_uint64(), C.c_uint64(), C.c_uint64()
        h, self._h = self._h, None
        L.check(L.lib().bhip_stream_write_ipc(h, path.encode(), C.byref(r), C.byref(n), C.byref(by)))
        return dict(num_rows=r.value, num_batches=n.value, num_bytes=by.value)

    def to_arrow_reader(self):
        """hand the stream to pyarrow through the Arrow C Stream Interface (consumes it)"""
        import pyarrow as pa
        c_stream = _ArrowArrayStream()
        L.check(L.lib().bhip_stream_export_arrow(self._h, C.addressof(c_stream)))
        self._h = None
        # the importer moves the struct out of `c_stream` (C Stream Interface ownership rules)
        return pa.RecordBatchReader._import_from_c(C.addressof(c_stream))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().bhip_stream_release(self._h)
                self._h = None
        except Exception:
            pass


def _read_schema(fn, handle):
    n = C.c_int32()
    L.check(fn(handle, 0, None, None, None, C.byref(n)))
    names = (C.c_char_p * max(1, n.value))()
    dts = (C.c_int32 * max(1, n.value))()
    nul = (C.c_int32 * max(1, n.value))()
    L.check(fn(handle, n.value, names, dts, nul, C.byref(n)))
    return [(names[i].decode(), DTYPE_NAME[dts[i]], bool(nul[i])) for i in range(n.value)]


# ---- plans ---------------------------------------------------------------------------------------------

class ExecutionPlan:
    """datafusion::physical_plan::ExecutionPlan"""

    def __init__(self, handle, ctx, children=()):
        self._h = handle
        self.ctx = ctx
        self._children = None if children is None else list(children)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().bhip_plan_release(self._h)
                self._h = None
        except Exception:
            pass

    def as_any(self) -> str:
        return L.lib().bhip_plan_name(self._h).decode()

    def schema(self):
        return _read_schema(L.lib().bhip_plan_schema, self._h)

    def output_partitioning(self) -> Partitioning:
        s, n = C.c_int32(), C.c_int32()
        L.check(L.lib().bhip_plan_output_partitioning(self._h, C.byref(s), C.byref(n)))
        return Partitioning(s.value, n.value)

    def children(self) -> List["ExecutionPlan"]:
        if self._children is None:             # a plan the library built (from_proto): ask it
            n = C.c_int32()
            L.check(L.lib().bhip_plan_children(self._h, 0, None, C.byref(n)))
            arr = (C.c_void_p * max(1, n.value))()
            L.check(L.lib().bhip_plan_children(self._h, n.value, arr, C.byref(n)))
            self._children = [ExecutionPlan(C.c_void_p(arr[i]), self.ctx, None) for i in range(n.value)]
        return list(self._children)

    @staticmethod
    def from_proto(ctx: Optional["Context"], data: bytes, resolver=None) -> "ExecutionPlan":
        """bhip_plan_from_proto: the protobuf bytes of a PhysicalPlanNode (rust/core/proto/ballista.proto:294-312) ->
        operator tree (rust/core/src/serde/physical_plan/from_proto.rs:58-346).  resolver(leaf: dict) -> ExecutionPlan | None
        is offered every scan / shuffle leaf.  ctx None: the plan can be inspected, not executed."""
        keep, err = [], []

        def _cb(_user, leaf_p, out_p):
            try:
                d = leaf_p.contents
                fields = [(d.fields[i].name.decode(), DTYPE_NAME[d.fields[i].dtype], bool(d.fields[i].nullable)) for i in range(d.n_fields)]
                leaf = dict(kind={1: "CsvScan", 2: "ParquetScan", 3: "ShuffleReader", 4: "UnresolvedShuffle"}[d.kind],
                            path=(d.path or b"").decode(), filenames=[d.filenames[i].decode() for i in range(d.n_filenames)],
                            projection=[d.projection[i] for i in range(d.n_projection)] if d.has_projection else None, fields=fields,
                            has_header=bool(d.has_header), delimiter=(d.delimiter or b"").decode(),
                            file_extension=(d.file_extension or b"").decode(), batch_size=d.batch_size, num_partitions=d.num_partitions,
                            locations=[dict(job_id=d.locations[i].job_id.decode(), stage_id=d.locations[i].stage_id,
                                            partition_id=d.locations[i].partition_id, executor_id=d.locations[i].executor_id.decode(),
                                            host=d.locations[i].host.decode(), port=d.locations[i].port,
                                            num_rows=d.locations[i].num_rows, num_batches=d.locations[i].num_batches,
                                            num_bytes=d.locations[i].num_bytes) for i in range(d.n_locations)],
                            stage_ids=[d.stage_ids[i] for i in range(d.n_stage_ids)], partition_count=d.partition_count)
                got = resolver(leaf)
                if got is None:
                    out_p[0] = None
                else:
                    keep.append(got)
                    L.lib().bhip_plan_retain(got._h)          # ownership of the handle passes to the library
                    out_p[0] = got._h.value if isinstance(got._h, C.c_void_p) else got._h
                return L.OK
            except BaseException as e:                          # noqa: BLE001 - must not unwind through C
                err.append(e)
                return L.EINVAL

        cb = L.LEAF_RESOLVER(_cb) if resolver is not None else None
        h = C.c_void_p()
        st = L.lib().bhip_plan_from_proto(ctx._h if ctx is not None else None, data, len(data), cb, None, C.byref(h))
        if err:
            raise err[0]
        L.check(st)
        plan = ExecutionPlan(h, ctx, None)
        plan._leaves = keep
        return plan

    def with_new_children(self, children: Sequence["ExecutionPlan"]) -> "ExecutionPlan":
        arr = (C.c_void_p * max(1, len(children)))(*[c._h for c in children])
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_with_new_children(self._h, len(children), arr, C.byref(h)))
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        ExecutionPlan.__init__(new, h, self.ctx, list(children))
        if len(children) == 1 and hasattr(self, "input"):
            new.input = children[0]
        if len(children) == 2 and hasattr(self, "left"):
            new.left, new.right = children
        return new

    def execute(self, partition: int) -> RecordBatchStream:
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_execute(self._h, partition, C.byref(h)))
        return RecordBatchStream(h, self.ctx)

    def collect(self) -> List[RecordBatch]:
        """datafusion::physical_plan::collect: every partition's batches, in partition order (bhip_plan_collect: one call)"""
        cap = 4096
        arr = (C.c_void_p * cap)()
        n = C.c_int32()
        L.check(L.lib().bhip_plan_collect(self._h, cap, C.cast(arr, C.POINTER(C.c_void_p)), C.byref(n)))
        return [RecordBatch(C.c_void_p(arr[i]), self.ctx) for i in range(n.value)]

    def display(self) -> str:
        buf = C.create_string_buffer(16384)
        L.check(L.lib().bhip_plan_display(self._h, buf, len(buf)))
        return buf.value.decode()

    def __repr__(self):
        return self.display()


class MemoryExec(ExecutionPlan):
    """datafusion MemoryExec::try_new(partitions, schema, projection): partitions = list of lists of
    RecordBatch.  Leaf standing in for CsvExec/ParquetExec/ShuffleReaderExec (from_proto.rs:93-121,
    277-286), whose file / Flight decoding stays on the host."""

    def __init__(self, partitions: Sequence[Sequence[RecordBatch]], ctx: Optional[Context] = None):
        flat, offs = [], [0]
        for p in partitions:
            flat.extend(p)
            offs.append(len(flat))
        if not flat:
            raise L.PlanError(L.EINVAL, "MemoryExec needs at least one batch")
        ctx = ctx or flat[0].ctx
        arr = (C.c_void_p * len(flat))(*[b._h for b in flat])
        o = (C.c_int32 * len(offs))(*offs)
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_memory(ctx._h, len(partitions), o, arr, C.byref(h)))
        super().__init__(h, ctx)
        self.partitions = [list(p) for p in partitions]


class ArrowStreamExec(ExecutionPlan):
    """bhip_plan_arrow_stream(s): leaf over host-side Arrow C streams (e.g. pyarrow.RecordBatchReader) — the C image of a CPU
    child operator's RecordBatchStream (rust/core/src/memory_stream.rs:57-92), one stream per output partition.  The streams
    are moved into the plan, a partition is drained on its first execute and replayed afterwards."""

    def __init__(self, readers, ctx: Context):
        if not isinstance(readers, (list, tuple)):
            readers = [readers]
        c_streams = [_ArrowArrayStream() for _ in readers]
        for r, cs in zip(readers, c_streams):
            r._export_to_c(C.addressof(cs))
        arr = (C.c_void_p * len(c_streams))(*[C.addressof(cs) for cs in c_streams])
        h = C.c_void_p()
        try:
            L.check(L.lib().bhip_plan_arrow_streams(ctx._h, len(c_streams), arr, C.byref(h)))
        finally:
            for cs in c_streams:
                if cs.release:                         # not taken over (an error): release our export
                    C.CFUNCTYPE(None, C.c_void_p)(cs.release)(C.addressof(cs))
        super().__init__(h, ctx)


class ParquetExec(ExecutionPlan):
    """ParquetExec::try_from_files(filenames, projection, None, batch_size, num_partitions)  (from_proto.rs:111-121): the files dealt
    out to `num_partitions` partitions, one batch per row group; projection = column indices (None: all)."""

    def __init__(self, filenames: Sequence[str], ctx: Context, projection: Optional[Sequence[int]] = None, num_partitions: int = 1):
        arr = (C.c_char_p * len(filenames))(*[f.encode() for f in filenames])
        proj = (C.c_uint32 * max(1, len(projection or [])))(*(projection or []))
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_parquet(ctx._h, len(filenames), arr, len(projection) if projection is not None else -1,
                                          proj if projection is not None else None, num_partitions, C.byref(h)))
        super().__init__(h, ctx)
        self.filenames, self.projection = list(filenames), projection


class FilterExec(ExecutionPlan):
    """FilterExec::try_new(predicate, input)  (from_proto.rs:81-92)"""

    def __init__(self, predicate: E.PhysicalExpr, input: ExecutionPlan):
        lw = _Lowered()
        ex = lw.expr(predicate)
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_filter(input._h, C.byref(ex), C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.predicate, self.input = predicate, input


class ProjectionExec(ExecutionPlan):
    """ProjectionExec::try_new(exprs: Vec<(expr, name)>, input)  (from_proto.rs:69-80)"""

    def __init__(self, exprs: Sequence[Tuple[E.PhysicalExpr, str]], input: ExecutionPlan):
        lw = _Lowered()
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_projection(input._h, len(exprs), lw.exprs([e for e, _ in exprs]),
                                             lw.strings([n for _, n in exprs]), C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.exprs, self.input = list(exprs), input


class HashAggregateExec(ExecutionPlan):
    """HashAggregateExec::try_new(mode, group_expr: Vec<(expr, name)>, aggr_expr, input)
    (from_proto.rs:173-252)"""

    _FN = {"SUM": 1, "AVG": 2, "COUNT": 3, "MIN": 4, "MAX": 5}

    def __init__(self, mode: str, group_expr: Sequence[Tuple[E.PhysicalExpr, str]],
                 aggr_expr: Sequence[E.AggregateExpr], input: ExecutionPlan):
        if mode not in (PARTIAL, FINAL):
            raise L.PlanError(L.EINVAL, f"Unsupported aggregate mode {mode}")
        lw = _Lowered()
        aggs = (L.Aggregate * max(1, len(aggr_expr)))()
        for i, a in enumerate(aggr_expr):
            nb = a.name.encode()
            lw.keep.append(nb)
            aggs[i] = L.Aggregate(self._FN[a.fun], lw.expr(a.expr), nb)
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_hash_aggregate(input._h, 0 if mode == PARTIAL else 1, len(group_expr),
                                                 lw.exprs([e for e, _ in group_expr]),
                                                 lw.strings([n for _, n in group_expr]),
                                                 len(aggr_expr), aggs, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.mode, self.group_expr, self.aggr_expr, self.input = mode, list(group_expr), list(aggr_expr), input


class HashJoinExec(ExecutionPlan):
    """HashJoinExec::try_new(left, right, on: &[(String, String)], join_type)  (from_proto.rs:253-276);
    left is the build side."""

    _JT = {INNER: 0, LEFT: 1, RIGHT: 2}

    def __init__(self, left: ExecutionPlan, right: ExecutionPlan, on: Sequence[Tuple[str, str]], join_type: str = INNER):
        if join_type not in self._JT:
            raise L.PlanError(L.EINVAL, f"Unsupported join type {join_type}")
        lw = _Lowered()
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_hash_join(left._h, right._h, len(on), lw.strings([a for a, _ in on]),
                                            lw.strings([b for _, b in on]), self._JT[join_type], C.byref(h)))
        super().__init__(h, left.ctx, [left, right])
        self.left, self.right, self.on, self.join_type = left, right, list(on), join_type


class SortExec(ExecutionPlan):
    """SortExec::try_new(expr: Vec<PhysicalSortExpr>, input)  (from_proto.rs:291-331)"""

    def __init__(self, expr: Sequence[E.PhysicalSortExpr], input: ExecutionPlan):
        lw = _Lowered()
        arr = (L.SortExprC * max(1, len(expr)))()
        for i, s in enumerate(expr):
            arr[i] = L.SortExprC(lw.expr(s.expr), 1 if s.descending else 0, 1 if s.nulls_first else 0)
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_sort(input._h, len(expr), arr, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.expr, self.input = list(expr), input


class RepartitionExec(ExecutionPlan):
    """RepartitionExec::try_new(input, partitioning)  (from_proto.rs:133-164)"""

    def __init__(self, input: ExecutionPlan, partitioning: Partitioning):
        lw = _Lowered()
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_repartition(input._h, partitioning.scheme, len(partitioning.exprs),
                                              lw.exprs(partitioning.exprs), partitioning.count, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.partitioning = input, partitioning


class CoalesceBatchesExec(ExecutionPlan):
    """CoalesceBatchesExec::new(input, target_batch_size)  (from_proto.rs:122-128)"""

    def __init__(self, input: ExecutionPlan, target_batch_size: int):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_coalesce_batches(input._h, target_batch_size, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.target_batch_size = input, target_batch_size


class MergeExec(ExecutionPlan):
    """MergeExec::new(input)  (from_proto.rs:129-132)"""

    def __init__(self, input: ExecutionPlan):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_merge(input._h, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input = input


class GlobalLimitExec(ExecutionPlan):
    """GlobalLimitExec::new(input, limit, concurrency)  (from_proto.rs:165-168)"""

    def __init__(self, input: ExecutionPlan, limit: int):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_global_limit(input._h, limit, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.limit = input, limit


class LocalLimitExec(ExecutionPlan):
    """LocalLimitExec::new(input, limit)  (from_proto.rs:169-172)"""

    def __init__(self, input: ExecutionPlan, limit: int):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_local_limit(input._h, limit, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.limit = input, limit


# ---- Arrow IPC files (the stage boundary on disk) --------------------------------------------------------------

def ipc_write_file(reader, path: str):
    """bhip_ipc_write_file: a host-side Arrow stream (pyarrow.RecordBatchReader) -> Arrow IPC file, written by the library's own
    writer; returns PartitionStats as a dict.  No GPU involved."""
    c_stream = _ArrowArrayStream()
    reader._export_to_c(C.addressof(c_stream))
    r, n, by = C.c_uint64(), C.c_uint64(), C.c_uint64()
    try:
        L.check(L.lib().bhip_ipc_write_file(C.addressof(c_stream), path.encode(), C.byref(r), C.byref(n), C.byref(by)))
    finally:
        if c_stream.release:
            C.CFUNCTYPE(None, C.c_void_p)(c_stream.release)(C.addressof(c_stream))
    return dict(num_rows=r.value, num_batches=n.value, num_bytes=by.value)


def ipc_open_file(path: str):
    """bhip_ipc_open_file: an Arrow IPC file read by the library's own reader -> pyarrow.RecordBatchReader (no GPU involved)"""
    import pyarrow as pa
    c_stream = _ArrowArrayStream()
    L.check(L.lib().bhip_ipc_open_file(path.encode(), C.addressof(c_stream)))
    return pa.RecordBatchReader._import_from_c(C.addressof(c_stream))


class IpcFileExec(ExecutionPlan):
    """bhip_plan_ipc_files: leaf over Arrow IPC files, one output partition per file — the local half of ShuffleReaderExec
    (rust/core/src/execution_plans/shuffle_reader.rs:77-99)."""

    def __init__(self, paths: Sequence[str], ctx: Context):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_ipc_files(ctx._h, len(paths), arr, C.byref(h)))
        super().__init__(h, ctx)
        self.paths = list(paths)


class Communicator:
    """bhip_comm: batches between the ranks of one node (one communicator per process and GPU; csrc/host/exchange.cpp).
    All calls are collective.  Transports: RCCL (the constructor), `loopback` (N communicators in one process on one device,
    one thread each: the N-rank code on a single-GPU box), `host` (the caller moves host bytes: a rehearsal over gloo)."""
    UNIQUE_ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Communicator.UNIQUE_ID_BYTES)
        L.check(L.lib().bhip_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx: Context, unique_id: bytes, world: int, rank: int, _handle=None, _keep=None):
        if _handle is None:
            if len(unique_id) != self.UNIQUE_ID_BYTES:
                raise ValueError("unique id must be 128 bytes")
            _handle = C.c_void_p()
            L.check(L.lib().bhip_comm_create(ctx._h, unique_id, world, rank, C.byref(_handle)))
        self._h, self.ctx, self.world, self.rank, self._keep = _handle, ctx, world, rank, _keep

    @classmethod
    def loopback(cls, ctx: Context, hub_id: bytes, world: int, rank: int) -> "Communicator":
        """rank `rank` of a world of communicators that live in THIS process (the same hub_id on every rank, any bytes)"""
        hub_id = (hub_id + b"\0" * cls.UNIQUE_ID_BYTES)[:cls.UNIQUE_ID_BYTES]
        h = C.c_void_p()
        L.check(L.lib().bhip_comm_create_loopback(ctx._h, hub_id, world, rank, C.byref(h)))
        return cls(ctx, b"", world, rank, _handle=h)

    @classmethod
    def host(cls, ctx: Context, world: int, rank: int, all_gather, exchange) -> "Communicator":
        """the caller moves host bytes.  all_gather(send: memoryview, recv: memoryview(world * len(send)));
        exchange(sends: [(memoryview, peer)], recvs: [(memoryview, peer)]) — regions of one peer pair are matched in order"""
        def view(ptr, n):
            return memoryview((C.c_uint8 * n).from_address(ptr)).cast("B") if n else memoryview(b"")

        def _ag(_user, send, recv, nbytes):
            try:
                all_gather(view(send, nbytes), view(recv, nbytes * world))
                return 0
            except BaseException as e:                       # noqa: BLE001 - must not unwind through C
                err.append(e)
                return 1

        def _ex(_user, ns, sends, nr, recvs):
            try:
                exchange([(view(sends[i].ptr, sends[i].bytes), sends[i].peer) for i in range(ns)],
                         [(view(recvs[i].ptr, recvs[i].bytes), recvs[i].peer) for i in range(nr)])
                return 0
            except BaseException as e:                       # noqa: BLE001
                err.append(e)
                return 1

        err = []
        t = L.CommHostTransport(None, L.HOST_ALL_GATHER(_ag), L.HOST_EXCHANGE(_ex))
        h = C.c_void_p()
        L.check(L.lib().bhip_comm_create_host(ctx._h, C.byref(t), world, rank, C.byref(h)))
        comm = cls(ctx, b"", world, rank, _handle=h, _keep=(t, err))
        return comm

    def _check(self, status):
        if self._keep and self._keep[1]:
            e = self._keep[1].pop()
            self._keep[1].clear()
            raise e
        L.check(status)

    def info(self):
        w, r, t = C.c_int32(), C.c_int32(), C.c_char_p()
        L.check(L.lib().bhip_comm_info(self._h, C.byref(w), C.byref(r), C.byref(t)))
        return dict(world=w.value, rank=r.value, transport=t.value.decode())

    def stats(self, reset=False):
        s, b, n = C.c_double(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().bhip_comm_stats(self._h, 1 if reset else 0, C.byref(s), C.byref(b), C.byref(n)))
        return dict(seconds=s.value, bytes_out=b.value, calls=n.value)

    def all_gather(self, batch: "RecordBatch") -> List["RecordBatch"]:
        out = (C.c_void_p * self.world)()
        self._check(L.lib().bhip_comm_all_gather(self._h, batch._h, out))
        return [RecordBatch(C.c_void_p(out[i]), self.ctx) for i in range(self.world)]

    def all_to_all(self, parts: Sequence["RecordBatch"]) -> List["RecordBatch"]:
        if len(parts) != self.world:
            raise ValueError(f"need one outgoing batch per rank ({self.world}), got {len(parts)}")
        arr = (C.c_void_p * self.world)(*[p._h for p in parts])
        out = (C.c_void_p * self.world)()
        self._check(L.lib().bhip_comm_all_to_all(self._h, arr, out))
        return [RecordBatch(C.c_void_p(out[i]), self.ctx) for i in range(self.world)]

    def shuffle(self, batch: "RecordBatch", key: str, chunk_rows: int = 0, with_stats=False):
        """bhip_comm_shuffle: my rows of every rank's batch under Hash([key], world), source-rank order"""
        h = C.c_void_p()
        st = L.ShuffleStats()
        self._check(L.lib().bhip_comm_shuffle(self._h, batch._h, key.encode(), chunk_rows, C.byref(h), C.byref(st)))
        out = RecordBatch(h, self.ctx)
        if not with_stats:
            return out
        d = {k: getattr(st, k) for k, _ in L.ShuffleStats._fields_ if k != "rows_to"}
        d["rows_to"] = [st.rows_to[i] for i in range(min(self.world, L.SHUFFLE_MAX_PEERS))]
        return out, d

    def close(self):
        if getattr(self, "_h", None):
            L.lib().bhip_comm_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AllGatherExec(ExecutionPlan):
    """bhip_plan_all_gather: the stage boundary of a partial aggregate inside one plan — executes every partition of `input`,
    all_gathers the result over `comm`; `world` output partitions (partition r = rank r's batch), what the next stage's
    MergeExec reads (rust/scheduler/src/planner.rs:136-171)."""

    def __init__(self, input: ExecutionPlan, comm: Communicator):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_all_gather(comm._h, input._h, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.comm = input, comm


class ShuffleExchangeExec(ExecutionPlan):
    """bhip_plan_shuffle: RepartitionExec(Hash([key], world)) (from_proto.rs:133-147) + the shuffle read of this rank's
    partition (shuffle_reader.rs:77-99) as one node; one output partition."""

    def __init__(self, input: ExecutionPlan, comm: Communicator, key: str, chunk_rows: int = 0):
        h = C.c_void_p()
        L.check(L.lib().bhip_plan_shuffle(comm._h, input._h, key.encode(), chunk_rows, C.byref(h)))
        super().__init__(h, input.ctx, [input])
        self.input, self.comm, self.key, self.chunk_rows = input, comm, key, chunk_rows


def pack_batch(batch: "RecordBatch"):
    """bhip_batch_pack -> (header int64[], block uint8[]): the block form the exchange moves (host copies)"""
    n_words = 2 + 3 * batch.num_columns
    header = np.zeros(n_words, np.int64)
    nbytes = C.c_int64()
    hp = header.ctypes.data_as(C.POINTER(C.c_int64))
    L.check(L.lib().bhip_batch_pack(batch._h, hp, n_words, None, 0, C.byref(nbytes)))
    block = np.zeros(max(1, nbytes.value), np.uint8)
    L.check(L.lib().bhip_batch_pack(batch._h, hp, n_words, block.ctypes.data, block.size, C.byref(nbytes)))
    return header, block[:nbytes.value]


def unpack_batch(ctx: Context, schema, header, block) -> "RecordBatch":
    """bhip_batch_unpack: schema = [(name, dtype, nullable)]"""
    descs, keep = [], []
    for name, dtype, nullable in schema:
        d = L.ColumnDesc()
        nb = name.encode()
        keep.append(nb)
        d.name, d.dtype, d.nullable = nb, DTYPE_ID[dtype], 1 if nullable else 0
        descs.append(d)
    arr = (L.ColumnDesc * max(1, len(descs)))(*descs)
    header = np.ascontiguousarray(header, np.int64)
    block = np.ascontiguousarray(block, np.uint8)
    h = C.c_void_p()
    L.check(L.lib().bhip_batch_unpack(ctx._h, len(descs), arr, header.ctypes.data_as(C.POINTER(C.c_int64)),
                                      block.ctypes.data if block.size else None, C.byref(h)))
    return RecordBatch(h, ctx)


# ---- batch-level helpers used by the multi-GPU exchange ---------------------------------------------------

def hash_partition(batch: RecordBatch, exprs: Sequence[E.PhysicalExpr], n: int) -> List[RecordBatch]:
    lw = _Lowered()
    out = (C.c_void_p * n)()
    L.check(L.lib().bhip_batch_hash_partition(batch._h, len(exprs), lw.exprs(exprs), n, out))
    return [RecordBatch(C.c_void_p(out[i]), batch.ctx) for i in range(n)]


def concat(ctx: Context, batches: Sequence[RecordBatch]) -> RecordBatch:
    arr = (C.c_void_p * len(batches))(*[b._h for b in batches])
    h = C.c_void_p()
    L.check(L.lib().bhip_batch_concat(ctx._h, len(batches), arr, C.byref(h)))
    return RecordBatch(h, ctx)


def _tpch_opts(key64, with_dates, sparse_keys, key_base, columns):
    o = L.TpchOpts()
    o.key64, o.with_dates, o.sparse_keys, o.key_base = int(bool(key64)), int(bool(with_dates)), int(bool(sparse_keys)), int(key_base)
    keep = None
    if columns:
        keep = (C.c_char_p * len(columns))(*[c.encode() for c in columns])
        o.columns, o.n_columns = keep, len(columns)
    return o, keep


def tpch_lineitem(ctx: Context, sf: float, seed: int, row0: int, n: int, key64=False, with_dates=False, sparse_keys=False, key_base=0,
                  columns=None) -> RecordBatch:
    """synthetic lineitem rows [row0, row0 + n) on the device.  sparse_keys: dbgen's order-key layout (8 of every 32 values);
    key_base: added to every order key; columns: only these"""
    h = C.c_void_p()
    o, _keep = _tpch_opts(key64, with_dates, sparse_keys, key_base, columns)
    L.check(L.lib().bhip_tpch_lineitem_opts(ctx._h, sf, seed, row0, n, C.byref(o), C.byref(h)))
    return RecordBatch(h, ctx)


def tpch_orders(ctx: Context, sf: float, seed: int, row0: int, n: int, key64=False, sparse_keys=False, key_base=0, columns=None) -> RecordBatch:
    h = C.c_void_p()
    o, _keep = _tpch_opts(key64, False, sparse_keys, key_base, columns)
    L.check(L.lib().bhip_tpch_orders_opts(ctx._h, sf, seed, row0, n, C.byref(o), C.byref(h)))
    return RecordBatch(h, ctx)
