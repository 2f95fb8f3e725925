"""ctypes binding of transferia_amd/libtfgpu.so (the C ABI in include/tfgpu.h)
and a thin host mirror of the reference's plugin interface.

The mirror keeps the reference's names and argument meaning
(abstract.Transformer: Type / Description / Suitable / ResultSchema / Apply —
pkg/abstract/transformer.go:32-38) so tests read like the reference's tests.
Nothing here computes: every call lands in a HIP kernel.  There is no CPU
fallback; if the library or a gfx950 device is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading
from typing import List, Optional, Sequence

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libtfgpu.so")
_lib = None

EXPORTS = [
    "tfgpu_abi_version", "tfgpu_last_error", "tfgpu_init", "tfgpu_shutdown", "tfgpu_device_count", "tfgpu_synchronize",
    "tfgpu_init_devices", "tfgpu_stream", "tfgpu_lane_count", "tfgpu_lane_use", "tfgpu_lane_current", "tfgpu_lane_device", "tfgpu_dbatch_slice", "tfgpu_dbatch_to_lane", "tfgpu_shard_rows", "tfgpu_dbatch_concat", "tfgpu_strictify", "tfgpu_parquet_read", "tfgpu_parquet_write", "tfgpu_parsequeue_create", "tfgpu_parsequeue_add", "tfgpu_parsequeue_error", "tfgpu_parsequeue_close", "tfgpu_parsequeue_set_release", "tfgpu_parsequeue_destroy", "tfgpu_bufferer_create", "tfgpu_bufferer_async_push", "tfgpu_bufferer_async_push_meta", "tfgpu_bufferer_wait", "tfgpu_bufferer_get_stats", "tfgpu_bufferer_close", "tfgpu_bufferer_destroy", "tfgpu_host_alloc", "tfgpu_host_free", "tfgpu_plan_create", "tfgpu_plan_destroy", "tfgpu_plan_type",
    "tfgpu_plan_description", "tfgpu_plan_suitable", "tfgpu_plan_result_schema", "tfgpu_schema_free", "tfgpu_registry_count",
    "tfgpu_registry_name", "tfgpu_batch_upload", "tfgpu_dbatch_view", "tfgpu_dbatch_download", "tfgpu_dbatch_free",
    "tfgpu_apply", "tfgpu_transformation_create", "tfgpu_transformation_from_config", "tfgpu_transformation_size", "tfgpu_transformation_plan_type", "tfgpu_transformation_errors_output", "tfgpu_transformation_destroy", "tfgpu_transformation_table_plan", "tfgpu_transformation_push",
    "tfgpu_transformation_get_stats", "tfgpu_executor_start", "tfgpu_transformation_push_async", "tfgpu_wait", "tfgpu_collapse", "tfgpu_keys_changed", "tfgpu_dbatch_deepsizeof", "tfgpu_partition", "tfgpu_comm_unique_id", "tfgpu_comm_init", "tfgpu_comm_destroy", "tfgpu_comm_rank", "tfgpu_comm_world", "tfgpu_exchange", "tfgpu_csv_options_default", "tfgpu_csv_parse", "tfgpu_csv_split_rows", "tfgpu_json_parse", "tfgpu_json_result_schema", "tfgpu_sr_frames", "tfgpu_sr_json_parse", "tfgpu_sr_compile_schema", "tfgpu_sr_schema_info", "tfgpu_sr_schema_free", "tfgpu_sr_compile_proto", "tfgpu_pb_schema_info", "tfgpu_pb_schema_free", "tfgpu_sr_proto_parse", "tfgpu_debezium_unpack", "tfgpu_debezium_unpack_cached", "tfgpu_debezium_parse", "tfgpu_debezium_compile_schema", "tfgpu_dbz_schema_info", "tfgpu_dbz_schema_free", "tfgpu_dbz_receiver_create", "tfgpu_dbz_receiver_destroy", "tfgpu_dbz_receiver_known", "tfgpu_dbz_receive", "tfgpu_dbz_receive_group", "tfgpu_dbz_receive_group_meta", "tfgpu_debezium_compile_registry_schema", "tfgpu_dbz_receiver_add_registry_schema", "tfgpu_debezium_registry_frames", "tfgpu_dbz_receive_registry", "tfgpu_serialize", "tfgpu_serialize_ex", "tfgpu_serialize_batch", "tfgpu_ch_native_block", "tfgpu_queue_serialize", "tfgpu_queue_raw_column", "tfgpu_queue_mirror", "tfgpu_queue_part_groups", "tfgpu_kafka_hash_partition", "tfgpu_kafka_partitions", "tfgpu_debezium_emit", "tfgpu_dbuf_size", "tfgpu_dbuf_ptr",
    "tfgpu_dbuf_download", "tfgpu_dbuf_free", "tfgpu_dbuf_upload", "tfgpu_dbuf_alloc", "tfgpu_dbuf_write", "tfgpu_prof_enable", "tfgpu_prof_reset", "tfgpu_prof_count",
    "tfgpu_prof_get", "tfgpu_prof_get_units", "tfgpu_parquet_read_object", "tfgpu_parquet_staging_size", "tfgpu_parquet_read_staged", "tfgpu_parquet_resolve_schema", "tfgpu_dbatch_nrows", "tfgpu_dbatch_dense",
]


class TfgpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"tfgpu error {code}: {msg}")
        self.code = code


ERR_INVALID, ERR_CONFIG, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM, ERR_UNKNOWN_TYPE = 1, 2, 3, 4, 5, 6


def load():
    """Load libtfgpu.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise RuntimeError(f"{_LIBPATH} is missing: run `python -m transferia_amd.build` (hipcc, gfx950). "
                           "transferia_amd has no CPU implementation.")
    L = C.CDLL(_LIBPATH)
    # the struct layouts of abi.py are version 2's: a library of another version would mis-stride tfgpu_column arrays (include/tfgpu.h)
    if L.tfgpu_abi_version() != abi.ABI_VERSION:
        raise RuntimeError("%s speaks ABI version %d, this binding version %d: rebuild one of them" % (_LIBPATH, L.tfgpu_abi_version(), abi.ABI_VERSION))
    P = C.c_void_p
    L.tfgpu_last_error.restype = C.c_char_p
    L.tfgpu_stream.restype = P
    L.tfgpu_plan_type.restype = C.c_char_p
    L.tfgpu_plan_type.argtypes = [P]
    L.tfgpu_registry_name.restype = C.c_char_p
    L.tfgpu_registry_name.argtypes = [C.c_int]
    L.tfgpu_init.argtypes = [C.c_int]
    L.tfgpu_lane_use.argtypes = [C.c_int]
    L.tfgpu_device_count.argtypes = [C.POINTER(C.c_int)]
    L.tfgpu_host_alloc.argtypes = [C.c_size_t, C.POINTER(P)]
    L.tfgpu_host_free.argtypes = [P]
    L.tfgpu_plan_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(P)]
    L.tfgpu_plan_destroy.argtypes = [P]
    L.tfgpu_plan_destroy.restype = None
    L.tfgpu_plan_description.argtypes = [P, C.c_char_p, C.c_size_t]
    L.tfgpu_plan_suitable.argtypes = [P, C.c_char_p, C.c_char_p, C.POINTER(abi.CSchema), C.POINTER(C.c_int)]
    L.tfgpu_plan_result_schema.argtypes = [P, C.POINTER(abi.CSchema), C.POINTER(C.POINTER(abi.CSchema))]
    L.tfgpu_schema_free.argtypes = [C.POINTER(abi.CSchema)]
    L.tfgpu_schema_free.restype = None
    L.tfgpu_batch_upload.argtypes = [C.POINTER(abi.CBatch), C.POINTER(P)]
    L.tfgpu_dbatch_view.argtypes = [P, C.POINTER(abi.CBatch)]
    L.tfgpu_dbatch_download.argtypes = [P, C.POINTER(abi.CBatch)]
    L.tfgpu_dbatch_free.argtypes = [P]
    L.tfgpu_dbatch_free.restype = None
    L.tfgpu_apply.argtypes = [C.POINTER(P), C.c_int, P, C.POINTER(P), C.POINTER(abi.CRowError), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_csv_split_rows.argtypes = [P, C.c_uint64, C.c_int, C.POINTER(P), C.POINTER(C.c_int64)]
    L.tfgpu_transformation_create.argtypes = [C.POINTER(P), C.c_int, C.POINTER(P)]
    L.tfgpu_transformation_destroy.argtypes = [P]
    L.tfgpu_transformation_table_plan.argtypes = [P, C.c_char_p, C.c_char_p, C.POINTER(abi.CSchema), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
    _push_tail = [C.POINTER(P), C.POINTER(P), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(abi.CRowError), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_transformation_push.argtypes = [P, P, C.POINTER(abi.CSchema)] + _push_tail
    L.tfgpu_transformation_get_stats.argtypes = [P, C.POINTER(abi.CTransformationStats)]
    L.tfgpu_executor_start.argtypes = [C.c_int]
    L.tfgpu_transformation_push_async.argtypes = [P, P, C.POINTER(abi.CSchema), C.POINTER(P)]
    L.tfgpu_wait.argtypes = [P] + _push_tail
    L.tfgpu_collapse.argtypes = [P, C.POINTER(P)]
    L.tfgpu_partition.argtypes = [P, C.c_int, C.POINTER(P), C.POINTER(C.c_int64)]
    L.tfgpu_dbatch_deepsizeof.argtypes = [P, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tfgpu_debezium_unpack.argtypes = [P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), P]
    L.tfgpu_debezium_unpack_cached.argtypes = [P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), C.POINTER(abi.CDbzPrefix), P]
    L.tfgpu_debezium_parse.argtypes = [C.POINTER(abi.CDbzOptions), P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), P, C.POINTER(P), P, C.c_int64,
                                       C.POINTER(abi.CRowError), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_ch_native_block.argtypes = [P, C.POINTER(abi.CChNativeColumn), C.c_int32, C.POINTER(P)]
    L.tfgpu_comm_unique_id.argtypes = [C.c_char_p]
    L.tfgpu_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(P)]
    L.tfgpu_comm_destroy.argtypes = [P]
    L.tfgpu_comm_destroy.restype = None
    L.tfgpu_comm_rank.argtypes = [P]
    L.tfgpu_comm_world.argtypes = [P]
    L.tfgpu_exchange.argtypes = [P, P, C.POINTER(C.c_int64), C.POINTER(P), C.POINTER(C.c_int64)]
    L.tfgpu_csv_options_default.argtypes = [C.POINTER(abi.CCsvOptions)]
    L.tfgpu_csv_options_default.restype = None
    L.tfgpu_csv_parse.argtypes = [C.POINTER(abi.CCsvOptions), C.POINTER(abi.CSchema), P, C.c_uint64, C.c_int, C.POINTER(P),
                                  C.POINTER(C.c_uint64), C.POINTER(abi.CRowError), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_json_parse.argtypes = [C.POINTER(abi.CJsonOptions), C.POINTER(abi.CSchema), P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages),
                                   C.POINTER(P), C.POINTER(abi.CRowError), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_json_result_schema.argtypes = [C.POINTER(abi.CJsonOptions), C.POINTER(abi.CSchema), C.POINTER(C.POINTER(abi.CSchema))]
    L.tfgpu_sr_frames.argtypes = [P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), C.POINTER(abi.CSrFrame), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_sr_json_parse.argtypes = [C.POINTER(abi.CSrJsonOptions), P, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), C.POINTER(P), C.POINTER(abi.CRowError),
                                      C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_serialize.argtypes = [C.c_int, P, C.POINTER(P)]
    L.tfgpu_serialize_ex.argtypes = [C.c_int, P, C.POINTER(abi.CSerializeOptions), C.POINTER(P)]
    L.tfgpu_serialize_batch.argtypes = [C.c_int, P, C.POINTER(abi.CSerializeOptions), C.POINTER(abi.CBatchSerializerConfig), C.c_int, C.POINTER(P), C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_debezium_emit.argtypes = [C.POINTER(abi.CDbzEmitOptions), P, C.POINTER(abi.CRowMeta), C.POINTER(P), P, C.POINTER(P), P, P, P, C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_queue_serialize.argtypes = [C.POINTER(abi.CQueueOptions), P, C.POINTER(abi.CRowMeta), C.POINTER(P), P, P, C.c_int64, C.POINTER(C.c_int64)]
    L.tfgpu_dbuf_size.argtypes = [P, C.POINTER(C.c_uint64)]
    L.tfgpu_dbuf_ptr.argtypes = [P]
    L.tfgpu_dbuf_ptr.restype = P
    L.tfgpu_dbuf_download.argtypes = [P, P, C.c_uint64]
    L.tfgpu_dbuf_free.argtypes = [P]
    L.tfgpu_dbuf_free.restype = None
    L.tfgpu_dbuf_upload.argtypes = [P, C.c_uint64, C.POINTER(P)]
    L.tfgpu_dbuf_alloc.argtypes = [C.c_uint64, C.POINTER(P)]
    L.tfgpu_dbuf_write.argtypes = [P, C.c_uint64, P, C.c_uint64]
    L.tfgpu_prof_enable.argtypes = [C.c_int]
    L.tfgpu_prof_get.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.tfgpu_prof_get_units.argtypes = [C.c_int, C.POINTER(C.c_int64)]
    L.tfgpu_dbatch_nrows.restype = C.c_int64
    L.tfgpu_dbatch_nrows.argtypes = [C.c_void_p]
    L.tfgpu_dbatch_dense.argtypes = [C.c_void_p]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise TfgpuError(rc, (load().tfgpu_last_error() or b"").decode("utf-8", "replace"))


_initialised = None


def init(device: Optional[int] = None) -> int:
    """Bind this process to one GPU (LOCAL_RANK by default): one process per GPU."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _initialised is not None:
        return _initialised
    _check(load().tfgpu_init(device))
    _initialised = device
    return device


def init_devices(devices) -> List[int]:
    """Bind this process to several GPUs: lane k lives on devices[k % len(devices)] (one worker process driving the node)."""
    global _initialised
    devices = [int(d) for d in devices]
    arr = (C.c_int * len(devices))(*devices)
    _check(load().tfgpu_init_devices(arr, len(devices)))
    _initialised = devices[0]
    return devices


def lane_device(lane: int) -> int:
    d = C.c_int(-1)
    _check(load().tfgpu_lane_device(int(lane), C.byref(d)))
    return int(d.value)


def parquet_read(data: bytes, schema: Optional[abi.Schema] = None, ns: str = "", table: str = "", file_name: Optional[str] = None) -> "DeviceBatch":
    """ReaderParquet.Read's decode on the device: a whole Parquet object (host bytes) → one device batch.  With file_name the
    schema's `__file_name` / `__row_index` are the reader's system columns (constructCI)."""
    init()
    out = C.c_void_p()
    buf = np.frombuffer(data, dtype=np.uint8)
    cs = schema.to_c() if schema is not None else None
    L = load()
    L.tfgpu_parquet_read_object.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    _check(L.tfgpu_parquet_read_object(C.c_void_p(buf.ctypes.data), C.c_uint64(len(buf)), abi.MEM_HOST, C.byref(cs) if cs is not None else None, ns.encode(), table.encode(),
                                       file_name.encode() if file_name is not None else None, C.byref(out)))
    return DeviceBatch(out)


def parquet_staging_size(data) -> int:
    """Bytes a staging buffer for tfgpu_parquet_read_staged must hold: the object, then room for what the reader decodes beside it."""
    ptr_, n, _mem, keep = _bytes_arg(data)
    need = C.c_uint64(0)
    L = load()
    L.tfgpu_parquet_staging_size.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    _check(L.tfgpu_parquet_staging_size(C.c_void_p(ptr_), C.c_uint64(n), C.byref(need)))
    return int(need.value)


def parquet_read_staged(data, staged: "DeviceBuffer", schema: Optional[abi.Schema] = None, ns: str = "", table: str = "", file_name: Optional[str] = None) -> "DeviceBatch":
    """tfgpu_parquet_read_object for an object already in HBM: `data` = the host copy (bytes or a pinned HostBuffer), `staged` = a device
    buffer of parquet_staging_size(data) bytes whose first len(data) bytes are the object."""
    init()
    ptr_, n, _mem, keep = _bytes_arg(data)
    out = C.c_void_p()
    cs = schema.to_c() if schema is not None else None
    L = load()
    L.tfgpu_parquet_read_staged.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    _check(L.tfgpu_parquet_read_staged(C.c_void_p(ptr_), C.c_uint64(n), staged._h, C.byref(cs) if cs is not None else None, ns.encode(), table.encode(),
                                       file_name.encode() if file_name is not None else None, C.byref(out)))
    return DeviceBatch(out)


def parquet_resolve_schema(data: bytes, hide_system_cols: bool = False) -> abi.Schema:
    """ParquetSchemaResolver.resolveSchema (+ the system columns): the TableSchema the reference infers from the object's footer"""
    L = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    out = C.POINTER(abi.CSchema)()
    L.tfgpu_parquet_resolve_schema.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.POINTER(abi.CSchema))]
    _check(L.tfgpu_parquet_resolve_schema(C.c_void_p(buf.ctypes.data), C.c_uint64(len(buf)), 1 if hide_system_cols else 0, C.byref(out)))
    try:
        return abi.Schema.from_c(out.contents)
    finally:
        L.tfgpu_schema_free(out)


def parquet_write(batch: "DeviceBatch", schema: abi.Schema, codec: str = "", row_group_max_rows: int = 0, row_group_max_bytes: int = 0) -> bytes:
    """parquetBatchSerializer.Serialize + Close of one batch: the Parquet object as bytes (codec: "", "SNAPPY", "GZIP", "ZSTD")"""
    init()
    p, n = C.c_void_p(), C.c_uint64(0)
    cs = schema.to_c()
    L = load()
    L.tfgpu_parquet_write.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    _check(L.tfgpu_parquet_write(batch._h, C.byref(cs), codec.encode(), row_group_max_rows, row_group_max_bytes, C.byref(p), C.byref(n)))
    try:
        return C.string_at(p.value, n.value)
    finally:
        L.tfgpu_host_free.argtypes = [C.c_void_p]
        L.tfgpu_host_free(p)


def strictify(batch: "DeviceBatch", schema: Optional[abi.Schema] = None) -> "DeviceBatch":
    """strictify.Strictify over a device batch (the strictifying serializers' first step).  Raises TfgpuError; .bad = (row, column)
    of the first value that cannot be converted, when that is the reason."""
    init()
    out, row, col = C.c_void_p(), C.c_int64(-1), C.c_int32(-1)
    cs = schema.to_c() if schema is not None else None
    rc = load().tfgpu_strictify(batch._h, C.byref(cs) if cs is not None else None, C.byref(out), C.byref(row), C.byref(col))
    if rc:
        e = TfgpuError(rc, (load().tfgpu_last_error() or b"").decode("utf-8", "replace"))
        e.bad = (int(row.value), int(col.value))
        raise e
    return DeviceBatch(out)


def lane_use(lane: int):
    """Bind the calling thread to lane `lane` (its own stream / HBM cache / pinned ring): calls made by
    threads on different lanes overlap on the device, the way the parsequeue's workers overlap."""
    init()
    _check(load().tfgpu_lane_use(int(lane)))


def lane_count() -> int:
    return int(load().tfgpu_lane_count())


def synchronize():
    _check(load().tfgpu_synchronize())


def registry() -> List[str]:
    L = load()
    return [L.tfgpu_registry_name(i).decode() for i in range(L.tfgpu_registry_count())]


# ---- device-resident containers ---------------------------------------------
class DeviceBatch:
    """A []ChangeItem run (one table, one schema) fanned out into HBM columns."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def upload(batch: abi.Batch) -> "DeviceBatch":
        init()
        cb = batch.to_c()
        h = C.c_void_p()
        _check(load().tfgpu_batch_upload(C.byref(cb), C.byref(h)))
        return DeviceBatch(h)

    def view(self) -> abi.CBatch:
        v = abi.CBatch()
        _check(load().tfgpu_dbatch_view(self._h, C.byref(v)))
        return v

    @property
    def nrows(self) -> int:
        return int(load().tfgpu_dbatch_nrows(self._h))  # (no column is touched: a filter's kept rows may still be a selection)

    def dense(self) -> "DeviceBatch":
        """gathers the kept rows now if they are still a selection (tfgpu_dbatch_dense); returns self"""
        _check(load().tfgpu_dbatch_dense(self._h))
        return self

    # ---- one process, several GPUs: row-range shards and their ordered merge (tf_shard.hip) ----
    def slice(self, row0: int, nrows: int) -> "DeviceBatch":
        h = C.c_void_p()
        _check(load().tfgpu_dbatch_slice(self._h, C.c_int64(row0), C.c_int64(nrows), C.byref(h)))
        return DeviceBatch(h)

    def to_lane(self, lane: int) -> "DeviceBatch":
        h = C.c_void_p()
        _check(load().tfgpu_dbatch_to_lane(self._h, int(lane), C.byref(h)))
        return DeviceBatch(h)

    def shard_rows(self, nshards: int, lanes=None):
        """-> ([DeviceBatch per shard, shard g on lane lanes[g] (default g)], [first row of each shard])"""
        hs = (C.c_void_p * nshards)()
        row0 = (C.c_int64 * nshards)()
        la = (C.c_int * nshards)(*[int(x) for x in lanes]) if lanes is not None else None
        _check(load().tfgpu_shard_rows(self._h, int(nshards), la, hs, row0))
        return [DeviceBatch(C.c_void_p(hs[g])) for g in range(nshards)], [int(row0[g]) for g in range(nshards)]

    @staticmethod
    def concat(parts, row_base=None) -> "DeviceBatch":
        n = len(parts)
        hs = (C.c_void_p * n)(*[p._h for p in parts])
        rb = (C.c_int64 * n)(*[int(x) for x in row_base]) if row_base is not None else None
        h = C.c_void_p()
        _check(load().tfgpu_dbatch_concat(hs, n, rb, C.byref(h)))
        return DeviceBatch(h)

    def table_schema(self) -> Optional[abi.Schema]:
        """ChangeItem.TableSchema where the batch carries one of its own, else None (= the columns)."""
        v = self.view()
        return abi.Schema.from_c(v.schema.contents) if v.schema else None

    def table_id(self):
        """(Schema, Table) of the batch's rows."""
        v = self.view()
        return (v.table_ns or b"").decode("utf-8"), (v.table_name or b"").decode("utf-8")

    def column_names(self) -> List[str]:
        v = self.view()
        return [v.cols[i].name.decode() for i in range(v.ncols)]

    def payload_bytes(self) -> int:
        """Bytes of column payload resident in HBM (values + offsets + data)."""
        v = self.view()
        n, tot = int(v.nrows), 0
        for i in range(v.ncols):
            c = v.cols[i]
            if c.repr in abi.VAR_REPRS:
                tot += (n + 1) * 4 + int(c.data_len)
            else:
                tot += n * np.dtype(abi.REPR_NP[c.repr]).itemsize + (n * 4 if c.nanos else 0)
        return tot

    def download(self) -> abi.Batch:
        v = self.view()
        n = int(v.nrows)
        nold = int(v.n_old_keys)
        cols, carr, keep = [], (abi.CColumn * max(v.ncols + nold, 1))(), []
        for i in range(v.ncols + nold):
            c = v.cols[i] if i < v.ncols else v.old_keys[i - v.ncols]
            col = abi.Column(c.name.decode(), abi.DTYPES[c.dtype], int(c.repr))
            carr[i].name, carr[i].dtype, carr[i].repr = c.name, c.dtype, c.repr
            if c.repr in abi.VAR_REPRS:
                col.offsets = np.zeros(n + 1, np.uint32)
                col.data = np.zeros(max(int(c.data_len), 1), np.uint8)
                carr[i].offsets, carr[i].data, carr[i].data_len = col.offsets.ctypes.data, col.data.ctypes.data, c.data_len
            else:
                col.values = np.zeros(max(n, 1), abi.REPR_NP[c.repr])
                carr[i].values = col.values.ctypes.data
                if c.nanos:
                    col.nanos = np.zeros(max(n, 1), np.int32)
                    carr[i].nanos = col.nanos.ctypes.data
            if c.validity:
                bm = np.zeros((n + 7) // 8 + 1, np.uint8)
                keep.append(bm)
                carr[i].validity = bm.ctypes.data
                col._bm = bm
            if c.absent:
                ab = np.zeros((n + 7) // 8 + 1, np.uint8)
                keep.append(ab)
                carr[i].absent = ab.ctypes.data
                col._ab = ab
            cols.append(col)
        hb = abi.CBatch()
        hb.nrows, hb.ncols, hb.cols, hb.mem = n, v.ncols, carr, abi.MEM_HOST
        opres = None
        if nold:
            hb.n_old_keys = nold
            hb.old_keys = C.cast(C.byref(carr, C.sizeof(abi.CColumn) * v.ncols), C.POINTER(abi.CColumn))
            if v.old_keys_present:
                opres = np.zeros((n + 7) // 8 + 1, np.uint8); hb.old_keys_present = opres.ctypes.data
        kind = src = part = None
        if v.kind:
            kind = np.zeros(max(n, 1), np.uint8); hb.kind = kind.ctypes.data
        if v.src_row:
            src = np.zeros(max(n, 1), np.int32); hb.src_row = src.ctypes.data
        if v.part_id:
            part = np.zeros(max(n, 1), np.uint32); hb.part_id = part.ctypes.data
        order = None
        if v.col_order:  # every row's own ColumnNames order (tfgpu_batch.col_order: a collapsed TOAST batch)
            order = np.zeros(max(n * v.ncols, 1), np.uint16); hb.col_order = order.ctypes.data
        _check(load().tfgpu_dbatch_download(self._h, C.byref(hb)))
        for col in cols:
            if col.repr in abi.VAR_REPRS:
                col.data = col.data[: int(col.offsets[-1])]
            else:
                col.values = col.values[:n]
                if col.nanos is not None:
                    col.nanos = col.nanos[:n]
            if hasattr(col, "_bm"):
                col.validity = abi.unpack_validity(col._bm, n)
            if hasattr(col, "_ab"):
                col.absent = abi.unpack_validity(col._ab, n)
        b = abi.Batch(cols[: v.ncols], n, (v.table_ns or b"").decode(), (v.table_name or b"").decode())
        if nold:
            b.old_keys = cols[v.ncols:]
            b.old_present = abi.unpack_validity(opres, n) if opres is not None else np.ones(n, bool)
        b.kind = kind[:n] if kind is not None else None
        # src_row NULL = identity (include/tfgpu.h): no row was dropped or reordered
        b.src_row = src[:n] if src is not None else np.arange(n, dtype=np.int32)
        b.part_id = part[:n] if part is not None else None
        if order is not None:
            b.col_order = order[: n * v.ncols].reshape(n, v.ncols)
        return b

    def free(self):
        if self._h:
            load().tfgpu_dbatch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def upload(data: bytes) -> "DeviceBuffer":
        init()
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        _check(load().tfgpu_dbuf_upload(buf.ctypes.data if len(buf) else None, len(buf), C.byref(h)))
        return DeviceBuffer(h)

    @staticmethod
    def alloc(nbytes: int) -> "DeviceBuffer":
        init()
        h = C.c_void_p()
        _check(load().tfgpu_dbuf_alloc(nbytes, C.byref(h)))
        return DeviceBuffer(h)

    def write(self, offset: int, arr: np.ndarray, nbytes: int):
        _check(load().tfgpu_dbuf_write(self._h, offset, arr.ctypes.data, nbytes))

    @property
    def size(self) -> int:
        n = C.c_uint64(0)
        load().tfgpu_dbuf_size(self._h, C.byref(n))
        return int(n.value)

    @property
    def ptr(self) -> int:
        return int(load().tfgpu_dbuf_ptr(self._h) or 0)

    def download(self) -> bytes:
        n = self.size
        out = np.zeros(max(n, 1), np.uint8)
        _check(load().tfgpu_dbuf_download(self._h, out.ctypes.data, n))
        return out[:n].tobytes()

    def free(self):
        if self._h:
            load().tfgpu_dbuf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Pinned (hipHostMalloc) staging memory: the source of a true asynchronous H2D copy."""

    def __init__(self, data: bytes):
        init()
        p = C.c_void_p()
        _check(load().tfgpu_host_alloc(max(len(data), 1), C.byref(p)))
        self.ptr, self.size = p.value, len(data)
        C.memmove(self.ptr, data, len(data))

    def free(self):
        if getattr(self, "ptr", None):
            load().tfgpu_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TransformerResult:
    """abstract.TransformerResult: Transformed rows + per-row Errors."""

    def __init__(self, transformed: DeviceBatch, errors):
        self.transformed = transformed
        self.errors = errors  # list of (row, code_name, step, column)


class Transformer:
    """abstract.Transformer backed by a tfgpu_plan."""

    def __init__(self, type_name: str, config):
        cfg = config if isinstance(config, str) else json.dumps(config)
        h = C.c_void_p()
        _check(load().tfgpu_plan_create(type_name.encode(), cfg.encode("utf-8"), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                load().tfgpu_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def type(self) -> str:
        return load().tfgpu_plan_type(self._h).decode()

    def description(self) -> str:
        buf = C.create_string_buffer(1024)
        _check(load().tfgpu_plan_description(self._h, buf, 1024))
        return buf.value.decode()

    def suitable(self, ns: str, table: str, schema: abi.Schema) -> bool:
        cs, out = schema.to_c(), C.c_int(0)
        _check(load().tfgpu_plan_suitable(self._h, ns.encode(), table.encode(), C.byref(cs), C.byref(out)))
        return bool(out.value)

    def result_schema(self, schema: abi.Schema) -> abi.Schema:
        cs = schema.to_c()
        out = C.POINTER(abi.CSchema)()
        _check(load().tfgpu_plan_result_schema(self._h, C.byref(cs), C.byref(out)))
        s = abi.Schema.from_c(out.contents)
        load().tfgpu_schema_free(out)
        return s

    def apply(self, batch: DeviceBatch) -> TransformerResult:
        return apply_chain([self], batch)


def apply_chain(transformers: Sequence[Transformer], batch: DeviceBatch, max_errors: int = 1 << 16) -> TransformerResult:
    """The Apply loop of transformation.do (pkg/transformer/transformation.go:252-274) in HBM."""
    init()
    n = len(transformers)
    arr = (C.c_void_p * max(n, 1))(*[t._h for t in transformers])
    out = C.c_void_p()
    errs = _errbuf(max_errors)
    nerr = C.c_int64(0)
    _check(load().tfgpu_apply(arr, n, batch._h, C.byref(out), errs, max_errors, C.byref(nerr)))
    el = [(int(errs[i].row), abi.ROWERR.get(int(errs[i].code), str(errs[i].code)), int(errs[i].step), int(errs[i].column))
          for i in range(min(int(nerr.value), max_errors))]
    return TransformerResult(DeviceBatch(out), el)


class PushResult:
    """What one run leaves of a transformation.Push: the transformed rows and, per failing transformer, the rows it refused
    AS IT SAW THEM (TransformerError.Input) with the reasons."""

    def __init__(self, transformed: DeviceBatch, error_batches, errors):
        self.transformed = transformed
        self.error_batches = error_batches  # [(index of the transformer, DeviceBatch)]
        self.errors = errors                # [(row in the original run, code_name, index of the transformer, column)], grouped like error_batches


class PushToken:
    def __init__(self, handle, owner, batch, max_errors):
        self._h, self._owner, self._batch, self._max = handle, owner, batch, max_errors

    def wait(self) -> PushResult:
        if self._h is None:
            raise RuntimeError("a push token is waited for exactly once")
        h, self._h = self._h, None
        return self._owner._collect(lambda *a: load().tfgpu_wait(h, *a), self._max)

    def __del__(self):
        # a token dropped without wait(): the job may still be running (the library keeps its own copy of the batch), and
        # the token + its event belong to the library until tfgpu_wait takes them
        try:
            if self._h is not None:
                self.wait().transformed.free()
        except Exception:
            pass


class Transformation:
    """transformation (pkg/transformer/transformation.go:20-45) behind the C ABI: the transformers in config order
    (+ ExtraTransformers), the plan cache per (TableID, schema), the Apply loop with error inputs, the stats."""

    def __init__(self, transformers: Sequence[Transformer]):
        init()
        self.transformers = list(transformers)  # keeps the plans alive
        n = len(self.transformers)
        arr = (C.c_void_p * max(n, 1))(*[t._h for t in self.transformers])
        h = C.c_void_p()
        _check(load().tfgpu_transformation_create(arr, n, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                load().tfgpu_transformation_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def table_plan(self, ns: str, table: str, schema: abi.Schema) -> List[int]:
        """AddTablePlan: indices of the transformers that are Suitable, each judged against its predecessors' ResultSchema."""
        cs = schema.to_c()
        idx, n = (C.c_int32 * max(len(self.transformers), 1))(), C.c_int32(0)
        _check(load().tfgpu_transformation_table_plan(self._h, ns.encode(), table.encode(), C.byref(cs), idx, len(self.transformers), C.byref(n)))
        return [int(idx[i]) for i in range(n.value)]

    def _collect(self, call, max_errors):
        out = C.c_void_p()
        cap = max(len(self.transformers), 1)
        eb, es, neb = (C.c_void_p * cap)(), (C.c_int32 * cap)(), C.c_int32(0)
        errs, nerr = (abi.CRowError * max_errors)(), C.c_int64(0)
        _check(call(C.byref(out), eb, es, cap, C.byref(neb), errs, max_errors, C.byref(nerr)))
        el = [(int(errs[i].row), abi.ROWERR.get(int(errs[i].code), str(errs[i].code)), int(errs[i].step), int(errs[i].column))
              for i in range(min(int(nerr.value), max_errors))]
        return PushResult(DeviceBatch(out), [(int(es[g]), DeviceBatch(C.c_void_p(eb[g]))) for g in range(min(neb.value, cap))], el)

    def push_run(self, batch: DeviceBatch, schema: Optional[abi.Schema] = None, max_errors: int = 1 << 16) -> PushResult:
        """One contiguous same-table, same-schema run through its table plan (transformation.do)."""
        cs = schema.to_c() if schema is not None else None
        return self._collect(lambda *a: load().tfgpu_transformation_push(self._h, batch._h, C.byref(cs) if cs is not None else None, *a), max_errors)

    def push_run_async(self, batch: DeviceBatch, schema: Optional[abi.Schema] = None, max_errors: int = 1 << 16) -> PushToken:
        cs = schema.to_c() if schema is not None else None
        tok = C.c_void_p()
        _check(load().tfgpu_transformation_push_async(self._h, batch._h, C.byref(cs) if cs is not None else None, C.byref(tok)))
        return PushToken(tok, self, batch, max_errors)

    def stats(self) -> dict:
        st = abi.CTransformationStats()
        _check(load().tfgpu_transformation_get_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in abi.CTransformationStats._fields_}


def executor_start(workers: int):
    init()
    _check(load().tfgpu_executor_start(int(workers)))


def collapse(batch: DeviceBatch) -> DeviceBatch:
    """abstract.Collapse (pkg/abstract/changeitem/change_item_collapse.go:48-134) on device."""
    init()
    out = C.c_void_p()
    _check(load().tfgpu_collapse(batch._h, C.byref(out)))
    return DeviceBatch(out)


def keys_changed(batch: DeviceBatch) -> np.ndarray:
    """ChangeItem.KeysChanged (change_item.go:237-286) of every row, as a bool array."""
    init()
    flags = np.zeros(max(batch.nrows, 1), dtype=np.uint8)
    cnt = C.c_int64(0)
    _check(load().tfgpu_keys_changed(batch._h, flags.ctypes.data_as(C.c_void_p), C.byref(cnt)))
    out = flags[:batch.nrows].astype(bool)
    assert int(out.sum()) == cnt.value
    return out


def split_updated_pkeys(batch: DeviceBatch):
    """abstract.SplitUpdatedPKeys (pkg/abstract/changeitem/utils.go:75-128) as a cut plan over the batch's rows:
    [("rows", a, b)] for a run of unchanged-key rows [a, b) kept in order, [("pkey_change", i)] for an Update whose
    primary key changed — the caller turns that one into its pair (Delete carrying the row's OldKeys and no columns,
    Insert carrying its ColumnValues and empty OldKeys, Counter + 1).  Empty runs are not emitted."""
    flags = keys_changed(batch)
    plan, a = [], 0
    for i in np.flatnonzero(flags):
        i = int(i)
        if i > a:
            plan.append(("rows", a, i))
        plan.append(("pkey_change", i))
        a = i + 1
    if a < batch.nrows:
        plan.append(("rows", a, batch.nrows))
    return plan


def partition(batch: DeviceBatch, nparts: int):
    """Regroup the rows by part_id (sharder_transformer's PartID), parts in order, row order kept inside a part.
    Returns (DeviceBatch, counts[nparts]): every column buffer is laid out for one all-to-all with those splits."""
    init()
    out = C.c_void_p()
    counts = (C.c_int64 * nparts)()
    _check(load().tfgpu_partition(batch._h, nparts, C.byref(out), counts))
    return DeviceBatch(out), [int(c) for c in counts]


def ch_native_block(batch: DeviceBatch, columns) -> "DeviceBuffer":
    """tfgpu_ch_native_block: the batch as one ClickHouse Native block; `columns` = [(batch column, ClickHouse type), ...]."""
    init()
    arr = (abi.CChNativeColumn * max(len(columns), 1))()
    for i, (n, t) in enumerate(columns):
        arr[i].name, arr[i].ch_type = n.encode(), t.encode()
    out = C.c_void_p()
    _check(load().tfgpu_ch_native_block(batch._h, arr, len(columns), C.byref(out)))
    return DeviceBuffer(out)


def deepsizeof(batch: DeviceBatch, per_row: bool = False, json_float64: bool = False):
    """util.DeepSizeof(ColumnValues) of every row (tfgpu_dbatch_deepsizeof): the total, or (total, per-row uint64 array)."""
    init()
    total = C.c_uint64(0)
    n = batch.nrows
    rows = np.zeros(max(n, 1), np.uint64) if per_row else None
    _check(load().tfgpu_dbatch_deepsizeof(batch._h, 1 if json_float64 else 0, rows.ctypes.data_as(C.POINTER(C.c_uint64)) if per_row else None, C.byref(total)))
    return (int(total.value), rows[:n]) if per_row else int(total.value)


COMM_ID_BYTES = 128


class Comm:
    """A communicator of the hash-partition exchange (tfgpu_comm_*): one per process, RCCL underneath."""

    def __init__(self, h, rank: int, world: int):
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id() -> bytes:
        """Rank 0's rendezvous id (ncclGetUniqueId): hand it to every rank over the control plane."""
        init()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load().tfgpu_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def create(uid: bytes, rank: int, world: int) -> "Comm":
        init()
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        _check(load().tfgpu_comm_init(uid, rank, world, C.byref(h)))
        return Comm(h, rank, world)

    def exchange(self, batch: DeviceBatch, counts):
        """tfgpu_exchange: `batch` grouped by destination rank (tfgpu_partition), counts[d] rows for rank d.
        Returns (DeviceBatch of the rows sent to this rank, rows received from each source rank)."""
        if len(counts) != self.world:
            raise ValueError("one count per rank")
        cin = (C.c_int64 * self.world)(*[int(c) for c in counts])
        cout = (C.c_int64 * self.world)()
        out = C.c_void_p()
        _check(load().tfgpu_exchange(self._h, batch._h, cin, C.byref(out), cout))
        return DeviceBatch(out), [int(c) for c in cout]

    def close(self):
        if self._h:
            load().tfgpu_comm_destroy(self._h)
            self._h = None


_tls = threading.local()


def _errbuf(max_errors: int):
    """the per-row error array of a call, kept per thread and size: allocating (and zeroing) 2^16 entries costs ~70 us of host time
    per call, which a Go caller with its own slice never pays"""
    cache = getattr(_tls, "errs", None)
    if cache is None:
        cache = _tls.errs = {}
    buf = cache.get(max_errors)
    if buf is None:
        buf = cache[max_errors] = (abi.CRowError * max_errors)()
    return buf


def csv_parse(opts: abi.CCsvOptions, schema, data, max_errors: int = 1 << 16):
    """s3 CSVReader.parseCSVRows + Strictify on device.  `data` is bytes or a DeviceBuffer; `schema` an abi.Schema or the
    tfgpu_schema it converts to (abi.Schema.to_c() of 105 columns is ~0.5 ms of Python: callers with a fixed schema convert once).
    Returns (DeviceBatch, consumed_bytes, errors)."""
    init()
    L = load()
    cs = schema if isinstance(schema, abi.CSchema) else schema.to_c()
    out, consumed, nerr = C.c_void_p(), C.c_uint64(0), C.c_int64(0)
    errs = _errbuf(max_errors)
    if isinstance(data, DeviceBuffer):
        _check(L.tfgpu_csv_parse(C.byref(opts), C.byref(cs), data.ptr, data.size, abi.MEM_DEVICE, C.byref(out), C.byref(consumed), errs,
                                 max_errors, C.byref(nerr)))
    elif isinstance(data, HostBuffer):
        _check(L.tfgpu_csv_parse(C.byref(opts), C.byref(cs), data.ptr, data.size, abi.MEM_HOST, C.byref(out), C.byref(consumed), errs,
                                 max_errors, C.byref(nerr)))
    else:
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(L.tfgpu_csv_parse(C.byref(opts), C.byref(cs), buf.ctypes.data if len(buf) else None, len(buf), abi.MEM_HOST, C.byref(out),
                                 C.byref(consumed), errs, max_errors, C.byref(nerr)))
    el = [(int(errs[i].row), abi.ROWERR.get(int(errs[i].code), str(errs[i].code)), int(errs[i].step), int(errs[i].column))
          for i in range(min(int(nerr.value), max_errors))]
    return DeviceBatch(out), int(consumed.value), el


def csv_split_rows(data) -> np.ndarray:
    """csv.Splitter (pkg/csv/splitter.go): offsets one past the '\\n' of every complete entry of the chunk."""
    init()
    out, n = C.c_void_p(), C.c_int64(0)
    p, ln, mem, keep = _bytes_arg(data)
    _check(load().tfgpu_csv_split_rows(p, ln, mem, C.byref(out), C.byref(n)))
    buf = DeviceBuffer(out)
    raw = buf.download()
    buf.free()
    return np.frombuffer(raw, dtype=np.uint32)[: n.value].copy()


def json_parse(opts: abi.CJsonOptions, fields: abi.Schema, data, msgs: Optional[abi.CMessages] = None, max_errors: int = 1 << 16):
    """GenericParser{Format: "json"}.DoBatch on device (pkg/parsers/generic/generic_parser.go:406-438).
    `data` is bytes or a DeviceBuffer holding the concatenated Message.Value bytes.
    Returns (DeviceBatch, errors); errors = (line ordinal, code name, message index, column)."""
    init()
    L = load()
    cs = fields.to_c()
    out, nerr = C.c_void_p(), C.c_int64(0)
    errs = _errbuf(max_errors)
    mp = C.byref(msgs) if msgs is not None else None
    if isinstance(data, DeviceBuffer):
        _check(L.tfgpu_json_parse(C.byref(opts), C.byref(cs), data.ptr, data.size, abi.MEM_DEVICE, mp, C.byref(out), errs, max_errors, C.byref(nerr)))
    else:
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(L.tfgpu_json_parse(C.byref(opts), C.byref(cs), buf.ctypes.data if len(buf) else None, len(buf), abi.MEM_HOST, mp, C.byref(out), errs,
                                  max_errors, C.byref(nerr)))
    el = [(int(errs[i].row), abi.ROWERR.get(int(errs[i].code), str(errs[i].code)), int(errs[i].step), int(errs[i].column))
          for i in range(min(int(nerr.value), max_errors))]
    return DeviceBatch(out), el


def _bytes_arg(data):
    """(pointer, length, mem, keep-alive) of host bytes, a pinned HostBuffer or a DeviceBuffer."""
    if isinstance(data, DeviceBuffer):
        return data.ptr, data.size, abi.MEM_DEVICE, data
    if isinstance(data, HostBuffer):
        return data.ptr, data.size, abi.MEM_HOST, data
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return buf.ctypes.data, len(data), abi.MEM_HOST, buf


def sr_frames(data, msgs: Optional[abi.CMessages] = None):
    """ConfluentSrImpl.DoBuf's walk over every message: [(msg, start, len, schema_id, code, index)] — the shim asks the
    registry for the schema ids it finds here."""
    init()
    ptr_, n, mem, keep = _bytes_arg(data)
    cap = max(int(msgs.nmsg) if msgs is not None else 1, 16)
    while True:
        arr = (abi.CSrFrame * cap)()
        nf = C.c_int64(0)
        rc = load().tfgpu_sr_frames(ptr_, n, mem, C.byref(msgs) if msgs is not None else None, arr, cap, C.byref(nf))
        if rc != 0 and nf.value > cap:
            cap = int(nf.value)
            continue
        _check(rc)
        return [(int(f.msg), int(f.start), int(f.len), int(f.schema_id), int(f.code), int(f.index)) for f in arr[:int(nf.value)]]


class ParseResult:
    def __init__(self, batch: "DeviceBatch", errors):
        self.device_batch, self.errors = batch, errors
        self._host = None

    @property
    def batch(self) -> abi.Batch:
        if self._host is None:
            self._host = self.device_batch.download()
        return self._host


def sr_json_parse(opts: abi.CSrJsonOptions, data, msgs: Optional[abi.CMessages] = None, max_errors: int = 1 << 16) -> ParseResult:
    """Confluent SR parser, JSON-schema frames of opts.schema_id → one device batch (rows: src_row = frame ordinal,
    part_id = message index) + [(frame ordinal, code, message index)] for the frames that become `_unparsed` items."""
    init()
    ptr_, n, mem, keep = _bytes_arg(data)
    errs = _errbuf(max_errors)
    ne, out = C.c_int64(0), C.c_void_p()
    _check(load().tfgpu_sr_json_parse(C.byref(opts), ptr_, n, mem, C.byref(msgs) if msgs is not None else None, C.byref(out), errs, max_errors, C.byref(ne)))
    return ParseResult(DeviceBatch(out), [(int(errs[i].row), int(errs[i].code), int(errs[i].step)) for i in range(min(int(ne.value), max_errors))])


def sr_proto_parse(schema_id: int, schema_text: bytes, data, msgs: Optional[abi.CMessages] = None, policy: str = "debezium_style", manual_table_name: str = ""):
    """Confluent SR parser, PROTOBUF schema `schema_id` (tfgpu_sr_compile_proto + tfgpu_sr_proto_parse).  Returns
    (ns, table, names, {message index: [(gotype, value)]}, {message index: TFGPU_ROW_* code}); with a schema the device does not take
    every message of the id carries the schema's code."""
    import sys
    from . import confluent_sr
    s = confluent_sr.ProtoSchema(sys.modules[__name__], schema_text, policy, manual_table_name)
    db, errors = s.parse(schema_id, data, msgs, report_frame_errors=False)
    b = db.download()
    rows = {int(b.src_row[r]): [c.pyvalue(r) for c in b.cols] for r in range(b.nrows)}
    return s.ns, s.table, [c.name for c in b.cols], rows, errors


def debezium_unpack(data, msgs: Optional[abi.CMessages] = None, known=None) -> np.ndarray:
    """IncludeSchema.Unpack for every message (tfgpu_debezium_unpack): a structured array (abi.DBZ_FRAME_DTYPE) of nmsg frames.
    known = (prefix bytes, schema_off, schema_len, (hash0, hash1)) of an earlier batch: tfgpu_debezium_unpack_cached."""
    init()
    ptr_, n, mem, keep = _bytes_arg(data)
    nmsg = msgs.nmsg if msgs is not None else 1
    frames = np.zeros(max(nmsg, 1), abi.DBZ_FRAME_DTYPE)
    pm = C.byref(msgs) if msgs is not None else None
    if known is None:
        _check(load().tfgpu_debezium_unpack(ptr_, n, mem, pm, frames.ctypes.data))
    else:
        pre, so, sl, h = known
        buf = np.frombuffer(pre, dtype=np.uint8)
        k = abi.CDbzPrefix()
        k.bytes, k.len, k.schema_off, k.schema_len = buf.ctypes.data, len(pre), so, sl
        k.schema_hash[0], k.schema_hash[1] = int(h[0]), int(h[1])
        _check(load().tfgpu_debezium_unpack_cached(ptr_, n, mem, pm, C.byref(k), frames.ctypes.data))
    return frames[:nmsg]


def debezium_parse(schema_hash, fields, data, frames: np.ndarray, msgs: Optional[abi.CMessages] = None, max_errors: int = 1 << 16, schema_code: int = 0):
    """Receiver.receive for the frames of one schema (tfgpu_debezium_parse).  fields = [(name, DBZ_*, optional, scale)].
    Returns (DeviceBatch, rows: structured array abi.DBZ_ROW_DTYPE, [(message, code)])."""
    init()
    ptr_, n, mem, keep = _bytes_arg(data)
    nmsg = len(frames)
    o = abi.CDbzOptions()
    o.schema_hash[0], o.schema_hash[1] = int(schema_hash[0]), int(schema_hash[1])
    arr = (abi.CDbzField * max(len(fields), 1))()
    for i, (name, op, optional, scale) in enumerate(fields):
        arr[i].name, arr[i].op, arr[i].optional, arr[i].scale = name.encode("utf-8"), op, 1 if optional else 0, scale
    o.nfields, o.fields, o.schema_code = len(fields), arr, schema_code
    rows = np.zeros(max(nmsg, 1), abi.DBZ_ROW_DTYPE)
    fr = np.ascontiguousarray(frames)
    errs = _errbuf(max_errors)
    ne, out = C.c_int64(0), C.c_void_p()
    _check(load().tfgpu_debezium_parse(C.byref(o), ptr_, n, mem, C.byref(msgs) if msgs is not None else None, fr.ctypes.data, C.byref(out), rows.ctypes.data, nmsg,
                                       errs, max_errors, C.byref(ne)))
    db = DeviceBatch(out)
    return db, rows[:db.nrows], [(int(errs[i].row), int(errs[i].code)) for i in range(min(int(ne.value), max_errors))]


def json_result_schema(opts: abi.CJsonOptions, fields: abi.Schema) -> abi.Schema:
    """GenericParser.ResultSchema(): the raw fields plus the aux columns the options add."""
    cs = fields.to_c()
    out = C.POINTER(abi.CSchema)()
    _check(load().tfgpu_json_result_schema(C.byref(opts), C.byref(cs), C.byref(out)))
    s = abi.Schema.from_c(out.contents)
    load().tfgpu_schema_free(out)
    return s


def serialize(fmt: int, batch: DeviceBatch, opts: Optional[abi.CSerializeOptions] = None) -> DeviceBuffer:
    """BatchSerializer.Serialize / MarshalCItoJSON per row, from device columns to device text."""
    init()
    out = C.c_void_p()
    if opts is None:
        _check(load().tfgpu_serialize(fmt, batch._h, C.byref(out)))
    else:
        _check(load().tfgpu_serialize_ex(fmt, batch._h, C.byref(opts), C.byref(out)))
    return DeviceBuffer(out)


def serialize_batch(fmt: int, batch: DeviceBatch, opts: Optional[abi.CSerializeOptions] = None, concurrency: int = 0, threshold: int = 0, disable_concurrency: bool = False,
                    gomaxprocs: int = 8, for_writer: bool = False):
    """batchSerializer.Serialize (→ DeviceBuffer) / SerializeAndWrite (→ (DeviceBuffer, [where each Write call ends]))."""
    init()
    cfg = abi.CBatchSerializerConfig(int(concurrency), int(threshold), int(bool(disable_concurrency)), int(gomaxprocs))
    out = C.c_void_p()
    cap = max(1, (batch.nrows + max(threshold or 25000, 1) - 1) // max(threshold or 25000, 1))
    ends = (C.c_uint64 * cap)()
    nparts = C.c_int64(0)
    _check(load().tfgpu_serialize_batch(fmt, batch._h, C.byref(opts) if opts is not None else None, C.byref(cfg), int(bool(for_writer)), C.byref(out), ends, cap, C.byref(nparts)))
    buf = DeviceBuffer(out)
    return (buf, [int(ends[i]) for i in range(nparts.value)]) if for_writer else buf


class QueueMessages:
    """queue.Serializer output for one table's rows: message values back to back in HBM + where each one starts.
    Keys stay with the caller: Fqtn() (`schema_table`) when batching is off, nil when it is on."""

    def __init__(self, values: DeviceBuffer, msg_start: np.ndarray, msg_row: np.ndarray):
        self.values, self.msg_start, self.msg_row = values, msg_start, msg_row

    def __len__(self):
        return len(self.msg_start) - 1

    def messages(self) -> List[bytes]:
        raw = self.values.download()
        return [raw[int(self.msg_start[i]):int(self.msg_start[i + 1])] for i in range(len(self))]


def queue_serialize(opts: abi.CQueueOptions, batch: DeviceBatch, meta: Optional[abi.CRowMeta] = None) -> QueueMessages:
    """NativeSerializer / JSONSerializer.Serialize (pkg/serializer/queue) for one table's rows, from device columns."""
    init()
    cap = batch.nrows
    start, row = np.zeros(cap + 1, np.uint64), np.zeros(cap + 1, np.int64)
    out, n = C.c_void_p(), C.c_int64(0)
    _check(load().tfgpu_queue_serialize(C.byref(opts), batch._h, C.byref(meta) if meta is not None else None, C.byref(out),
                                        start.ctypes.data, row.ctypes.data, cap, C.byref(n)))
    k = int(n.value)
    return QueueMessages(DeviceBuffer(out), start[:k + 1].copy(), row[:k + 1].copy())


class DebeziumMessages:
    """Emitter.EmitKV's (key, value) pairs for one table's rows: keys and values back to back in HBM, where each one starts, which
    values are nil (tombstones) and the batch row every message came from."""

    def __init__(self, keys: DeviceBuffer, key_start, values: DeviceBuffer, val_start, val_null, msg_row):
        self.keys, self.key_start, self.values, self.val_start, self.val_null, self.msg_row = keys, key_start, values, val_start, val_null, msg_row

    def __len__(self):
        return len(self.msg_row)

    def messages(self):
        """[(key bytes, value bytes or None)]"""
        k, v = self.keys.download(), self.values.download()
        return [(k[int(self.key_start[i]):int(self.key_start[i + 1])], None if self.val_null[i] else v[int(self.val_start[i]):int(self.val_start[i + 1])])
                for i in range(len(self))]


def debezium_emit(opts: abi.CDbzEmitOptions, batch: DeviceBatch, meta: Optional[abi.CRowMeta] = None) -> DebeziumMessages:
    """queue.DebeziumSerializer (pkg/serializer/queue/debezium_serializer.go) for one table's rows, from device columns."""
    init()
    cap = 3 * batch.nrows
    ks, vs = np.zeros(cap + 1, np.uint64), np.zeros(cap + 1, np.uint64)
    nul, row = np.zeros(cap + 1, np.uint8), np.zeros(cap + 1, np.int64)
    ko, vo, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
    _check(load().tfgpu_debezium_emit(C.byref(opts), batch._h, C.byref(meta) if meta is not None else None, C.byref(ko), ks.ctypes.data, C.byref(vo),
                                      vs.ctypes.data, nul.ctypes.data, row.ctypes.data, cap, C.byref(n)))
    k = int(n.value)
    return DebeziumMessages(DeviceBuffer(ko), ks[:k + 1].copy(), DeviceBuffer(vo), vs[:k + 1].copy(), nul[:k].copy(), row[:k].copy())


def fqtn(ns: str, table: str) -> str:
    """ChangeItem.Fqtn() (change_item.go:139-141): the Kafka message key of the unbatched queue serializers."""
    return ns + "_" + table


def prof_enable(on: bool):
    _check(load().tfgpu_prof_enable(1 if on else 0))


def prof_reset():
    _check(load().tfgpu_prof_reset())


def prof_get():
    """[(kernel name, launches, total_ms)] measured with HIP events on the library stream."""
    L = load()
    out = []
    for i in range(L.tfgpu_prof_count()):
        name, n, ms = C.c_char_p(), C.c_int64(0), C.c_double(0)
        L.tfgpu_prof_get(i, C.byref(name), C.byref(n), C.byref(ms))
        out.append((name.value.decode(), int(n.value), float(ms.value)))
    return out


def prof_units():
    """{kernel name: units (rows) its timed launches were issued over} — tfgpu_prof_get_units; 0 where the call site states none."""
    L = load()
    out = {}
    for i in range(L.tfgpu_prof_count()):
        name, n, ms, u = C.c_char_p(), C.c_int64(0), C.c_double(0), C.c_int64(0)
        L.tfgpu_prof_get(i, C.byref(name), C.byref(n), C.byref(ms))
        L.tfgpu_prof_get_units(i, C.byref(u))
        out[name.value.decode()] = int(u.value)
    return out
