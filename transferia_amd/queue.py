"""The queue serializers whose messages ARE column bytes, and the Kafka writer's partitioner, as the C ABI exposes
them — ctypes pass-throughs over device batches; the rules live in transferia_amd/csrc/tf_serialize.hip:

    RawColumnSerializer      raw_column_serializer.go:21-73    tfgpu_queue_raw_column
    MirrorSerializer         mirror_serializer.go:15-52        tfgpu_queue_mirror
    splitByTablePartID       split.go:5-12                     tfgpu_queue_part_groups
    kafka-go Hash balancer   balancer.go:153-181               tfgpu_kafka_hash_partition / tfgpu_kafka_partitions
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import abi, lib

RAW_DATA_COLUMNS = ("topic", "partition", "seq_no", "write_time", "data", "meta", "sequence_key")  # changeitem/mirror.go:23-32
# RawDataSchema (changeitem/mirror.go:23-31): (name, DataType, PrimaryKey, path, OriginalType, Required)
RAW_DATA_SCHEMA = abi.Schema.of([["topic", "utf8", True, "", "", True], ["partition", "uint32", True, "", "", True], ["seq_no", "uint64", True, "", "", True],
                                 ["write_time", "datetime", True, "", "", True], ["data", "utf8", False, "", "mirror:binary"], ["meta", "any"],
                                 ["sequence_key", "string"]])


def fqtn(ns: str, table: str) -> str:
    """ChangeItem.Fqtn() (change_item.go:139-141): the key of the native / JSON serializers' one-item messages."""
    return ns + "_" + table


def _cut(buf: lib.DeviceBuffer, starts, n: int) -> List[bytes]:
    raw = buf.download()
    return [raw[int(starts[i]):int(starts[i + 1])] for i in range(n)]


def raw_column_messages(batch: lib.DeviceBatch, column_name: str, schema: Optional[abi.Schema] = None) -> List[bytes]:
    lib.init()
    cap = max(batch.nrows, 1)
    starts = (C.c_uint32 * (cap + 1))()
    out, n = C.c_void_p(), C.c_int64(0)
    cs = schema.to_c() if schema is not None else None
    lib._check(lib.load().tfgpu_queue_raw_column(batch._h, column_name.encode(), C.byref(cs) if cs is not None else None, C.byref(out), starts, cap, C.byref(n)))
    return _cut(lib.DeviceBuffer(out), starts, n.value)


def mirror_messages(batch: lib.DeviceBatch, schema: Optional[abi.Schema] = None) -> List[Tuple[Optional[bytes], bytes]]:
    lib.init()
    cap = max(batch.nrows, 1)
    vs, ks = (C.c_uint32 * (cap + 1))(), (C.c_uint32 * (cap + 1))()
    nil = (C.c_uint8 * cap)()
    v, k, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
    cs = schema.to_c() if schema is not None else None
    lib._check(lib.load().tfgpu_queue_mirror(batch._h, C.byref(cs) if cs is not None else None, C.byref(v), vs, C.byref(k), ks, nil, cap, C.byref(n)))
    vals, keys = _cut(lib.DeviceBuffer(v), vs, n.value), _cut(lib.DeviceBuffer(k), ks, n.value)
    return [(None if nil[i] else keys[i], vals[i]) for i in range(n.value)]


def part_groups(batch: lib.DeviceBatch):
    """→ (row order grouping equal PartIDs, rows per group, PartID text per group)"""
    lib.init()
    n = batch.nrows
    order = np.zeros(max(n, 1), np.int32)
    cap = max(n, 1)
    rows, ids, ng = (C.c_int64 * cap)(), (C.c_uint32 * cap)(), C.c_int64(0)
    lib._check(lib.load().tfgpu_queue_part_groups(batch._h, order.ctypes.data_as(C.POINTER(C.c_int32)), rows, ids, cap, C.byref(ng)))
    has = batch.view().part_id is not None and bool(batch.view().part_id)
    return order[:n], [int(rows[g]) for g in range(ng.value)], [str(int(ids[g])) if has else "" for g in range(ng.value)]


def kafka_hash_partition(key: Optional[bytes], npartitions: int) -> Optional[int]:
    if key is None:
        return None
    lib.load().tfgpu_kafka_hash_partition.restype = C.c_int32
    return int(lib.load().tfgpu_kafka_hash_partition(key, len(key), npartitions))


def kafka_partitions(batch: lib.DeviceBatch, key_column: str, npartitions: int) -> List[Optional[int]]:
    lib.init()
    out = np.zeros(max(batch.nrows, 1), np.int32)
    lib._check(lib.load().tfgpu_kafka_partitions(batch._h, key_column.encode(), npartitions, out.ctypes.data_as(C.POINTER(C.c_int32))))
    return [None if p < 0 else int(p) for p in out[:batch.nrows]]
