"""CPU restatement of the queue serializers whose messages are column bytes, and of the Kafka writer's partitioner
(pkg/serializer/queue).  TEST INFRASTRUCTURE: only tests/ may import this; the product path is
transferia_amd/queue.py → tfgpu_queue_raw_column / tfgpu_queue_mirror / tfgpu_queue_part_groups / tfgpu_kafka_* (C ABI).
Pinned by tests/test_queue_host.py to raw_column_serializer_test.go, mirror_serializer_test.go and kafka-go's balancer_test.go.

    Serializer.Serialize's message keys          native_serializer.go:24, json_serializer.go:37, *_batcher.go (Key: nil)
    splitByTablePartID                           split.go:5-12 (rows grouped by ChangeItem.PartID, one group per call)
    RawColumnSerializer                          raw_column_serializer.go:21-73 (a message per row: one column's bytes)
    MirrorSerializer                             mirror_serializer.go:15-52 (a message per row: `data` → value, `sequence_key` → key)

RawColumnSerializer needs no kernel: its message values ARE the (offsets, data) buffers of the named column, as
tfgpu_dbatch_view / download expose them.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from transferia_amd import abi


def fqtn(ns: str, table: str) -> str:
    """ChangeItem.Fqtn() (change_item.go:139-141)."""
    return ns + "_" + table


def message_keys(ns: str, table: str, nmsg: int, batching_enabled: bool) -> List[Optional[bytes]]:
    """Keys of the native / JSON serializers' messages: Fqtn() for one-item messages, nil for batched ones."""
    return [None] * nmsg if batching_enabled else [fqtn(ns, table).encode("utf-8")] * nmsg


def part_groups(part_id: Optional[np.ndarray], nrows: int) -> Tuple[np.ndarray, List[int], List[str]]:
    """splitByTablePartID for one table: (stable row order grouping equal PartIDs, rows per group, PartID text per group).
    Go maps iterate in no defined order; the groups come out by first appearance."""
    if part_id is None:
        return np.arange(nrows), [nrows], [""]
    _, first, inv = np.unique(part_id, return_index=True, return_inverse=True)
    rank = np.argsort(np.argsort(first, kind="stable"), kind="stable")  # group number by first appearance
    g = rank[inv]
    order = np.argsort(g, kind="stable")
    counts = np.bincount(g, minlength=len(first)).tolist()
    ids = [str(int(part_id[np.sort(first)[k]])) for k in range(len(first))]
    return order, counts, ids


def raw_column_messages(batch: abi.Batch, column_name: str, schema: Optional[abi.Schema] = None) -> List[bytes]:
    """RawColumnSerializer.Serialize for one table / PartID: the value of `column_name` of every row, as bytes; rows the
    reference skips with a warning (column absent from ColumnNames or from the TableSchema, DataType neither "utf8" nor
    "string", a value that is not a Go string / []byte — nil included) are skipped here too."""
    col = next((c for c in batch.cols if c.name == column_name), None)
    if col is None:
        return []
    sch = schema if schema is not None else getattr(batch, "schema", None)
    dtype = col.dtype
    if sch is not None:
        sc = next((c for c in sch.cols if c.name == column_name), None)
        if sc is None:
            return []  # "table schema does not contain column"
        dtype = sc.dtype
    if dtype not in ("utf8", "string"):
        return []
    if col.repr not in (abi.R_STRING, abi.R_BYTES):
        return []  # "unexpected column value type"
    return [col.get_bytes(i) for i in range(batch.nrows) if col.is_valid(i)]


RAW_DATA_COLUMNS = ("topic", "partition", "seq_no", "write_time", "data", "meta", "sequence_key")  # changeitem/mirror.go:23-32
# RawDataSchema (changeitem/mirror.go:23-31): (name, DataType, PrimaryKey, path, OriginalType, Required)
RAW_DATA_SCHEMA = abi.Schema.of([["topic", "utf8", True, "", "", True], ["partition", "uint32", True, "", "", True], ["seq_no", "uint64", True, "", "", True],
                                 ["write_time", "datetime", True, "", "", True], ["data", "utf8", False, "", "mirror:binary"], ["meta", "any"],
                                 ["sequence_key", "string"]])


def is_mirror(batch: abi.Batch) -> bool:
    """ChangeItem.IsMirror (change_item.go:385-395): ColumnNames are exactly RawDataColumns, in order."""
    return tuple(c.name for c in batch.cols) == RAW_DATA_COLUMNS


def _is_raw_data_schema(schema: abi.Schema) -> bool:
    """GetSequenceKey compares the TableSchema POINTER with RawDataSchema (mirror.go:71-73).  Value identity is the
    closest a columnar batch has: same names, types, key / required flags and original types, position by position."""
    want = RAW_DATA_SCHEMA.cols
    return len(schema.cols) == len(want) and all(
        (a.name, a.dtype, a.key, a.required, a.original_type) == (b.name, b.dtype, b.key, b.required, b.original_type) for a, b in zip(schema.cols, want))


def mirror_messages(batch: abi.Batch) -> List[Tuple[Optional[bytes], bytes]]:
    """MirrorSerializer.Serialize for one table / PartID: (key, value) per row = (`sequence_key`, `data`) — two column
    views, no kernel.  Raises ValueError where the reference fails: rows that are not mirror items
    (mirror_serializer.go:16-18); a TableSchema that is not RawDataSchema (GetSequenceKey, mirror.go:71-73 — checked when
    the batch carries its schema); a `sequence_key` column that does not hold []byte (the unchecked `.([]byte)` at
    mirror.go:74 panics on a string); a `data` value that is neither text nor bytes (mirror.go:78-87; nil included).
    A null in the []byte `sequence_key` column is MakeRawMessage's typed nil []byte: the message has no key.  (An UNTYPED
    nil interface would panic in Go; a column cannot tell the two apart and every producer in the reference passes a
    []byte, so null is read as the typed nil.)"""
    if not is_mirror(batch):
        raise ValueError("MirrorSerializer should be used only with 'Mirror' changeItems")
    schema = getattr(batch, "schema", None)
    if schema is not None and not _is_raw_data_schema(schema):
        raise ValueError("unable to get sequence key: changeItem should be 'mirror'")
    data, key = batch.cols[RAW_DATA_COLUMNS.index("data")], batch.cols[RAW_DATA_COLUMNS.index("sequence_key")]
    if key.repr != abi.R_BYTES and batch.nrows and any(key.is_valid(i) for i in range(batch.nrows)):
        raise ValueError("interface conversion: sequence_key is not []byte")
    if data.repr not in (abi.R_STRING, abi.R_BYTES):
        raise ValueError("unable to get message: unexpected data type, expected string or []byte")
    out = []
    for i in range(batch.nrows):
        if not data.is_valid(i):
            raise ValueError("unable to get message: unexpected data type: <nil>, expected string or []byte")
        out.append((key.get_bytes(i) if key.is_valid(i) else None, data.get_bytes(i)))
    return out


def kafka_hash_partition(key: Optional[bytes], npartitions: int) -> Optional[int]:
    """kafka-go's Hash balancer (vendor_patched/github.com/segmentio/kafka-go/balancer.go:153-181), the writer the Kafka
    sink uses: FNV-1a(32) of the key, the hash taken as an int32, Go's truncated remainder, a negative result negated
    — Sarama's hashPartitioner.  None for a nil key (round robin there).  All messages of one unbatched table share
    the key Fqtn(), so a table lands in ONE partition; batched messages (nil keys) are spread round robin."""
    if key is None:
        return None
    h = 0x811C9DC5
    for b in key:
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    v = h - (1 << 32) if h & 0x80000000 else h  # int32(hasher.Sum32())
    r = abs(v) % npartitions                       # Go's % keeps the dividend's sign; the balancer negates a negative result
    return r
