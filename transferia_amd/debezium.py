"""Host side of the Debezium parser with inline schemas: what the reference does ONCE PER SCHEMA before any
value is touched — unmarshal the Kafka Connect schema, find the `before` / `after` structs, resolve every field to its
receiver and column type — and the loop that hands each group of messages to the device.

    Receiver.receiveSchema / receiveTableSchema    pkg/debezium/receiver.go:60-96, 45-59
    receiveFieldColSchema                          pkg/debezium/receiver_engine.go:108-146
    TypeToDefault, Point / VariableScaleDecimal /
    Decimal matchers                               pkg/debezium/common/field_receiver_default.go:14-31, 258-355
    Schema                                         pkg/debezium/common/debezium_schema.go:12-29, 84-101
    DebeziumImpl.DoBatch                           pkg/parsers/registry/debezium/engine/parser.go:120-130

Scope: NewDebeziumImpl(logger, nil, threads) — no schema registry, no original-type table.  The reference caches the
compiled schema by a hash of its bytes (receiver.go:61-66); so does `Parser`, keyed by the device's hash of the same bytes.
The per-message work is tfgpu_debezium_unpack / tfgpu_debezium_parse (csrc/tf_debezium.hip).
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi

INT32_MIN = -(1 << 31)
OP_DTYPE = {abi.DBZ_BOOLEAN: "boolean", abi.DBZ_INT8: "int8", abi.DBZ_INT16: "int16", abi.DBZ_INT32: "int32", abi.DBZ_INT64: "int64", abi.DBZ_FLOAT64: "double",
            abi.DBZ_STRING: "utf8", abi.DBZ_BYTES: "string", abi.DBZ_DECIMAL: "utf8", abi.DBZ_POINT: "utf8", abi.DBZ_VSD: "double", abi.DBZ_HOST: "any"}


class SchemaError(ValueError):
    """receiveSchema fails: every message of the schema becomes an `_unparsed` item (TFGPU_ROW_DBZ_SCHEMA)."""


class HostOnly(ValueError):
    """The stock code must handle this schema (arrays, __dt_original_type_info, keys bound by case folding, a Go panic)."""


class _Obj(list):
    """A JSON object as its (key, value) pairs in document order (json.loads object_pairs_hook); arrays stay plain lists."""


def _member(pairs, name: str):
    """encoding/json's struct-field binding: a key names the field exactly or under case folding; the LAST such key wins.
    Returns (value, found); a key that matches only by folding is left to the host (the device binds exact keys only)."""
    out, found = None, False
    for k, v in pairs:
        if k == name:
            out, found = v, True
        elif k.lower() == name.lower():
            raise HostOnly("key %r binds %r by case folding" % (k, name))
    return out, found


def _string(pairs, name: str) -> str:
    v, found = _member(pairs, name)
    if not found or v is None:
        return ""
    if not isinstance(v, str):
        raise SchemaError("json: cannot unmarshal %s into Go struct field Schema.%s of type string" % (type(v).__name__, name))
    return v


class _Schema:
    """debezium_schema.go:12-22, as json.Unmarshal fills it."""

    def __init__(self, node, depth=0):
        self.field = self.name = self.type = self.scale = ""
        self.optional, self.fields, self.has_parameters, self.has_dt_info = False, [], False, False
        if node is None:
            return
        if not isinstance(node, _Obj) or depth > 64:
            raise SchemaError("json: cannot unmarshal into Go value of type common.Schema")
        self.field, self.name, self.type = _string(node, "field"), _string(node, "name"), _string(node, "type")
        v, found = _member(node, "optional")
        if found and v is not None:
            if not isinstance(v, bool):
                raise SchemaError("Schema.optional")
            self.optional = v
        v, found = _member(node, "version")
        if found and v is not None and (isinstance(v, bool) or not isinstance(v, int) or not -(1 << 63) <= v < (1 << 63)):
            raise SchemaError("Schema.version")
        v, found = _member(node, "parameters")
        if found and v is not None:
            if not _is_object(v):
                raise SchemaError("Schema.parameters")
            self.has_parameters = True
            for k in ("length", "connect.decimal.precision", "allowed"):
                _string(v, k)
            self.scale = _string(v, "scale")
        v, found = _member(node, "items")
        if found and v is not None:
            if not _is_object(v):
                raise SchemaError("Schema.items")
            _Schema(v, depth + 1)
        v, found = _member(node, "__dt_original_type_info")
        self.has_dt_info = found and v is not None
        v, found = _member(node, "fields")
        if found and v is not None:
            if not isinstance(v, list) or isinstance(v, _Obj):
                raise SchemaError("Schema.fields")
            self.fields = [_Schema(x, depth + 1) for x in v]

    def find(self, field_name: str) -> Optional["_Schema"]:
        for f in self.fields:  # FindSchemaDescr: the first one
            if f.field == field_name:
                return f
        return None


def _is_object(v) -> bool:
    return isinstance(v, _Obj)


def _receiver(f: _Schema) -> Tuple[int, int]:
    """(DBZ_* op, scale) of one field with an empty original type: receiveFieldColSchema + the default matchers."""
    if f.has_dt_info or f.type == "array":
        return abi.DBZ_HOST, 0
    simple = {"int8": abi.DBZ_INT8, "int16": abi.DBZ_INT16, "int32": abi.DBZ_INT32, "int64": abi.DBZ_INT64, "boolean": abi.DBZ_BOOLEAN, "string": abi.DBZ_STRING,
              "float": abi.DBZ_FLOAT64, "double": abi.DBZ_FLOAT64}
    if f.type in simple:
        return simple[f.type], 0
    if f.type == "struct":
        if f.name == "io.debezium.data.geometry.Point":
            return abi.DBZ_POINT, 0
        if f.name == "io.debezium.data.VariableScaleDecimal":
            return abi.DBZ_VSD, 0
    if f.type == "bytes":
        if f.name == "org.apache.kafka.connect.data.Decimal":
            scale = 0
            if f.has_parameters and f.scale != "":
                try:  # strconv.Atoi
                    if not f.scale.lstrip("+-").isdigit() or not f.scale.lstrip("+-").isascii():
                        raise ValueError
                    scale = int(f.scale)
                    if not INT32_MIN <= scale < (1 << 31):
                        raise ValueError
                except ValueError:
                    scale = INT32_MIN  # every non-nil value fails ("unable to parse scale")
            return abi.DBZ_DECIMAL, scale
        return abi.DBZ_BYTES, 0
    raise SchemaError("unable to find field receiver - even default, for kafka type: %s" % f.type)


def compile_schema(schema_bytes: bytes):
    """receiveSchema for one distinct schema: [(name, DBZ_*, optional, scale)] of the `after` struct.
    Raises SchemaError (→ TFGPU_ROW_DBZ_SCHEMA for every message of the schema) or HostOnly (→ TFGPU_ROW_HOST_FALLBACK)."""
    if not schema_bytes:
        raise SchemaError("unexpected end of JSON input")
    try:
        node = json.loads(schema_bytes.decode("utf-8", "replace"), object_pairs_hook=_Obj)
    except RecursionError:
        raise HostOnly("schema nested too deep")
    if node is not None and not _is_object(node):
        raise SchemaError("json: cannot unmarshal into Go value of type common.Schema")
    top = _Schema(node)
    before, after = top.find("before"), top.find("after")
    if before is None or after is None:
        raise HostOnly("receiveTableSchema(nil): the reference dereferences a nil schema")
    out = []
    for which in (before, after):
        fields = [(f.field,) + _receiver(f) + (f.optional,) for f in which.fields]
        out.append([(n, op, opt, scale) for (n, op, scale, opt) in fields])
    if out[0] != out[1]:
        raise HostOnly("before and after structs differ: Delete rows would have another TableSchema than the rest")
    if len({f[0] for f in out[1]}) != len(out[1]):
        raise HostOnly("a field name repeats")
    return out[1]


def table_schema(fields, ns: str, table: str) -> abi.Schema:
    """receiveTableSchema's result: PrimaryKey = !optional, TableSchema / TableName = source.schema / source.table."""
    return abi.Schema([abi.ColSchema(n, OP_DTYPE[op], not opt, "", "", False, ns, table) for (n, op, opt, _s) in fields])


class Parsed:
    """One table's rows of a message batch: the device batch (src_row = message index), per-row ID / LSN / CommitTime /
    names_form (abi.DBZ_ROW_DTYPE) and the TableSchema the items carry."""

    def __init__(self, batch, rows: np.ndarray, schema: abi.Schema):
        self.batch, self.rows, self.schema = batch, rows, schema


class Parser:
    """DebeziumImpl over message batches.  parse() returns ([Parsed per distinct schema, in order of first appearance],
    {message index: TFGPU_ROW_* code} for the messages that become `_unparsed` items or go to the stock code)."""

    def __init__(self, lib):
        self.lib = lib
        self.cache: Dict[Tuple[int, int], object] = {}
        self.known = None   # (prefix bytes, schema_off, schema_len, hash) of the last batch's opening message: the cache hit path

    def parse(self, data, msgs: Optional[abi.CMessages] = None, host_bytes: Optional[bytes] = None):
        lib = self.lib
        frames = lib.debezium_unpack(data, msgs, self.known)
        bad = np.nonzero(frames["code"])[0]
        errors = dict(zip(bad.tolist(), frames["code"][bad].tolist()))
        ok = np.nonzero(frames["code"] == 0)[0]
        groups: Dict[Tuple[int, int], int] = {}
        if len(ok):  # first message of every distinct schema
            hashes = frames["schema_hash"][ok]
            if (hashes[:, 0] == hashes[0, 0]).all() and (hashes[:, 1] == hashes[0, 1]).all():  # the usual topic: one schema
                groups[(int(hashes[0][0]), int(hashes[0][1]))] = int(ok[0])
            else:
                _u, first = np.unique(hashes, axis=0, return_index=True)
                for i in sorted(first):
                    groups[(int(hashes[i][0]), int(hashes[i][1]))] = int(ok[i])
        out = []
        for key, m in groups.items():
            if key not in self.cache:
                a, n = int(frames["schema_start"][m]), int(frames["schema_len"][m])
                raw = self._bytes(data, host_bytes, a, n)
                head = host_bytes if host_bytes is not None else (data if isinstance(data, (bytes, bytearray)) else None)
                if self.known is None and m == 0 and head is not None and int(frames["payload_len"][m]):
                    # the opening message's head up to its payload value, for the next batches (tfgpu_debezium_unpack_cached);
                    # usable when the payload is the message's last member (message 0 starts at offset 0)
                    ps = int(frames["payload_start"][m])
                    if a < ps:
                        self.known = (bytes(head[:ps]), a, n, key)
                try:
                    self.cache[key] = compile_schema(raw)
                except SchemaError:
                    self.cache[key] = abi.ROW_DBZ_SCHEMA
                except HostOnly:
                    self.cache[key] = abi.ROW_HOST_FALLBACK
            comp = self.cache[key]
            if isinstance(comp, int):  # receiveSchema's fate comes after the payload / op checks: the device still runs those
                _db, _rows, errs = lib.debezium_parse(key, [], data, frames, msgs, schema_code=comp)
                for mm, code in errs:
                    errors[mm] = code
                continue
            db, rows, errs = lib.debezium_parse(key, comp, data, frames, msgs)
            for mm, code in errs:
                errors[mm] = code
            if db.nrows:
                ns, table = db.table_id()
                out.append(Parsed(db, rows, table_schema(comp, ns, table)))
        return out, errors

    @staticmethod
    def _bytes(data, host_bytes, a, n) -> bytes:
        if host_bytes is not None:
            return host_bytes[a:a + n]
        if isinstance(data, (bytes, bytearray, memoryview)):
            return bytes(data[a:a + n])
        return data.download()[a:a + n]  # a DeviceBuffer without a host copy: rare (one schema per table per process)
