"""ctypes pass-through to the Debezium receiver's host half, which lives behind the C ABI (csrc/tf_dbzrecv.cpp: schema compilation,
the schema cache, DoBatch's loop) — what the reference does ONCE PER SCHEMA before any value is touched, and the loop that hands
each group of messages to the device.  A cgo shim binds the same entry points.

    Receiver.receiveSchema / receiveTableSchema    pkg/debezium/receiver.go:60-96, 45-59
    receiveFieldColSchema                          pkg/debezium/receiver_engine.go:108-146
    TypeToDefault, Point / VariableScaleDecimal /
    Decimal matchers                               pkg/debezium/common/field_receiver_default.go:14-31, 258-355
    Schema                                         pkg/debezium/common/debezium_schema.go:12-29, 84-101
    DebeziumImpl.DoBatch                           pkg/parsers/registry/debezium/engine/parser.go:120-130

Scope: NewDebeziumImpl(logger, nil, threads) — no original-type table; with a schema registry (events framed 0x00 | schema id |
payload, pkg/debezium/unpacker/schema_registry.go) through Parser.add_registry_schema / parse_registry.  The reference caches the
compiled schema by a hash of its bytes (receiver.go:61-66); so does `Parser`, keyed by the device's hash of the same bytes.
The per-message work is tfgpu_debezium_unpack / tfgpu_debezium_parse (csrc/tf_debezium.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi

INT32_MIN = -(1 << 31)
SR_FRAME_DTYPE = np.dtype([("msg", "<i8"), ("start", "<u8"), ("len", "<u4"), ("schema_id", "<u4"), ("code", "<i4"), ("index", "<i4")])   # tfgpu_sr_frame
OP_DTYPE = {abi.DBZ_BOOLEAN: "boolean", abi.DBZ_INT8: "int8", abi.DBZ_INT16: "int16", abi.DBZ_INT32: "int32", abi.DBZ_INT64: "int64", abi.DBZ_FLOAT64: "double",
            abi.DBZ_STRING: "utf8", abi.DBZ_BYTES: "string", abi.DBZ_DECIMAL: "utf8", abi.DBZ_POINT: "utf8", abi.DBZ_VSD: "double", abi.DBZ_HOST: "any"}


class SchemaError(ValueError):
    """receiveSchema fails: every message of the schema becomes an `_unparsed` item (TFGPU_ROW_DBZ_SCHEMA)."""


class HostOnly(ValueError):
    """The stock code must handle this schema (arrays, __dt_original_type_info, keys bound by case folding, a Go panic)."""


def compile_schema(schema_bytes: bytes):
    """receiveSchema for one distinct schema (tfgpu_debezium_compile_schema, csrc/tf_dbzrecv.cpp): [(name, DBZ_*, optional, scale)] of
    the `after` struct.  Raises SchemaError (→ TFGPU_ROW_DBZ_SCHEMA for every message of the schema) or HostOnly (→ TFGPU_ROW_HOST_FALLBACK)."""
    from . import lib
    L = lib.load()
    buf = np.frombuffer(schema_bytes, dtype=np.uint8) if len(schema_bytes) else np.zeros(1, np.uint8)
    h = C.c_void_p()
    L.tfgpu_debezium_compile_schema.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib._check(L.tfgpu_debezium_compile_schema(C.c_void_p(buf.ctypes.data), len(schema_bytes), C.byref(h)))
    try:
        code, why, fields = _schema_info(L, h)
    finally:
        L.tfgpu_dbz_schema_free.argtypes = [C.c_void_p]
        L.tfgpu_dbz_schema_free(h)
    if code == abi.ROW_DBZ_SCHEMA:
        raise SchemaError(why)
    if code == abi.ROW_HOST_FALLBACK:
        raise HostOnly(why)
    return fields


def compile_registry_schema(schema_text: bytes):
    """convertSchemaFormat + receiveSchema for a registry (ConfluentJSONSchema) text: like compile_schema."""
    from . import lib
    L = lib.load()
    h = C.c_void_p()
    L.tfgpu_debezium_compile_registry_schema.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib._check(L.tfgpu_debezium_compile_registry_schema(schema_text, len(schema_text), C.byref(h)))
    try:
        code, why, fields = _schema_info(L, h)
    finally:
        L.tfgpu_dbz_schema_free.argtypes = [C.c_void_p]
        L.tfgpu_dbz_schema_free(h)
    if code == abi.ROW_DBZ_SCHEMA:
        raise SchemaError(why)
    if code == abi.ROW_HOST_FALLBACK:
        raise HostOnly(why)
    return fields


def _fields_of(ptr, n):
    return [((ptr[i].name or b"").decode("utf-8", "replace"), int(ptr[i].op), bool(ptr[i].optional), int(ptr[i].scale)) for i in range(n)]


def _schema_info(L, h):
    code, n, why = C.c_int32(0), C.c_int32(0), C.c_char_p()
    fp = C.POINTER(abi.CDbzField)()
    L.tfgpu_dbz_schema_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(abi.CDbzField)), C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]
    L.tfgpu_dbz_schema_info(h, C.byref(code), C.byref(fp), C.byref(n), C.byref(why))
    return int(code.value), (why.value or b"").decode("utf-8", "replace"), _fields_of(fp, int(n.value))


def table_schema(fields, ns: str, table: str) -> abi.Schema:
    """receiveTableSchema's result: PrimaryKey = !optional, TableSchema / TableName = source.schema / source.table."""
    return abi.Schema([abi.ColSchema(n, OP_DTYPE[op], not opt, "", "", False, ns, table) for (n, op, opt, _s) in fields])


class Parsed:
    """One table's rows of a message batch: the device batch (src_row = message index), per-row ID / LSN / CommitTime /
    names_form (abi.DBZ_ROW_DTYPE) and the TableSchema the items carry."""

    def __init__(self, batch, rows: np.ndarray, schema: abi.Schema, meta=None):
        self.batch, self.rows, self.schema = batch, rows, schema
        self.meta = meta   # () -> (ids, lsns, commit_times, names_form) by message index (tfgpu_dbz_receive_group_meta); valid until the next parse()


class Parser:
    """DebeziumImpl over message batches: a pass-through to tfgpu_dbz_receiver (csrc/tf_dbzrecv.cpp), which keeps the schema cache and
    the opening message's head.  parse() returns ([Parsed per distinct schema, in order of first appearance],
    {message index: TFGPU_ROW_* code} for the messages that become `_unparsed` items or go to the stock code)."""

    def __init__(self, lib):
        self.lib = lib
        L = lib.load()
        lib.init()
        self._schemas: Dict[tuple, abi.Schema] = {}
        self.cache: Dict[tuple, abi.Schema] = {}  # the distinct (table, fields) seen so far — a view for diagnostics; the cache itself is the receiver's
        self._h = C.c_void_p()
        L.tfgpu_dbz_receiver_create.argtypes = [C.POINTER(C.c_void_p)]
        lib._check(L.tfgpu_dbz_receiver_create(C.byref(self._h)))
        L.tfgpu_dbz_receive.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
        L.tfgpu_dbz_receive_group.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.POINTER(abi.CDbzField)), C.POINTER(C.c_int32)]
        L.tfgpu_dbz_receiver_destroy.argtypes = [C.c_void_p]

    def __del__(self):
        try:
            if self._h:
                self.lib.load().tfgpu_dbz_receiver_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    def _group_schema(self, db, fp, nf):
        """The TableSchema of one group.  The receiver keeps a compiled schema for its whole life and hands out the same field array for it in
        every batch: the Python objects built from it (one per field) are kept by (that array's address, table) instead of rebuilt per batch."""
        ns, table = db.table_id()
        key = (C.addressof(fp.contents) if nf else 0, nf, ns, table)
        hit = self._schemas.get(key)
        if hit is None:
            fields = _fields_of(fp, nf)
            hit = table_schema(fields, ns, table)
            self._schemas[key] = hit
            self.cache.setdefault((ns, table, tuple(fields)), hit)
        return hit

    def _meta(self, g, nmsg):
        ids, lsns, cts, nf = np.empty(nmsg, np.uint32), np.empty(nmsg, np.uint64), np.empty(nmsg, np.uint64), np.empty(nmsg, np.uint8)
        L = self.lib.load()
        L.tfgpu_dbz_receive_group_meta.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib._check(L.tfgpu_dbz_receive_group_meta(self._h, g, nmsg, ids.ctypes.data, lsns.ctypes.data, cts.ctypes.data, nf.ctypes.data))
        return ids, lsns, cts, nf

    @property
    def known(self):
        """(prefix bytes, schema_off, schema_len, hash) of the head the receiver keeps for tfgpu_debezium_unpack_cached, or None"""
        k = abi.CDbzPrefix()
        L = self.lib.load()
        L.tfgpu_dbz_receiver_known.argtypes = [C.c_void_p, C.POINTER(abi.CDbzPrefix)]
        self.lib._check(L.tfgpu_dbz_receiver_known(self._h, C.byref(k)))
        if not k.len:
            return None
        return (C.string_at(k.bytes, k.len), int(k.schema_off), int(k.schema_len), (int(k.schema_hash[0]), int(k.schema_hash[1])))

    def parse(self, data, msgs: Optional[abi.CMessages] = None, host_bytes: Optional[bytes] = None):
        lib = self.lib
        L = lib.load()
        ptr_, n, mem, keep = lib._bytes_arg(data)
        nmsg = int(msgs.nmsg) if msgs is not None else 1
        if getattr(self, "_msg_codes", None) is None or len(self._msg_codes) < max(nmsg, 1):   # kept across batches; the call writes every entry
            self._msg_codes = np.zeros(max(nmsg, 1), np.int32)
        codes = self._msg_codes
        ng = C.c_int32(0)
        hb = np.frombuffer(host_bytes, dtype=np.uint8) if host_bytes else None
        lib._check(L.tfgpu_dbz_receive(self._h, ptr_, n, mem, C.c_void_p(hb.ctypes.data) if hb is not None else None, C.byref(msgs) if msgs is not None else None,
                                       C.byref(ng), C.c_void_p(codes.ctypes.data)))
        out = []
        for g in range(int(ng.value)):
            bh, rp, nr, fp, nf = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.POINTER(abi.CDbzField)(), C.c_int32(0)
            lib._check(L.tfgpu_dbz_receive_group(self._h, g, C.byref(bh), C.byref(rp), C.byref(nr), C.byref(fp), C.byref(nf)))
            db = lib.DeviceBatch(bh)
            rows = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint8)), shape=(int(nr.value) * abi.DBZ_ROW_DTYPE.itemsize,)).view(abi.DBZ_ROW_DTYPE) if nr.value else np.zeros(0, abi.DBZ_ROW_DTYPE)  # (a view of the receiver's buffer: valid until the next parse())
            sch = self._group_schema(db, fp, int(nf.value))
            out.append(Parsed(db, rows, sch, meta=(lambda g=g, nmsg=nmsg: self._meta(g, nmsg))))
        bad = np.nonzero(codes[:nmsg])[0]
        return out, dict(zip(bad.tolist(), codes[bad].tolist()))

    # ---- events framed by a schema registry (NewDebeziumImpl(logger, client, threads)) ----
    def add_registry_schema(self, schema_id: int, schema_text: bytes):
        """The registry's answer for one schema id (ConfluentJSONSchema text): converted and compiled once, kept by the receiver."""
        L = self.lib.load()
        L.tfgpu_dbz_receiver_add_registry_schema.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64]
        self.lib._check(L.tfgpu_dbz_receiver_add_registry_schema(self._h, schema_id, schema_text, len(schema_text)))

    def parse_registry(self, data, msgs: Optional[abi.CMessages] = None):
        """DoBatch with a registry (tfgpu_dbz_receive_registry).  Returns None, missing ids when some schema id was never registered
        (nothing parsed: fetch them, add_registry_schema, call again); otherwise ([Parsed per schema id; src_row = event ordinal],
        {event ordinal: TFGPU_ROW_* code}, events = tfgpu_sr_frames' list as a structured array: msg, start, len, schema_id, code, index)."""
        lib = self.lib
        L = lib.load()
        ptr_, n, mem, keep = lib._bytes_arg(data)
        L.tfgpu_dbz_receive_registry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(abi.CSrFrame), C.c_int64, C.POINTER(C.c_int64), C.c_void_p,
                                                 C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        cap = getattr(self, "_ev_cap", 1024)
        while True:
            if getattr(self, "_ev", None) is None or len(self._ev) < cap:   # kept across batches: 36 bytes per event are not allocated per call
                self._ev, self._codes, self._ev_cap = np.zeros(cap, SR_FRAME_DTYPE), np.zeros(cap, np.int32), cap
            ev, codes = self._ev, self._codes
            missing = np.zeros(64, np.uint32)
            nev, nmiss, ng = C.c_int64(0), C.c_int32(0), C.c_int32(0)
            rc = L.tfgpu_dbz_receive_registry(self._h, ptr_, n, mem, C.byref(msgs) if msgs is not None else None, C.cast(C.c_void_p(ev.ctypes.data), C.POINTER(abi.CSrFrame)), cap, C.byref(nev),
                                              C.c_void_p(codes.ctypes.data), C.c_void_p(missing.ctypes.data), 64, C.byref(nmiss), C.byref(ng))
            if rc and nev.value > cap:
                cap = int(nev.value) + int(nev.value) // 8
                continue
            lib._check(rc)
            break
        if nmiss.value:
            return None, [int(x) for x in missing[:min(int(nmiss.value), 64)]], None
        nevents = int(nev.value)
        out = []
        for g in range(int(ng.value)):
            bh, rp, nr, fp, nf = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.POINTER(abi.CDbzField)(), C.c_int32(0)
            lib._check(L.tfgpu_dbz_receive_group(self._h, g, C.byref(bh), C.byref(rp), C.byref(nr), C.byref(fp), C.byref(nf)))
            db = lib.DeviceBatch(bh)
            rows = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint8)), shape=(int(nr.value) * abi.DBZ_ROW_DTYPE.itemsize,)).view(abi.DBZ_ROW_DTYPE) if nr.value else np.zeros(0, abi.DBZ_ROW_DTYPE)  # (a view of the receiver's buffer: valid until the next call)
            out.append(Parsed(db, rows, self._group_schema(db, fp, int(nf.value)), meta=(lambda g=g, nevents=nevents: self._meta(g, nevents))))
        bad = np.nonzero(codes[:nevents])[0]
        return out, dict(zip(bad.tolist(), codes[bad].tolist())), ev[:nevents]   # events: a view (msg, start, len, schema_id, code, index), valid until the next call
