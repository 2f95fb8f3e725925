"""ctypes image of include/tfgpu.h plus numpy-side batch containers.

This module only describes memory layout and performs no computation; the
product binding (transferia_amd.lib) and the test harness both build their
batches through it.
"""
from __future__ import annotations

import ctypes as C
import datetime as _dt
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# ---- enums (keep in sync with include/tfgpu.h) ---------------------------
DTYPES = ["invalid", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float", "double",
          "boolean", "string", "utf8", "date", "datetime", "timestamp", "interval", "any"]
DTYPE_ID = {n: i for i, n in enumerate(DTYPES)}

R_INT8, R_INT16, R_INT32, R_INT64, R_UINT8, R_UINT16, R_UINT32, R_UINT64, R_FLOAT32, R_FLOAT64, R_BOOL, \
    R_STRING, R_BYTES, R_JSONNUM, R_JSON, R_TIME, R_DURATION = range(1, 18)
REPR_NAMES = ["invalid", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64",
              "bool", "string", "bytes", "jsonnum", "json", "time", "duration"]
REPR_NP = {R_INT8: np.int8, R_INT16: np.int16, R_INT32: np.int32, R_INT64: np.int64, R_UINT8: np.uint8,
           R_UINT16: np.uint16, R_UINT32: np.uint32, R_UINT64: np.uint64, R_FLOAT32: np.float32,
           R_FLOAT64: np.float64, R_BOOL: np.uint8, R_TIME: np.int64, R_DURATION: np.int64}
VAR_REPRS = (R_STRING, R_BYTES, R_JSONNUM, R_JSON)

K_INSERT, K_UPDATE, K_DELETE, K_OTHER, K_SYNCHRONIZE = 0, 1, 2, 3, 4
KIND_ID = {"insert": 0, "Insert": 0, "update": 1, "Update": 1, "delete": 2, "Delete": 2, "synchronize": 4}

MEM_HOST, MEM_DEVICE = 0, 1
COL_KEY, COL_REQUIRED, COL_FAKE_KEY = 1, 2, 4

ROWERR = {0: "OK", 1: "UNSUPPORTED_KIND", 2: "COLUMN_NOT_FOUND", 3: "INT_OVERFLOW", 4: "TYPE_PAIR", 5: "MISSING_CELL",
          6: "CAST", 7: "RANGE", 8: "QUOTE", 9: "DOUBLE_QUOTE", 10: "QUOTING_DISABLED", 11: "HOST_FALLBACK",
          12: "JSON_SYNTAX", 13: "PARSE_VAL", 14: "NIL_KEY", 15: "SR_SHORT", 16: "SR_MAGIC", 17: "SR_TYPE", 18: "SR_REQUIRED",
          19: "DBZ_UNPACK", 20: "DBZ_PAYLOAD", 21: "DBZ_OP", 22: "DBZ_SCHEMA", 23: "DBZ_FIELD", 24: "DROPPED", 25: "SR_PROTO"}
ROWERR_ID = {v: k for k, v in ROWERR.items()}
for _k, _v in ROWERR.items():
    globals()["ROW_" + _v] = _k

FMT_CH_JSON_EACH_ROW, FMT_JSON, FMT_CSV, FMT_RAW = 1, 2, 3, 4


# ---- C structs -----------------------------------------------------------
class CColSchema(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("flags", C.c_uint32), ("path", C.c_char_p),
                ("original_type", C.c_char_p), ("table_schema", C.c_char_p), ("table_name", C.c_char_p), ("expression", C.c_char_p),
                ("properties_json", C.c_char_p)]


class CSchema(C.Structure):
    _fields_ = [("ncols", C.c_int32), ("cols", C.POINTER(CColSchema))]


ABI_VERSION = 2  # include/tfgpu.h TFGPU_ABI_VERSION: the struct layouts below are this version's


class CColumn(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("repr", C.c_int32), ("values", C.c_void_p),
                ("offsets", C.c_void_p), ("data", C.c_void_p), ("data_len", C.c_uint64), ("nanos", C.c_void_p),
                ("validity", C.c_void_p), ("absent", C.c_void_p)]


class CBatch(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int32), ("cols", C.POINTER(CColumn)), ("table_ns", C.c_char_p),
                ("table_name", C.c_char_p), ("kind", C.c_void_p), ("src_row", C.c_void_p), ("part_id", C.c_void_p),
                ("mem", C.c_int32), ("n_old_keys", C.c_int32), ("old_keys", C.POINTER(CColumn)), ("old_keys_present", C.c_void_p),
                ("schema", C.POINTER(CSchema)), ("col_order", C.c_void_p)]


class CRowError(C.Structure):
    _fields_ = [("row", C.c_int64), ("code", C.c_int32), ("step", C.c_int32), ("column", C.c_int32)]


class CDbzFrame(C.Structure):
    _fields_ = [("schema_start", C.c_uint64), ("payload_start", C.c_uint64), ("schema_len", C.c_uint32), ("payload_len", C.c_uint32),
                ("schema_hash", C.c_uint64 * 2), ("code", C.c_int32), ("reserved", C.c_int32)]


class CDbzPrefix(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("len", C.c_uint32), ("schema_off", C.c_uint32), ("schema_len", C.c_uint32), ("reserved", C.c_uint32), ("schema_hash", C.c_uint64 * 2)]


class CDbzField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("op", C.c_int32), ("optional", C.c_int32), ("scale", C.c_int32), ("reserved", C.c_int32)]


class CDbzOptions(C.Structure):
    _fields_ = [("schema_hash", C.c_uint64 * 2), ("nfields", C.c_int32), ("fields", C.POINTER(CDbzField)), ("schema_code", C.c_int32), ("reserved", C.c_int32)]


class CDbzRow(C.Structure):
    _fields_ = [("msg", C.c_int64), ("lsn", C.c_uint64), ("commit_time", C.c_uint64), ("id", C.c_uint32), ("names_form", C.c_uint8), ("reserved", C.c_uint8 * 3)]


DBZ_BOOLEAN, DBZ_INT8, DBZ_INT16, DBZ_INT32, DBZ_INT64, DBZ_FLOAT64, DBZ_STRING, DBZ_BYTES, DBZ_DECIMAL, DBZ_POINT, DBZ_VSD, DBZ_HOST = range(1, 13)
DBZ_ROW_DTYPE = np.dtype([("msg", "<i8"), ("lsn", "<u8"), ("commit_time", "<u8"), ("id", "<u4"), ("names_form", "u1"), ("reserved", "u1", 3)])
DBZ_FRAME_DTYPE = np.dtype([("schema_start", "<u8"), ("payload_start", "<u8"), ("schema_len", "<u4"), ("payload_len", "<u4"), ("schema_hash", "<u8", 2),
                            ("code", "<i4"), ("reserved", "<i4")])


class CChNativeColumn(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ch_type", C.c_char_p)]


class CTransformationStats(C.Structure):
    _fields_ = [("pushes", C.c_int64), ("items_in", C.c_int64), ("items_out", C.c_int64), ("dropped", C.c_int64), ("errors", C.c_int64),
                ("elapsed_ns", C.c_int64), ("plans_built", C.c_int64)]


class CCsvOptions(C.Structure):
    _fields_ = [("delimiter", C.c_uint8), ("quote_char", C.c_uint8), ("escape_char", C.c_uint8),
                ("double_quote", C.c_uint8), ("newlines_in_value", C.c_uint8), ("include_missing_columns", C.c_uint8),
                ("strings_can_be_null", C.c_uint8), ("quoted_strings_can_be_null", C.c_uint8),
                ("n_null_values", C.c_int32), ("null_values", C.POINTER(C.c_char_p)),
                ("n_true_values", C.c_int32), ("true_values", C.POINTER(C.c_char_p)),
                ("n_false_values", C.c_int32), ("false_values", C.POINTER(C.c_char_p)),
                ("n_timestamp_parsers", C.c_int32), ("timestamp_parsers", C.POINTER(C.c_char_p)),
                ("decimal_point", C.c_char_p), ("skip_rows", C.c_int64),
                ("file_name", C.c_char_p), ("row_number_base", C.c_uint64), ("hide_system_cols", C.c_uint8),
                ("encoding_table", C.POINTER(C.c_uint32))]


class CJsonOptions(C.Structure):
    _fields_ = [("add_rest", C.c_uint8), ("add_dedupe_keys", C.c_uint8), ("null_keys_allowed", C.c_uint8),
                ("use_numbers_in_any", C.c_uint8), ("unescape_string_values", C.c_uint8), ("unpack_bytes_base64", C.c_uint8),
                ("ignore_column_paths", C.c_uint8), ("mark_dedupe_keys_as_system", C.c_uint8), ("topic", C.c_char_p), ("partition", C.c_char_p),
                ("format", C.c_uint8)]


class CMessages(C.Structure):
    _fields_ = [("nmsg", C.c_int64), ("start", C.c_void_p), ("offset", C.c_void_p), ("write_time_ns", C.c_void_p)]


def json_options(add_rest=False, add_dedupe_keys=False, null_keys_allowed=False, use_numbers_in_any=False,
                 unescape_string_values=False, unpack_bytes_base64=False, ignore_column_paths=False, mark_dedupe_keys_as_system=False, topic="",
                 partition='{"partition":0,"topic":""}', format="json") -> CJsonOptions:
    o = CJsonOptions()
    o.format = {"json": 0, "tskv": 1}[format]
    o.add_rest, o.add_dedupe_keys, o.null_keys_allowed = int(add_rest), int(add_dedupe_keys), int(null_keys_allowed)
    o.use_numbers_in_any, o.unescape_string_values = int(use_numbers_in_any), int(unescape_string_values)
    o.unpack_bytes_base64, o.ignore_column_paths = int(unpack_bytes_base64), int(ignore_column_paths)
    o.mark_dedupe_keys_as_system = int(mark_dedupe_keys_as_system)
    t, p = _b(topic), _b(partition)
    o.topic, o.partition = t, p
    o._keep = (t, p)
    return o


def messages(values: Sequence[bytes], offsets=None, write_times_ns=None):
    """parsers.MessageBatch → (concatenated bytes, CMessages).  Values are concatenated as they are."""
    n = len(values)
    start = np.zeros(n + 1, np.uint64)
    if n:
        start[1:] = np.cumsum([len(v) for v in values])
    off = np.asarray(offsets if offsets is not None else np.zeros(n), dtype=np.uint64)
    wt = np.asarray(write_times_ns if write_times_ns is not None else np.zeros(n), dtype=np.int64)
    m = CMessages()
    m.nmsg, m.start, m.offset, m.write_time_ns = n, start.ctypes.data, off.ctypes.data, wt.ctypes.data
    m._keep = (start, off, wt)
    return b"".join(values), m


class CSerializeOptions(C.Structure):
    _fields_ = [("add_closing_newline", C.c_int32), ("any_as_string", C.c_int32), ("ncols", C.c_int32),
                ("ch_flags", C.POINTER(C.c_uint32)), ("ch_precision", C.POINTER(C.c_uint8))]


class CBatchSerializerConfig(C.Structure):
    _fields_ = [("concurrency", C.c_int32), ("threshold", C.c_int32), ("disable_concurrency", C.c_int32), ("gomaxprocs", C.c_int32)]


CH_STRING, CH_DATE, CH_DATETIME64, CH_DECIMAL, CH_ARRAY = 1, 2, 4, 8, 16


def serialize_options(add_closing_newline=False, any_as_string=False, ch_types=None) -> CSerializeOptions:
    """ch_types: per batch column (flags, precision) of the ClickHouse target column, or None to
    derive them from the DataType."""
    o = CSerializeOptions()
    o.add_closing_newline, o.any_as_string = int(bool(add_closing_newline)), int(bool(any_as_string))
    if ch_types:
        n = len(ch_types)
        fl = (C.c_uint32 * n)(*[int(t[0]) for t in ch_types])
        pr = (C.c_uint8 * n)(*[int(t[1]) for t in ch_types])
        o.ncols, o.ch_flags, o.ch_precision = n, fl, pr
        o._keep = (fl, pr)
    return o


SRT_BOOLEAN, SRT_INTEGER, SRT_NUMBER, SRT_STRING, SRT_ANY = 1, 2, 3, 4, 5


class CSrFrame(C.Structure):
    _fields_ = [("msg", C.c_int64), ("start", C.c_uint64), ("len", C.c_uint32), ("schema_id", C.c_uint32), ("code", C.c_int32), ("index", C.c_int32)]


class CSrProperty(C.Structure):
    _fields_ = [("name", C.c_char_p), ("json_type", C.c_int32), ("required", C.c_int32)]


class CSrJsonOptions(C.Structure):
    _fields_ = [("schema_id", C.c_uint32), ("nprops", C.c_int32), ("props", C.POINTER(CSrProperty)), ("table_ns", C.c_char_p), ("table_name", C.c_char_p),
                ("is_generate_updates", C.c_int32), ("report_frame_errors", C.c_int32)]


def sr_json_options(schema_id, props, table_ns="", table_name="", is_generate_updates=False, report_frame_errors=True) -> CSrJsonOptions:
    """props: [(name, SRT_*, required)] sorted by name (util.MapKeysInOrder of the schema's properties)."""
    o = CSrJsonOptions()
    arr = (CSrProperty * max(len(props), 1))()
    keep = [arr]
    for i, (name, jt, req) in enumerate(props):
        nb = _b(name)
        keep.append(nb)
        arr[i].name, arr[i].json_type, arr[i].required = nb, int(jt), int(bool(req))
    ns, tn = _b(table_ns), _b(table_name)
    keep += [ns, tn]
    o.schema_id, o.nprops, o.props, o.table_ns, o.table_name = int(schema_id), len(props), arr, ns, tn
    o.is_generate_updates, o.report_frame_errors = int(bool(is_generate_updates)), int(bool(report_frame_errors))
    o._keep = keep
    return o


QFMT_NATIVE, QFMT_JSON = 1, 2


class CRowMeta(C.Structure):
    _fields_ = [("id", C.c_void_p), ("lsn", C.c_void_p), ("commit_time", C.c_void_p), ("counter", C.c_void_p),
                ("tx_id_offsets", C.c_void_p), ("tx_id_data", C.c_void_p), ("query_offsets", C.c_void_p), ("query_data", C.c_void_p),
                ("names_form", C.c_void_p), ("n", C.c_int64), ("mem", C.c_int32)]


class CQueueOptions(C.Structure):
    _fields_ = [("format", C.c_int32), ("batching_enabled", C.c_int32), ("max_change_items", C.c_int32), ("max_message_size", C.c_int64),
                ("table_schema_json", C.c_char_p), ("table_schema", C.POINTER(CSchema)), ("omit_table_schema", C.c_int32),
                ("old_key_types", C.POINTER(C.c_char_p)), ("ngroups", C.c_int32), ("group_rows", C.c_void_p), ("group_part_ids", C.POINTER(C.c_char_p))]


class CDbzEmitOptions(C.Structure):
    _fields_ = [("param_keys", C.POINTER(C.c_char_p)), ("param_values", C.POINTER(C.c_char_p)), ("nparams", C.c_int32), ("version", C.c_char_p),
                ("drop_keys", C.c_int32), ("snapshot", C.c_int32), ("table_schema", C.POINTER(CSchema))]


def dbz_emit_options(params: dict, table_schema: "Schema", version=None, drop_keys=False, snapshot=False) -> CDbzEmitOptions:
    """NewDebeziumSerializer's arguments (pkg/serializer/queue/debezium_serializer.go:117-127): the format settings, dropKeys, isSnapshot."""
    o = CDbzEmitOptions()
    items = list((params or {}).items())
    ks = (C.c_char_p * max(len(items), 1))(*[_b(k) for k, _ in items])
    vs = (C.c_char_p * max(len(items), 1))(*[_b(v) for _, v in items])
    cs = table_schema.to_c()
    o.param_keys, o.param_values, o.nparams = ks, vs, len(items)
    o.version = _b(version) if version is not None else None
    o.drop_keys, o.snapshot = int(bool(drop_keys)), int(bool(snapshot))
    o.table_schema = C.pointer(cs)
    o._keep = [ks, vs, cs]
    return o


def _var(strings, n):
    parts = [_b(x) for x in strings]
    off = np.zeros(n + 1, np.uint32)
    if n:
        off[1:] = np.cumsum([len(x) for x in parts])
    dat = np.frombuffer(b"".join(parts) or b"\0", np.uint8).copy()
    return off, dat


def row_meta(n, ids=None, lsns=None, commit_times=None, counters=None, tx_ids=None, queries=None, names_form=None) -> CRowMeta:
    """The ChangeItem fields that are not columns, one entry per pipeline INPUT row (HOST memory)."""
    m = CRowMeta()
    keep = []

    def arr(v, dt):
        if v is None:
            return None
        a = np.ascontiguousarray(v, dtype=dt)
        assert len(a) == n
        keep.append(a)
        return a.ctypes.data if n else None
    m.id, m.lsn, m.commit_time, m.counter = arr(ids, np.uint32), arr(lsns, np.uint64), arr(commit_times, np.uint64), arr(counters, np.int64)
    m.names_form = arr(names_form, np.uint8)
    if tx_ids is not None:
        off, dat = _var(tx_ids, n)
        keep += [off, dat]
        m.tx_id_offsets, m.tx_id_data = off.ctypes.data, dat.ctypes.data
    if queries is not None:
        off, dat = _var(queries, n)
        keep += [off, dat]
        m.query_offsets, m.query_data = off.ctypes.data, dat.ctypes.data
    m.n, m.mem = n, MEM_HOST
    m._keep = keep
    return m


def queue_options(fmt, enabled=False, max_change_items=0, max_message_size=0, table_schema_json=None, table_schema: "Schema" = None,
                  omit_table_schema=False, old_key_types=None, group_rows=None, group_part_ids=None) -> CQueueOptions:
    """model.Batching + what ChangeItem.MarshalJSON needs beyond the batch (queue serializers)."""
    o = CQueueOptions()
    o.format, o.batching_enabled, o.max_change_items, o.max_message_size = int(fmt), int(bool(enabled)), int(max_change_items), int(max_message_size)
    keep = []
    if table_schema_json is not None:
        t = _b(table_schema_json)
        keep.append(t)
        o.table_schema_json = t
    if table_schema is not None:
        cs = table_schema.to_c()
        keep.append(cs)
        o.table_schema = C.pointer(cs)
    o.omit_table_schema = int(bool(omit_table_schema))
    if old_key_types is not None:
        a = (C.c_char_p * max(len(old_key_types), 1))(*[_b(x) for x in old_key_types])
        keep.append(a)
        o.old_key_types = a
    if group_rows is not None:
        g = np.ascontiguousarray(group_rows, dtype=np.int64)
        keep.append(g)
        o.ngroups, o.group_rows = len(g), g.ctypes.data
    if group_part_ids is not None:
        a = (C.c_char_p * max(len(group_part_ids), 1))(*[_b(x) for x in group_part_ids])
        keep.append(a)
        o.group_part_ids = a
    o._keep = keep
    return o


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else str(s).encode("utf-8")


# csv.Reader.Encoding names (x/text charmap.All .String(), lower-cased with '-' for ' ' by prepareEncodingName,
# pkg/csv/reader.go:22-37) -> the Python codec holding the same single-byte table.
_CHARMAPS = {"ibm-code-page-037": "cp037", "ibm-code-page-437": "cp437", "ibm-code-page-850": "cp850", "ibm-code-page-852": "cp852",
             "ibm-code-page-855": "cp855", "ibm-code-page-858": "cp858", "ibm-code-page-860": "cp860", "ibm-code-page-862": "cp862",
             "ibm-code-page-863": "cp863", "ibm-code-page-865": "cp865", "ibm-code-page-866": "cp866", "ibm-code-page-1047": "cp1047",
             "ibm-code-page-1140": "cp1140", "koi8-r": "koi8_r", "koi8-u": "koi8_u", "macintosh": "mac_roman",
             "macintosh-cyrillic": "mac_cyrillic", "windows-874": "cp874",
             **{"iso-8859-%d" % i: "iso8859_%d" % i for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16)},
             **{"windows-%d" % i: "cp%d" % i for i in range(1250, 1259)}}


def charmap_table(encoding: str):
    """The 256 code points tfgpu_csv_options.encoding_table carries.  The Go shim reads them off the reference's own
    decoder (decoderFactoryByEncoding(name).NewDecoder(), pkg/csv/reader.go:27-37); this host mirror takes them from
    Python's codec of the same code page.  A byte the code page leaves undefined decodes to U+FFFD as x/text does, except
    in the windows-125x pages, whose WHATWG tables (x/text's source) pass such bytes through as C1 controls."""
    name = encoding.lower().replace(" ", "-")
    if name not in _CHARMAPS:
        raise ValueError("csv: unknown Encoding %r (not in x/text charmap.All)" % encoding)
    codec = _CHARMAPS[name]
    import codecs
    try:
        codecs.lookup(codec)
    except LookupError:
        raise ValueError("csv: this Python has no table for Encoding %r (%s); pass the 256 code points yourself" % (encoding, codec))
    out = []
    for b in range(256):
        try:
            out.append(ord(bytes([b]).decode(codec)))
        except UnicodeDecodeError:
            out.append(b if codec.startswith("cp125") else 0xFFFD)
    return out


# ---- python-side containers ------------------------------------------------
@dataclass
class ColSchema:
    name: str
    dtype: str
    key: bool = False
    path: str = ""
    original_type: str = ""
    required: bool = False
    table_schema: str = ""     # ColSchema.TableSchema / TableName / Expression / FakeKey / Properties: read by the wire form only
    table_name: str = ""
    expression: str = ""
    fake_key: bool = False
    properties_json: str = ""


@dataclass
class Schema:
    cols: List[ColSchema]

    @staticmethod
    def of(spec: Sequence) -> "Schema":
        out = []
        for i, c in enumerate(spec):
            if isinstance(c, ColSchema):
                out.append(c)
            else:
                c = list(c)
                out.append(ColSchema(c[0], c[1], bool(c[2]) if len(c) > 2 else False, c[3] if len(c) > 3 else "",
                                     c[4] if len(c) > 4 else "", bool(c[5]) if len(c) > 5 else False))
        return Schema(out)

    def dtype_of(self, name: str) -> str:
        for c in self.cols:
            if c.name == name:
                return c.dtype
        return "invalid"

    def to_c(self):
        # memoised on the content: building 105 tfgpu_colschema structs through ctypes is ~0.5 ms of Python, paid by every call
        # that takes a schema; a Go caller converts its TableSchema once
        fp = tuple((c.name, c.dtype, c.key, c.path, c.original_type, c.required, c.table_schema, c.table_name, c.expression, c.fake_key, c.properties_json) for c in self.cols)
        hit = self.__dict__.get("_c_memo")
        if hit is not None and hit[0] == fp:
            return hit[1]
        s = self._to_c()
        self.__dict__["_c_memo"] = (fp, s)
        return s

    def _to_c(self):
        arr = (CColSchema * max(len(self.cols), 1))()
        keep = []
        for i, c in enumerate(self.cols):
            n, p, o = _b(c.name), _b(c.path), _b(c.original_type)
            ts, tn, ex = _b(c.table_schema), _b(c.table_name), _b(c.expression)
            pj = _b(c.properties_json) if c.properties_json else None
            keep += [n, p, o, ts, tn, ex, pj]
            arr[i].name, arr[i].dtype, arr[i].path, arr[i].original_type = n, DTYPE_ID[c.dtype], p, o
            arr[i].table_schema, arr[i].table_name, arr[i].expression, arr[i].properties_json = ts, tn, ex, pj
            arr[i].flags = (COL_KEY if c.key else 0) | (COL_REQUIRED if c.required else 0) | (COL_FAKE_KEY if c.fake_key else 0)
        s = CSchema(len(self.cols), arr)
        s._keep = (arr, keep)
        return s

    @staticmethod
    def from_c(cs: CSchema) -> "Schema":
        out = []
        for i in range(cs.ncols):
            c = cs.cols[i]
            out.append(ColSchema(c.name.decode("utf-8"), DTYPES[c.dtype], bool(c.flags & COL_KEY),
                                 (c.path or b"").decode(), (c.original_type or b"").decode(), bool(c.flags & COL_REQUIRED),
                                 (c.table_schema or b"").decode(), (c.table_name or b"").decode(), (c.expression or b"").decode(),
                                 bool(c.flags & COL_FAKE_KEY), (c.properties_json or b"").decode()))
        return Schema(out)

    def triples(self):
        return [[c.name, c.dtype, c.key] for c in self.cols]


@dataclass
class Column:
    name: str
    dtype: str
    repr: int
    values: Optional[np.ndarray] = None   # fixed width
    offsets: Optional[np.ndarray] = None  # uint32[n+1]
    data: Optional[np.ndarray] = None     # uint8
    nanos: Optional[np.ndarray] = None    # int32
    validity: Optional[np.ndarray] = None  # bool[n] (python side), bitmap on the C side
    absent: Optional[np.ndarray] = None    # bool[n]: rows whose ColumnNames leave this column out (tfgpu_column.absent); None = none

    def nrows(self) -> int:
        return int(len(self.offsets) - 1) if self.repr in VAR_REPRS else int(len(self.values))

    def is_valid(self, i: int) -> bool:
        return True if self.validity is None else bool(self.validity[i])

    def get_bytes(self, i: int) -> bytes:
        return bytes(self.data[int(self.offsets[i]):int(self.offsets[i + 1])])

    def pyvalue(self, i: int):
        """Go-typed python value [gotype, v] for row i (test convenience)."""
        if not self.is_valid(i):
            return ["nil", None]
        r = self.repr
        if r in VAR_REPRS:
            b = self.get_bytes(i)
            return [{R_STRING: "string", R_BYTES: "bytes", R_JSONNUM: "jsonnum", R_JSON: "json"}[r], b]
        if r == R_TIME:
            return ["time", (int(self.values[i]), int(self.nanos[i]) if self.nanos is not None else 0)]
        if r == R_DURATION:
            return ["duration", int(self.values[i])]
        if r == R_BOOL:
            return ["bool", bool(self.values[i])]
        if r in (R_FLOAT32, R_FLOAT64):
            return [REPR_NAMES[r], float(self.values[i])]
        return [REPR_NAMES[r], int(self.values[i])]


def pack_validity(v: Optional[np.ndarray], n: int) -> Optional[np.ndarray]:
    if v is None:
        return None
    return np.packbits(np.asarray(v, dtype=bool), bitorder="little")


def unpack_validity(bits: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(bits, bitorder="little")[:n].astype(bool)


@dataclass
class Batch:
    cols: List[Column]
    nrows: int
    table_ns: str = ""
    table_name: str = ""
    kind: Optional[np.ndarray] = None     # uint8
    src_row: Optional[np.ndarray] = None  # int32
    part_id: Optional[np.ndarray] = None  # uint32

    def col(self, name: str) -> Column:
        for c in self.cols:
            if c.name == name:
                return c
        raise KeyError(name)

    def to_c(self) -> CBatch:
        old = list(getattr(self, "old_keys", None) or [])
        arr = (CColumn * max(len(self.cols) + len(old), 1))()
        keep = []
        for i, c in enumerate(list(self.cols) + old):
            nm = _b(c.name)
            keep.append(nm)
            arr[i].name, arr[i].dtype, arr[i].repr = nm, DTYPE_ID[c.dtype], c.repr
            if c.repr in VAR_REPRS:
                off = np.ascontiguousarray(c.offsets, dtype=np.uint32)
                dat = np.ascontiguousarray(c.data if c.data is not None and len(c.data) else np.zeros(1, np.uint8), dtype=np.uint8)
                keep += [off, dat]
                arr[i].offsets, arr[i].data, arr[i].data_len = off.ctypes.data, dat.ctypes.data, int(off[-1]) if len(off) else 0
            else:
                vals = np.ascontiguousarray(c.values, dtype=REPR_NP[c.repr])
                if len(vals) == 0:
                    vals = np.zeros(1, REPR_NP[c.repr])
                keep.append(vals)
                arr[i].values = vals.ctypes.data
                if c.nanos is not None:
                    nn = np.ascontiguousarray(c.nanos, dtype=np.int32)
                    keep.append(nn)
                    arr[i].nanos = nn.ctypes.data
            if c.validity is not None:
                bm = pack_validity(c.validity, self.nrows)
                if len(bm) == 0:
                    bm = np.zeros(1, np.uint8)
                keep.append(bm)
                arr[i].validity = bm.ctypes.data
            if getattr(c, "absent", None) is not None:  # rows whose ColumnNames leave this column out (tfgpu_column.absent)
                ab = pack_validity(c.absent, self.nrows)
                if len(ab) == 0:
                    ab = np.zeros(1, np.uint8)
                keep.append(ab)
                arr[i].absent = ab.ctypes.data
        cb = CBatch()
        cb.nrows, cb.ncols, cb.cols = self.nrows, len(self.cols), arr
        ns, tn = _b(self.table_ns), _b(self.table_name)
        keep += [ns, tn]
        cb.table_ns, cb.table_name, cb.mem = ns, tn, MEM_HOST
        if old:  # ChangeItem.OldKeys: columns by KeyNames + which rows carry them
            cb.n_old_keys = len(old)
            cb.old_keys = C.cast(C.byref(arr, C.sizeof(CColumn) * len(self.cols)), C.POINTER(CColumn))
            pres = getattr(self, "old_present", None)
            if pres is not None:
                pb = pack_validity(np.asarray(pres, dtype=bool), self.nrows)
                if len(pb) == 0:
                    pb = np.zeros(1, np.uint8)
                keep.append(pb)
                cb.old_keys_present = pb.ctypes.data
        if self.part_id is not None:
            pid = np.ascontiguousarray(self.part_id, dtype=np.uint32)
            keep.append(pid)
            cb.part_id = pid.ctypes.data
        if getattr(self, "schema", None) is not None:  # ChangeItem.TableSchema when it differs from ColumnNames
            cs = self.schema.to_c()
            keep.append(cs)
            cb.schema = C.pointer(cs)
        if getattr(self, "col_order", None) is not None:  # every row's own ColumnNames order (an output of tfgpu_collapse; the oracle reads it, upload refuses it)
            co = np.ascontiguousarray(self.col_order, dtype=np.uint16)
            keep.append(co)
            cb.col_order = co.ctypes.data if co.size else None
        if self.kind is not None:
            k = np.ascontiguousarray(self.kind, dtype=np.uint8)
            keep.append(k)
            cb.kind = k.ctypes.data if len(k) else None
        if self.src_row is not None:
            s = np.ascontiguousarray(self.src_row, dtype=np.int32)
            keep.append(s)
            cb.src_row = s.ctypes.data if len(s) else None
        cb._keep = (arr, keep)
        return cb


def _np_from_ptr(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def batch_from_c(cb: CBatch) -> Batch:
    """Copy a HOST-memory tfgpu_batch into numpy-owned storage."""
    n = int(cb.nrows)
    cols = [_column_from_c(cb.cols[i], n) for i in range(cb.ncols)]
    return _batch_from_c_tail(cb, cols, n)


def _column_from_c(c, n):
    if True:
        col = Column((c.name or b"").decode("utf-8"), DTYPES[c.dtype], int(c.repr))
        if c.repr in VAR_REPRS:
            col.offsets = _np_from_ptr(c.offsets, n + 1, np.uint32) if n or c.offsets else np.zeros(1, np.uint32)
            if len(col.offsets) == 0:
                col.offsets = np.zeros(1, np.uint32)
            col.data = _np_from_ptr(c.data, int(col.offsets[-1]), np.uint8)
        else:
            col.values = _np_from_ptr(c.values, n, REPR_NP[c.repr])
            if c.repr == R_TIME and c.nanos:
                col.nanos = _np_from_ptr(c.nanos, n, np.int32)
        if c.validity:
            col.validity = unpack_validity(_np_from_ptr(c.validity, (n + 7) // 8, np.uint8), n)
        if c.absent:
            col.absent = unpack_validity(_np_from_ptr(c.absent, (n + 7) // 8, np.uint8), n)
        return col


def _batch_from_c_tail(cb, cols, n):
    b = Batch(cols, n, (cb.table_ns or b"").decode(), (cb.table_name or b"").decode())
    if cb.n_old_keys:
        b.old_keys = [_column_from_c(cb.old_keys[i], n) for i in range(cb.n_old_keys)]
        b.old_present = unpack_validity(_np_from_ptr(cb.old_keys_present, (n + 7) // 8, np.uint8), n) if cb.old_keys_present else np.ones(n, bool)
    if cb.kind:
        b.kind = _np_from_ptr(cb.kind, n, np.uint8)
    if cb.src_row:
        b.src_row = _np_from_ptr(cb.src_row, n, np.int32)
    if cb.part_id:
        b.part_id = _np_from_ptr(cb.part_id, n, np.uint32)
    return b


# ---- building batches from Go-typed python values (tests, host mirror) ----
_GOTYPE_REPR = {"int8": R_INT8, "int16": R_INT16, "int32": R_INT32, "int64": R_INT64, "int": R_INT64,
                "uint8": R_UINT8, "uint16": R_UINT16, "uint32": R_UINT32, "uint64": R_UINT64, "uint": R_UINT64,
                "float32": R_FLOAT32, "float64": R_FLOAT64, "bool": R_BOOL, "string": R_STRING, "bytes": R_BYTES,
                "bytes-utf8": R_BYTES, "jsonnum": R_JSONNUM, "json": R_JSON, "time": R_TIME, "duration": R_DURATION}

_RFC = re.compile(r"^(-?\d{4,})-(\d\d)-(\d\d)T(\d\d):(\d\d):(\d\d)(?:\.(\d+))?(Z|[+-]\d\d:\d\d)$")


def days_from_civil(y: int, m: int, d: int) -> int:
    y -= m <= 2
    era = y // 400  # python floor division
    yoe = y - era * 400
    doy = (153 * (m - 3 if m > 2 else m + 9) + 2) // 5 + d - 1
    doe = yoe * 365 + yoe // 4 - yoe // 100 + doy
    return era * 146097 + doe - 719468


def parse_rfc3339(s: str):
    """RFC3339Nano text (any year, any numeric zone) → (unix seconds, nanoseconds)."""
    m = _RFC.match(s)
    if not m:
        raise ValueError("bad time " + s)
    y, mo, d, h, mi, se = (int(m.group(i)) for i in range(1, 7))
    frac = (m.group(7) or "")[:9].ljust(9, "0")
    z = m.group(8)
    off = 0 if z == "Z" else (1 if z[0] == "+" else -1) * (int(z[1:3]) * 3600 + int(z[4:6]) * 60)
    return days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + se - off, int(frac)


def _to_bytes(gotype: str, v) -> bytes:
    if isinstance(v, (bytes, bytearray)):
        return bytes(v)
    if isinstance(v, list):
        return bytes(v)
    if gotype == "bytes":
        return v.encode("latin-1")
    return v.encode("utf-8")


def column_from_values(name: str, dtype: str, vals: Sequence) -> Column:
    """vals: list of [gotype, value]; all non-nil entries must share a gotype."""
    n = len(vals)
    gts = {g for g, _ in vals if g != "nil"}
    if len(gts) > 1:
        raise ValueError(f"column {name}: mixed Go types {gts}")
    gt = next(iter(gts)) if gts else "string"
    r = _GOTYPE_REPR[gt]
    valid = np.array([g != "nil" for g, _ in vals], dtype=bool)
    col = Column(name, dtype, r, validity=None if valid.all() else valid)
    if r in VAR_REPRS:
        parts = [(_to_bytes(g, v) if g != "nil" else b"") for g, v in vals]
        off = np.zeros(n + 1, np.uint32)
        if n:
            off[1:] = np.cumsum([len(p) for p in parts])
        col.offsets = off
        col.data = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    elif r == R_TIME:
        secs, nanos = np.zeros(n, np.int64), np.zeros(n, np.int32)
        for i, (g, v) in enumerate(vals):
            if g == "nil":
                continue
            s, ns = parse_rfc3339(v) if isinstance(v, str) else v
            secs[i], nanos[i] = s, ns
        col.values, col.nanos = secs, nanos
    else:
        arr = np.zeros(n, REPR_NP[r])
        for i, (g, v) in enumerate(vals):
            if g != "nil":
                arr[i] = v
        col.values = arr
    return col


def batch_from_rows(schema: Schema, names: Sequence[str], rows: Sequence[Sequence], ns="", table="", kinds=None) -> Batch:
    cols = [column_from_values(nm, schema.dtype_of(nm), [r[i] for r in rows]) for i, nm in enumerate(names)]
    b = Batch(cols, len(rows), ns, table)
    if kinds is not None:
        b.kind = np.array([KIND_ID.get(k, K_OTHER) for k in kinds], dtype=np.uint8)
    return b


def norm_value(v):
    """Canonical comparable form of a [gotype, value] pair."""
    g, x = v
    if g == "nil":
        return ("nil", None)
    if g in ("int", "int64"):
        return ("int64", int(x))
    if g in ("uint", "uint64"):
        return ("uint64", int(x))
    if g in ("string", "bytes", "bytes-utf8", "jsonnum", "json"):
        return ({"bytes-utf8": "bytes"}.get(g, g), _to_bytes(g, x))
    if g == "time":
        return ("time", tuple(parse_rfc3339(x)) if isinstance(x, str) else tuple(x))
    if g == "float32":
        return ("float32", float(np.float32(x)))
    if g == "float64":
        return ("float64", float(x))
    if g == "bool":
        return ("bool", bool(x))
    return (g, int(x))


def batch_rows(b: Batch):
    return [[norm_value(c.pyvalue(i)) for c in b.cols] for i in range(b.nrows)]


def csv_options(delimiter=",", quote_char='"', escape_char="\\", double_quote=1, newlines_in_value=0,
                include_missing_columns=0, strings_can_be_null=0, quoted_strings_can_be_null=0, null_values=(),
                true_values=(), false_values=(), timestamp_parsers=(), decimal_point="", skip_rows=0, file_name="",
                row_number_base=1, hide_system_cols=0, encoding="") -> CCsvOptions:
    o = CCsvOptions()

    def ch(x):
        return 0 if x in (0, None, "") else (x if isinstance(x, int) else ord(x))
    o.delimiter, o.quote_char, o.escape_char = ch(delimiter), ch(quote_char), ch(escape_char)
    o.double_quote, o.newlines_in_value = int(double_quote), int(newlines_in_value)
    o.include_missing_columns, o.strings_can_be_null = int(include_missing_columns), int(strings_can_be_null)
    o.quoted_strings_can_be_null = int(quoted_strings_can_be_null)
    keep = []

    def lst(vals):
        arr = (C.c_char_p * max(len(vals), 1))(*[_b(v) for v in vals])
        keep.append(arr)
        return len(vals), arr
    o.n_null_values, o.null_values = lst(list(null_values))
    o.n_true_values, o.true_values = lst(list(true_values))
    o.n_false_values, o.false_values = lst(list(false_values))
    o.n_timestamp_parsers, o.timestamp_parsers = lst(list(timestamp_parsers))
    dp = _b(decimal_point)
    keep.append(dp)
    o.decimal_point, o.skip_rows = dp, int(skip_rows)
    fn = _b(file_name)
    keep.append(fn)
    o.file_name, o.row_number_base, o.hide_system_cols = fn, int(row_number_base), int(hide_system_cols)
    if encoding:
        tab = (C.c_uint32 * 256)(*charmap_table(encoding))
        keep.append(tab)
        o.encoding_table = tab
    o._keep = keep
    return o
