"""ctypes pass-through to the Confluent Schema Registry parser's per-schema set-up, which lives behind the C ABI
(tfgpu_sr_compile_schema, csrc/tf_dbzrecv.cpp): unmarshal the JSON schema, resolve every property to a column (type through
`oneOf`, required set), derive the table id from the title.  It runs once per schema id (the shim caches it next to the
registry client); the per-message work is tfgpu_sr_frames / tfgpu_sr_json_parse.  What stays here is the test harness's
`ConfluentSrParser`, which turns device batches back into reference-ordered items for comparison with the oracle.

    JSONProperties, jsonPropertyToJSONSchemaRow   engine/utils_json.go:15-21, 71-95
    jsonSchemaTypes                               engine/types_json.go:25-32
    BuildJSONTableID                              table_name_policy/table_name_policy.go:73-92
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

from . import abi

# types_json.go:25-32 — JSON-schema `type` → the column's ytschema type
_JSON_TYPES = {"array": abi.SRT_ANY, "boolean": abi.SRT_BOOLEAN, "integer": abi.SRT_INTEGER, "number": abi.SRT_NUMBER,
               "object": abi.SRT_ANY, "string": abi.SRT_STRING}
SRT_DTYPE = {abi.SRT_BOOLEAN: "boolean", abi.SRT_INTEGER: "int64", abi.SRT_NUMBER: "double", abi.SRT_STRING: "utf8", abi.SRT_ANY: "any"}

POLICY_DEBEZIUM_STYLE, POLICY_TITLE = "debezium_style", "title"


def _compile(schema_text: str, policy: str, manual_table_name: str):
    """tfgpu_sr_compile_schema (csrc/tf_dbzrecv.cpp): (title, [(name, SRT_*, required)], ns, table).  The reference's errors are ValueError."""
    from . import lib
    L = lib.load()
    raw = schema_text.encode("utf-8") if isinstance(schema_text, str) else bytes(schema_text)
    h = C.c_void_p()
    L.tfgpu_sr_compile_schema.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    rc = L.tfgpu_sr_compile_schema(raw, len(raw), policy.encode(), (manual_table_name or "").encode(), C.byref(h))
    if rc:
        raise ValueError((L.tfgpu_last_error() or b"").decode("utf-8", "replace"))
    try:
        pp, n, ns, tn, title = C.POINTER(abi.CSrProperty)(), C.c_int32(0), C.c_char_p(), C.c_char_p(), C.c_char_p()
        L.tfgpu_sr_schema_info.argtypes = [C.c_void_p, C.POINTER(C.POINTER(abi.CSrProperty)), C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.tfgpu_sr_schema_info(h, C.byref(pp), C.byref(n), C.byref(ns), C.byref(tn), C.byref(title))
        rows = [((pp[i].name or b"").decode("utf-8", "replace"), int(pp[i].json_type), bool(pp[i].required)) for i in range(int(n.value))]
        return (title.value or b"").decode("utf-8", "replace"), rows, (ns.value or b"").decode("utf-8", "replace"), (tn.value or b"").decode("utf-8", "replace")
    finally:
        L.tfgpu_sr_schema_free.argtypes = [C.c_void_p]
        L.tfgpu_sr_schema_free(h)


def json_schema_rows(schema_text: str) -> Tuple[str, List[Tuple[str, int, bool]]]:
    """(title, [(name, SRT_*, required)]) in util.MapKeysInOrder order — what processPayload iterates."""
    title, rows, _ns, _tn = _compile(schema_text, POLICY_TITLE, "")
    return title, rows


def build_json_table_id(title: str, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = "") -> Tuple[str, str]:
    """(ChangeItem.Schema, ChangeItem.Table) — BuildJSONTableID through the same entry point (a schema that holds only the title)."""
    import json
    _t, _rows, ns, table = _compile(json.dumps({"type": "object", "title": title}), policy, manual_table_name)
    return ns, table


def sr_json_options(schema_id: int, schema_text: str, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = "",
                    report_frame_errors: bool = True, is_generate_updates: bool = False) -> abi.CSrJsonOptions:
    _title, rows, ns, table = _compile(schema_text, policy, manual_table_name)
    return abi.sr_json_options(schema_id, rows, ns, table, is_generate_updates=is_generate_updates, report_frame_errors=report_frame_errors)


def table_schema(rows, ns: str = "", table: str = "") -> abi.Schema:
    """abstract.NewTableSchema(rows) of processPayload: one ColSchema per property, TableSchema / TableName set to the
    message's table (jsonPropertyToJSONSchemaRow, utils_json.go:71-95) — the native wire form carries them."""
    return abi.Schema([abi.ColSchema(n, SRT_DTYPE[t], False, "", "", bool(r), ns, table) for n, t, r in rows])


PROTO_POLICY = {"debezium_style": "debezium_style", "title": "message_name"}  # JSON table-name policy of the parser → the protobuf one it goes with (DefaultDerivedTableNamePolicy: both debezium_style)


class ConfluentSrParser:
    """ConfluentSrImpl.DoBatch for JSON schemas (engine/parser.go:108-152) over an engine that provides
    `sr_frames(data, msgs)` and `sr_json_parse(opts, data, msgs)` — transferia_amd.lib on the device (the tests also run it
    over the oracle).  The registry is a mapping schema id → {"schema": text, "schemaType": "JSON" | ...}: the network
    client stays with the caller, as it stays in Go.

    do_batch returns the items in the reference's order — message by message, frame by frame — as dicts:
      {"kind": "insert", "schema", "table", "names", "values": [(gotype, value)], "msg", "index"}   a parsed frame
      {"unparsed": code, "msg", "index"}                                                            generic.NewUnparsed
      {"fallback": True, "msg"}                                                  the message goes through the stock parser
    An `_unparsed` frame ends its message for the frames after it, whatever their schema id (DoBuf stops at the first
    nil rest); frames of a schema the registry does not know, or of a non-JSON schema, are the stock parser's business.
    """

    def __init__(self, registry, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = "", is_generate_updates: bool = False):
        self.is_generate_updates = bool(is_generate_updates)  # ParserConfig*.IsGenerateUpdates: JSON items become Updates listing the fields their payload holds
        self.registry, self.policy, self.manual = registry, policy, manual_table_name
        self._plans = {}

    def _plan(self, sid: int):
        if sid not in self._plans:
            ent = self.registry.get(sid) or self.registry.get(str(sid))
            self._plans[sid] = None if not ent or ent.get("schemaType") != "JSON" else ent["schema"]
        return self._plans[sid]

    def _proto(self, sid: int):
        ent = self.registry.get(sid) or self.registry.get(str(sid))
        return ent["schema"] if ent and ent.get("schemaType") == "PROTOBUF" else None

    def do_batch(self, engine, data, msgs=None):
        frames = engine.sr_frames(data, msgs)
        fate = {}  # frame ordinal → item
        first = True
        # PROTOBUF schemas: the payload is the whole rest of its Kafka message (doWithSchema, parser.go:46-48), so only a message's
        # FIRST frame is one — what the JSON cut found behind it are bytes of the same protobuf message
        proto_msgs = {}
        for k, f in enumerate(frames):
            if f[5] == 0 and f[4] == 0 and self._proto(f[3]) is not None:
                proto_msgs[f[0]] = (k, f[3])
        ignored = set()
        if proto_msgs:
            ignored = {k for k, f in enumerate(frames) if f[0] in proto_msgs and proto_msgs[f[0]][0] != k}
            for sid in sorted({v[1] for v in proto_msgs.values()}):
                text = self._proto(sid)
                ns, table, names, rows, errors = engine.sr_proto_parse(sid, text.encode() if isinstance(text, str) else text, data, msgs,
                                                                       PROTO_POLICY.get(self.policy, "debezium_style"), self.manual)
                for m, (k, s2) in proto_msgs.items():
                    if s2 != sid:
                        continue
                    if m in rows:
                        fate[k] = {"kind": "insert", "schema": ns, "table": table, "names": names, "values": rows[m], "msg": m, "index": 0}
                    elif errors.get(m) == abi.ROWERR_ID["HOST_FALLBACK"]:
                        fate[k] = {"fallback": True, "msg": m}
                    elif m in errors:
                        fate[k] = {"unparsed": errors[m], "msg": m, "index": 0}
        proto_frames = {v[0] for v in proto_msgs.values()}
        for sid in sorted({f[3] for k, f in enumerate(frames) if f[4] == 0 and k not in ignored and k not in proto_frames}):
            text = self._plan(sid)
            if text is None:
                for k, f in enumerate(frames):
                    if f[4] == 0 and f[3] == sid:
                        fate[k] = {"fallback": True, "msg": f[0]}
                continue
            opts = sr_json_options(sid, text, self.policy, self.manual, report_frame_errors=first, is_generate_updates=self.is_generate_updates)
            first = False
            res = engine.sr_json_parse(opts, data, msgs)
            b = res.batch
            for e in res.errors:
                k, code = int(e[0]), int(e[1])
                fate[k] = {"fallback": True, "msg": frames[k][0]} if code == abi.ROWERR_ID["HOST_FALLBACK"] else {"unparsed": code, "msg": frames[k][0], "index": frames[k][5]}
            names = [c.name for c in b.cols]
            for r in range(b.nrows):
                k = int(b.src_row[r])
                fate[k] = {"kind": "insert", "schema": b.table_ns, "table": b.table_name, "names": names, "values": [c.pyvalue(r) for c in b.cols],
                           "msg": frames[k][0], "index": frames[k][5]}
        if first:  # no JSON schema in the batch: the frame errors were reported by nobody
            for k, f in enumerate(frames):
                if f[4]:
                    fate[k] = {"unparsed": f[4], "msg": f[0], "index": f[5]}
        out, dead_msg, host_msgs = [], set(), set()
        for k, f in enumerate(frames):  # the reference's order; the first `_unparsed` frame of a message ends it
            it = fate.get(k)
            if k in ignored or f[0] in dead_msg or f[0] in host_msgs or it is None:
                continue
            if it.get("fallback"):
                host_msgs.add(f[0])
                out = [x for x in out if x["msg"] != f[0]]  # the whole message is re-parsed by the stock parser
                out.append({"fallback": True, "msg": f[0]})
                continue
            out.append(it)
            if "unparsed" in it:
                dead_msg.add(f[0])
        return out


# ---- PROTOBUF schemas (engine/format_protobuf.go, utils_protobuf.go, types_protobuf.go) — pass-throughs to csrc/tf_protoschema.cpp / tf_protobuf.hip ----
PB_DTYPE = {abi_pb: dt for abi_pb, dt in enumerate([None, "double", "float", "int64", "uint64", "int32", "uint64", "uint32", "boolean", "utf8", "string", "uint32", "int32", "int64",
                                                     "int32", "int64", "utf8", "any"])}   # protoSchemaTypes by TFGPU_PB_*


class _CPbMember(C.Structure):
    _fields_ = [("name", C.c_char_p), ("number", C.c_int32), ("ptype", C.c_int32)]


class _CPbField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("number", C.c_int32), ("ptype", C.c_int32), ("nmembers", C.c_int32), ("members", C.POINTER(_CPbMember)), ("repeated", C.c_int32), ("oneof", C.c_int32)]


class ProtoSchema:
    """One registry .proto text compiled for the device (tfgpu_sr_compile_proto): code = abi.ROW_OK with `fields` = [(name, number,
    TFGPU_PB_*, [(member, number, TFGPU_PB_*)], repeated)], or ROW_SR_PROTO / ROW_HOST_FALLBACK with `why`; ns / table / record as the reference
    derives them (BuildProtobufTableID, getRecordName)."""

    def __init__(self, lib, text: bytes, policy: str = "debezium_style", manual_table_name: str = "", message_name: str = ""):
        self.lib = lib
        L = lib.load()
        self._h = C.c_void_p()
        L.tfgpu_sr_compile_proto.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib._check(L.tfgpu_sr_compile_proto(text, len(text), policy.encode(), manual_table_name.encode(), message_name.encode(), C.byref(self._h)))
        code, nf = C.c_int32(0), C.c_int32(0)
        fp = C.POINTER(_CPbField)()
        ns, table, record, why = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        L.tfgpu_pb_schema_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(_CPbField)), C.POINTER(C.c_int32)] + [C.POINTER(C.c_char_p)] * 4
        lib._check(L.tfgpu_pb_schema_info(self._h, C.byref(code), C.byref(fp), C.byref(nf), C.byref(ns), C.byref(table), C.byref(record), C.byref(why)))
        self.code, self.why = int(code.value), (why.value or b"").decode()
        self.ns, self.table, self.record = (ns.value or b"").decode(), (table.value or b"").decode(), (record.value or b"").decode()
        self.fields = [(fp[i].name.decode(), int(fp[i].number), int(fp[i].ptype), [(fp[i].members[k].name.decode(), int(fp[i].members[k].number), int(fp[i].members[k].ptype)) for k in range(fp[i].nmembers)], bool(fp[i].repeated))
                       for i in range(nf.value)]
        self.oneofs = [int(fp[i].oneof) for i in range(nf.value)]   # k > 0: the field is a member of the message's k-th oneof

    def __del__(self):
        try:
            if self._h:
                self.lib.load().tfgpu_pb_schema_free.argtypes = [C.c_void_p]
                self.lib.load().tfgpu_pb_schema_free(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    def table_schema(self) -> abi.Schema:
        return abi.Schema([abi.ColSchema(n, "any" if rep else PB_DTYPE[t], False, "", "", False, self.ns, self.table) for (n, _k, t, _m, rep) in self.fields])

    def parse(self, schema_id: int, data, msgs=None, report_frame_errors: bool = True):
        """tfgpu_sr_proto_parse → (DeviceBatch, {message index: TFGPU_ROW_* code})"""
        lib = self.lib
        L = lib.load()
        ptr_, n, mem, keep = lib._bytes_arg(data)
        nmsg = int(msgs.nmsg) if msgs is not None else 1
        errs = (abi.CRowError * max(nmsg, 1))()
        ne, out = C.c_int64(0), C.c_void_p()
        L.tfgpu_sr_proto_parse.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib._check(L.tfgpu_sr_proto_parse(self._h, schema_id, 1 if report_frame_errors else 0, ptr_, n, mem, C.byref(msgs) if msgs is not None else None, C.byref(out), errs, max(nmsg, 1), C.byref(ne)))
        return lib.DeviceBatch(out), {int(errs[i].row): int(errs[i].code) for i in range(min(int(ne.value), max(nmsg, 1)))}
