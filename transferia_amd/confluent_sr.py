"""Host side of the Confluent Schema Registry parser's JSON-schema path: the per-schema set-up the
reference does in Go before any message is touched — unmarshal the JSON schema, resolve every property to a column
(type through `oneOf`, required set) and derive the table id from the title.  It runs once per schema id (the shim
caches it next to the registry client); the per-message work is tfgpu_sr_frames / tfgpu_sr_json_parse.

    JSONProperties, jsonPropertyToJSONSchemaRow   engine/utils_json.go:15-21, 71-95
    jsonSchemaTypes                               engine/types_json.go:25-32
    BuildJSONTableID                              table_name_policy/table_name_policy.go:73-92
"""
from __future__ import annotations

import json
from typing import List, Optional, Tuple

from . import abi

# types_json.go:25-32 — JSON-schema `type` → the column's ytschema type
_JSON_TYPES = {"array": abi.SRT_ANY, "boolean": abi.SRT_BOOLEAN, "integer": abi.SRT_INTEGER, "number": abi.SRT_NUMBER,
               "object": abi.SRT_ANY, "string": abi.SRT_STRING}
SRT_DTYPE = {abi.SRT_BOOLEAN: "boolean", abi.SRT_INTEGER: "int64", abi.SRT_NUMBER: "double", abi.SRT_STRING: "utf8", abi.SRT_ANY: "any"}

POLICY_DEBEZIUM_STYLE, POLICY_TITLE = "debezium_style", "title"


def _field(obj: dict, name: str):
    """encoding/json binds every key of the document to the struct field it names — exactly or, failing that, ignoring case —
    in document order, so the LAST key that names the field is the one whose value stays (decode.go object())."""
    out, want = None, name.lower()
    for k, v in obj.items():  # json.loads keeps document order (a repeated identical key: the last one, as in Go)
        if k == name or k.lower() == want:
            out = v
    return out


def json_schema_rows(schema_text: str) -> Tuple[str, List[Tuple[str, int, bool]]]:
    """(title, [(name, SRT_*, required)]) in util.MapKeysInOrder order — what processPayload iterates."""
    js = json.loads(schema_text)
    if not isinstance(js, dict) or _field(js, "type") != "object":
        raise ValueError("json schema type must be 'object'")  # utils_json.go:35-37
    required = set(_field(js, "required") or [])
    props = _field(js, "properties") or {}
    rows = []
    for name in sorted(props, key=lambda s: s.encode("utf-8")):  # Go compares strings bytewise
        p = props[name]
        t = _JSON_TYPES.get(_field(p, "type") or "")
        req = name in required
        one_of = _field(p, "oneOf")
        if one_of is not None:
            for q in one_of:
                qt = _field(q, "type") or ""
                if qt == "null":
                    req = False
                else:
                    t = _JSON_TYPES.get(qt)
        if t is None:
            raise ValueError("property %r: JSON-schema type without a column type (DataType \"\" in the reference)" % name)
        rows.append((name, t, req))
    return _field(js, "title") or "", rows


def build_json_table_id(title: str, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = "") -> Tuple[str, str]:
    """(ChangeItem.Schema, ChangeItem.Table)."""
    if manual_table_name:
        return "", manual_table_name
    if policy == POLICY_DEBEZIUM_STYLE:
        parts = title.split(".", 1)
        if len(parts) != 2:
            raise ValueError("Can't split title '%s' from json into schema and table names" % title)
        return parts[0], parts[1]
    if policy == POLICY_TITLE:
        return "", title
    raise ValueError("invalid JSONTableNamePolicy")


def sr_json_options(schema_id: int, schema_text: str, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = "",
                    report_frame_errors: bool = True) -> abi.CSrJsonOptions:
    title, rows = json_schema_rows(schema_text)
    ns, table = build_json_table_id(title, policy, manual_table_name)
    return abi.sr_json_options(schema_id, rows, ns, table, report_frame_errors=report_frame_errors)


def table_schema(rows, ns: str = "", table: str = "") -> abi.Schema:
    """abstract.NewTableSchema(rows) of processPayload: one ColSchema per property, TableSchema / TableName set to the
    message's table (jsonPropertyToJSONSchemaRow, utils_json.go:71-95) — the native wire form carries them."""
    return abi.Schema([abi.ColSchema(n, SRT_DTYPE[t], False, "", "", bool(r), ns, table) for n, t, r in rows])


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

    def __init__(self, registry, policy: str = POLICY_DEBEZIUM_STYLE, manual_table_name: str = ""):
        self.registry, self.policy, self.manual = registry, policy, manual_table_name
        self._plans = {}

    def _plan(self, sid: int):
        if sid not in self._plans:
            ent = self.registry.get(sid) or self.registry.get(str(sid))
            self._plans[sid] = None if not ent or ent.get("schemaType") != "JSON" else ent["schema"]
        return self._plans[sid]

    def do_batch(self, engine, data, msgs=None):
        frames = engine.sr_frames(data, msgs)
        fate = {}  # frame ordinal → item
        first = True
        for sid in sorted({f[3] for f in frames if f[4] == 0}):
            text = self._plan(sid)
            if text is None:
                for k, f in enumerate(frames):
                    if f[4] == 0 and f[3] == sid:
                        fate[k] = {"fallback": True, "msg": f[0]}
                continue
            opts = sr_json_options(sid, text, self.policy, self.manual, report_frame_errors=first)
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
            if f[0] in dead_msg or f[0] in host_msgs or it is None:
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
