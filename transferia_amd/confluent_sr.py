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
    """encoding/json matches struct fields by exact name first, then case-insensitively."""
    if name in obj:
        return obj[name]
    for k, v in obj.items():
        if k.lower() == name:
            return v
    return None


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


def table_schema(rows) -> abi.Schema:
    """abstract.NewTableSchema(rows) of processPayload: one ColSchema per property."""
    return abi.Schema([abi.ColSchema(n, SRT_DTYPE[t], False, "", "", bool(r)) for n, t, r in rows])
