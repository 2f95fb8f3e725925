"""TEST INFRASTRUCTURE — the checker's restatement of the schema-registry JSON-schema formats, never part of the product path.

    ConfluentJSONSchema / KafkaJSONSchema          pkg/schemaregistry/format/json_schema_format.go:36-68
    confluentTypeToKafka / kafkaTypeToConfluent    :70-118
    ToKafkaJSONSchema                              :120-164
    ToConfluentSchema / makeOneOfConfluentSchema   :166-258
    Receiver.convertSchemaFormat                   pkg/debezium/receiver.go:118-139

Both structs are handled as Python dicts in the shape json.Marshal gives them (field order of the Go struct, `omitempty` applied), built from
parsed JSON the way json.Unmarshal binds it (unknown keys dropped, exact key names only — the tests hold no keys that bind by case folding).
Pinned to the reference's fixtures (tests/golden/sr_format.json, tests/test_debezium_sr.py): TestKafkaToConfluentToKafka[Arrays] and the canon
file of TestCanonizeMakeClosedContentModelTrue."""
import json

_INTS = ("int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64")


class Unbindable(ValueError):
    """json.Unmarshal into the struct fails (a field of the wrong JSON type)."""


class GoPanic(RuntimeError):
    """the reference dereferences a nil connect.index while sorting, or the order is pdqsort's business"""


def _s(o, k):
    v = o.get(k)
    if v is None:
        return ""
    if not isinstance(v, str):
        raise Unbindable(k)
    return v


def _i(o, k):
    v = o.get(k)
    if v is None:
        return 0
    if isinstance(v, bool) or not isinstance(v, int):
        raise Unbindable(k)
    return v


def _params(v, k):
    if v is None:
        return None
    if not isinstance(v, dict):
        raise Unbindable(k)
    out = {}
    for name in ("length", "connect.decimal.precision", "scale", "allowed"):   # JSONSchemaParameters, omitempty strings
        t = _s(v, name)
        if t:
            out[name] = t
    return out


def bind_confluent(o):
    """json.Unmarshal(text, &ConfluentJSONSchema): a dict with the struct's fields (absent = zero value)."""
    if o is None:
        return {}
    if not isinstance(o, dict):
        raise Unbindable("ConfluentJSONSchema")
    c = {}
    ix = o.get("connect.index")
    if ix is not None:
        if isinstance(ix, bool) or not isinstance(ix, int):
            raise Unbindable("connect.index")
        c["connect.index"] = ix
    p = _params(o.get("connect.parameters"), "connect.parameters")
    if p is not None:
        c["connect.parameters"] = p
    for k in ("connect.type", "description", "title", "type"):
        t = _s(o, k)
        if t:
            c[k] = t
    v = _i(o, "connect.version")
    if v:
        c["connect.version"] = v
    if o.get("default") is not None:
        c["default"] = o["default"]
    if o.get("items") is not None:
        if not isinstance(o["items"], dict):
            raise Unbindable("items")
        c["items"] = bind_confluent(o["items"])
    if o.get("oneOf") is not None:
        if not isinstance(o["oneOf"], list):
            raise Unbindable("oneOf")
        c["oneOf"] = [bind_confluent(x) for x in o["oneOf"]]
    if o.get("properties") is not None:
        if not isinstance(o["properties"], dict):
            raise Unbindable("properties")
        c["properties"] = {k: bind_confluent(v) for k, v in o["properties"].items()}
    if o.get("__dt_original_type_info") is not None:
        c["__dt_original_type_info"] = o["__dt_original_type_info"]
    ap = o.get("additionalProperties")
    if ap is not None:
        if not isinstance(ap, bool):
            raise Unbindable("additionalProperties")
        c["additionalProperties"] = ap
    return c


def bind_kafka(o):
    """json.Unmarshal(text, &KafkaJSONSchema)"""
    if o is None:
        return {"type": "", "optional": False}
    if not isinstance(o, dict):
        raise Unbindable("KafkaJSONSchema")
    k = {"type": _s(o, "type")}
    if o.get("fields") is not None:
        if not isinstance(o["fields"], list):
            raise Unbindable("fields")
        if o["fields"]:
            k["fields"] = [bind_kafka(x) for x in o["fields"]]
    opt = o.get("optional")
    if opt is not None and not isinstance(opt, bool):
        raise Unbindable("optional")
    k["optional"] = bool(opt)
    if _s(o, "name"):
        k["name"] = o["name"]
    if _i(o, "version"):
        k["version"] = o["version"]
    if _s(o, "doc"):
        k["doc"] = o["doc"]
    p = _params(o.get("parameters"), "parameters")
    if p is not None:
        k["parameters"] = p
    if o.get("default") is not None:
        k["default"] = o["default"]
    if o.get("items") is not None:
        if not isinstance(o["items"], dict):
            raise Unbindable("items")
        k["items"] = bind_kafka(o["items"])
    if _s(o, "field"):
        k["field"] = o["field"]
    if o.get("__dt_original_type_info") is not None:
        k["__dt_original_type_info"] = o["__dt_original_type_info"]
    return k


def confluent_type_to_kafka(json_type, connect_type):   # :70-96
    if json_type == "object":
        return "struct"
    if json_type == "string":
        return "bytes" if connect_type == "bytes" else "string"
    if json_type == "boolean":
        return "boolean"
    if json_type == "integer":
        return connect_type
    if json_type == "number":
        return "double" if connect_type == "float64" else "float" if connect_type == "float32" else "bytes"
    if json_type == "array":
        return "array"
    return ""


def kafka_type_to_confluent(t):   # :98-118
    if t in _INTS:
        return "integer", t
    return {"float": ("number", "float32"), "double": ("number", "float64"), "string": ("string", ""), "struct": ("object", ""),
            "bytes": ("string", "bytes"), "boolean": ("boolean", ""), "array": ("array", "")}.get(t, ("", ""))


def to_kafka(c):
    """ConfluentJSONSchema.ToKafkaJSONSchema (:120-164) over a bound struct"""
    for one in c.get("oneOf", []):
        if one.get("type", "") == "null":
            continue
        f = to_kafka(one)
        f["optional"] = True
        return f
    props = list(c.get("properties", {}).items())
    if len(props) >= 2:
        if any("connect.index" not in p for _n, p in props):
            raise GoPanic("nil connect.index")
        idx = [p["connect.index"] for _n, p in props]
        if len(set(idx)) != len(idx):
            raise GoPanic("equal connect.index: sort.Slice is not stable")
        props.sort(key=lambda np: np[1]["connect.index"])
    fields = []
    for name, p in props:
        f = to_kafka(p)
        f["field"] = name
        fields.append(f)
    k = {"type": confluent_type_to_kafka(c.get("type", ""), c.get("connect.type", ""))}
    if fields:
        k["fields"] = fields
    k["optional"] = False
    if c.get("title"):
        k["name"] = c["title"]
    if c.get("connect.version"):
        k["version"] = c["connect.version"]
    if c.get("description"):
        k["doc"] = c["description"]
    if "connect.parameters" in c:
        k["parameters"] = c["connect.parameters"]
    if "default" in c:
        k["default"] = c["default"]
    if "items" in c:
        k["items"] = to_kafka(c["items"])
    if "__dt_original_type_info" in c:
        k["__dt_original_type_info"] = c["__dt_original_type_info"]
    return _kafka_order(k)


def _kafka_order(k):
    return {n: k[n] for n in ("type", "fields", "optional", "name", "version", "doc", "parameters", "default", "items", "field", "__dt_original_type_info") if n in k}


def to_confluent(k, closed=False, depth=0, into_after_or_before=False):
    """KafkaJSONSchema.ToConfluentSchema (:166-258) over a bound struct"""
    if k.get("optional"):
        inner = dict(k)
        inner["optional"] = False
        return {"oneOf": [{"type": "null"}, to_confluent(inner, closed, depth + 1, into_after_or_before)]}
    c = {}
    if "parameters" in k:
        c["connect.parameters"] = k["parameters"]
    jt, ct = kafka_type_to_confluent(k.get("type", ""))
    if ct:
        c["connect.type"] = ct
    if k.get("version"):
        c["connect.version"] = k["version"]
    if "default" in k:
        c["default"] = k["default"]
    if k.get("doc"):
        c["description"] = k["doc"]
    if "items" in k:
        c["items"] = to_confluent(k["items"], closed, depth + 1, into_after_or_before)
    if k.get("fields"):
        props = {}
        for i, f in enumerate(k["fields"]):
            p = to_confluent(f, closed, depth + 1, f.get("field", "") in ("before", "after"))
            p = dict(p)
            p["connect.index"] = i
            props[f.get("field", "")] = p
        c["properties"] = props
    if k.get("name"):
        c["title"] = k["name"]
    if jt:
        c["type"] = jt
    if "__dt_original_type_info" in k:
        c["__dt_original_type_info"] = k["__dt_original_type_info"]
    if closed and depth == 2 and into_after_or_before:
        c["additionalProperties"] = False
    return c


def convert_schema_format(text: bytes) -> bytes:
    """Receiver.convertSchemaFormat with ConverterConfluentJSON: the Kafka Connect schema text UnmarshalSchema then reads.
    Raises Unbindable (→ "can't convert schema format": every event of the schema is `_unparsed`) or GoPanic (→ host)."""
    try:
        o = json.loads(text.decode("utf-8", "surrogateescape"))
    except ValueError as e:
        raise Unbindable(str(e))
    if o is not None and not isinstance(o, dict):
        raise Unbindable("ConfluentJSONSchema")
    return json.dumps(to_kafka(bind_confluent(o)), ensure_ascii=False, separators=(",", ":")).encode("utf-8", "surrogateescape")
