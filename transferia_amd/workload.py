"""Synthetic ClickBench-`hits` workload: the 105-column schema
of the reference fixture pkg/providers/postgres/testdata/hits_data.json (kept
as tests/golden/hits_schema.json) and a deterministic, row-addressable CSV
generator (tools/hitsgen.c).  Used by bench.py and the tests; produces input
data only."""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

from . import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(_ROOT, "tools", "hitsgen.c")
_SO = os.path.join(_ROOT, "tools", "libhitsgen.so")
SEED = 0x5EEDC11C

_lib = None


def build_generator(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, _SRC, "-lm"])
    return _SO


def _gen():
    global _lib
    if _lib is None:
        build_generator()
        L = C.CDLL(_SO)
        L.hits_csv.restype = C.c_uint64
        L.hits_csv.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
        L.hits_row_cap.restype = C.c_uint64
        L.hits_row_cap.argtypes = [C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def hits_columns():
    with open(os.path.join(_ROOT, "tests", "golden", "hits_schema.json"), encoding="utf-8") as f:
        return json.load(f)["columns"]


def hits_schema() -> abi.Schema:
    """TableSchema of `hits` as the s3 CSV reader sees it: column i reads CSV field i."""
    return abi.Schema([abi.ColSchema(n, t, bool(k), str(i)) for i, (n, t, k) in enumerate(hits_columns())])


_UNIFORM64 = {"watchid", "userid", "funiqid", "refererhash", "urlhash"}
_UNIFORM32 = {"clientip", "remoteip"}
_ZIPF = {"counterid", "regionid"}
_URLS = {"url", "referer", "originalurl"}


def hits_roles() -> np.ndarray:
    roles = []
    for name, typ, _ in hits_columns():
        if typ == "int16":
            roles.append(0 if name.startswith(("is", "has", "java", "cookie", "dontcount", "withhash", "goodevent")) else 1)
        elif typ == "int32":
            roles.append(2 if name in _UNIFORM32 else 3 if name in _ZIPF else 12)
        elif typ == "int64":
            roles.append(4 if name in _UNIFORM64 else 5)
        elif typ == "timestamp":
            roles.append(6)
        elif typ == "date":
            roles.append(7)
        elif name == "title":
            roles.append(8)
        elif name in _URLS:
            roles.append(9)
        elif name in ("useragentminor", "hitcolor"):
            roles.append(11)
        else:
            roles.append(10)
    return np.asarray(roles, dtype=np.int32)


def hits_csv(nrows: int, row0: int = 0, seed: int = SEED, header: bool = True) -> bytes:
    """CSV text (',' delimiter, '"' quote, header line) of rows [row0, row0+nrows)."""
    L = _gen()
    roles = hits_roles()
    cap = int(L.hits_row_cap(roles.ctypes.data, len(roles)))
    # expected ~700 B/row; size for 1.5x the expectation, retry with the hard cap if needed
    for est in (max(1 << 16, int(nrows * 1100) + cap), nrows * cap + cap):
        buf = np.empty(est, dtype=np.uint8)
        n = int(L.hits_csv(seed, row0, nrows, roles.ctypes.data, len(roles), buf.ctypes.data, est))
        if n or nrows == 0:
            body = buf[:n].tobytes()
            break
    else:
        raise RuntimeError("hits generator: buffer too small")
    if header:
        return (",".join(c[0] for c in hits_columns()) + "\n").encode() + body
    return body


class HitsStream:
    """Chunked generator writing into ONE reused host buffer (fresh pages are
    expensive on sandboxed hosts), e.g. for staging a large table into HBM."""

    def __init__(self, chunk_rows: int = 1 << 15, seed: int = SEED):
        self.L = _gen()
        self.roles = hits_roles()
        self.seed = seed
        self.chunk_rows = chunk_rows
        cap = int(self.L.hits_row_cap(self.roles.ctypes.data, len(self.roles)))
        self.buf = np.zeros(chunk_rows * 1300 + cap, dtype=np.uint8)
        self.header = (",".join(c[0] for c in hits_columns()) + "\n").encode()

    def chunk(self, row0: int, nrows: int):
        """→ (buffer, nbytes) for rows [row0, row0+nrows); buffer is reused."""
        n = int(self.L.hits_csv(self.seed, row0, nrows, self.roles.ctypes.data, len(self.roles), self.buf.ctypes.data, len(self.buf)))
        if not n and nrows:
            raise RuntimeError("hits generator: chunk buffer too small")
        return self.buf, n

    def total_bytes(self, row0: int, nrows: int, header: bool = True) -> int:
        tot = len(self.header) if header else 0
        r = row0
        while r < row0 + nrows:
            k = min(self.chunk_rows, row0 + nrows - r)
            tot += self.chunk(r, k)[1]
            r += k
        return tot


def hits_csv_options(header: bool = True) -> abi.CCsvOptions:
    return abi.csv_options(skip_rows=1 if header else 0)


def cdc_batch(nrows: int, keys: int = 0, seed: int = SEED, p_old: float = 0.6, p_pk_change: float = 0.15, toast: float = 0.0):
    """A CDC stream slice for abstract.Collapse (BASELINE.json configs[4]: debezium stream → hash-partition → dedup):
    (Batch, Schema) of `nrows` row events over `keys` primary keys (default nrows / 4: chains of ~4 events per key),
    35 % inserts, 45 % updates, 20 % deletes; Update / Delete rows carry OldKeys with probability p_old and
    p_pk_change of those change the primary key.  Columns: id int64 (PK), ver int64, payload utf8 (24 bytes), ts int64.
    toast > 0: that share of the Updates leaves `payload` out of its ColumnNames (an unchanged TOASTed column: Column.absent),
    so Collapse runs compareColumns' merge (change_item_collapse.go:7-35, :86-100)."""
    rng = np.random.default_rng(seed)
    keys = keys or max(nrows // 4, 1)
    schema = abi.Schema([abi.ColSchema("id", "int64", True, "", ""), abi.ColSchema("ver", "int64", False, "", ""),
                         abi.ColSchema("payload", "utf8", False, "", ""), abi.ColSchema("ts", "int64", False, "", "")])
    ids = rng.integers(0, keys, size=nrows, dtype=np.int64)
    kind = rng.choice(np.array([abi.K_INSERT, abi.K_UPDATE, abi.K_DELETE], dtype=np.uint8), size=nrows, p=[0.35, 0.45, 0.20])
    pay = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)[rng.integers(0, 16, size=(nrows, 24))]
    paycol = abi.Column("payload", "utf8", abi.R_STRING, offsets=(np.arange(nrows + 1, dtype=np.uint32) * 24), data=np.ascontiguousarray(pay.reshape(-1)))
    if toast > 0:
        ab = (kind == abi.K_UPDATE) & (np.random.default_rng(seed ^ 0x70A57).random(nrows) < toast)
        lens = np.where(ab, 0, 24).astype(np.uint32)
        off = np.zeros(nrows + 1, np.uint32)
        np.cumsum(lens, out=off[1:])
        paycol = abi.Column("payload", "utf8", abi.R_STRING, offsets=off, data=np.ascontiguousarray(pay[~ab].reshape(-1)), validity=~ab, absent=ab)
    cols = [abi.Column("id", "int64", abi.R_INT64, values=ids), abi.Column("ver", "int64", abi.R_INT64, values=np.arange(nrows, dtype=np.int64)), paycol,
            abi.Column("ts", "int64", abi.R_INT64, values=1_700_000_000_000 + np.arange(nrows, dtype=np.int64))]
    b = abi.Batch(cols, nrows, "public", "events", kind=kind)
    b.schema = schema
    present = (kind != abi.K_INSERT) & (rng.random(nrows) < p_old)
    old = ids.copy()
    change = present & (rng.random(nrows) < p_pk_change)
    old[change] = rng.integers(0, keys, size=int(change.sum()), dtype=np.int64)
    old[~present] = 0
    b.old_keys = [abi.Column("id", "int64", abi.R_INT64, values=old, validity=present.copy())]
    b.old_present = present
    return b, schema


def debezium_cdc_messages(nmsg: int, keys: int = 0, seed: int = SEED, table: str = "events"):
    """BASELINE.json configs[4] from its real source format: a Postgres CDC stream in Debezium's JSON envelope with the Kafka
    Connect schema inlined in every message (the JsonConverter default).  Table public.<table>: id int64 (PK), ver int32,
    payload text (24 bytes), amount numeric(10,2) (Decimal), ts MicroTimestamp; 35 % inserts / 45 % updates / 20 % deletes over
    `keys` primary keys (default nmsg / 4).  Returns the list of message values (bytes), ~2 KB each."""
    import base64
    import json
    rng = np.random.default_rng(seed)
    keys = keys or max(nmsg // 4, 1)

    def struct(field):
        return {"type": "struct", "optional": True, "name": "srv.public.%s.Value" % table, "field": field, "fields": [
            {"type": "int64", "optional": False, "field": "id"}, {"type": "int32", "optional": True, "field": "ver"},
            {"type": "string", "optional": True, "field": "payload"},
            {"type": "bytes", "optional": True, "name": "org.apache.kafka.connect.data.Decimal", "version": 1,
             "parameters": {"scale": "2", "connect.decimal.precision": "10"}, "field": "amount"},
            {"type": "int64", "optional": True, "name": "io.debezium.time.MicroTimestamp", "version": 1, "field": "ts"}]}
    source = {"type": "struct", "optional": False, "name": "io.debezium.connector.postgresql.Source", "field": "source", "fields": [
        {"type": "string", "optional": False, "field": f} for f in ("version", "connector", "name")] + [
        {"type": "int64", "optional": False, "field": "ts_ms"}, {"type": "string", "optional": True, "field": "snapshot"},
        {"type": "string", "optional": False, "field": "db"}, {"type": "string", "optional": False, "field": "schema"},
        {"type": "string", "optional": False, "field": "table"}, {"type": "int64", "optional": True, "field": "txId"},
        {"type": "int64", "optional": True, "field": "lsn"}, {"type": "int64", "optional": True, "field": "xmin"}]}
    schema = json.dumps({"type": "struct", "optional": False, "name": "srv.public.%s.Envelope" % table, "fields": [
        struct("before"), struct("after"), source, {"type": "string", "optional": False, "field": "op"}, {"type": "int64", "optional": True, "field": "ts_ms"}]},
        separators=(",", ":"))
    ids = rng.integers(0, keys, size=nmsg)
    ops = rng.choice(np.array(["c", "u", "d"]), size=nmsg, p=[0.35, 0.45, 0.20])
    pay = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)[rng.integers(0, 16, size=(nmsg, 24))]
    cents = rng.integers(-10**9, 10**9, size=nmsg)
    out = []
    for k in range(nmsg):
        c = int(cents[k])
        row = '{"id":%d,"ver":%d,"payload":"%s","amount":"%s","ts":%d}' % (
            int(ids[k]), k, pay[k].tobytes().decode(), base64.b64encode(c.to_bytes(max(1, (c.bit_length() + 8) // 8), "big", signed=True)).decode(),
            1_700_000_000_000_000 + k)
        op = str(ops[k])
        before = row if op in ("u", "d") else "null"
        after = "null" if op == "d" else row
        payload = ('{"before":%s,"after":%s,"source":{"version":"2.4.0.Final","connector":"postgresql","name":"srv","ts_ms":%d,"snapshot":"false","db":"db",'
                   '"schema":"public","table":"%s","txId":%d,"lsn":%d,"xmin":null},"op":"%s","ts_ms":%d,"transaction":null}') % (
            before, after, 1_700_000_000_000 + k, table, 500 + k, 10_000 + 8 * k, op, 1_700_000_000_123 + k)
        out.append(('{"schema":%s,"payload":%s}' % (schema, payload)).encode())
    return out


def kafka_to_confluent_schema(k: dict) -> dict:
    """A Kafka Connect JSON schema in the form a Confluent schema registry stores it (what the Debezium serializer registers:
    KafkaJSONSchema.ToConfluentSchema, pkg/schemaregistry/format/json_schema_format.go:166-258) — input generation for the
    registry-framed Debezium workload."""
    if k.get("optional"):
        inner = dict(k)
        inner["optional"] = False
        return {"oneOf": [{"type": "null"}, kafka_to_confluent_schema(inner)]}
    ints = ("int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64")
    t = k.get("type", "")
    jt, ct = ("integer", t) if t in ints else {"float": ("number", "float32"), "double": ("number", "float64"), "string": ("string", ""), "struct": ("object", ""),
                                                "bytes": ("string", "bytes"), "boolean": ("boolean", ""), "array": ("array", "")}.get(t, ("", ""))
    c = {}
    if k.get("parameters") is not None:
        c["connect.parameters"] = k["parameters"]
    if ct:
        c["connect.type"] = ct
    if k.get("version"):
        c["connect.version"] = k["version"]
    if k.get("default") is not None:
        c["default"] = k["default"]
    if k.get("doc"):
        c["description"] = k["doc"]
    if k.get("items") is not None:
        c["items"] = kafka_to_confluent_schema(k["items"])
    if k.get("fields"):
        c["properties"] = {}
        for i, f in enumerate(k["fields"]):
            p = kafka_to_confluent_schema(f)
            p["connect.index"] = i
            c["properties"][f.get("field", "")] = p
    if k.get("name"):
        c["title"] = k["name"]
    if jt:
        c["type"] = jt
    if k.get("__dt_original_type_info") is not None:
        c["__dt_original_type_info"] = k["__dt_original_type_info"]
    return c


def registry_framed(message: bytes, schema_id: int):
    """An inline-schema Debezium message as (registry schema text, event bytes 0x00 | schema id | payload)."""
    import json
    doc = json.loads(message)
    i = message.index(b'"payload":') + len(b'"payload":')
    payload = message[i:message.rindex(b"}")].strip()
    assert json.loads(payload) == doc["payload"]
    text = json.dumps(kafka_to_confluent_schema(doc["schema"]), separators=(",", ":")).encode()
    return text, b"\x00" + int(schema_id).to_bytes(4, "big") + payload
