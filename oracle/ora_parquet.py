"""CPU restatement of the reference's Parquet source for tests (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's
cpu_baseline leg may import this; the product path never does).

What is restated is the REFERENCE's own code:
  * the schema resolver — pkg/providers/s3/reader/registry/parquet/parquet_schema_resolver.go:81-158 (physical type, then
    logical type, then converted type decide the ytschema type; OriginalType = "parquet:" + el.Type().String()), with the
    system columns of s3_reader.AppendSystemColsTableSchema in front (__file_name utf8 key, __row_index uint64 key);
  * the row path — reader_parquet.go:137-283 (pr.Read(&row) → constructCI: system columns, a column the file lacks is nil),
    parseParquetField :299-323 (deprecated.Int96 → its decimal String(); DATE → time.Unix(0, 0).Add(24h * days); everything
    else through abstract.Restore, pkg/abstract/restore.go:20-260: a DECIMAL column is typed `double` and its int32 / int64 /
    []byte value falls to Restore's default → nil; []byte under a `string` / `utf8` column → string; float32 under `double`
    → float64; an int64 under a `timestamp` column → ytschema.Timestamp(v) = MICROseconds whatever the file's unit says).

What is NOT restated is the file-format decoder: the reference reads through github.com/parquet-go/parquet-go (go.mod; not under
/root/reference), here pyarrow decodes the pages — an independent implementation of the published format.  The restatement is
pinned to the reference's 30 reader canon outputs (tests/golden/parquet_reader.json ← tests/canon/s3/parquet/canondata, see
tests/test_parquet_canon.py): type mapping, OriginalType strings, Go value types and the first rows of every flat file.
Nested columns (`any`: the map / slice tree parquet-go builds) are named, not restated: `Nested` is raised for them."""
from __future__ import annotations

import io

import pyarrow as pa
import pyarrow.parquet as pq

SYSTEM_COLS = ("__file_name", "__row_index")


class Nested(Exception):
    """a top-level field that is a group (list / map / struct): its value is parquet-go's own tree, not restated here"""


def _logical_string(col) -> str | None:
    """parquet-go's LogicalType.String() for the annotations the resolver meets (format/parquet.go); None = no logical type."""
    lt = col.logical_type
    t = lt.type
    if t in ("NONE", "UNKNOWN"):
        # parquet-go lifts the legacy converted types into logical ones when it opens a file (the canon's
        # fixed_length_decimal_legacy prints DECIMAL(13,2)); pyarrow does the same for DECIMAL / UTF8 / DATE
        return None
    if t == "STRING":
        return "STRING"
    if t == "DECIMAL":
        return "DECIMAL(%d,%d)" % (col.precision, col.scale)
    if t == "INT":
        import json
        j = json.loads(lt.to_json())
        return "INT(%d,%s)" % (j["bitWidth"], "true" if j["isSigned"] else "false")
    if t == "TIMESTAMP":
        import json
        j = json.loads(lt.to_json())
        return "TIMESTAMP(isAdjustedToUTC=%s,unit=%s)" % ("true" if j["isAdjustedToUTC"] else "false", {"milliseconds": "MILLIS", "microseconds": "MICROS", "nanoseconds": "NANOS"}[j["timeUnit"]])
    return t  # DATE, UUID, ENUM, JSON, BSON, TIME…


def _physical_string(col) -> str:
    if col.physical_type == "FIXED_LEN_BYTE_ARRAY":
        return "FIXED_LEN_BYTE_ARRAY(%d)" % col.length
    return col.physical_type


def _leaf_schema(col):
    """(ytschema type, OriginalType) of one leaf — parquet_schema_resolver.go:96-153, in its order."""
    typ = {"BOOLEAN": "boolean", "INT32": "int32", "INT64": "int64", "FLOAT": "float", "DOUBLE": "double", "INT96": "utf8",
           "BYTE_ARRAY": "string", "FIXED_LEN_BYTE_ARRAY": "string"}[col.physical_type]
    lt = col.logical_type.type
    if lt == "DATE":
        typ = "date"
    elif lt == "STRING":
        typ = "utf8"
    elif lt == "INT":
        import json
        typ = "int64" if json.loads(col.logical_type.to_json())["isSigned"] else "uint64"
    elif lt == "DECIMAL":
        typ = "utf8" if col.precision > 8 else "double"
    elif lt == "TIMESTAMP":
        typ = "timestamp"
    elif lt in ("UUID", "ENUM"):
        typ = "utf8"
    ct = col.converted_type
    if ct == "UTF8":
        typ = "utf8"
    elif ct == "DATE":
        typ = "date"
    elif ct == "DECIMAL":
        typ = "double"
    return typ, "parquet:" + (_logical_string(col) or _physical_string(col))


def _open(data: bytes):
    return pq.ParquetFile(io.BytesIO(data))


def _fields(pf):
    """top-level fields in file order: (name, leaf column or None for a group, arrow type)"""
    leaves = {}
    for i in range(pf.metadata.num_columns):
        c = pf.schema.column(i)
        leaves.setdefault(c.path.split(".")[0], []).append(c)
    out = []
    for f in pf.schema_arrow:
        cs = leaves.get(f.name, [])
        flat = len(cs) == 1 and cs[0].path == f.name and cs[0].max_repetition_level == 0
        out.append((f.name, cs[0] if flat else None, f.type))
    return out


def resolve_schema(data: bytes, hide_system_cols: bool = False):
    """[(name, type, original_type, key, required)] — resolveSchema + the system columns (keys, since an inferred schema has none)."""
    cols = []
    for name, leaf, at in _fields(_open(data)):
        if leaf is None:
            # a group: TypeAny; Type().String() is the group's logical annotation or "group" (a converted-type-only LIST / MAP stays
            # "group" in parquet-go — the canon's datapage_v2 `e`; pyarrow cannot tell the two apart, files written here carry both)
            ot = "LIST" if pa.types.is_list(at) or pa.types.is_large_list(at) else "MAP" if pa.types.is_map(at) else "group"
            cols.append((name, "any", "parquet:" + ot, False, False))
        else:
            t, ot = _leaf_schema(leaf)
            cols.append((name, t, ot, False, False))
    if not hide_system_cols:
        cols = [("__file_name", "utf8", "", True, False), ("__row_index", "uint64", "", True, False)] + cols
    return cols


def _int96_string(ts_ns: int) -> str:
    """deprecated.Int96.String(): the 96-bit integer (nanoseconds of the day | Julian day << 64) in decimal."""
    day, nanos = divmod(ts_ns, 86400 * 10**9)
    return str(((day + 2440588) << 64) | nanos)


def read(data: bytes, schema, file_name: str):
    """rows of [go type, value] cells for the columns of `schema` ((name, type, …) tuples) — Read + constructCI."""
    pf = _open(data)
    fields = {n: (leaf, at) for n, leaf, at in _fields(pf)}
    # pages decoded by pyarrow; INT96 comes back as timestamp[ns], decimals as Decimal, dates as date32
    tab = pf.read(use_pandas_metadata=False, columns=[n for n, (leaf, _) in fields.items() if leaf is not None and any(s[0] == n for s in schema)] or None) if fields else None
    n = pf.metadata.num_rows
    cols = []
    for s in schema:
        name, typ = s[0], s[1]
        if name == "__file_name":
            cols.append([["string", file_name]] * n)
        elif name == "__row_index":
            cols.append([["uint64", i + 1] for i in range(n)])
        elif name not in fields:
            cols.append([["nil", None]] * n)
        else:
            leaf, at = fields[name]
            if leaf is None:
                raise Nested(name)
            cols.append(_column(tab.column(name), leaf, typ))
    return [[c[r] for c in cols] for r in range(n)]


def _column(arr, leaf, typ):
    phys, lt, ct = leaf.physical_type, leaf.logical_type.type, leaf.converted_type
    out = []
    if phys == "INT96":
        arr = arr.cast(pa.timestamp("ns")).cast(pa.int64())
    elif lt == "TIMESTAMP":
        arr = arr.cast(pa.int64())   # the stored integer, in the file's unit
    elif lt == "DATE" or ct == "DATE":
        arr = arr.cast(pa.int32())
    vals = arr.to_pylist()
    for v in vals:
        if v is None:
            out.append(["nil", None])
        elif phys == "INT96":
            out.append(["string", _int96_string(v)])                       # parseParquetField: legacyInt96.String()
        elif lt == "DATE" or ct == "DATE":
            out.append(["time", (int(v) * 86400, 0)])                       # parseLogicalDate
        elif lt == "DECIMAL" or ct == "DECIMAL":
            if typ == "double":
                out.append(["nil", None])                                   # Restore "double": int32 / int64 / []byte → default → nil
            else:
                raise NotImplementedError("DECIMAL under a %s column: not pinned by the canon" % typ)
        elif typ == "timestamp" and phys == "INT64":
            out.append(["time", (v // 10**6, v % 10**6 * 1000)])            # Restore: ytschema.Timestamp(v).Time() — microseconds
        elif phys == "BOOLEAN":
            out.append(["bool", bool(v)])
        elif phys == "INT32":
            out.append(["int32", int(v)] if typ in ("int32",) else _restore_int(typ, int(v), "int32"))
        elif phys == "INT64":
            out.append(["int64", int(v)] if typ in ("int64",) else _restore_int(typ, int(v), "int64"))
        elif phys == "FLOAT":
            out.append(["float64", float(v)] if typ == "double" else ["float32", float(v)])
        elif phys == "DOUBLE":
            out.append(["float64", float(v)])
        else:  # BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY: []byte → string under "string" / "utf8" (restore.go:222-229)
            b = v.encode() if isinstance(v, str) else bytes(v)
            out.append(["string", b])
    return out


def _restore_int(typ, v, go):
    """Restore's integer cases (cast.ToIntNN of an int of another width): the value, retyped; out of range is not pinned."""
    bits = {"int8": 8, "int16": 16, "int32": 32, "int64": 64, "uint8": 8, "uint16": 16, "uint32": 32, "uint64": 64}
    if typ in bits:
        lo, hi = (0, 2 ** bits[typ] - 1) if typ.startswith("u") else (-(2 ** (bits[typ] - 1)), 2 ** (bits[typ] - 1) - 1)
        if not lo <= v <= hi:
            raise NotImplementedError("cast.To%s of %d: not pinned" % (typ, v))
        return [typ, v]
    return [go, v]


def canon_text(b: bytes) -> str:
    """how the canon file shows a Go string: encoding/json replaces every invalid UTF-8 byte with U+FFFD"""
    return b.decode("utf-8", errors="replace")
