"""Re-creates the inputs of the reference's Parquet reader canon (tests/canon/s3/parquet: the apache/parquet-testing corpus, which is
NOT in the reference tree) from the canon's own values (tests/golden/parquet_reader.json): same column names, physical / logical
types, encodings, codecs and page versions, the canon's rows at the canon's row positions (Counter), filler rows elsewhere.
Written with pyarrow at test time — nothing binary is committed.

What pyarrow cannot write, stated per file in NOTES: a DECIMAL over BYTE_ARRAY (byte_array_decimal: FIXED_LEN_BYTE_ARRAY here), a
converted-type-only annotation (fixed_length_decimal_legacy, datapage_v2's `e`), a dictionary_page_offset of zero
(dict-page-offset-zero).  The ten files with nested columns are built in an approximate shape only: the device names them
and refuses (tests/test_parquet_canon.py asserts that per file)."""
import datetime
import decimal
import io
import json
import os
import re

import pyarrow as pa
import pyarrow.parquet as pq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet_reader.json")

NESTED = {"datapage_v2.snappy.parquet", "list_columns.parquet", "nested_lists.snappy.parquet", "nested_maps.snappy.parquet", "nested_structs.rust.parquet",
          "nonnullable.impala.parquet", "null_list.parquet", "nullable.impala.parquet", "nulls.snappy.parquet", "repeated_no_annotation.parquet"}
NOTES = {
    "byte_array_decimal.parquet": "DECIMAL(4,2) over FIXED_LEN_BYTE_ARRAY (pyarrow cannot write it over BYTE_ARRAY); the reference turns every DECIMAL into nil",
    "fixed_length_decimal_legacy.parquet": "logical + converted DECIMAL (pyarrow cannot write the converted type alone)",
    "dict-page-offset-zero.parquet": "a correct dictionary_page_offset (the original's is 0: a writer bug pyarrow does not reproduce)",
}
# per file: write options (the encodings / codecs / page versions the original files are known for)
OPTIONS = {
    "alltypes_plain.parquet": dict(use_dictionary=False, compression="NONE"),
    "alltypes_dictionary.parquet": dict(use_dictionary=True, compression="NONE"),
    "alltypes_plain.snappy.parquet": dict(use_dictionary=False, compression="SNAPPY"),
    "binary.parquet": dict(use_dictionary=False, compression="NONE"),
    "data_index_bloom_encoding_stats.parquet": dict(use_dictionary=True, compression="NONE", write_page_index=True, write_statistics=True),
    "delta_encoding_optional_column.parquet": dict(use_dictionary=False, compression="NONE", delta=True),
    "delta_encoding_required_column.parquet": dict(use_dictionary=False, compression="NONE", delta=True, required=True),
    "delta_length_byte_array.parquet": dict(use_dictionary=False, compression="ZSTD", column_encoding={"FRUIT": "DELTA_LENGTH_BYTE_ARRAY"}),
    "dict-page-offset-zero.parquet": dict(use_dictionary=True, compression="NONE"),
    "fixed_length_byte_array.parquet": dict(use_dictionary=False, compression="NONE", data_page_size=512),
    "int32_with_null_pages.parquet": dict(use_dictionary=False, compression="NONE", data_page_size=64, write_batch_size=8),
    "lz4_raw_compressed.parquet": dict(use_dictionary=False, compression="LZ4"),
    "plain-dict-uncompressed-checksum.parquet": dict(use_dictionary=True, compression="NONE", write_page_checksum=True),
    "rle_boolean_encoding.parquet": dict(use_dictionary=False, compression="GZIP", column_encoding={"datatype_boolean": "RLE"}),
    "int32_decimal.parquet": dict(store_decimal_as_integer=True, compression="NONE"),
    "int64_decimal.parquet": dict(store_decimal_as_integer=True, compression="NONE"),
}


def golden():
    with open(GOLDEN) as f:
        return json.load(f)["files"]


def text_bytes(s: str) -> bytes:
    """bytes whose canon rendering (invalid UTF-8 → U+FFFD) is s: every U+FFFD stands for one byte that is not UTF-8 (0xE8 here)"""
    return b"".join(b"\xe8" if ch == "�" else ch.encode() for ch in s)


def _arrow_type(ot: str):
    t = ot[len("parquet:"):]
    m = re.match(r"FIXED_LEN_BYTE_ARRAY\((\d+)\)", t)
    if m:
        return pa.binary(int(m.group(1)))
    m = re.match(r"DECIMAL\((\d+),(\d+)\)", t)
    if m:
        return pa.decimal128(int(m.group(1)), int(m.group(2)))
    return {"INT32": pa.int32(), "INT64": pa.int64(), "FLOAT": pa.float32(), "DOUBLE": pa.float64(), "BOOLEAN": pa.bool_(), "BYTE_ARRAY": pa.binary(), "STRING": pa.string(),
            "INT96": pa.timestamp("ns")}[t]


def _value(cell, at, k):
    gt, v = cell
    if pa.types.is_decimal(at):  # the canon holds nil for every DECIMAL (Restore's default): the file needs SOME value there
        return decimal.Decimal(k % 90 + 1).scaleb(-at.scale) if gt == "nil" else v
    if gt == "nil":
        return None
    if pa.types.is_timestamp(at):  # INT96: the decimal string of (Julian day << 64 | nanoseconds of the day)
        x = int(v)
        return ((x >> 64) - 2440588) * 86400 * 10**9 + (x & (2**64 - 1))
    if pa.types.is_binary(at) or pa.types.is_fixed_size_binary(at):
        return text_bytes(v)
    return v


def build(name: str, total_rows: int = 0):
    """(file bytes, file name as the canon's __file_name has it).  The canon's rows sit at their Counter positions (1-based);
    the rows in between repeat the last canon row in front of them (rows past the last one cycle through the canon rows)."""
    g = golden()[name]
    names = g["names"][2:]
    sch = {c[0]: c for c in g["table_schema"]}
    opts = dict(OPTIONS.get(name, dict(compression="NONE")))
    required = opts.pop("required", False)
    delta = opts.pop("delta", False)
    counters = g["counters"]
    n = max(max(counters), total_rows)
    at_row = {c - 1: r for c, r in zip(counters, g["rows"])}
    fields, arrays = [], []
    for ci, cn in enumerate(names):
        ot = sch[cn][2]
        if ot in ("parquet:group", "parquet:LIST"):
            raise ValueError("%s: nested column %s — see build_nested" % (name, cn))
        at = _arrow_type(ot)
        vals, last = [], None
        for r in range(n):
            row = at_row.get(r)
            if row is None and r > max(at_row):
                row = g["rows"][r % len(g["rows"])]
            if row is not None:
                last = row
            vals.append(_value(last[2 + ci], at, r))
        if pa.types.is_timestamp(at):
            arr = pa.array(vals, pa.int64()).cast(at)
        else:
            arr = pa.array(vals, at)
        fields.append(pa.field(cn, at, nullable=not required))
        arrays.append(arr)
    if delta:
        opts["column_encoding"] = {f.name: ("DELTA_BYTE_ARRAY" if pa.types.is_string(f.type) else "DELTA_BINARY_PACKED") for f in fields}
    t = pa.Table.from_arrays(arrays, schema=pa.schema(fields))
    buf = io.BytesIO()
    pq.write_table(t, buf, use_deprecated_int96_timestamps=any(pa.types.is_timestamp(f.type) for f in fields), store_schema=False, **opts)
    return buf.getvalue(), "data/" + name


def build_nested(name: str):
    """The ten files with nested columns, in the shape their canon rows show (lists, maps, structs next to flat leaves): enough for
    the device to meet a group where the reference builds an `any` tree."""
    ints = pa.array([[1, 2, 3], None, [4]], pa.list_(pa.int64()))
    strs = pa.array([["abc", "efg"], [], None], pa.list_(pa.string()))
    struct = pa.array([{"b_c_int": None}, {"b_c_int": 1}, None], pa.struct([("b_c_int", pa.int32())]))
    maps = pa.array([[("a", 1)], [], None], pa.map_(pa.string(), pa.int32()))
    flat_i, flat_d = pa.array([1, 2, 3], pa.int32()), pa.array([1.0, 2.0, 3.0], pa.float64())
    tables = {
        "datapage_v2.snappy.parquet": ({"a": pa.array(["abc", "abc", "abc"]), "b": flat_i, "c": flat_d, "d": pa.array([True, False, True]), "e": ints}, dict(compression="SNAPPY", data_page_version="2.0")),
        "list_columns.parquet": ({"int64_list": ints, "utf8_list": strs}, {}),
        "nested_lists.snappy.parquet": ({"a": pa.array([[[["a", "b"], ["c"]], [None, ["d"]]], None, [[["e"]]]], pa.list_(pa.list_(pa.list_(pa.string())))), "b": flat_i}, dict(compression="SNAPPY")),
        "nested_maps.snappy.parquet": ({"a": pa.array([[("a", [(1, True)])], [], None], pa.map_(pa.string(), pa.map_(pa.int32(), pa.bool_()))), "b": flat_i, "c": flat_d}, dict(compression="SNAPPY")),
        "nested_structs.rust.parquet": ({"roll_num": pa.array([{"count": 495, "max": 1, "mean": 0.5}], pa.struct([("count", pa.int64()), ("max", pa.int64()), ("mean", pa.float64())]))}, {}),
        "nonnullable.impala.parquet": ({"ID": pa.array([8], pa.int64()), "Int_Array": pa.array([[-1]], pa.list_(pa.int32())), "Int_Map": pa.array([[("k1", -1)]], pa.map_(pa.string(), pa.int32()))}, {}),
        "null_list.parquet": ({"emptylist": pa.array([None], pa.list_(pa.int64()))}, {}),
        "nullable.impala.parquet": ({"id": pa.array([1, 2, 3], pa.int64()), "int_array": ints, "int_map": maps, "nested_struct": struct}, {}),
        "nulls.snappy.parquet": ({"b_struct": struct}, dict(compression="SNAPPY")),
        "repeated_no_annotation.parquet": ({"id": flat_i, "phoneNumbers": pa.array([None, None, {"phone": [{"number": 5555555555, "kind": None}]}],
                                                                              pa.struct([("phone", pa.list_(pa.struct([("number", pa.int64()), ("kind", pa.string())])))]))}, {}),
    }
    cols, opts = tables[name]
    buf = io.BytesIO()
    pq.write_table(pa.table(cols), buf, store_schema=False, **opts)
    return buf.getvalue(), "data/" + name


def go_time(days: int):
    return datetime.date(1970, 1, 1) + datetime.timedelta(days=days)
