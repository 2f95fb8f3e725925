"""Pins the oracle (oracle/) to the reference's own golden vectors
(tests/golden/, built by tools/extract_golden.py from the reference's canon
files and test tables).  CPU only."""
import numpy as np
import pytest

from transferia_amd import abi
from util import golden, item_to_batch, json_value


def test_hmac_and_crc_primitives(oracle):
    import hashlib
    import hmac
    import zlib
    rng = np.random.default_rng(1)
    for n in [0, 1, 31, 32, 55, 56, 63, 64, 65, 119, 120, 200, 1000]:
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for klen in [0, 5, 36, 64, 65, 200]:
            key = rng.integers(0, 256, klen, dtype=np.uint8).tobytes()
            assert oracle.hmac_sha256_hex(key, msg) == hmac.new(key, msg, hashlib.sha256).hexdigest()
        assert oracle.crc32(msg) == zlib.crc32(msg)


def test_mask_canon(oracle):
    g = golden("mask.json")
    t = oracle.Transformer("mask_field", g["config"])
    for case in g["cases"]:
        b, schema = item_to_batch(case["item"])
        assert t.suitable(case["item"]["ns"], case["item"]["table"], schema)
        r = t.apply(b, schema)
        assert len(r.errors) == case["expect_errors"]
        got = [json_value(c, 0) for c in r.batch.cols]
        assert got == case["expect_values"]
        exp_schema = [[c[0], c[1], c[2], c[3]] for c in case["expect_schema"]]
        assert [[c.name, c.dtype, c.key, c.original_type] for c in r.schema.cols] == exp_schema
        rs = t.result_schema(schema)
        assert [[c.name, c.dtype] for c in rs.cols] == [[c[0], c[1]] for c in case["expect_schema"]]


def test_sharder_canon(oracle):
    g = golden("sharder.json")
    for case in g["cases"]:
        t = oracle.Transformer("sharder_transformer", case["config"])
        b, schema = item_to_batch(case["item"])
        assert t.suitable(case["item"]["ns"], case["item"]["table"], schema) == case["suitable"]
        if case["suitable"]:
            r = t.apply(b, schema)
            assert str(int(r.batch.part_id[0])) == case["expect_part"]


def test_to_string_canon(oracle):
    g = golden("to_string.json")
    for case in g["cases"]:
        t = oracle.Transformer("convert_to_string", case["config"])
        b, schema = item_to_batch(case["item"])
        assert t.suitable(case["item"]["ns"], case["item"]["table"], schema) == case["suitable"]
        if not case["suitable"]:
            continue
        r = t.apply(b, schema)
        got = [json_value(c, 0) for c in r.batch.cols]
        # canon renders time.Duration as its integer nanoseconds and float32 via float64 JSON
        exp = case["expect_values"]
        for gv, ev in zip(got, exp):
            if isinstance(ev, float):
                assert abs(gv - ev) < 1e-4
            else:
                assert gv == ev
        assert [c.dtype for c in r.schema.cols] == case["expect_types"]


def test_serialize_to_string_kats(oracle):
    g = golden("to_string.json")
    # route each KAT through convert_to_string on a one-column table
    for val, dtype, expected in g["serialize_kats"]:
        schema = abi.Schema.of([["c", dtype, False]])
        b = abi.batch_from_rows(schema, ["c"], [[val]], "db", "t")
        t = oracle.Transformer("convert_to_string", {})
        r = t.apply(b, schema)
        assert r.batch.cols[0].get_bytes(0).decode("utf-8") == expected, (val, dtype)


def test_to_datetime_canon(oracle):
    g = golden("to_datetime.json")
    for case in g["cases"]:
        t = oracle.Transformer("convert_to_datetime", case["config"])
        b, schema = item_to_batch(case["item"])
        assert t.suitable(case["item"]["ns"], case["item"]["table"], schema) == case["suitable"]
        if not case["suitable"]:
            continue
        r = t.apply(b, schema)
        assert [json_value(c, 0) for c in r.batch.cols] == case["expect_values"]
        assert [c.dtype for c in r.schema.cols] == case["expect_types"]


def test_filter_rows_table(oracle):
    g = golden("filter_rows.json")
    for bad in g["unparseable"]:
        with pytest.raises(ValueError):
            oracle.Transformer("filter_rows", {"filter": bad})
    for case in g["cases"]:
        t = oracle.Transformer("filter_rows", case["config"])
        b, schema = item_to_batch(case)
        assert t.suitable(case["ns"], case["table"], schema) == case["suitable"], case["name"]
        r = t.apply(b, schema)
        exp = [[abi.norm_value(v) for v in row] for row in case["expect_rows"]]
        assert abi.batch_rows(r.batch) == exp, case["name"]
        assert len(r.errors) == case["expect_errors"], case["name"]
        if "expect_error_code" in case:
            assert {abi.ROWERR[e[1]] for e in r.errors} == {case["expect_error_code"]}, case["name"]


def test_csv_reader_cases(oracle):
    g = golden("csv_reader.json")
    for case in g["cases"]:
        o = dict(case["opts"])
        opts = abi.csv_options(**o)
        data = case["input_latin1"].encode("latin-1") if "input_latin1" in case else case["input"].encode("utf-8")
        lines, errs, consumed = oracle.csv_read_all(opts, data)
        if "expect_error" in case:
            assert abi.ROWERR_ID[case["expect_error"]] in errs, case["name"]
            continue
        assert not any(errs), case["name"]
        if "expect" in case:
            assert [[f.decode("utf-8") for f in ln] for ln in lines] == case["expect"], case["name"]
        if "expect_latin1" in case:
            assert [[f.decode("latin-1") for f in ln] for ln in lines] == case["expect_latin1"], case["name"]
        if "expect_nlines" in case:
            assert len(lines) == case["expect_nlines"], case["name"]
        if "expect_len_line1" in case:
            assert len(lines[1]) == case["expect_len_line1"]
        if "expect_field" in case:
            li, fi, val = case["expect_field"]
            assert lines[li][fi].decode("utf-8") == val
