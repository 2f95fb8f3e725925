"""Parity of the HIP CSV ingest (tfgpu_csv_parse: csv.Reader + constructCI +
Strictify on device) with the oracle.  Through the C ABI; needs an MI355X."""
import numpy as np
import pytest

from transferia_amd import abi, workload
from util import golden
from test_gpu_transformers import assert_batches_equal
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def compare(tf, oracle, opts_kw, schema, data: bytes, ctx="", max_fallback=0):
    opts = abi.csv_options(**opts_kw)
    ref = oracle.csv_parse(opts, schema, data, "ns", "t")
    db, consumed, errs = tf.csv_parse(opts, schema, data, max_errors=1 << 18)
    out = db.download()
    assert consumed == ref.consumed, ctx
    gpu_err = sorted((e[0], e[1]) for e in errs)
    ref_err = sorted((e[0], abi.ROWERR[e[1]]) for e in ref.errors)
    hf = {e[0] for e in gpu_err if e[1] == "HOST_FALLBACK"}
    assert len(hf) <= max_fallback, (ctx, "rows handed back to the host path", len(hf), max_fallback)
    if hf:
        # rows the device hands back to the host path: everything else must still agree
        gpu_err = [e for e in gpu_err if e[0] not in hf]
        ref_err = [e for e in ref_err if e[0] not in hf]
        keep = ~np.isin(ref.batch.src_row, list(hf))
        assert np.array_equal(out.src_row if out.src_row is not None else np.arange(out.nrows), ref.batch.src_row[keep]), ctx
    else:
        if ref.batch.nrows == 0:
            assert out.nrows == 0, ctx
        else:
            assert_batches_equal(out, ref.batch, ctx)
            src = out.src_row if out.src_row is not None else np.arange(out.nrows, dtype=np.int32)
            assert np.array_equal(src, ref.batch.src_row), ctx
    assert gpu_err == ref_err, ctx
    return out, errs


S4 = abi.Schema.of([["a", "int32", True, "0"], ["b", "utf8", False, "1"], ["c", "int64", False, "2"], ["d", "utf8", False, "3"]])


def test_reference_reader_cases(tf, oracle):
    """pkg/csv/reader_test.go inputs through the typed path (all-utf8 schema)."""
    g = golden("csv_reader.json")
    for case in g["cases"]:
        o = dict(case["opts"])
        data = case["input_latin1"].encode("latin-1") if "input_latin1" in case else case["input"].encode("utf-8")
        nf = max((len(r) for r in case.get("expect", [[0] * 8])), default=8)
        schema = abi.Schema.of([[f"f{i}", "utf8", False, str(i)] for i in range(nf)])
        o["include_missing_columns"] = 1
        compare(tf, oracle, o, schema, data, case["name"])


def test_unsupported_options_fail_loudly(tf):
    """What is left outside the device: an Encoding that moves the line feed, '\\n' as the escape character of multi-line rows."""
    for kw in [dict(newlines_in_value=1, escape_char="\n"), dict(encoding="IBM Code Page 037")]:
        with pytest.raises(tf.TfgpuError) as ei:
            tf.csv_parse(abi.csv_options(**kw), S4, b"1,a,2,b\n")
        assert ei.value.code == tf.ERR_UNSUPPORTED


def test_newlines_in_value(tf, oracle):
    """NewlinesInValue (readMultiline, reader.go:110-137): a row runs until its quotes are complete, empty physical lines are
    skipped.  The reference's own case is in test_reference_reader_cases; here: shapes and a random sweep against the oracle."""
    kw = dict(newlines_in_value=1)
    data = (b'1,"a\nb",2,"c"\n' b'\n' b'3,d,4,"e\n\tf  "\n' b'5,"g ""h""\ni",6,j\n' b'7,"k\\"\nl",8,m\n' b'\n\n' b'9,n,10,"open\nnever closed\n11,o,12,p\n')
    out, errs = compare(tf, oracle, kw, S4, data, "multiline shapes")
    assert out.nrows == 4 and out.col("b").get_bytes(0) == b"a\nb"
    compare(tf, oracle, dict(newlines_in_value=1, escape_char=0), S4, data, "multiline no escape", max_fallback=1)  # without the escape rule row 7 swallows the empty lines
    compare(tf, oracle, dict(newlines_in_value=1, skip_rows=2), S4, data, "multiline skip")
    compare(tf, oracle, kw, S4, b'1,"x\n\ny",2,z\n3,a,4,b\n', "blank line inside a value", max_fallback=1)
    rng = np.random.default_rng(31)
    alphabet = np.frombuffer(b'0123456789,,,""\\ \n\nab\r', dtype=np.uint8)
    for seed in range(4):
        blob = rng.choice(alphabet, 60_000).tobytes()
        schema = abi.Schema.of([["a", "utf8", False, "0"], ["b", "utf8", False, "1"], ["c", "int16", False, "2"]])
        compare(tf, oracle, dict(newlines_in_value=1, include_missing_columns=1), schema, blob, "multiline random %d" % seed, max_fallback=400)


EDGE_LINES = [
    b"1,a,2,b\n", b" 1 , a , 2 , b \n", b"1,\"a,b\",2,\"c\"\"d\"\n", b"\n", b"\r\n", b"x\n", b"1\n", b",,,\n", b"1,a,2\n",
    b"1,a,2,b,extra,more\n", b"08,a,2,b\n", b"0x1F,a,0b101,b\n", b"1_000,a,0x_1,b\n", b"12.00,a,5.0,b\n", b"12.5,a,5,b\n", b"-0,a,+7,b\n",
    b"2147483647,a,9223372036854775807,b\n", b"2147483648,a,1,b\n", b"-2147483649,a,1,b\n", b"1,a,9223372036854775808,b\n",
    b"1,a,-9223372036854775808,b\n", b"1,a,-9223372036854775809,b\n", b"\"1\",\"a\",\"2\",\"b\"\n", b"\"1,a,2,b\n", b"1,a\\,2,b\n",
    b"1,\"a\\\",2,b\"\n", b"1,\"a\\\\\",2,b\n", b"1,\",2,b\n", b"1,\"\",2,\"\"\"\"\n", b"1,\"\"\"\",2,b\n", b"1,\xc2\xa0a\xe2\x80\x83,2,\xe3\x80\x80b\n",
    b"1,\xd0\xb9\xd1\x86\xd1\x83,2,\xf0\x9f\x98\x80\n", b"1,a\tb,2,\x0bc\x0c\n", b"1,a,2,b\r\n", b"  \n", b"1,a,2,b,\n", b"0,a,0o17,b\n",
    b"1,a,0X1f,B\n", b"1,'a',2,'b'\n", b"1,a,__1,b\n", b"1,a,1__0,b\n", b"1,a,0_7,b\n", b"+,a,1,b\n", b"1,a,.0,b\n", b"1.,a,1,b\n", b"1,a,00.000,b\n",
]


@pytest.mark.parametrize("kw", [dict(), dict(include_missing_columns=1), dict(delimiter=";"), dict(escape_char=0),
                                dict(double_quote=0), dict(quote_char="'"), dict(strings_can_be_null=1, null_values=["", "a", "1"]),
                                dict(quoted_strings_can_be_null=1, null_values=["a", "NULL", "2"])])
def test_edge_lines(tf, oracle, kw):
    data = b"".join(EDGE_LINES)
    if kw.get("delimiter") == ";":
        data = data.replace(b",", b";")
    compare(tf, oracle, kw, S4, data + b"trailing,no,newline", str(kw))
    # each line on its own too (first/last line handling)
    for ln in EDGE_LINES[:12]:
        compare(tf, oracle, kw, S4, ln, str(kw) + repr(ln))
    compare(tf, oracle, kw, S4, b"", "empty")
    compare(tf, oracle, kw, S4, b"no newline at all", "no-nl")


def test_typed_columns(tf, oracle):
    schema = abi.Schema.of([["i8", "int8", False, "0"], ["u8", "uint8", False, "1"], ["u64", "uint64", False, "2"], ["b", "boolean", False, "3"],
                            ["d", "date", False, "4"], ["dt", "datetime", False, "5"], ["ts", "timestamp", False, "6"], ["f", "double", False, "7"],
                            ["by", "string", False, "8"], ["any", "any", False, "9"], ["dup", "utf8", False, "0"], ["neg", "int16", False, "-1"]])
    lines = [
        b"1,2,3,true,2013-07-15,2013-07-15 10:47:34,1374100000,1.5,bytes,x\n",
        b"-128,255,18446744073709551615,F,0001-01-01,2013-07-15T10:47:34Z,2013-07-15 10:47:34.123456789,-1e5,,\n",
        b"127,0,0,1,9999-12-31,2013-07-15T10:47:34+03:00,-5,inf,\"q\"\"q\",\"a,b\"\n",
        b"128,1,1,t,2013-07-15,2013-07-15,0,nan,a,b\n", b"1,256,1,t,2013-07-15,2013-07-15,0,1,a,b\n", b"1,-1,1,t,2013-07-15,2013-07-15,0,1,a,b\n",
        b"1,1,-1,t,2013-07-15,2013-07-15,0,1,a,b\n", b"1,1,1,yes,2013-07-15,2013-07-15,0,1,a,b\n", b"1,1,1,TRUE,2013-02-30,2013-07-15,0,1,a,b\n",
        b"1,1,1,False,2013-13-01,2013-07-15,0,1,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15 24:00:00,0,1,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15 9:05:06,0,1,a,b\n",
        b"1,1,1,0,2013-07-15,15 Jul 2013,0,1,a,b\n", b"1,1,1,0,13-07-15,2013-07-15,0,1,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15,x,1,a,b\n",
        b"1,1,1,0,2013-07-15,2013-07-15,0,1.,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15,0,.5,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15,0,+5,a,b\n",
        b"1,1,1,0,2013-07-15,2013-07-15,0,1e,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15,0,abc,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15,0,0x10,a,b\n",
        b"1,1,1,0,2013-07-15,2013-07-15T10:47:34.5,2013-07-15T00:00:00.000000001Z,1,a,b\n", b"1,1,1,0,2013-07-15,2013-07-15 10:47:34 +0300,0,1,a,b\n",
    ]
    for kw in [dict(), dict(include_missing_columns=1), dict(true_values=["yes"], false_values=["t"], strings_can_be_null=1, null_values=["TRUE"])]:
        compare(tf, oracle, kw, schema, b"".join(lines), "typed " + str(kw))


@pytest.mark.parametrize("n", [1, 257, 5000])
def test_hits_small(tf, oracle, n):
    schema = workload.hits_schema()
    data = workload.hits_csv(n)
    out, errs = compare(tf, oracle, dict(skip_rows=1), schema, data, f"hits {n}")
    assert out.nrows == n and not errs
    # a chunk cut in the middle of a line leaves the tail unconsumed
    compare(tf, oracle, dict(skip_rows=1), schema, data[: len(data) - 7], f"hits-cut {n}")


@pytest.mark.parametrize("seed", range(6))
def test_random_bytes(tf, oracle, seed):
    rng = np.random.default_rng(SEED0 + (seed))
    alphabet = np.frombuffer(b"0123456789,,,,\"\"\\ \t\n\n\nabc-+._xX\xc2\xa0\r'", dtype=np.uint8)
    data = rng.choice(alphabet, 200_000).tobytes()
    schema = abi.Schema.of([["a", "int32", True, "0"], ["b", "utf8", False, "1"], ["c", "int16", False, "2"], ["d", "string", False, "4"],
                            ["e", "double", False, "3"], ["g", "timestamp", False, "2"]])
    for kw in [dict(include_missing_columns=1), dict(), dict(include_missing_columns=1, escape_char=0), dict(include_missing_columns=1, double_quote=0)]:
        compare(tf, oracle, kw, schema, data, f"random {seed} {kw}")


def test_tile_path_shapes(tf, oracle):
    """Inputs that stress the tile path's windows: lines longer than the LDS look-behind (per-row
    hand-off), lines spanning several tiles, more fields / lines per tile than one pass indexes."""
    rng = np.random.default_rng(SEED0 + (11))
    kw = dict(include_missing_columns=1)

    def row(n_fields, width):
        return b",".join(b"%d" % rng.integers(0, 10 ** width) for _ in range(n_fields)) + b"\n"

    normal = b"".join(b"%d,s%d,%d,\"q,%d\"\n" % (i, i, i * 7, i) for i in range(3000))
    long9k = b"5,\"" + b"x" * 9000 + b"\",6,y\n"            # longer than the 8 KiB look-behind
    long40k = b"7,\"" + b"ab" * 20000 + b"\",8,\"z\"\"z\"\n"  # spans three 16 KiB tiles
    wide = b"9," + b"," * 7000 + b"\n"                        # more fields than one pass indexes
    tiny = b"1\n" * 30000                                     # thousands of lines per tile
    empties = b",,,\n" * 20000                                # field ends at every byte
    blank = b"\n" * 20000                                     # nil lines only
    for name, data in [
        ("long-first", long9k + normal), ("long-mid", normal + long9k + normal + long40k + normal),
        ("long-only", long40k), ("long-last", normal + long40k), ("wide", normal + wide + normal + wide + wide + normal),
        ("tiny", tiny), ("empties", empties + normal), ("blank", blank + normal + blank),
        ("mix", tiny + long9k + empties + wide + normal + blank + long40k + b"tail without newline"),
    ]:
        compare(tf, oracle, kw, S4, data, name)
        compare(tf, oracle, dict(), S4, data, name + " strict")
    # a header to skip, sitting in a long line
    compare(tf, oracle, dict(skip_rows=1, include_missing_columns=1), S4, long9k + normal, "skip-long")
    compare(tf, oracle, dict(skip_rows=3, include_missing_columns=1), S4, normal, "skip3")


def test_single_pass_form_and_its_fallbacks():
    """The single-pass CSV form is opt-in (TFGPU_CSV_SPEC=1, read once per process): its own cases and the tile-shape stress run
    in a process that has it on."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TFGPU_CSV_SPEC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "single_pass_cases or tile_path_shapes or hits_small or random_bytes"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]


def test_single_pass_cases(tf, oracle):
    """A lane that has parsed a chunk of this shape sizes the next chunk's buffers from it — by default without waiting for
    csv_count_newlines' total (the sized-ahead form), with TFGPU_CSV_SPEC=1 letting csv_parse_regular count the lines itself (no
    csv_count_newlines pass); a chunk with more lines than expected, a header longer than the chunk's lines, or
    a last line that spans the last tiles is parsed again the two-pass way.  Every answer must be the oracle's."""
    rng = np.random.default_rng(SEED0 + 23)
    kw = dict(include_missing_columns=1)

    def rows(n, width):
        return b"".join(b"%d,s%d,%d,\"q,%d\"\n" % (i, rng.integers(0, 10 ** width), i * 7, i) for i in range(n))

    def kernels():
        return {name for name, _n, _ms in tf.prof_get()}

    a = rows(6000, 6)
    compare(tf, oracle, kw, S4, a, "first chunk (count pass)")
    tf.prof_enable(True)
    try:
        tf.prof_reset()
        compare(tf, oracle, kw, S4, a, "same chunk again")
        k = kernels()
        spec = _os.environ.get("TFGPU_CSV_SPEC") == "1"
        ahead = not spec and _os.environ.get("TFGPU_CSV_AHEAD") != "0"   # the default: two passes, buffers sized ahead of the count

        def launches(name):
            return sum(n for nm, n, _ms in tf.prof_get() if nm == name)
        if spec:
            assert "csv_parse_regular" in k and "csv_count_newlines" not in k, k
        tf.prof_reset()
        compare(tf, oracle, kw, S4, rows(5000, 9), "longer lines: fewer than expected")
        if spec:
            assert "csv_count_newlines" not in kernels()
        if ahead:
            assert launches("csv_count_newlines") == 1 and launches("csv_parse_regular") == 1
        tf.prof_reset()
        compare(tf, oracle, kw, S4, rows(20000, 1), "shorter lines: more than the buffers hold")
        if spec:
            assert "csv_count_newlines" in kernels()  # parsed again
        if ahead:
            assert launches("csv_count_newlines") == 2 and launches("csv_parse_regular") == 2   # parsed again, the count read back first
    finally:
        tf.prof_enable(False)
    compare(tf, oracle, kw, S4, rows(20000, 1), "the same density again")
    compare(tf, oracle, kw, S4, rows(9000, 1) + b"7,\"" + b"ab" * 40000 + b"\",8,z\n", "a last line across the last three tiles")
    compare(tf, oracle, kw, S4, rows(9000, 1) + b"7,\"" + b"ab" * 40000, "an unfinished last line across the last three tiles")
    compare(tf, oracle, dict(skip_rows=3, include_missing_columns=1), S4, b"h\n" + b"x" * 70000 + b"\n", "fewer lines than the header skips")
    compare(tf, oracle, dict(skip_rows=2, include_missing_columns=1), S4, rows(9000, 2), "skip 2")
    compare(tf, oracle, kw, S4, b"\n" * 70000, "nil lines only")
    compare(tf, oracle, kw, S4, rows(3000, 3) + b"1\n" * 40000 + rows(3000, 3), "a dense stretch in the middle")


def test_unmapped_fields_are_still_sanitized(tf, oracle):
    """sanitizeElement runs on every field of a line, also those no column reads (reader.go:240-266)."""
    schema = abi.Schema.of([["a", "int32", True, "0"], ["d", "utf8", False, "3"]])
    lines = [b"1,x,y,z\n", b"2,\",y,z\n", b"3,x,y,z,\"\n", b"4,x,\"a\"\"b\",z\n", b"5\n", b"6,x,y\n", b"7,x,y,z,w,v,\"\n"] * 50
    for kw in [dict(), dict(include_missing_columns=1), dict(double_quote=0)]:
        compare(tf, oracle, kw, schema, b"".join(lines), "unmapped " + str(kw))


def test_row_path_equals_tile_path():
    """TFGPU_CSV_ROWPATH=1 routes whole chunks through the per-row kernel; both must match the oracle."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TFGPU_CSV_ROWPATH="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_csv.py", "-m", "gpu", "-q", "-x", "-k",
                        "edge_lines or typed_columns or random_bytes or tile_path_shapes or unmapped"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_column_lane_form_equals_item_form():
    """TFGPU_CSV_COL_LANES=2 (pieces of 8 columns): csv_parse_cols, the column-lane cell phase of the regular-tile kernel —
    slower than the item form on the MI355X and off by default, kept with its measurements; it must pass the same cases."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TFGPU_CSV_COL_LANES="2", TFGPU_CSV_COL_PIECE="8")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_csv.py", "-m", "gpu", "-q", "-x", "-k",
                        "edge_lines or typed_columns or random_bytes or tile_path_shapes or unmapped or hits_small"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_lanes_overlap_and_agree(tf, oracle):
    """Two host threads on two device lanes (tfgpu_lane_use) parse + transform concurrently; each result is
    identical to the single-lane result (the parsequeue's parallel workers, parsequeue.go:57-154)."""
    import threading
    schema = workload.hits_schema()
    opts = workload.hits_csv_options()
    chain = [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["clientip"]}), ("filter_rows", {"filter": "eventdate >= 2013-07-15"})]
    datas = [workload.hits_csv(4000), workload.hits_csv(2500)]
    def run(data):
        db, consumed, errs = tf.csv_parse(opts, schema, data)
        res = tf.apply_chain([tf.Transformer(t, c) for t, c in chain], db)
        return res.transformed.download()
    base = [run(d) for d in datas]
    outs, errs = {}, []
    def worker(k):
        try:
            tf.lane_use(k + 1)
            for _ in range(5):
                outs[k] = run(datas[k])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for k in range(2):
        assert_batches_equal(outs[k], base[k], "lane %d" % k)
    tf.lane_use(0)


def test_late_materialisation(tf, oracle):
    """Text columns stay (offsets, position in the source text) after the tile path; row compaction packs the kept
    cells straight from the text.  Checked against the oracle's parse + filter, against the same chain over a batch
    that was packed first, with the source buffer freed / overwritten while the batch is still lazy."""
    schema = abi.Schema.of([["a", "int32", True, "0"], ["b", "utf8", False, "1"], ["c", "double", False, "2"], ["d", "utf8", False, "3"], ["e", "utf8", False, "7"]])
    rng = np.random.default_rng(SEED0 + (3))
    words = ['plain', '"quoted, text"', '"say ""hi"" twice ""x"""', '', '""', '"a longer cell with many bytes to cross several eight-byte words, é and more"', 'x' * 70, '"""";"""']
    lines = []
    for i in range(5000):
        lines.append("%d,%s,%s,%s" % (i, words[int(rng.integers(0, len(words)))], ["1.5", "", "2e3"][int(rng.integers(0, 3))], words[int(rng.integers(0, len(words)))]))
    data = ("\n".join(lines) + "\n").encode("utf-8")
    opts = abi.csv_options(include_missing_columns=1)
    chain = [("filter_rows", {"filter": "a > 10 AND a < 4000"}), ("filter_rows", {"filter": "a != 77"})]
    ref = oracle.csv_parse(opts, schema, data, "ns", "t")
    want = oracle.apply_chain([oracle.Transformer(t, c) for t, c in chain], ref.batch, ref.schema).batch
    plans = [tf.Transformer(t, c) for t, c in chain]
    # lazy: parse → filter → download
    db, _, errs = tf.csv_parse(opts, schema, data)
    assert sorted((e[0], e[1]) for e in errs) == sorted((e[0], abi.ROWERR[e[1]]) for e in ref.errors) and len(errs) > 1000  # "" is no double: those lines are dropped
    got = tf.apply_chain(plans, db).transformed.download()
    assert_batches_equal(got, want, "lazy gather")
    # packed first (download materialises), then the same chain
    db2, _, _ = tf.csv_parse(opts, schema, data)
    full = db2.download()
    assert_batches_equal(full, ref.batch, "materialised")
    assert_batches_equal(tf.apply_chain(plans, db2).transformed.download(), want, "packed gather")
    # a predicate on a text column packs just what it reads; the result is the same
    sfilter = [tf.Transformer("filter_rows", {"filter": "b = \"plain\""})]
    db3, _, _ = tf.csv_parse(opts, schema, data)
    want3 = oracle.apply_chain([oracle.Transformer("filter_rows", {"filter": "b = \"plain\""})], ref.batch, ref.schema).batch
    assert_batches_equal(tf.apply_chain(sfilter, db3).transformed.download(), want3, "string predicate")
    # the source text outlives its handle: free / overwrite the device buffer while the batch is still unpacked
    buf = tf.DeviceBuffer.upload(data)
    db4, _, _ = tf.csv_parse(opts, schema, buf)
    buf.write(0, np.zeros(4096, np.uint8), 4096)  # copy-on-write: the batch keeps reading the old text
    buf.free()
    junk = [tf.DeviceBuffer.upload(b"z" * len(data)) for _ in range(3)]  # recycle the block cache
    assert_batches_equal(tf.apply_chain(plans, db4).transformed.download(), want, "source freed")
    assert_batches_equal(db4.download(), ref.batch, "source freed, full")
    for j in junk:
        j.free()


def test_quoting_disabled_and_encoding(tf, oracle):
    """QuoteChar = 0 (reader.go:185-187: a '"' anywhere in the line is errQuotingDisabled, otherwise quotes mean nothing) and
    Encoding (reader.go:171-179: every line through the charmap decoder), against the oracle; the reference's own cases for
    both are in test_reference_reader_cases."""
    lines = b"1,a,2,b\n3,'x,4,y\n5,\"q\",6,z\n7,\\,8,w\n\n9,tab\there,10,\xc2\xa0v\n"
    compare(tf, oracle, dict(quote_char=0), S4, lines, "quote0")
    compare(tf, oracle, dict(quote_char=0, escape_char=0), S4, lines + b"11,no newline", "quote0-esc0")
    rng = np.random.default_rng(7)
    for enc in ["ISO 8859-1", "Windows 1251", "KOI8-R", "IBM Code Page 437", "Macintosh", "Windows 1252", "ISO 8859-7"]:
        rows = []
        for i in range(3000):
            txt = bytes(int(x) for x in rng.integers(0x20, 0x100, size=int(rng.integers(0, 24))) if x not in (0x22, 0x2C, 0x5C))
            rows.append(b"%d,%s,%d,\"%s\"" % (i, txt, i * 7, txt[::-1]))
        data = b"\n".join(rows) + b"\n" + b"tail without newline \xe4"
        out, errs = compare(tf, oracle, dict(encoding=enc), S4, data, enc)
        assert out.nrows == 3000 and not errs
    with pytest.raises(tf.TfgpuError) as ei:  # EBCDIC moves the line feed: refused, not mis-split
        tf.csv_parse(abi.csv_options(encoding="IBM Code Page 037"), S4, b"1,a,2,b\n")
    assert ei.value.code == tf.ERR_UNSUPPORTED


def test_typed_goldens_of_the_reference(tf):
    """The reference's own expectations for the typed half (tests/golden/csv_typed.json: TestConstructCI incl. system columns,
    the s3 canon rows) straight against the device — no oracle in between."""
    from test_oracle_golden import _csv_line
    g = golden("csv_typed.json")

    def typed(col, i):
        k, v = col.pyvalue(i)
        if k in ("string", "bytes", "jsonnum", "json"):
            return [k, v.decode("utf-8") if isinstance(v, bytes) else v]
        return [k, v]
    for case in g["construct_ci"]:
        schema = abi.Schema.of(case["schema"])
        db, consumed, errs = tf.csv_parse(abi.csv_options(**case["opts"]), schema, _csv_line(case["row"]))
        if "expect_error" in case:
            assert db.nrows == 0 and [e[1] for e in errs] == [case["expect_error"]], case["name"]
            continue
        out = db.download()
        assert not errs and out.nrows == 1, case["name"]
        assert [abi.norm_value(typed(c, 0)) for c in out.cols] == [abi.norm_value(v) for v in case["expect"]], case["name"]
    for case in g["s3_canon"]:
        schema = abi.Schema.of(case["schema"])
        user = [(i, c) for i, c in enumerate(case["schema"]) if not c[0].startswith("__")]
        width = 1 + max([int(c[3]) for _, c in user if int(c[3]) < 20], default=0)
        text = b""
        for row in case["expect_rows"]:
            cells = ["0"] * width
            for i, c in user:
                if int(c[3]) < width:
                    v = row[i][1]
                    cells[int(c[3])] = ("true" if v else "false") if isinstance(v, bool) else str(v)
            text += _csv_line(cells)
        db, consumed, errs = tf.csv_parse(abi.csv_options(**case["opts"]), schema, text)
        out = db.download()
        assert not errs and consumed == len(text) and out.nrows == len(case["expect_rows"])
        for r, exp in enumerate(case["expect_rows"]):
            assert [abi.norm_value(typed(c, r)) for c in out.cols] == [abi.norm_value(v) for v in exp], (case["name"], r)


LAYOUTS = ["2006-01-02", "02-Jan-2006", "January 2, 2006", "02/01/06 15:04", "2006-01-02T15:04:05.000Z07:00", "Mon, 02 Jan 2006 15:04:05 MST",
           "2006.01.02 3:04:05 PM", "Jan _2 2006", "20060102", "2006-002", "15:04:05.999999 2006-01-02 -0700"]


def test_user_timestamp_layouts_and_cast_layouts(tf, oracle):
    """TimestampParsers (reader_csv.go:405-415: time.Parse with each layout in turn) and, for what stays a string, cast's
    StringToDate list (strictify): the reference's TestParseDateValue values first, then random instants rendered in
    every layout (and damaged) against the oracle's restatement of time.Parse."""
    import datetime
    g = golden("csv_typed.json")
    for case in [c for c in g["corresponding_value"] if c["fn"] == "parseDateValue"]:
        schema = abi.Schema.of([["d", "date", False, "1"]])
        db, _, errs = tf.csv_parse(abi.csv_options(**case["opts"]), schema, ("x," + case["in"] + "\n").encode() if "," not in case["in"] else ('x,"' + case["in"] + '"\n').encode())
        out = db.download()
        if case["expect"][0] == "time":
            assert not errs and out.nrows == 1
            sec = int(out.cols[0].values[0])
            assert datetime.datetime.utcfromtimestamp(sec).strftime("%Y-%m-%dT%H:%M:%SZ") == case["expect"][1], case["in"]
        else:  # the value stays a string: StringToDate then refuses "2024/03/22" (strictify error)
            assert out.nrows == 0 and [e[1] for e in errs] == ["CAST"], case["in"]
    rng = np.random.default_rng(99)
    py = {"2006": "%Y", "01": "%m", "02": "%d", "15": "%H", "04": "%M", "05": "%S", "Jan": "%b", "January": "%B", "Mon": "%a", "06": "%y"}
    lines = []
    for i in range(4000):
        t = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=int(rng.integers(0, 4_000_000_000)), microseconds=int(rng.integers(0, 10**6)))
        forms = [t.strftime("%Y-%m-%d"), t.strftime("%d-%b-%Y"), t.strftime("%B ") + str(t.day) + t.strftime(", %Y"), t.strftime("%d/%m/%y %H:%M"),
                 t.strftime("%Y-%m-%dT%H:%M:%S.") + "%03d" % (t.microsecond // 1000) + ["Z", "+03:00", "-11:30"][i % 3], t.strftime("%a, %d %b %Y %H:%M:%S ") + ["UTC", "MSK", "GMT+3", "CEST"][i % 4],
                 t.strftime("%Y.%m.%d ") + str((t.hour % 12) or 12) + t.strftime(":%M:%S ") + ("PM" if t.hour >= 12 else "AM"), t.strftime("%b ") + "%2d" % t.day + t.strftime(" %Y"),
                 t.strftime("%Y%m%d"), t.strftime("%Y-%j"), t.strftime("%H:%M:%S.") + "%06d" % t.microsecond + t.strftime(" %Y-%m-%d ") + ["+0000", "-0730", "+1245"][i % 3],
                 t.strftime("%Y-%m-%d %H:%M:%S"), t.strftime("%d %b %Y"), t.strftime("%Y-%m-%dT%H:%M:%S") + ",5", str(int(t.timestamp())), t.strftime("%a %b ") + "%2d" % t.day + t.strftime(" %H:%M:%S %Y"),
                 t.strftime("%I:%M%p").lstrip("0"), t.strftime("%Y-%m-%d %H:%M:%S.%f +0300 MSK")]
        v = forms[i % len(forms)]
        if i % 7 == 3:  # damage: a character dropped or doubled, an out-of-range field
            k = int(rng.integers(0, len(v)))
            v = [v[:k] + v[k + 1:], v[:k] + v[k] + v[k:], v.replace("1", "9", 1)][i % 3]
        q = '"' + v + '"' if "," in v else v
        lines.append("%d,%s,%s,%s" % (i, q, q, q))
    data = ("\n".join(lines) + "\n").encode()
    schema = abi.Schema.of([["id", "int32", True, "0"], ["d", "date", False, "1"], ["dt", "datetime", False, "2"], ["ts", "timestamp", False, "3"]])
    for tp in [LAYOUTS, LAYOUTS[::-1], []]:
        out, errs = compare(tf, oracle, dict(timestamp_parsers=tp), schema, data, "layouts %d" % len(tp))
        assert out.nrows > 1500 and len(errs) > 300


def test_go_time_parse_golden(tf, oracle):
    """time.Parse corner cases the ADVICE review listed (space runs, signed-hour zone names, ChST / MeST, GMT before the
    capitals are counted, the lower-case guard of nextStdChunk, __2 / 002 day-of-year rules): the inputs of
    tests/golden/gotime.json (hand-derived from time/format.go; the oracle is pinned to it in test_oracle_golden.py) through
    the device's compiled layouts — a datetime cell under one user layout.  A layout without a year gets one in front
    (year 0 is outside what the CSV path stores), so the expectation is the oracle's time.Parse of the same pair."""
    g = golden("gotime.json")
    schema = abi.Schema.of([["id", "int32", True, "0"], ["ts", "datetime", False, "1"]])  # (TimestampParsers are tried on date / datetime cells: reader_csv.go:405-415)
    checked = 0
    for k, (layout, value, _expect, why) in enumerate(g["cases"]):
        if "2006" not in layout:
            layout, value = "2006 " + layout, "2021 " + value
        expect = oracle.time_parse(layout, value)
        cell = '"' + value + '"' if ("," in value or value != value.strip()) else value
        opts = abi.csv_options(timestamp_parsers=[layout])
        data = ("%d,%s\n" % (k, cell)).encode()
        db, _, errs = tf.csv_parse(opts, schema, data, max_errors=16)
        out = db.download()
        assert not [e for e in errs if e[1] == "HOST_FALLBACK"], (layout, value, "handed to the host")
        ref = oracle.csv_parse(opts, schema, data, "ns", "t")
        assert out.nrows == ref.batch.nrows and len(errs) == len(ref.errors), (layout, value, why, errs, ref.errors)
        if expect is not None and ref.batch.nrows == 1:
            c = out.col("ts")
            assert [int(c.values[0]), int(c.nanos[0]) if c.nanos is not None else 0] == [int(expect[0]), int(expect[1])], (layout, value, why)
            checked += 1
    assert checked >= 18


def test_decimal_point(tf, oracle):
    """DecimalPoint (reader_csv.go:363-378): the first occurrence becomes '.', kept only when strconv.ParseFloat takes the
    result.  The reference's TestParseFloatValue values, then a sweep against the oracle."""
    g = golden("csv_typed.json")
    for case in [c for c in g["corresponding_value"] if c["fn"] == "parseFloatValue"]:
        schema = abi.Schema.of([["f", "utf8" if case["expect"][1] == "abc" else "double", False, "1"]])
        cell = '"' + case["in"] + '"' if "," in case["in"] else case["in"]
        db, _, errs = tf.csv_parse(abi.csv_options(**case["opts"]), schema, ("x," + cell + "\n").encode())
        out = db.download()
        assert not errs and out.cols[0].get_bytes(0).decode() == case["expect"][1], case
    cells = ["1,5", "-0,25", "1,5e3", "1,", ",5", ",", "1,2,3", "abc", "1.5", "1e400", "1,e5", "inf", "-Infinity", "nan", "+nan", "0x1,8p3", "1_0,5", "", "12", "1,,5",
             "1;5", "1;;5", "٣,٥", "1,5 ", "+1,0E-2", "1,5,", "9" * 400 + ",5", "1,5e", "--1,5"]
    rows = []
    for i, c in enumerate(cells * 3):
        rows.append('%d,"%s","%s"' % (i, c, c))
    data = ("\n".join(rows) + "\n").encode()
    schema = abi.Schema.of([["id", "int32", True, "0"], ["f", "double", False, "1"], ["s", "utf8", False, "2"]])
    for dpv in [",", ";", ";;", ".", "5"]:
        compare(tf, oracle, dict(decimal_point=dpv), schema, data, "dp " + dpv, max_fallback=12)


def test_csv_splitter(tf, oracle):
    """tfgpu_csv_split_rows = csv.Splitter (pkg/csv/splitter.go:37-85): the reference's ConsumeRow cases, then random quoted text
    against the oracle's restatement of its three-state machine."""
    for case in golden("csv_splitter.json")["cases"]:
        data = case["input"].encode()
        ends = [int(x) for x in tf.csv_split_rows(data)]
        rows = [data[a:b].decode() for a, b in zip([0] + ends[:-1], ends)]
        assert rows == case["rows"] and data[ends[-1] if ends else 0:].decode() == case["eof_rest"], case["name"]
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b'ab,"""\n\n 1', dtype=np.uint8)
    for n in [0, 1, 17, 5000, 300_000]:
        data = rng.choice(alphabet, n).tobytes()
        assert [int(x) for x in tf.csv_split_rows(data)] == oracle.csv_split_rows(data), n


def test_float32_and_interval_cells_on_device(tf, oracle):
    """strictify's last two string targets (strictify.go:44-157): float32 = strconv.ParseFloat(s, 32) (atof32exact + Eisel-Lemire;
    what Go leaves to its decimal slow path goes back to the host, bounded below) and interval = cast.ToDurationE(string) =
    time.ParseDuration, "ns" appended to a text without unit letters.  Go's own ParseDuration table first, then random cells."""
    g = golden("goduration.json")
    S = abi.Schema.of([["k", "int32", True, "0"], ["d", "interval", False, "1"], ["f", "float", False, "2"]])
    texts = [bytes.fromhex(h) for h, _, _ in g["parse_duration"] + g["cast"]]
    texts = [t for t in texts if not any(c in t for c in b',"\r\n') and t == t.strip() and t.isascii() or t in ("12µs".encode(), "12μs".encode())]
    floats = [b"1", b"0.1", b"3.4028235e38", b"3.4028236e38", b"1e39", b"-1e-46", b"16777217", b"16777216.5", b"1.17549435e-38", b"0.000001", b"123456789012345678901234567890",
              b"1e10", b"1e11", b"9999999e10", b"7.038531e-26", b"inf", b"-Infinity", b"nan", b"+nan", b"1_0", b"0x1p3", b"1e", b".", b"5.", b".5", b"1,5", b""]
    lines = []
    for i, t in enumerate(texts):
        f = floats[i % len(floats)]
        f = b'"%s"' % f if b"," in f else f
        lines.append(b"%d,%s,%s" % (i, t, f))
    out, errs = compare(tf, oracle, dict(), S, b"\n".join(lines) + b"\n", "go tables", max_fallback=len(lines) // 4)
    assert out.nrows >= len(lines) // 3
    rng = np.random.default_rng(SEED0 + 41)
    units = [b"ns", b"us", b"ms", b"s", b"m", b"h", "µs".encode(), b"", b"d"]
    rows = []
    for i in range(6000):
        parts = b"".join(b"%d%s%s" % (rng.integers(0, 10 ** int(rng.integers(1, 6))), (b".%d" % rng.integers(0, 1000)) if rng.integers(0, 3) == 0 else b"", units[int(rng.integers(0, len(units)))])
                         for _ in range(int(rng.integers(1, 4))))
        mant = b"%d.%d" % (rng.integers(0, 10 ** int(rng.integers(1, 12))), rng.integers(0, 10 ** int(rng.integers(1, 10))))
        ftxt = mant + ((b"e%d" % rng.integers(-40, 40)) if rng.integers(0, 2) else b"")
        rows.append(b"%d,%s%s,%s%s" % (i, b"-" if rng.integers(0, 5) == 0 else b"", parts, b"-" if rng.integers(0, 4) == 0 else b"", ftxt))
    out, errs = compare(tf, oracle, dict(), S, b"\n".join(rows) + b"\n", "random", max_fallback=60)
    assert out.nrows > 1500
    # DecimalPoint applies to float32 cells as to doubles; null values to intervals
    S2 = abi.Schema.of([["d", "interval", False, "0"], ["f", "float", False, "1"]])
    compare(tf, oracle, dict(decimal_point="#", null_values=["NULL", ""], strings_can_be_null=1), S2, b"NULL,1#5\n1h,2#25e1\n,#5\n3,1#2#3\n", "decimal point")
