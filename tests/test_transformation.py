"""transformation.Push around the Apply chain: table plans from Suitable + ResultSchema, the plan cache,
multi-table pushes, TransformerError inputs → `__transform_error` rows, stats, the async token.  The first test is the
reference's own TestMultipleTransformers (pkg/transformer/transformation_test.go:29-111) with its literals."""
import numpy as np
import pytest

from transferia_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init(0)
    return lib


def test_multiple_transformers_reference_case(tf):
    """transformation_test.go:29-111: replace_primary_key(field2, field1) + filter_columns(field2, field1, field4) through
    the whole stage.  Expected (lines 96-110): two items reach the sink; the insert's TableSchema is [field2 key, field1 key,
    field4 non-key] while its ColumnValues stay in ColumnNames order: ["test", 2, "{}"]."""
    from transferia_amd.transformation import ControlItem, Stage
    table = "test_table"
    chain = [tf.Transformer("replace_primary_key", {"keys": ["field2", "field1"], "tables": {"includeTables": [table]}}),
             tf.Transformer("filter_columns", {"tables": {"includeTables": [table]}, "columns": {"includeColumns": ["field2", "field1", "field4"]}})]
    # the test's ColSchemas carry no DataType; the values are a string, an int, a float64 and a string
    schema = abi.Schema.of([["field1", "utf8", True], ["field2", "int64", True], ["field3", "double", True], ["field4", "utf8", True]])
    b = abi.batch_from_rows(schema, ["field1", "field2", "field3", "field4"], [[["string", "test"], ["int", 2], ["float64", 1.23], ["string", "{}"]]], "", table)
    b.schema = schema
    got = []
    stage = Stage(chain, got.extend)
    stage.push([ControlItem("", table, "init_load_table"), tf.DeviceBatch.upload(b)])
    assert len(got) == 2 and isinstance(got[0], ControlItem)
    out = got[1]
    sch = out.table_schema()
    assert [(c.name, c.key) for c in sch.cols] == [("field2", True), ("field1", True), ("field4", False)]
    h = out.download()
    assert [c.name for c in h.cols] == ["field1", "field2", "field4"]
    assert [abi.norm_value(c.pyvalue(0)) for c in h.cols] == [abi.norm_value(v) for v in (["string", "test"], ["int", 2], ["string", "{}"])]
    assert int(h.kind[0]) == abi.K_INSERT if h.kind is not None else True
    st = stage.stats()
    assert st["pushes"] == 1 and st["items_in"] == 1 and st["items_out"] == 1 and st["dropped"] == 0 and st["errors"] == 0 and st["plans_built"] == 1


def _hits_like(tf, n, table, seed, kinds=None):
    rng = np.random.default_rng(seed)
    schema = abi.Schema.of([["id", "int64", True], ["ip", "int32", False], ["day", "int32", False], ["name", "utf8", False]])
    rows = [[["int64", int(i)], ["int32", int(rng.integers(-2**31, 2**31))], ["int32", int(rng.integers(0, 31))], ["string", "n%d" % i]] for i in range(n)]
    b = abi.batch_from_rows(schema, ["id", "ip", "day", "name"], rows, "db", table)
    if kinds is not None:
        b.kind = np.asarray(kinds, np.uint8)
    b.schema = schema
    return b, schema


def test_table_plans_cache_and_multi_table_push(tf, oracle):
    """Plans are per (TableID, schema): a transformer that is not Suitable for a table is not in its plan (and therefore not
    applied), the plan is built once, runs of two tables go through their own plans in one Push."""
    from transferia_amd.transformation import Stage
    chain_cfg = [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["ip"], "tables": {"includeTables": ["^db.a$"]}}),
                 ("filter_rows", {"filter": "day >= 10", "tables": {"includeTables": ["^db.a$", "^db.b$"]}}),
                 ("convert_to_datetime", {"columns": {"includeColumns": ["^day$"]}, "tables": {}}),
                 ("convert_to_string", {"columns": {"includeColumns": ["^nosuch$"]}, "tables": {}})]
    chain = [tf.Transformer(t, c) for t, c in chain_cfg]
    ba, schema = _hits_like(tf, 5000, "a", 1)
    bb, _ = _hits_like(tf, 3000, "b", 2)
    ba2, _ = _hits_like(tf, 700, "a", 3)
    got = []
    stage = Stage(chain, got.extend)
    assert stage.t.table_plan("db", "a", schema) == [0, 1, 2]   # convert_to_string matches no column → not Suitable
    assert stage.t.table_plan("db", "b", schema) == [1, 2]
    stage.push([tf.DeviceBatch.upload(ba), tf.DeviceBatch.upload(bb), tf.DeviceBatch.upload(ba2)])
    st = stage.stats()
    assert st["plans_built"] == 2 and st["pushes"] == 3 and st["items_in"] == 8700
    # the oracle applies, per table, exactly the transformers of that table's plan
    by_table = {}
    for g in got:
        v = g.view()
        by_table.setdefault(v.table_name.decode(), []).append(g.download())
    for name, batches, inputs, plan in (("a", by_table["a"], [ba, ba2], [0, 1, 2]), ("b", by_table["b"], [bb], [1, 2])):
        assert len(batches) == len(inputs)
        for out, src in zip(batches, inputs):
            ref = oracle.apply_chain([oracle.Transformer(*chain_cfg[i]) for i in plan], src, schema).batch
            assert out.nrows == ref.nrows and [c.name for c in out.cols] == [c.name for c in ref.cols]
            for a, b in zip(out.cols, ref.cols):
                assert a.repr == b.repr, (name, a.name)
                if a.repr in abi.VAR_REPRS:
                    assert np.array_equal(a.offsets, b.offsets) and bytes(a.data) == bytes(b.data[: int(b.offsets[-1])]), (name, a.name)
                else:
                    assert np.array_equal(a.values, b.values), (name, a.name)
    assert st["dropped"] == 8700 - sum(g.nrows for g in got) and st["items_out"] == sum(g.nrows for g in got)


def test_errors_keep_the_failing_transformers_input(tf):
    """filter_rows refuses Update / Delete items with a fatal error (filter_rows.go:99-125).  The error item is what THAT
    transformer was handed — here: with `ip` already masked by its predecessor — plus `__transform_error`; with
    ErrorsOutput devnull it is dropped instead (transformation.go:173-190)."""
    from transferia_amd.transformation import TRANSFORM_ERROR_COLUMN, Stage
    n = 400
    kinds = np.full(n, abi.K_INSERT, np.uint8)
    kinds[7::10] = abi.K_UPDATE
    kinds[3::50] = abi.K_DELETE
    b, schema = _hits_like(tf, n, "a", 5, kinds=kinds)
    chain = [tf.Transformer("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["ip"]}),
             tf.Transformer("filter_rows", {"filter": "day >= 0"})]
    bad = np.flatnonzero(kinds != abi.K_INSERT)
    pushed = []
    stage = Stage(chain, pushed.append)
    stage.push([tf.DeviceBatch.upload(b)])
    assert len(pushed) == 2  # errors first (pushErrors), then the transformed items
    (errs,), (ok,) = pushed
    assert errs.nrows == len(bad) and ok.nrows == n - len(bad)
    assert [c.name for c in errs.cols] == ["id", "ip", "day", "name", TRANSFORM_ERROR_COLUMN]
    assert [c.name for c in errs.schema.cols][-1] == TRANSFORM_ERROR_COLUMN and errs.schema.cols[-1].dtype == "utf8"
    assert list(errs.col("id").values) == list(bad)
    assert errs.col("ip").repr == abi.R_STRING and all(len(errs.col("ip").get_bytes(i)) == 64 for i in range(errs.nrows))  # masked, as filter_rows saw it
    assert all(errs.col(TRANSFORM_ERROR_COLUMN).get_bytes(i).startswith(b"fatal") for i in range(errs.nrows))
    # (the library runs this filter in front of the mask it does not read — chain_sequence, tf_transform.hip — and masks the rows
    #  it refused afterwards: they must carry the same HMACs the mask alone gives those rows)
    alone = chain[0].apply(tf.DeviceBatch.upload(b)).transformed.download()
    assert [errs.col("ip").get_bytes(i) for i in range(errs.nrows)] == [alone.col("ip").get_bytes(int(r)) for r in bad]
    ok = ok.download() if hasattr(ok, "download") else ok
    assert [ok.col("ip").get_bytes(i) for i in range(ok.nrows)] == [alone.col("ip").get_bytes(int(r)) for r in np.flatnonzero(kinds == abi.K_INSERT)]
    st = stage.stats()
    assert st["errors"] == len(bad) and st["dropped"] == len(bad)
    quiet = []
    s2 = Stage(chain, quiet.append, errors_output="devnull")
    s2.push([tf.DeviceBatch.upload(b)])
    assert len(quiet) == 1 and s2.dropped_errors == len(bad)


def test_async_tokens_agree_with_sync_pushes(tf):
    """tfgpu_transformation_push_async / tfgpu_wait: pushes submitted from one thread, run on the library's lane workers,
    waited for in submission order (the parsequeue's in-order push) — same rows as the synchronous call."""
    chain = [tf.Transformer("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["ip"]}),
             tf.Transformer("filter_rows", {"filter": "day >= 12"})]
    t = tf.Transformation(chain)
    tf.executor_start(3)
    batches = [tf.DeviceBatch.upload(_hits_like(tf, 2000 + 137 * i, "a", 40 + i)[0]) for i in range(9)]
    tokens = [t.push_run_async(b) for b in batches]
    for b, tok in zip(batches, tokens):
        got = tok.wait().transformed.download()
        ref = t.push_run(b).transformed.download()
        assert got.nrows == ref.nrows and np.array_equal(got.src_row, ref.src_row)
        for a, r in zip(got.cols, ref.cols):
            if a.repr in abi.VAR_REPRS:
                assert np.array_equal(a.offsets, r.offsets) and bytes(a.data) == bytes(r.data)
            else:
                assert np.array_equal(a.values, r.values)
    with pytest.raises(RuntimeError):
        tokens[0].wait()
