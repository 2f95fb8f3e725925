"""abstract.Collapse: the oracle's restatement against the reference's TestCollapse cases."""
import json
import os

import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "collapse.json")
CASES = json.load(open(GOLD, encoding="utf-8"))["cases"]


def _norm(v):
    """[gotype, value] as the oracle reports it: Go int is int64, text comes back as bytes"""
    t, x = v
    if t == "int":
        t = "int64"
    if t in ("string", "json") and isinstance(x, str):
        x = x.encode("utf-8")
    return [t, x]


def _item_as_output(it, idx):
    old = [[n, _norm(v)] for n, v in zip(it.get("old_names", []), it.get("old_values", []))]
    return {"kind": it["kind"], "names": it["names"], "values": [_norm(v) for v in it["values"]], "old": old, "src": idx}


@pytest.mark.parametrize("case", CASES, ids=["%s@%s" % (c["name"], i) for i, c in enumerate(CASES)])
def test_reference_cases(case):
    res = oracle.collapse_rows(case["items"])
    e = case["expect"]
    assert len(res) == e["len"], res
    for k, kind in e.get("kinds", {}).items():
        assert res[int(k)]["kind"] == kind
    for k, vals in e.get("values_equal", {}).items():
        assert res[int(k)]["values"] == [_norm(v) for v in vals]
    for k, vals in e.get("values_contain", {}).items():
        for v in vals:
            assert _norm(v) in res[int(k)]["values"]
    for k, vals in e.get("values_not_contain", {}).items():
        for v in vals:
            assert _norm(v) not in res[int(k)]["values"]
    for k, names in e.get("names_contain", {}).items():
        for n in names:
            assert n in res[int(k)]["names"]
    for k, n in e.get("nvalues", {}).items():
        assert len(res[int(k)]["values"]) == n and len(res[int(k)]["names"]) == n
    for k, cols in e.get("col_value", {}).items():
        r = res[int(k)]
        for name, v in cols.items():
            assert r["values"][r["names"].index(name)] == _norm(v)
    for k, v in e.get("old_value0", {}).items():
        assert res[int(k)]["old"][0][1] == _norm(v)
    if e.get("equals_input"):
        assert res == [_item_as_output(it, i) for i, it in enumerate(case["items"])]
    for k, src in e.get("equals_item", {}).items():
        assert res[int(k)] == _item_as_output(case["items"][src], src)


# ---- columnar form (what the device sees) -------------------------------------------------------------------------
from collapse_cases import random_batch, model, expected_from_model, rows_of  # noqa: E402

SHAPES = [dict(n=2), dict(n=40), dict(n=300, domain=5), dict(n=300, domain=40, two_keys=True), dict(n=200, p_old=1.0),
          dict(n=200, p_old=0.0), dict(n=200, weights=(1, 0, 0, 0)), dict(n=200, weights=(1, 0, 0, 1)), dict(n=250, null_keys=0.2, two_keys=True),
          dict(n=250, domain=3, weights=(1, 6, 3, 0)), dict(n=200, two_keys=True, bytes_key=True)]


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_device_algorithm_model_matches_oracle(shape):
    """tf_collapse.hip replays the reference's loop per connected component of keys over integer key ids; this is that
    algorithm in Python against the oracle's restatement of the Go maps, on random CDC batches."""
    for seed in range(6):
        b, schema = random_batch(1000 * shape + seed, **SHAPES[shape])
        ref = oracle.collapse(b, schema)
        want = expected_from_model(b, model(b, schema)) if b.nrows >= 2 and (b.kind != 0).any() else rows_of(b)
        assert rows_of(ref.batch) == want, (shape, seed)


def test_columnar_equals_rowwise():
    """The oracle's columnar I/O (OldKeys as columns + presence bitmap) against its row-wise builder."""
    b, schema = random_batch(7, 60, domain=4, two_keys=True, nstrs=9)  # valid UTF-8 only: the row-wise builder reads JSON
    items = []
    for i in range(b.nrows):
        it = {"kind": ["insert", "update", "delete", "other"][int(b.kind[i])], "keys": ["id", "k2"], "names": [c.name for c in b.cols],
              "values": [[g, (x.decode("utf-8") if isinstance(x, bytes) else x)] for g, x in (c.pyvalue(i) for c in b.cols)]}
        if b.old_present[i]:
            it["old_names"] = [c.name for c in b.old_keys]
            it["old_values"] = [[g, (x.decode("utf-8") if isinstance(x, bytes) else x)] for g, x in (c.pyvalue(i) for c in b.old_keys)]
        items.append(it)
    rw = oracle.collapse_rows(items)
    col = rows_of(oracle.collapse(b, schema).batch)
    assert len(rw) == len(col)
    for r, c in zip(rw, col):
        assert ["insert", "update", "delete", "other"][c[0]] == r["kind"] and c[1] == r["src"]
        assert [tuple(v) for v in r["values"]] == [tuple(v) for v in c[2]]
        assert (c[3] is None) == (not r["old"])
        if c[3] is not None:
            assert [tuple(v[1]) for v in r["old"]] == [tuple(x) for x in c[3]]


# ---- ChangeItem.KeysChanged / SplitUpdatedPKeys ---------------------------------------------------------------------
KC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keys_changed.json"), encoding="utf-8"))


def test_keys_changed_reference_cases():
    items = [c["item"] for c in KC["keys_changed"]]
    for c in KC["keys_changed"]:
        assert oracle.keys_changed_rows([c["item"]]) == [c["changed"]], c["ref"]
    # the second half of every MySQL-ported case: a TableSchema with one more non-key column changes nothing
    assert len(items) == 14


def test_split_updated_pkeys_reference_case():
    sp = KC["split"]
    got = oracle.split_updated_pkeys_rows(sp["items"])
    assert len(got) == len(sp["expected"])
    for gl, el in zip(got, sp["expected"]):
        assert len(gl) == len(el)
        for g, e in zip(gl, el):
            assert g["kind"] == e["kind"] and g["names"] == e["names"]
            assert g["values"] == [_norm(v) for v in e["values"]]
            assert [[n, v] for n, v in g["old"]] == [[n, _norm(v)] for n, v in zip(e.get("old_names", []), e.get("old_values", []))]


def test_keys_changed_columnar_equals_rowwise():
    b, schema = random_batch(11, 80, domain=3, two_keys=True, nstrs=9, null_keys=0.2)
    items = []
    for i in range(b.nrows):
        it = {"kind": ["insert", "update", "delete", "other"][int(b.kind[i])], "keys": ["id", "k2"], "names": [c.name for c in b.cols],
              "values": [[g, (x.decode("utf-8") if isinstance(x, bytes) else x)] for g, x in (c.pyvalue(i) for c in b.cols)]}
        if b.old_present[i]:
            it["old_names"] = [c.name for c in b.old_keys]
            it["old_values"] = [[g, (x.decode("utf-8") if isinstance(x, bytes) else x)] for g, x in (c.pyvalue(i) for c in b.old_keys)]
        items.append(it)
    col = oracle.keys_changed(b, schema)
    assert list(col) == oracle.keys_changed_rows(items)
    assert 0 < col.sum() < b.nrows
