"""Shared helpers for the test-suite (golden loading, batch comparison)."""
import json
import os

from transferia_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def item_to_batch(item):
    """One reference-test ChangeItem (tools/extract_golden.py encoding) → (Batch, Schema)."""
    cols = []
    ot = item.get("original_types", {})
    for c in item["schema"]:
        cols.append(abi.ColSchema(c[0], c[1], bool(c[2]), "", ot.get(c[0], "")))
    schema = abi.Schema(cols)
    rows = [item["values"]] if "values" in item else item["rows"]
    if "values" in item and not item["names"]:
        b = abi.Batch([], 1, item.get("ns", ""), item.get("table", ""))
    else:
        b = abi.batch_from_rows(schema, item["names"], rows, item.get("ns", ""), item.get("table", ""), item.get("kinds"))
    b.schema = schema  # the item's TableSchema travels with the batch (ColumnNames may differ from it)
    return b, schema


def json_value(col, i):
    """Render row i of a column the way encoding/json renders the Go value in
    the reference's canon files (enough for the canon comparisons)."""
    g, v = col.pyvalue(i)
    if g == "nil":
        return None
    if g in ("string", "jsonnum", "json"):
        return v.decode("utf-8")
    if g == "time":
        import datetime
        s, ns = v
        d = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=s)
        frac = ("." + ("%09d" % ns).rstrip("0")) if ns else ""
        return d.strftime("%Y-%m-%dT%H:%M:%S") + frac + "Z"
    if g in ("float32", "float64"):
        return v
    return v
