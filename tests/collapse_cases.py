"""Random CDC batches for abstract.Collapse (columnar form: one ColumnNames list, OldKeys as columns) and a model of the
device algorithm (transferia_amd/csrc/tf_collapse.hip) over the same inputs."""
import numpy as np

from transferia_amd import abi
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)

KINDS = ["insert", "update", "delete", "other"]
STRS = [b"", b"a", b"b", b"<x>&", "é".encode(), b"\\u003c", b'"', b"a\x00", "\ufffd".encode(), b"\xff", b"\xfe", b"a\xc3"]  # the last three: invalid UTF-8


def random_batch(seed, n, domain=8, two_keys=False, p_old=0.6, weights=(3, 4, 2, 1), null_keys=0.0, bytes_key=False, nstrs=len(STRS)):
    """(Batch, Schema).  Keys come from a small domain so rows chain; Update / Delete rows carry OldKeys with
    probability p_old (an Insert never does here); OldKeys also list a non-key name, which Collapse must ignore."""
    rng = np.random.default_rng(SEED0 + (seed))
    cols = [abi.ColSchema("id", "int64", True, "", ""), abi.ColSchema("v", "int64", False, "", ""), abi.ColSchema("s", "utf8", False, "", "")]
    if two_keys:
        cols.insert(1, abi.ColSchema("k2", "string" if bytes_key else "utf8", True, "", ""))
    schema = abi.Schema(cols)
    names = [c.name for c in cols]
    kinds = [KINDS[k] for k in rng.choice(4, size=n, p=np.array(weights) / sum(weights))]
    sgo = "bytes" if bytes_key else "string"

    def key_vals():
        kv = [["int64", int(rng.integers(0, domain))] if rng.random() >= null_keys else ["nil", None]]
        if two_keys:
            kv.append([sgo, STRS[int(rng.integers(0, nstrs))]])
        return kv
    rows, old_rows, present = [], [], np.zeros(n, bool)
    for i in range(n):
        kv = key_vals()
        row = [kv[0]] + ([kv[1]] if two_keys else []) + [["int64", i], ["string", "row%d" % i]]
        rows.append(row)
        if kinds[i] in ("update", "delete") and rng.random() < p_old:
            present[i] = True
            okv = key_vals() if rng.random() < 0.5 else kv  # half of them change the primary key
            old_rows.append(okv + [["string", "old%d" % i]])
        else:
            old_rows.append([["nil", None]] * ((2 if two_keys else 1) + 1))
    b = abi.batch_from_rows(schema, names, rows, "ns", "t", kinds)
    b.schema = schema
    if present.any():
        onames = ["id"] + (["k2"] if two_keys else []) + ["s"]
        oschema = abi.Schema([abi.ColSchema(nm, schema.dtype_of(nm), False, "", "") for nm in onames])
        ob = abi.batch_from_rows(oschema, onames, old_rows)
        b.old_keys = ob.cols
        b.old_present = present
    return b, schema


def key_text(b, schema, i, old):
    """What the walk keys a row by — any injective image of the reference's key string serves the model."""
    names = sorted({c.name for c in schema.cols if c.key})
    cols = {c.name: c for c in (b.old_keys if old else b.cols)}
    def canon(v):  # json.Marshal writes every invalid UTF-8 byte of a string as \\ufffd: such strings coincide
        g, x = v
        if g == "string":
            x = "".join("<bad>" if 0xDC80 <= ord(ch) <= 0xDCFF else ch for ch in x.decode("utf-8", "surrogateescape"))
        return (g, x)
    return repr([canon(cols[nm].pyvalue(i)) if nm in cols else ("nil", None) for nm in names])


def model(b, schema):
    """The device algorithm in Python: key ids, components, per-component walk, result order.
    Returns [(meta_row, values_row, oldkeys_row)] in output order."""
    n = b.nrows
    kind = [int(k) for k in b.kind]
    pres = getattr(b, "old_present", None)
    has_old = getattr(b, "old_keys", None) is not None
    ids = {}
    kc = [ids.setdefault(key_text(b, schema, i, False), len(ids)) for i in range(n)]
    kro = [ids.setdefault(key_text(b, schema, i, True), len(ids)) if has_old and pres[i] else None for i in range(n)]
    parent = list(range(len(ids)))

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x
    for i in range(n):
        if kro[i] is not None:
            a, c = find(kc[i]), find(kro[i])
            if a != c:
                parent[max(a, c)] = min(a, c)
    comp = {}
    for i in range(n):
        comp.setdefault(find(kc[i]), []).append(i)
    rows_m, rows_v, k2idx, del_i, del_o, ak = {}, {}, {}, {}, {}, {}
    U, D = abi.K_UPDATE, abi.K_DELETE
    for members in comp.values():  # any order: components never touch each other's keys
        for i in members:
            ko = kro[i] if kind[i] in (U, D) and kro[i] is not None else kc[i]
            if kind[i] == abi.K_INSERT:
                del_i.pop(ko, None)
                rows_m[ko], rows_v[ko], k2idx[ko], ak[i] = i, i, i, ko
            elif kind[i] == U:
                del_i.pop(ko, None)
                m = rows_m.get(ko)
                if m is None:
                    rows_m[kc[i]], rows_v[kc[i]], k2idx[kc[i]], ak[i] = i, i, i, kc[i]
                else:
                    newk = kro[m] if kind[m] in (U, D) and kro[m] is not None else kc[i]
                    if newk != ko:
                        rows_m.pop(ko)
                    rows_m[newk], rows_v[newk], k2idx[newk], ak[i] = m, i, i, newk
            elif kind[i] == D:
                m = rows_m.pop(ko, None)
                k, o = ko, i
                if m is not None and kro[m] is not None:
                    k, o = kro[m], m
                del_i[k], del_o[k], ak[i] = i, o, k
    out = [(i, i, i) for i in range(n) if kind[i] == abi.K_OTHER]
    for i in range(n):
        if kind[i] in (abi.K_INSERT, U) and ak[i] in rows_m and k2idx[ak[i]] == i:
            out.append((rows_m[ak[i]], rows_v[ak[i]], rows_m[ak[i]]))
    for i in range(n):
        if kind[i] == D and del_i.get(ak[i]) == i:
            out.append((i, i, del_o[ak[i]]))
    return out


def expected_from_model(b, sel):
    """Rows of the result as comparable tuples, picked from the input by the model's selection."""
    old = getattr(b, "old_keys", None) or []
    pres = getattr(b, "old_present", None)
    res = []
    for m, v, o in sel:
        oldv = [abi.norm_value(c.pyvalue(o)) for c in old] if old and pres[o] else None
        res.append((int(b.kind[m]), m, [abi.norm_value(c.pyvalue(v)) for c in b.cols], oldv))
    return res


def rows_of(out):
    """A result batch (oracle or device) in the same comparable form."""
    old = getattr(out, "old_keys", None) or []
    pres = getattr(out, "old_present", None)
    res = []
    for i in range(out.nrows):
        oldv = [abi.norm_value(c.pyvalue(i)) for c in old] if old and (pres is None or pres[i]) else None
        res.append((int(out.kind[i]) if out.kind is not None else 0, int(out.src_row[i]) if out.src_row is not None else i,
                    [abi.norm_value(c.pyvalue(i)) for c in out.cols], oldv))
    return res


# ---- items whose ColumnNames differ (TOAST-style Updates): the ABSENT cell state of the columnar form -----------------------------
def batch_from_items(items, ns="", table="t", names=None):
    """(Batch, Schema) for row-wise items [{kind, keys, names, values, old_names?, old_values?}] whose ColumnNames may differ: the batch's
    columns are the names in order of first appearance, a row that does not list a column is ABSENT there (Column.absent).  None when the
    items cannot be one columnar batch (OldKeys under different KeyNames, one name under two Go types)."""
    names = list(names or [])
    for it in items:
        for nm in it["names"]:
            if nm not in names:
                names.append(nm)
    gotype = {}
    for it in items:
        for nm, v in zip(it["names"], it["values"]):
            if v[0] != "nil" and gotype.setdefault(nm, v[0]) != v[0]:
                return None
    dt = {"int": "int64", "int64": "int64", "string": "utf8", "bytes": "string", "float64": "double"}
    keys = items[0]["keys"]
    knames = [k for k in keys if k not in names]  # a key column no item lists still belongs to the TableSchema
    schema = abi.Schema([abi.ColSchema(nm, dt.get(gotype.get(nm, "string"), "utf8"), nm in keys, "", "") for nm in names + knames])
    rows = [[dict(zip(it["names"], it["values"])).get(nm, ["nil", None]) for nm in names] for it in items]
    b = abi.batch_from_rows(schema, names, rows, ns, table, [it["kind"] for it in items])
    b.schema = schema
    for c in b.cols:
        ab = np.array([c.name not in it["names"] for it in items], bool)
        if ab.any():
            c.absent = ab
            if c.validity is None:
                c.validity = ~ab
    onames = next((it["old_names"] for it in items if it.get("old_names")), None)
    if onames:
        if any(it.get("old_names") and it["old_names"] != onames for it in items):
            return None
        ogo = {}
        for it in items:
            for nm, v in zip(it.get("old_names") or [], it.get("old_values") or []):
                if v[0] != "nil":
                    ogo.setdefault(nm, v[0])
        osch = abi.Schema([abi.ColSchema(nm, dt.get(ogo.get(nm, "string"), "utf8"), False, "", "") for nm in onames])
        ob = abi.batch_from_rows(osch, onames, [it.get("old_values") or [["nil", None]] * len(onames) for it in items])
        b.old_keys, b.old_present = ob.cols, np.array([bool(it.get("old_names")) for it in items])
    return b, schema


def items_of(out, keys=None):
    """A batch (input or result) as the row-wise items the oracle's collapse_rows reports: a row lists the columns it is not ABSENT from."""
    old = getattr(out, "old_keys", None) or []
    pres = getattr(out, "old_present", None)
    res = []
    order = getattr(out, "col_order", None)
    for i in range(out.nrows):
        seq = out.cols if order is None else [out.cols[int(j)] for j in order[i]]  # tfgpu_batch.col_order: the row's own ColumnNames order
        cols = [c for c in seq if getattr(c, "absent", None) is None or not c.absent[i]]
        it = {"kind": KINDS[int(out.kind[i])] if out.kind is not None else "insert", "names": [c.name for c in cols],
              "values": [list(abi.norm_value(c.pyvalue(i))) for c in cols],
              "old": [[c.name, list(abi.norm_value(c.pyvalue(i)))] for c in old] if old and (pres is None or pres[i]) else [],
              "src": int(out.src_row[i]) if out.src_row is not None else i}
        res.append(it)
    return res


def norm_items(rows):
    """the oracle's row-wise result in the same comparable form"""
    return [{"kind": r["kind"], "names": list(r["names"]), "values": [list(abi.norm_value(v)) for v in r["values"]],
             "old": [[n, list(abi.norm_value(v))] for n, v in r["old"]], "src": r["src"]} for r in rows]


def random_toast_items(seed, n, ncols=5, domain=6, p_absent=0.35, weights=(3, 6, 1, 1), front_ok=False, toastable=None, p_nokey=0.3):
    """Row-wise CDC items over columns id (key), c0..c{ncols-1}: an Update leaves each non-key column out with probability p_absent
    (only the columns in `toastable`, when given; sometimes the key too, then it carries OldKeys); Inserts list everything unless front_ok (then they may leave columns out as well —
    chains whose merged name order leaves batch order, which the device must refuse)."""
    rng = np.random.default_rng(SEED0 + seed)
    names = ["id"] + ["c%d" % j for j in range(ncols)]
    items = []
    for i in range(n):
        kind = KINDS[int(rng.choice(4, p=np.array(weights) / sum(weights)))]
        key = int(rng.integers(0, domain))
        full = {"id": ["int64", key]}
        for j in range(ncols):
            full["c%d" % j] = ["string", "r%d.%d" % (i, j)] if j % 2 == 0 else (["int64", i * 10 + j] if rng.random() > 0.15 else ["nil", None])
        listed = list(names)
        if kind == "update" or (front_ok and kind == "insert"):
            listed = ["id"] + [nm for j, nm in enumerate(names[1:]) if (toastable is not None and j not in toastable) or rng.random() >= p_absent]
        it = {"kind": kind, "keys": ["id"], "names": listed, "values": [full[nm] for nm in listed]}
        if kind in ("update", "delete") and rng.random() < 0.5:
            okey = key if rng.random() < 0.6 else int(rng.integers(0, domain))
            it["old_names"], it["old_values"] = ["id"], [["int64", okey]]
            if kind == "update" and rng.random() < p_nokey:  # the key itself left out: the row is named by its OldKeys alone
                it["names"] = it["names"][1:]
                it["values"] = it["values"][1:]
        if kind == "delete" and rng.random() < 0.5:
            it["names"], it["values"] = ["id"], [full["id"]]
        items.append(it)
    return items
