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
