"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/tfgpu.h declares, and every compute entry point fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from transferia_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tfgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == syms
    assert L.tfgpu_abi_version() == 2  # 2: tfgpu_column.absent, tfgpu_batch.col_order (the struct layouts changed)


def test_enum_tables_match_header():
    src = open(os.path.join(ROOT, "include", "tfgpu.h")).read()
    m = re.search(r"typedef enum tfgpu_rowerr \{(.*?)\}", src, re.S)
    names = re.findall(r"TFGPU_ROW_([A-Z_]+)\s*=\s*(\d+)", m.group(1))
    assert {int(v): k for k, v in names} == abi.ROWERR
    m = re.search(r"typedef enum tfgpu_repr \{(.*?)\}", src, re.S)
    reprs = [x.lower() for x in re.findall(r"TFGPU_R_([A-Z0-9]+)", m.group(1)) if x != "_COUNT"]
    assert reprs[: len(abi.REPR_NAMES)] == [r.upper().lower() for r in abi.REPR_NAMES]


def test_struct_sizes():
    # plain C structs, natural alignment on x86-64
    assert C.sizeof(abi.CColumn) == 8 + 4 + 4 + 8 * 3 + 8 + 8 + 8 + 8  # … validity, absent
    assert C.sizeof(abi.CRowError) == 24
    assert C.sizeof(abi.CColSchema) == 64


def test_registry_lists_reference_type_names():
    # YAML keys of the reference (pkg/transformer/registry/*/): same names, drop-in
    assert set(lib.registry()) == {"mask_field", "rename_tables", "filter_columns", "skip_events", "filter_rows",
                                   "convert_to_string", "convert_to_datetime", "sharder_transformer", "replace_primary_key", "sql"}


def _no_gpu():
    n = C.c_int(0)
    lib.load().tfgpu_device_count(C.byref(n))
    return n.value == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(lib.TfgpuError) as ei:
        lib.init()
    assert ei.value.code == lib.ERR_DEVICE
    L = lib.load()
    out = C.c_void_p()
    b = abi.Batch([], 0).to_c()
    assert L.tfgpu_batch_upload(C.byref(b), C.byref(out)) == lib.ERR_DEVICE
    assert b"no CPU fallback" in L.tfgpu_last_error() or b"tfgpu_init" in L.tfgpu_last_error()
    opts = abi.csv_options()
    s = abi.Schema.of([["a", "int32", False, "0"]]).to_c()
    consumed, nerr = C.c_uint64(0), C.c_int64(0)
    assert L.tfgpu_csv_parse(C.byref(opts), C.byref(s), b"1\n", 2, 0, C.byref(out), C.byref(consumed), None, 0, C.byref(nerr)) == lib.ERR_DEVICE


def test_unknown_transformer_type():
    with pytest.raises(lib.TfgpuError) as ei:
        lib.Transformer("no_such_transformer", {})
    assert ei.value.code == lib.ERR_UNKNOWN_TYPE
    # registered in the reference, host-only here: the shim learns it when it builds the chain
    for name in ("lambda", "dbt", "number_to_float_transformer"):
        with pytest.raises(lib.TfgpuError) as ei:
            lib.Transformer(name, {"query": "select * from table"})
        assert ei.value.code == lib.ERR_UNSUPPORTED, name
    # a14 `sql`: the predicate + cast subset has a device plan, anything else is refused by name (tests/test_sql.py)
    assert lib.Transformer("sql", {"query": "select * from table"}).type() == "sql"
    with pytest.raises(lib.TfgpuError) as ei:
        lib.Transformer("sql", {"query": "select count() from table"})
    assert ei.value.code == lib.ERR_UNSUPPORTED

