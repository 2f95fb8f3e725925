"""Synthetic ClickBench-`hits` workload: the 105-column schema
of the reference fixture pkg/providers/postgres/testdata/hits_data.json (kept
as tests/golden/hits_schema.json) and a deterministic, row-addressable CSV
generator (tools/hitsgen.c).  Used by bench.py and the tests; produces input
data only."""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

from . import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(_ROOT, "tools", "hitsgen.c")
_SO = os.path.join(_ROOT, "tools", "libhitsgen.so")
SEED = 0x5EEDC11C

_lib = None


def build_generator(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, _SRC, "-lm"])
    return _SO


def _gen():
    global _lib
    if _lib is None:
        build_generator()
        L = C.CDLL(_SO)
        L.hits_csv.restype = C.c_uint64
        L.hits_csv.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
        L.hits_row_cap.restype = C.c_uint64
        L.hits_row_cap.argtypes = [C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def hits_columns():
    with open(os.path.join(_ROOT, "tests", "golden", "hits_schema.json"), encoding="utf-8") as f:
        return json.load(f)["columns"]


def hits_schema() -> abi.Schema:
    """TableSchema of `hits` as the s3 CSV reader sees it: column i reads CSV field i."""
    return abi.Schema([abi.ColSchema(n, t, bool(k), str(i)) for i, (n, t, k) in enumerate(hits_columns())])


_UNIFORM64 = {"watchid", "userid", "funiqid", "refererhash", "urlhash"}
_UNIFORM32 = {"clientip", "remoteip"}
_ZIPF = {"counterid", "regionid"}
_URLS = {"url", "referer", "originalurl"}


def hits_roles() -> np.ndarray:
    roles = []
    for name, typ, _ in hits_columns():
        if typ == "int16":
            roles.append(0 if name.startswith(("is", "has", "java", "cookie", "dontcount", "withhash", "goodevent")) else 1)
        elif typ == "int32":
            roles.append(2 if name in _UNIFORM32 else 3 if name in _ZIPF else 12)
        elif typ == "int64":
            roles.append(4 if name in _UNIFORM64 else 5)
        elif typ == "timestamp":
            roles.append(6)
        elif typ == "date":
            roles.append(7)
        elif name == "title":
            roles.append(8)
        elif name in _URLS:
            roles.append(9)
        elif name in ("useragentminor", "hitcolor"):
            roles.append(11)
        else:
            roles.append(10)
    return np.asarray(roles, dtype=np.int32)


def hits_csv(nrows: int, row0: int = 0, seed: int = SEED, header: bool = True) -> bytes:
    """CSV text (',' delimiter, '"' quote, header line) of rows [row0, row0+nrows)."""
    L = _gen()
    roles = hits_roles()
    cap = int(L.hits_row_cap(roles.ctypes.data, len(roles)))
    # expected ~700 B/row; size for 1.5x the expectation, retry with the hard cap if needed
    for est in (max(1 << 16, int(nrows * 1100) + cap), nrows * cap + cap):
        buf = np.empty(est, dtype=np.uint8)
        n = int(L.hits_csv(seed, row0, nrows, roles.ctypes.data, len(roles), buf.ctypes.data, est))
        if n or nrows == 0:
            body = buf[:n].tobytes()
            break
    else:
        raise RuntimeError("hits generator: buffer too small")
    if header:
        return (",".join(c[0] for c in hits_columns()) + "\n").encode() + body
    return body


class HitsStream:
    """Chunked generator writing into ONE reused host buffer (fresh pages are
    expensive on sandboxed hosts), e.g. for staging a large table into HBM."""

    def __init__(self, chunk_rows: int = 1 << 15, seed: int = SEED):
        self.L = _gen()
        self.roles = hits_roles()
        self.seed = seed
        self.chunk_rows = chunk_rows
        cap = int(self.L.hits_row_cap(self.roles.ctypes.data, len(self.roles)))
        self.buf = np.zeros(chunk_rows * 1300 + cap, dtype=np.uint8)
        self.header = (",".join(c[0] for c in hits_columns()) + "\n").encode()

    def chunk(self, row0: int, nrows: int):
        """→ (buffer, nbytes) for rows [row0, row0+nrows); buffer is reused."""
        n = int(self.L.hits_csv(self.seed, row0, nrows, self.roles.ctypes.data, len(self.roles), self.buf.ctypes.data, len(self.buf)))
        if not n and nrows:
            raise RuntimeError("hits generator: chunk buffer too small")
        return self.buf, n

    def total_bytes(self, row0: int, nrows: int, header: bool = True) -> int:
        tot = len(self.header) if header else 0
        r = row0
        while r < row0 + nrows:
            k = min(self.chunk_rows, row0 + nrows - r)
            tot += self.chunk(r, k)[1]
            r += k
        return tot


def hits_csv_options(header: bool = True) -> abi.CCsvOptions:
    return abi.csv_options(skip_rows=1 if header else 0)
