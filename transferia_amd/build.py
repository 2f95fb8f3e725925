"""Builds transferia_amd/libtfgpu.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libtfgpu.so")
SOURCES = ["tf_runtime.hip", "tf_shard.hip", "tf_scan.hip", "tf_plan.cpp", "tf_sql.cpp", "tf_transformation.cpp", "tf_pipeline.cpp", "tf_dbzrecv.cpp", "tf_protoschema.cpp", "tf_transform.hip", "tf_api.hip", "tf_csv.hip", "tf_parquet.hip", "tf_parquetw.hip", "tf_dbzemit.hip", "tf_serialize.hip", "tf_json.hip", "tf_collapse.hip", "tf_srjson.hip", "tf_exchange.hip", "tf_sizeof.hip", "tf_chnative.hip", "tf_debezium.hip", "tf_protobuf.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-function",
         "-x", "hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtfgpu.so cannot be built (and there is no CPU fallback)")
    return exe


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src, *deps])


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "tfgpu.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or _newer(src, obj, headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *os.environ.get("TFGPU_EXTRA_HIPCC_FLAGS", "").split(), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
