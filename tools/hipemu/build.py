#!/usr/bin/env python3
"""Builds tools/hipemu/_build/libtfgpu_emu.so: tf_runtime.hip + tf_serialize.hip compiled as plain C++ (g++) against
tools/hipemu/hip/hip_runtime.h, every `kernel<<<g, b, s, st>>>(args)` rewritten to emu_launch(kernel, g, b, args).

TEST INFRASTRUCTURE ONLY: a way to run the lane-per-item kernels' logic in the GPU-less build container.  Only
tests/test_hipemu.py loads the result; transferia_amd/ never does (it has no CPU path)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "transferia_amd", "csrc")
SANITIZE = os.environ.get("HIPEMU_SANITIZE") == "1"   # AddressSanitizer + UBSan build (tools/hipemu/run_gpu_tests.py under LD_PRELOAD=libasan): hostile-input runs
OUT = os.path.join(HERE, "_build_asan" if SANITIZE else "_build")
UNITS = ["tf_runtime.hip", "tf_shard.hip", "tf_scan.hip", "tf_plan.cpp", "tf_sql.cpp", "tf_transformation.cpp", "tf_pipeline.cpp", "tf_dbzrecv.cpp", "tf_protoschema.cpp", "tf_transform.hip", "tf_api.hip", "tf_csv.hip", "tf_parquet.hip", "tf_parquetw.hip", "tf_dbzemit.hip", "tf_serialize.hip", "tf_json.hip",
         "tf_collapse.hip", "tf_srjson.hip", "tf_exchange.hip", "tf_sizeof.hip", "tf_chnative.hip", "tf_debezium.hip", "tf_protobuf.hip"]  # = transferia_amd/build.py SOURCES
LAUNCH = re.compile(r"(\b[A-Za-z_][A-Za-z0-9_:]*(?:<[^<>;]*>)?)\s*<<<\s*([^;]*?)>>>\s*\(")


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def rewrite(src: str) -> str:
    def sub(m):
        cfg = _split_args(m.group(2))
        return "emu_launch(%s, dim3(%s), dim3(%s), " % (m.group(1), cfg[0], cfg[1])
    out = LAUNCH.sub(sub, src)
    assert "<<<" not in out, "unrewritten kernel launch"
    return out


def build() -> str:
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libtfgpu_emu.so")
    srcs = []
    for u in UNITS:
        with open(os.path.join(CSRC, u)) as f:
            text = rewrite(f.read())
        text = text.replace(", )", ")")  # kernels without arguments
        dst = os.path.join(OUT, os.path.splitext(u)[0] + "_emu.cpp")
        with open(dst, "w") as f:
            f.write(text)
        srcs.append(dst)
    objs = []

    def compile_one(src):
        obj = os.path.splitext(src)[0] + ".o"
        deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + [os.path.join(HERE, "hip", "hip_runtime.h")]
        if os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
            return obj
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-c", "-Wno-unused-result", "-I", HERE, "-I", CSRC, "-o", obj, src]
        if SANITIZE:
            cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize=alignment,shift"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-6000:])
            raise RuntimeError("hipemu build failed: " + src)
        return obj
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run(["g++", "-shared", "-fPIC", *(["-fsanitize=address,undefined"] if SANITIZE else []), "-o", so, *objs, "-ldl", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-6000:])
        raise RuntimeError("hipemu link failed")
    # the stand-in for RCCL the emulated tfgpu_exchange binds through TFGPU_RCCL_LIB (world_size-2 tests)
    fake_src, fake = os.path.join(HERE, "fake_rccl.cpp"), os.path.join(OUT, "libfakerccl.so")
    if not os.path.exists(fake) or os.path.getmtime(fake) < os.path.getmtime(fake_src):
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-o", fake, fake_src, "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-6000:])
            raise RuntimeError("fake rccl build failed")
    return so


if __name__ == "__main__":
    print(build())
