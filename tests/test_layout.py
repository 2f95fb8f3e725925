"""The oracle is test infrastructure: nothing in the product package may
import, load or link it, and the product must not contain a CPU fallback."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "transferia_amd")


def product_files():
    for d, _, files in os.walk(PKG):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                yield os.path.join(d, f)


def test_product_never_touches_oracle():
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liboracle|oracle/|ora_[a-z_]+\()")
    bad = []
    for p in product_files():
        src = open(p, encoding="utf-8").read()
        for m in pat.finditer(src):
            line = src[: m.start()].count("\n") + 1
            bad.append(f"{os.path.relpath(p, ROOT)}:{line}: {m.group(0)}")
    assert not bad, bad


def test_libtfgpu_does_not_link_oracle():
    so = os.path.join(PKG, "libtfgpu.so")
    if not os.path.exists(so):
        return
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert " ora_" not in syms


def test_only_allowed_importers_of_oracle():
    allowed_dirs = {"tests", "oracle", "bench"}  # bench/ = the modules behind bench.py: the cpu_baseline / parity legs of its lines
    allowed_files = {"bench.py", "__graft_entry__.py"}
    offenders = []
    for d, _, files in os.walk(ROOT):
        rel = os.path.relpath(d, ROOT)
        top = rel.split(os.sep)[0]
        if top in (".git", "gpurun_out", "profiles") or "__pycache__" in rel:
            continue
        for f in files:
            if not f.endswith(".py"):
                continue
            p = os.path.join(d, f)
            src = open(p, encoding="utf-8").read()
            if re.search(r"^\s*(from oracle|import oracle)", src, re.M):
                if top in allowed_dirs or (rel == "." and f in allowed_files):
                    continue
                offenders.append(os.path.relpath(p, ROOT))
    assert not offenders, offenders


def test_no_reference_path_at_runtime():
    """/root/reference does not exist on the GPU box: only tools/ may mention it."""
    bad = []
    for d, _, files in os.walk(ROOT):
        rel = os.path.relpath(d, ROOT)
        top = rel.split(os.sep)[0]
        if top in (".git", "gpurun_out", "tools") or "__pycache__" in rel:
            continue
        for f in files:
            if f.endswith(".py") and f != "test_layout.py":
                src = open(os.path.join(d, f), encoding="utf-8").read()
                if re.search(r"open\([^)]*/root/reference|REF\s*=\s*\"/root/reference", src):
                    bad.append(os.path.join(rel, f))
    assert not bad, bad


def test_product_never_touches_the_emulator():
    """tools/hipemu is a CPU pre-flight for tests: the product binding knows one library, transferia_amd/libtfgpu.so,
    has no environment override for it, and nothing in the package mentions the emulator."""
    bad = []
    for p in product_files():
        src = open(p, encoding="utf-8").read()
        if re.search(r"hipemu|libtfgpu_emu|emu_launch", src):
            bad.append(os.path.relpath(p, ROOT))
    assert not bad, bad
    lib_src = open(os.path.join(PKG, "lib.py"), encoding="utf-8").read()
    assert re.search(r'_LIBPATH\s*=\s*os\.path\.join\(_HERE,\s*"libtfgpu\.so"\)', lib_src)
    assert not re.search(r"environ[^\n]*LIB", lib_src)


def test_collecting_the_suite_does_not_import_torch():
    """torch brings its own HIP / HSA / RCCL copies; loaded into the test process before libtfgpu.so (at collection time, say,
    through a skipif that counts GPUs) they are what the library's RCCL binding then finds, and its init fails on the GPU box.
    Tests that need torch import it inside a worker process."""
    import subprocess
    import sys
    code = ("import sys, pytest\n"
            "pytest.main(['--collect-only', '-q', '-p', 'no:cacheprovider', %r])\n"
            "print('TORCH-LOADED' if 'torch' in sys.modules else 'TORCH-FREE')\n") % os.path.join(ROOT, "tests")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT).stdout
    assert "TORCH-FREE" in out, out[-2000:]
