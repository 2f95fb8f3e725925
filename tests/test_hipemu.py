"""Every kernel of libtfgpu runs as plain C++ in this GPU-less container (tools/hipemu: one fiber per thread, wavefront
operations and barriers exchanged in lockstep) and must already agree with the oracle byte for byte — a pre-flight for
the `-m gpu` run, not a substitute for it: it proves the kernels' logic, not the gfx950 build.  Runs in a subprocess so that the product
binding of THIS process never points at anything but libtfgpu.so."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_emulated_kernels_match_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hipemu", "run_gpu_tests.py")], capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
