#!/usr/bin/env python3
"""Runs the `-m gpu` parity tests of the emulated translation units against tools/hipemu's build of the same kernels.

TEST INFRASTRUCTURE ONLY.  This process — and only this process — points the ctypes binding at libtfgpu_emu.so, so the
kernels' byte-level logic is checked against the oracle here, in the GPU-less container, before a GPU box is spent.
It proves nothing about the gfx950 build: the `-m gpu` run on the MI355X remains the parity gate."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import build as emu_build  # noqa: E402

so = emu_build.build()
from transferia_amd import lib  # noqa: E402

lib._LIBPATH = so
os.environ["TFGPU_TEST_EMU_LIB"] = so
os.environ["TFGPU_RCCL_LIB"] = os.path.join(os.path.dirname(so), "libfakerccl.so")  # tfgpu_exchange's RCCL stand-in
import pytest  # noqa: E402

# everything marked gpu, except what needs the real device: full-size batches (minutes here), torch.cuda + real RCCL, the static
# resource table of the gfx950 build.  Known emulator gap: the experimental JSON wave path (TFGPU_JSON_WAVES=1) parses the
# subnormal-boundary literal 1e-308 as nil here and correctly on the MI355X.
K = "not fullsize and not rccl and not resources and not pipeline_single_rank and not cross_check"
args = sys.argv[1:]
if "-k" in args:
    i = args.index("-k")
    K = "(%s) and (%s)" % (K, args[i + 1])
    del args[i:i + 2]
# workers test by test (pytest-xdist, when it is there; TFGPU_EMU_DIST=loadfile: by file): the pre-flight is most of the CPU suite's wall-clock; each worker is a
# process of its own with the same TFGPU_TEST_EMU_LIB, and the conftest points lib at it
par = []
if "-n" not in args and os.environ.get("TFGPU_EMU_WORKERS", str(min(8, os.cpu_count() or 4))) != "1":
    try:
        import xdist  # noqa: F401
        par = ["-n", os.environ.get("TFGPU_EMU_WORKERS", str(min(8, os.cpu_count() or 4))), "--dist", os.environ.get("TFGPU_EMU_DIST", "load")]
    except ImportError:
        pass
sys.exit(pytest.main(["-m", "gpu", "-q", "-p", "no:cacheprovider", *par, "-k", K, os.path.join(ROOT, "tests"), *args]))
