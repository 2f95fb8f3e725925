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
import pytest  # noqa: E402

FILES = ["tests/test_queue_serializers.py", "tests/test_serializers.py", "tests/test_confluent_sr.py", "tests/test_gpu_json.py"]
# test_gpu_hits_chain_to_jsoneachrow needs the CSV tile parser (LDS, wave scans), the JSON wave path wave ballots: not emulated
sys.exit(pytest.main(["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", "not hits_chain and not wave_path", *[os.path.join(ROOT, f) for f in FILES], *sys.argv[1:]]))
