import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / -m gpu)")
    # tools/hipemu/run_gpu_tests.py (the CPU pre-flight of the -m gpu tests, test infrastructure) exports this so that the
    # tests it runs — and the pytest subprocesses some of them spawn — bind its build of the kernels; nothing else sets it.
    emu = os.environ.get("TFGPU_TEST_EMU_LIB")
    if emu:
        from transferia_amd import lib
        lib._LIBPATH = emu


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as ora
    ora.build()
    return ora
