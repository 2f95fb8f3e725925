#!/bin/bash
# The CPU pre-flight of the -m gpu tests (tools/hipemu/run_gpu_tests.py) with every kernel compiled under AddressSanitizer + UBSan:
# hostile-input runs (corrupt Parquet objects, damaged protobuf / Debezium / JSON bytes) must not read or write outside their buffers.
# libstdc++ is preloaded next to libasan so that C++ exceptions thrown inside the dlopen'ed library reach ASan's __cxa_throw
# interceptor with its real target resolved.  Reports, if any: /tmp/asanlog.*, /tmp/ubsanlog.*   usage: run_asan.sh [pytest args]
cd "$(dirname "$0")/../.." || exit 1
rm -f /tmp/asanlog.* /tmp/ubsanlog.*
export HIPEMU_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=/tmp/asanlog UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/ubsanlog
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" python tools/hipemu/run_gpu_tests.py "$@"
rc=$?
if grep -l "ERROR: AddressSanitizer\|runtime error" /tmp/asanlog.* /tmp/ubsanlog.* 2>/dev/null; then echo "sanitizer reports found"; exit 1; fi
echo "no sanitizer reports"; exit $rc
