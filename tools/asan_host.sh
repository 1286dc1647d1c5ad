#!/bin/bash
# The CPU test suite on libohw built with AddressSanitizer + UBSan on the HOST code (host_engine.cpp, pool.cpp, tracker.cpp,
# vad.cpp, dsp.cpp and the host side of the .hip files).  GPU sanitizers are not available on the pool: this runs HERE, no GPU.
#   tools/asan_host.sh [pytest args]          (default: the host-logic test files)
set -e
cd "$(dirname "$0")/.."
OHW_BUILD_VARIANT=asan python -m openhush_amd.build
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export OHW_LIB="$PWD/openhush_amd/libohw_asan.so"
# python itself leaks by design and is not instrumented: leak checking off, everything else on, first error aborts the test run
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
ARGS=("$@")
[ ${#ARGS[@]} -eq 0 ] && ARGS=(tests/test_host_cpu.py tests/test_dsp.py tests/test_tracker.py tests/test_vad_resample.py tests/test_policy_cpu.py tests/test_cli_wav.py tests/test_streaming.py tests/test_c_abi.py)
LD_PRELOAD="$RT" python -m pytest "${ARGS[@]}" -x -q -m "not gpu" -p no:cacheprovider
