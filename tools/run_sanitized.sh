#!/bin/bash
# Host code of the library (bds_codes.cpp and the host parts of bds_api / bds_acq / bds_track / bds_sync / bds_multi) under
# AddressSanitizer + UndefinedBehaviorSanitizer through the CPU test suite:
#   BDS_SAN=1 ./build.sh && tools/run_sanitized.sh            (no GPU needed; -m gpu tests can be run the same way on a GPU box)
# Python itself is not instrumented, so the sanitizer runtime is preloaded; leak checking is off (the interpreter never frees
# everything), every other report aborts the run.
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
RT="$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)"
[ -f "$RT" ] || RT="$(/opt/rocm/lib/llvm/bin/clang --print-runtime-dir)/libclang_rt.asan-x86_64.so"
export BDS_LIB_PATH="$ROOT/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_san.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
cd "$ROOT" && LD_PRELOAD="$RT" python -m pytest tests -x -q -m "${1:-not gpu}" -p no:cacheprovider
