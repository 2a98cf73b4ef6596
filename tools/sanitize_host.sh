#!/bin/bash
# CPU-side sanitizer pass over the HOST half of the engine (SURVEY section 5, row "sanitizers").
#   * climsim_amd/csrc/climsim_hip.hip compiled with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side only
#     (-fno-gpu-sanitize: the gfx950 code object is the production one and is never run here), loaded through the same ctypes binding (CLIMSIM_HIP_LIB), and driven by
#     tests/test_cabi_cpu.py + tests/test_cabi_args_cpu.py: every entry point's argument validation, error paths, the
#     thread-local error string, handle teardown on failed creation.
#   * the two C programs that write the HDF5 fixtures (tests/golden/hdf5/*.c) compiled with gcc -fsanitize=address,undefined
#     and run into a scratch directory (they link the libhdf5 of /opt/conda when it is there; skipped otherwise).
# Never on the GPU box (GPU AddressSanitizer / XNACK runs are refused there).  Writes profiles/r06_sanitizer_host.log.
# Fails (exit 1) when the build fails, when pytest fails, or when a sanitizer report is found - a status file carries the group's outcome
# out of the `{ ... } | tee` subshell (round-3 advisor: the old script printed 'clean' after a failed build).
set -u
set -o pipefail
cd "$(dirname "$0")/.."
REPO=$(pwd)
OUT=${TMPDIR:-/tmp}/climsim_asan
LOG=$REPO/profiles/r06_sanitizer_host.log
mkdir -p "$OUT"
STATUS=$OUT/status; echo fail > "$STATUS"
{
echo "== host-side ASan + UBSan build of climsim_hip.hip ($(date -u +%Y-%m-%dT%H:%MZ), $(hipcc --version | grep -m1 -i 'hip version'))"
hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -std=c++17 \
      -shared -fPIC -Wall -Wno-unused-function climsim_amd/csrc/climsim_hip.hip -o "$OUT/libclimsim_hip_asan.so" || { echo "BUILD FAILED"; exit 1; }
[ -s "$OUT/libclimsim_hip_asan.so" ] || { echo "BUILD FAILED (no library)"; exit 1; }
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
echo "runtime: $RT"
echo "== tests/test_cabi_cpu.py + tests/test_cabi_args_cpu.py against the sanitized library"
# (test_header_symbols... would rebuild the production .so when sources are newer: harmless; the library under test is CLIMSIM_HIP_LIB)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  CLIMSIM_HIP_LIB="$OUT/libclimsim_hip_asan.so" python -m pytest tests/test_cabi_cpu.py tests/test_cabi_args_cpu.py -q -p no:cacheprovider 2>&1 | tail -15
PYRC=${PIPESTATUS[0]}
echo "rc=$PYRC"
[ "$PYRC" -eq 0 ] || { echo "PYTEST FAILED under the sanitizers"; exit 1; }
echo "== HDF5 fixture writers under gcc -fsanitize=address,undefined"
H5INC=/opt/conda/include; H5LIB=/opt/conda/lib
if [ -f "$H5INC/hdf5.h" ]; then
  for src in make_hdf5_fixtures make_keras_h5_fixture; do
    gcc -fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -I"$H5INC" "tests/golden/hdf5/$src.c" -L"$H5LIB" -lhdf5_hl -lhdf5 -Wl,-rpath,"$H5LIB" -o "$OUT/$src" 2>&1 | tail -3
    (cd "$OUT" && ASAN_OPTIONS=detect_leaks=0 ./$src > "$OUT/$src.out" 2>&1; echo "$src: rc=$? ($(wc -l < "$OUT/$src.out") lines of output, $(grep -c -i 'runtime error\|AddressSanitizer' "$OUT/$src.out") sanitizer reports)")
  done
else
  echo "libhdf5 headers not found under /opt/conda: skipped"
fi
echo ok > "$STATUS"
} 2>&1 | tee "$LOG"
[ "$(cat "$STATUS")" = ok ] || { echo "SANITIZER PASS DID NOT COMPLETE (build or pytest failed: see $LOG)"; exit 1; }
grep -q "AddressSanitizer\|runtime error" "$LOG" && { echo "SANITIZER REPORTS FOUND"; exit 1; }
echo "clean"
