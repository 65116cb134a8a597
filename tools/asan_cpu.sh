#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libegotap_hip.so (round-2 verdict item 9).
# The ~3 k lines of host planning in csrc/egotap_abi.hip (workspace layouts, parameter resolution, pointer arithmetic, argument checks)
# are compiled with -fsanitize=address,undefined for the HOST pass only (-Xarch_host; the device code is built as usual) and the CPU
# test-suite's ABI tests run against that library: everything the ABI does before its first kernel launch -- create / bind /
# resolve / workspace sizing / intermediates / the error paths of every entry point.  CPU build container only: GPU sanitizers are
# not available on the pool.  Exit code 0 = no sanitizer report.
#   usage: tools/asan_cpu.sh        (about 5 minutes: four translation units; host pass at -O1 -g)
set -euo pipefail
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=${EGOTAP_ASAN_DIR:-/tmp/egotap_asan}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || { echo "asan runtime not found ($RT)"; exit 2; }
# [r5] the DEVICE pass is compiled exactly as the product is (-O3, no debug info): at -O1 -g hipcc 7.2's ADCE pass crashes on gemm_bf16s64_kernel;
# only the host pass gets -O1 -g and the sanitizers
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$REPO/include -I$REPO/egotap_amd/csrc -Xarch_host -O1 -Xarch_host -g -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined"
pids=()
for part in 0 1 2 3; do
  $HIPCC $FLAGS -DEGOTAP_PART=$part -c egotap_amd/csrc/egotap_abi.hip -o "$OUT/part$part.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT/libegotap_hip_asan.so" "$OUT"/part?.o
echo "built $OUT/libegotap_hip_asan.so"
# python itself is not instrumented: preload the sanitizer runtime; leak checking off (the interpreter never frees its arenas)
export EGOTAP_LIB="$OUT/libegotap_hip_asan.so"
export LD_PRELOAD="$RT"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
python -m pytest tests/test_abi_cpu.py -x -q -p no:cacheprovider
echo "asan_cpu: OK (no sanitizer report)"
