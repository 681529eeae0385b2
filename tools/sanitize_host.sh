#!/bin/bash
# Host code of libxsmm_amd.so under AddressSanitizer + UBSan, no GPU needed: builds an instrumented copy of the library under /tmp
# (device code is left alone: -fno-gpu-sanitize), swaps it in for the duration of the CPU test-suite and of examples/registry_check.c in
# dry-run mode (capacity, exhaustion, hit path, init / finalize cycles; with leak detection), and restores the real library afterwards.
# Round 2: 766 CPU tests (incl. the dry-run code generators) and the registry program ran without a report (the few CPU tests that start children with a cleaned environment or
# link a C program with gcc cannot preload the sanitizer runtime and are not counted); the threaded dispatch check under ThreadSanitizer: clean.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/libxsmm_amd_asan}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
mkdir -p "$OUT/obj" "$OUT/lib"
cd "$ROOT/libxsmm_amd/csrc" || exit 1
# the five HOST translation units are instrumented; the kernel translation units (device code + their launch stubs) are taken as the shipped build compiled them
# (libxsmm_amd/lib/obj: run `make -C libxsmm_amd/csrc` first) -- since round 5 they are too many (shards, per-kind units) to rebuild for a host check
for f in runtime.cpp frontend.cpp utils.cpp jit.cpp meqn.cpp; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o "$OUT/obj/$f.o" & done
wait
KOBJ=$(ls "$ROOT"/libxsmm_amd/lib/obj/*.o | grep -v "\.cpp\.o$" | grep -v mono)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize "$OUT"/obj/*.cpp.o $KOBJ -ldl -o "$OUT/lib/libxsmm_amd.so" || exit 1

REAL="$ROOT/libxsmm_amd/lib/libxsmm_amd.so"
cp "$REAL" "$OUT/real.so"
trap 'cp "$OUT/real.so" "$REAL"' EXIT
cp "$OUT/lib/libxsmm_amd.so" "$REAL"
cd "$ROOT" || exit 1
rm -f "$OUT"/asan.* "$OUT"/ubsan.*
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$OUT/asan UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/ubsan LD_PRELOAD=$RT \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_kernel_resources_cpu.py | tail -3

gcc -std=c99 -D_POSIX_C_SOURCE=200809L -O1 -g -I"$ROOT/include" "$ROOT/examples/registry_check.c" -L"$OUT/lib" -lxsmm_amd -Wl,-rpath,"$OUT/lib" \
  -Wl,--allow-shlib-undefined -o "$OUT/registry_check" || exit 1
export ASAN_OPTIONS=detect_leaks=1:halt_on_error=0:log_path=$OUT/asan UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/ubsan LIBXSMM_HIP_DRYRUN=1
run() { env LD_PRELOAD=$RT "$@" | tail -1; }          # the preload (and its leak check) for the program only, not for the pipeline's tail
run "$OUT/registry_check" capacity 136072 131072
run "$OUT/registry_check" hit 200000
run "$OUT/registry_check" cycle
run "$OUT/registry_check" info
LIBXSMM_HIP_MAX_HANDLES=500 run "$OUT/registry_check" capacity 700 500
LIBXSMM_HIP_THUNKS=0 run "$OUT/registry_check" capacity 300 256
unset LD_PRELOAD

# the registry's miss path (lock), hit path (thread cache) and thunk pool under ThreadSanitizer: eight threads, the same descriptors
TS="$OUT/tsan"; mkdir -p "$TS/obj" "$TS/lib"
TFLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -fsanitize=thread -fno-gpu-sanitize"
cd "$ROOT/libxsmm_amd/csrc" || exit 1
for f in runtime.cpp frontend.cpp utils.cpp jit.cpp meqn.cpp; do /opt/rocm/bin/hipcc $TFLAGS -x hip -c $f -o "$TS/obj/$f.o" & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize "$TS"/obj/*.cpp.o $KOBJ -ldl -o "$TS/lib/libxsmm_amd.so" || exit 1
gcc -std=c99 -D_POSIX_C_SOURCE=200809L -O1 -g -I"$ROOT/include" "$ROOT/examples/registry_check.c" -L"$TS/lib" -lxsmm_amd -Wl,-rpath,"$TS/lib" \
  -Wl,--allow-shlib-undefined -lpthread -o "$TS/registry_check" || exit 1
TSAN_OPTIONS=halt_on_error=0:log_path=$OUT/tsan_report LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1) \
  LIBXSMM_HIP_DRYRUN=1 "$TS/registry_check" threads 8 5000 | tail -1
echo "sanitizer reports:"; ls "$OUT"/asan.* "$OUT"/ubsan.* "$OUT"/tsan_report.* 2>/dev/null || echo "  none"
