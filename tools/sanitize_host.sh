#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer pass over the C ABI's argument paths (SURVEY 5 stance).
# Every tbx_* entry point validates pointers / sizes / alignment and fills launch descriptors on the HOST before it enqueues a
# kernel; this builds libtbx_hip.so with the host halves instrumented (-fsanitize=address,undefined; device code left alone:
# -fno-gpu-sanitize - GPU ASan / XNACK are not available on this pool) and runs the no-GPU tests against it
# (tests/test_abi_and_host.py: symbol table, argument validation of every entry-point family, ctypes struct layouts against gcc's,
# host schedules). Runs in the container (no GPU needed): hipcc cross-compiles, the tests make no compute calls.
#   tools/sanitize_host.sh            # build into /tmp/tbx_asan, run, print the summary; non-zero exit on any report
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TBX_ASAN_DIR:-/tmp/tbx_asan}
src=$root/trafficbotsv1.5_amd/csrc
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$out"
flags="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wall -Wno-unused-function"
srcs=$(sed -n 's/^SRCS = //p' "$src/Makefile")
objs=""
newest_dep() { ls -t "$src"/*.h "$src"/*.inc "$root"/include/*.h "$1" | head -1; }  # (a header changed: every object is stale)
throttle() { while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 0.2; done; }  # at most 4 compiles at a time (8 CPUs, ~1 GB each)
for f in $srcs; do
  o=$out/${f%.hip}.o
  [ "$o" -nt "$(newest_dep "$src/$f")" ] || $hipcc $flags -c "$src/$f" -o "$o" &
  objs="$objs $o"
  throttle
done
for f in tile_layer tile_heads tile_window tall_linear tile_tail; do
  o=$out/${f}_p1.o
  [ "$o" -nt "$(newest_dep "$src/$f.hip")" ] || $hipcc $flags -DTBX_TILE_SINGLE=1 -c "$src/$f.hip" -o "$o" &
  objs="$objs $o"
  throttle
done
wait
$hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize $objs -o "$out/libtbx_hip.so"
asan_rt=$($hipcc -print-file-name=libclang_rt.asan-x86_64.so)
# python itself is not instrumented: the ASan runtime is preloaded; leak checking off (the interpreter's own arenas), everything else fatal
export LD_PRELOAD=$asan_rt
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:exitcode=99
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export TBX_HIP_LIB=$out/libtbx_hip.so
cd "$root"
set +e
python -m pytest tests/test_abi_and_host.py -x -q -p no:cacheprovider 2>&1 | tee "$out/report.txt"
rc=${PIPESTATUS[0]}
set -e
n_ub=$(grep -c 'runtime error:' "$out/report.txt" || true)
n_as=$(grep -c 'ERROR: AddressSanitizer' "$out/report.txt" || true)
if [ "$n_ub" != 0 ] || [ "$n_as" != 0 ]; then rc=99; fi
echo "[sanitize_host] exit $rc ($n_ub UBSan reports, $n_as ASan reports; library $out/libtbx_hip.so)"
exit $rc
