#!/bin/bash
# Host-side sanitizer run (GPU AddressSanitizer is not available on this pool): libmp2gpu built with ASan + UBSan on the HOST code
# only (-fno-gpu-sanitize), then the CPU tests that exercise host code of the library -- the work plan, the bincode wire format,
# the witness-program executor -- run against it with the ASan runtime preloaded.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg/asan
CLANG=/opt/rocm/lib/llvm/bin/clang
for f in mapreduce-plonky2_amd/csrc/*.hip; do
  o=build_dbg/asan/$(basename ${f%.hip}).o
  [ "$o" -nt "$f" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Iinclude -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -c "$f" -o "$o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -fno-gpu-sanitize -o build_dbg/asan/libmp2gpu_asan.so build_dbg/asan/*.o
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 MP2G_LIB=$PWD/build_dbg/asan/libmp2gpu_asan.so
python -m pytest tests/test_workplan.py tests/test_wire_host.py tests/test_recursion.py tests/test_witness_tape.py -q -x -k "not map_reduce_with" "$@"
