#!/bin/bash
# The host C objects under AddressSanitizer + UBSan, CPU tests only (no GPU needed; sanitizers are not
# available on the GPU pool): builds /tmp/asan/libturtle_amd.so (host objects instrumented, the device
# object as built) and runs the host-side tests against it.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $ROOT/turtle_amd/csrc >/dev/null
mkdir -p /tmp/asan
cd $ROOT/turtle_amd/csrc
for f in *.c; do
  gcc -O1 -g -std=gnu99 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I. -c $f -o /tmp/asan/${f%.c}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/asan/libturtle_amd.so /tmp/asan/*.o build/device.o -lm -lz -lpthread -fsanitize=address,undefined
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1 TURTLE_AMD_LIBRARY=/tmp/asan/libturtle_amd.so \
  python -m pytest tests/test_host_scalar.py tests/test_host_logic.py tests/test_cabi_symbols.py -x -q
