#!/bin/bash
# build_variant.sh <name> [-DFLAG ...]: a build of the library with extra flags, as variants/<name>.so
# (no GPU needed; for A/B runs with TURTLE_AMD_LIBRARY)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $ROOT/variants $ROOT/turtle_amd/csrc/build
make -s -C $ROOT/turtle_amd/csrc >/dev/null
obj=$ROOT/turtle_amd/csrc/build/device_$name.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I$ROOT/include -I$ROOT/turtle_amd/csrc "$@" \
    -c ${SRC:-$ROOT/turtle_amd/csrc/device.hip} -o $obj
others=$(ls $ROOT/turtle_amd/csrc/build/*.o | grep -v "/device")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/variants/$name.so $others $obj -lm -lz
echo $ROOT/variants/$name.so
