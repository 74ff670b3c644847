#!/bin/bash
# Instruction mix of the trace kernel variants (cross-compiles; no GPU needed).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/asm}
mkdir -p $OUT
cd $ROOT/turtle_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I../../include -I. \
    -S --cuda-device-only device.hip -o $OUT/device.s 2>/dev/null
cd $OUT
for k in $(grep -oE "^_ZN12_GLOBAL__N_1[0-9]+k_(trace|step)[A-Za-z0-9_]+:" device.s | tr -d ':'); do
  ln=$(grep -n "^$k:" device.s | head -1 | cut -d: -f1)
  awk -v s=$ln 'NR>=s{print} NR>s && /s_endpgm/{exit}' device.s > $k.s
  echo "$k total=$(grep -cE '^\s+[a-z]' $k.s) f64=$(grep -cE '^\s+v_[a-z_0-9]+_f64' $k.s) valu=$(grep -cE '^\s+v_' $k.s) salu=$(grep -cE '^\s+s_' $k.s) vmem=$(grep -cE '^\s+(global|flat|buffer)_' $k.s)"
done
