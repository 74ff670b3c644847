#!/bin/bash
# Register / scratch / occupancy of every kernel of device.hip (runs anywhere hipcc does).
cd "$(dirname "$0")/.."
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iturtle_amd/csrc \
  -c turtle_amd/csrc/device.hip -o /tmp/device_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r'remark: (.*?)\s*\[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        if name: print(name[:70].ljust(70), row)
        name = t.split(':', 1)[1].strip(); row = {}
    else:
        k, _, val = t.partition(':')
        if k.strip() in ('VGPRs', 'AGPRs', 'SGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]'):
            row[k.strip().split(' ')[0]] = val.strip()
if name: print(name[:70].ljust(70), row)
" | grep ${1:-k_}
