#!/bin/bash
# Calibrates FETCH_SIZE for 4-byte gathers (traffic).  Runs on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/calib; mkdir -p $out; rm -rf $out/pmc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/calib_fetch.hip -o $out/calib_fetch || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -- $out/calib_fetch > $out/run.log 2>&1
echo "exit $?"; grep expected $out/run.log
python3 - $out/pmc <<'PY' | tee $out/calibration.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
per = {}
for r in rows:
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    name = r["Kernel_Name"].split("(")[0]
    per.setdefault((name, r.get("Grid_Size", "")), []).append(float(r["Counter_Value"]))
GiB = 4 << 30
expect = {"gather_one_per_line": None, "gather_two_per_line": (GiB // 128 // 4) * 128, "stream_read": GiB // 4}
for (name, grid), vals in sorted(per.items()):
    kb = sum(vals) / len(vals)
    print(f"{name:24s} grid {grid:>10s}: FETCH_SIZE {kb:14.0f} KB = {kb * 1024:.4g} B per launch ({len(vals)} launches)")
PY
rm -rf $out/pmc $out/calib_fetch
