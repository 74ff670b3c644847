#!/bin/bash
# Counters of the passes of a large trace (k_flow<MODE, FLOW>), per dispatch in launch order
# usage: exp_flow_pmc.sh [workload] ["ENV=.."]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=${1:-c4}; [ -n "$2" ] && export $2
out=gpurun_out/flow_pmc; rm -rf $out; mkdir -p $out
run() { name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- \
      python3 bench.py --workload $wl --also none --steps 2 --warmup 1 --no-cpu --in-flight 1 > $out/pmc_$name.log 2>&1; echo "pmc $name exit $?"; }
run A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run B SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_flow" not in name and "k_trace" not in name and "k_cross" not in name: continue
        key = (int(r["Dispatch_Id"]), name[name.find("k_"):][:28])
        rows.setdefault(key, {})[r["Counter_Name"]] = rows.get(key, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
# the LAST trace of each run: group dispatch ids by pass order
keys = sorted(rows)
per = [k for k in keys]
names = ["SQ_WAVES","SQ_INSTS_VALU","SQ_WAVE_CYCLES","GRBM_GUI_ACTIVE","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_ACTIVE_INST_VALU","SQ_THREAD_CYCLES_VALU","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VMEM"]
print("dispatch kernel " + " ".join(names))
for k in keys[-24:]:
    d = rows[k]
    lanes = d.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * d["SQ_ACTIVE_INST_VALU"]) if d.get("SQ_ACTIVE_INST_VALU") else float("nan")
    print(k[0], k[1], " ".join("%.3g" % d.get(n, float("nan")) for n in names), "lanes %.2f" % lanes)
PY
