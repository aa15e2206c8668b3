#!/bin/bash
# usage (on the GPU box, through gpurun): tools/regime_probe.sh <n-processes>
# The pass runs in a "fast" or a "slow" regime from process to process (K1 and K2 both ~10 % apart, the core passes unchanged).
# For every process: durations of K1 / K2 and a few memory-side counters, to see what differs.
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/regime; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
for i in $(seq 1 $1); do
  for GROUP in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $D/p; rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/p -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 1 > $D/p.log 2>&1 || { echo failed; continue; }
    python3 - $D/p <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:16]
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:16]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_partition<2>", "k_count_slices<1"):
    if k in dur:
        print(k, "us %.0f" % (sum(dur[k][-2:]) / len(dur[k][-2:])), " ".join("%s=%.3g" % (c, sum(v[-2:]) / len(v[-2:])) for c, v in sorted(acc[k].items())))
PY
  done
done
