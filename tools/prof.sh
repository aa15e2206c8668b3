#!/bin/bash
# usage (on the GPU box, through gpurun): tools/prof.sh <name> <program args...>   e.g. tools/prof.sh r01_v4 bench.py --steps 5
# rocprofv3 kernel-trace + stats of `python3 <args>`; prints the per-kernel summary, keeps the csv under gpurun_out/<name>/
name=$1; shift
prog=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $prog "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$name" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/*/*kernel_stats.csv" % sys.argv[1])
for r in csv.DictReader(open(f[0])):
    print("%-90s calls %6s avg %10.1f us  total %6.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
