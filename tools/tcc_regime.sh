#!/bin/bash
# usage (on the GPU box): tools/tcc_regime.sh <out-dir> <processes>
# Per-channel L2 <-> memory counters of the pass in several fresh processes (each lands in whatever placement regime its allocations
# fall into): three --pmc passes per process would mean three different placements, so every process collects ONE counter group and
# the kernel durations of the very same dispatches come from its own kernel trace.  tools/tcc_channels.py prints the tables.
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/$1; N=${2:-6}; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
GROUPS_=("TCC_EA0_WRREQ TCC_EA0_RDREQ TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ_LEVEL" "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ TCC_BUSY")
for i in $(seq 1 $N); do
  g=$(( (i - 1) % 3 ))
  rocprofv3 --kernel-trace --pmc ${GROUPS_[$g]} --output-format json -d $D/p$i -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 2 > $D/p$i.log 2>&1 || { echo "process $i failed" >> $D/errors.txt; continue; }
  grep "^variant" $D/p$i.log > $D/p$i.tune.txt
  (cd $R && python3 tools/tcc_channels.py $D/p$i) > $D/p$i.channels.txt
  rm -rf $D/p$i
done
# for comparison: the same processes without a profiler
for i in $(seq 1 3); do python3 $R/tools/tune.py --variants 4 --rounds 2 --steps 3 2>/dev/null | grep "^variant" >> $D/unprofiled.txt; done
ls $D
