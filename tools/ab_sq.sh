#!/bin/bash
# usage (GPU box): tools/ab_sq.sh <out-dir> <lib.so>...   — SQ instruction/lane counters of the count kernels for builds of the library
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/$1; shift; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  n=$(basename $L .so)
  i=0
  for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/$n.$i -- python3 $R/tools/tune.py --lib-path $R/$L --variants 4 --rounds 1 --steps 1 > $D/$n.$i.log 2>&1 || { echo "$n group $i failed" >> $D/errors.txt; continue; }
    (cd $R && python3 tools/pmc_table.py $(ls -t $D/$n.$i/*/*counter_collection.csv | head -1) k_count_slices k_core k_partition) > $D/$n.sq$i.txt
    rm -rf $D/$n.$i
  done
done
