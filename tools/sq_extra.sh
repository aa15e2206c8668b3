#!/bin/bash
# usage (GPU box, through gpurun): tools/sq_extra.sh [lib.so]   — instruction-fetch side of the count kernels: SQC instruction-cache requests,
# busy cycles and back-pressure, branches, scalar-unit cycles (one --pmc pass per group); output under gpurun_out/r04/sqx/
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/r04/sqx; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
LIB=${1:+--lib-path $R/$1}
i=0
for GROUP in "SQC_ICACHE_REQ SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VSKIPPED"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/g$i -- python3 $R/tools/tune.py $LIB --variants 4 --rounds 1 --steps 1 > $D/g$i.log 2>&1 || { echo "group $i failed: $GROUP" >> $D/errors.txt; continue; }
  (cd $R && python3 tools/pmc_table.py $(ls -t $D/g$i/*/*counter_collection.csv | head -1) k_count_slices k_core k_partition) > $D/sqx_$i.txt
  rm -rf $D/g$i
done
cat $D/sqx_*.txt; cat $D/errors.txt 2>/dev/null
