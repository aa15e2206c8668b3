#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc_sq.sh <out-dir under gpurun_out>
# SQ counters of the count kernels, one --pmc pass per group (the three groups of tools/evidence.sh + tools/sq_more.sh's six)
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/$1; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
i=0
for GROUP in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/g$i -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 1 > $D/g$i.log 2>&1 || { echo "group $i failed: $GROUP" >> $D/errors.txt; continue; }
  (cd $R && python3 tools/pmc_table.py $(ls -t $D/g$i/*/*counter_collection.csv | head -1) k_count_slices k_core k_partition) > $D/sq_$i.txt
  rm -rf $D/g$i
done
cat $D/sq_*.txt; cat $D/errors.txt 2>/dev/null
