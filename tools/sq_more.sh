R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/sqmore; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
i=0
for GROUP in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/g$i -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 1 > $D/g$i.log 2>&1 || { echo "group $i failed: $GROUP" >> $D/errors.txt; continue; }
  (cd $R && python3 tools/pmc_table.py $(ls -t $D/g$i/*/*counter_collection.csv | head -1) k_count_slices k_core k_partition) > $D/sq_more_$i.txt
  rm -rf $D/g$i
done
cat $D/sq_more_*.txt; cat $D/errors.txt 2>/dev/null
