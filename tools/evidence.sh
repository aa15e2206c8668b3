#!/bin/bash
# usage (on the GPU box, through gpurun): tools/evidence.sh <round-dir>      e.g. tools/evidence.sh r02
# Collects everything DESIGN.md §6 cites into gpurun_out/<round-dir>/ (copy it to profiles/<round-dir>/ afterwards):
#   kernel_stats_{1mm,exact}.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 10` (count pipeline; all dispatches, the placement
#                                  trials' included) + kernel_stats_timed_*.csv: the last 10 dispatches of every kernel = the timed region
#   kernel_stats_ingest.csv        the same of tools/tune_ingest.py on 10M reads (FASTQ ingest kernels) + ingest.txt (TB/s of text)
#   pmc_traffic.json               HBM bytes per pass from separate --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
#   sq_*.txt                       SQ counters of the count kernels (tools/pmc_table.py), one --pmc pass per group
#   stamps.txt                     s_memtime phase stamps of k_count_slices / k_core (dbg 512)
#   wg_timeline.txt                per kernel: lifetimes of the workgroups, slot use, phases (dbg 1048576, tools/wg_timeline.py)
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/$1; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 10 --warmup 2 --cpu-seconds 0 --e2e-reads 0 --placement-trials 0 --dominant 0 --other-configs 0 --multi-sample-reads 0"
for W in 1mm exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_$W -- python3 $R/bench.py $BENCH --workload $W > $D/bench_prof_$W.log 2>&1 || exit 1
  cp $(ls -t $D/prof_$W/*/*kernel_stats.csv | head -1) $D/kernel_stats_$W.csv
  python3 $R/tools/trace_timed.py $(ls -t $D/prof_$W/*/*kernel_trace.csv | head -1) 10 > $D/kernel_stats_timed_$W.csv
  grep '^{' $D/bench_prof_$W.log > $D/bench_under_rocprof_$W.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_ingest -- python3 $R/tools/tune_ingest.py --reads 10000000 > $D/ingest_prof.log 2>&1 || exit 1
cp $(ls -t $D/prof_ingest/*/*kernel_stats.csv | head -1) $D/kernel_stats_ingest.csv
grep -e '-> records' $D/ingest_prof.log > $D/ingest.txt
for W in 1mm exact; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch_$W -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-reads 0 --placement-trials 0 --dominant 0 --other-configs 0 --multi-sample-reads 0 --workload $W > $D/pmc_fetch_$W.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_write_$W -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-reads 0 --placement-trials 0 --dominant 0 --other-configs 0 --multi-sample-reads 0 --workload $W > $D/pmc_write_$W.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_traffic.py $D/pmc_fetch_1mm $D/pmc_write_1mm $D/pmc_fetch_exact $D/pmc_write_exact "$1" > $D/pmc_traffic.txt && cp profiles/pmc_traffic.json $D/pmc_traffic.json
cd /tmp
i=0
for GROUP in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/sq$i -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 1 > $D/sq$i.log 2>&1 || { echo "counter group $i failed" >> $D/sq_errors.txt; continue; }
  (cd $R && python3 tools/pmc_table.py $(ls -t $D/sq$i/*/*counter_collection.csv | head -1) k_count_slices k_core k_partition k_export) > $D/sq_$i.txt
done
# the phase stamps are compiled out of the shipped kernels (they cost scalar registers): a second build of the library carries them
# (sgcount_amd/build.py STAMPS_SO, built on the spot if stale; the shipped library is not touched)
cd $R && python3 tools/tune.py --lib stamps --variants "4:dbg=512" --rounds 1 --steps 1 --nocheck 2>&1 | grep -E "^K2 wg|^k_core" > $D/stamps.txt
# ... and the workgroup timelines of the last of four back-to-back passes (where every workgroup ran, when, and its phases)
python3 tools/tune.py --lib stamps --variants "4:dbg=1048576" --rounds 1 --steps 4 --notiming --nocheck > $D/tl_raw.txt 2>&1 && python3 tools/wg_timeline.py $D/tl_raw.txt > $D/wg_timeline.txt; rm -f $D/tl_raw.txt
rm -rf $D/prof_* $D/pmc_fetch_* $D/pmc_write_* $D/sq[0-9] 2>/dev/null
ls -la $D
