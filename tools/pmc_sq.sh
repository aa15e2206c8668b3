#!/bin/bash
# usage (GPU box): tools/pmc_sq.sh <name> "<counters...>"   -> per-kernel means of the given SQ counters on one tune.py step
R=$GRAFT_REPO_ROOT; name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $R/gpurun_out/$name -- python3 $R/tools/tune.py --variants 4 --rounds 1 --steps 1 > $R/gpurun_out/$name.log 2>&1 || exit 1
cd $R && python3 tools/pmc_table.py $(ls -t gpurun_out/$name/*/*counter_collection.csv | head -1) k_count_slices k_core k_cp_ k_partition
