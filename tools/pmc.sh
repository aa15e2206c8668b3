#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc.sh     -> gpurun_out/pmc_{fetch,write}_{1mm,exact}/ + profiles/pmc_traffic.json
# HBM traffic of the count pipeline: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in 1mm exact; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$W -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload $W > $R/gpurun_out/pmc_fetch_$W.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$W -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload $W > $R/gpurun_out/pmc_write_$W.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_1mm gpurun_out/pmc_write_1mm gpurun_out/pmc_fetch_exact gpurun_out/pmc_write_exact &&
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
