#!/bin/bash
# usage (on the GPU box): tools/pad_sweep.sh <out-file>   — block-stride padding of the pool (PART_PAD, 8-byte words) against the
# placement regimes: a physically contiguous pool (the deterministic slow case) and two ordinary allocations, per padding
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
cd $R
for PAD in 0 16 32 64 160; do
  SGC_HIPCC_FLAGS=-DPART_PAD=${PAD}u python3 -c "from sgcount_amd import build as b; b.build_one(b.SO, force=True)" >> $OUT 2>&1 || exit 1
  echo "== PART_PAD $PAD (stride $((4096 + 8 * PAD)) bytes)" >> $OUT
  python3 tools/regime_sweep.py vmm_chunk_kb=1 vmm_chunk_mb=0 vmm_chunk_mb=0 2>&1 | grep -v amdgpu >> $OUT
done
python3 -c "from sgcount_amd import build as b; b.build_one(b.SO, force=True)" >> $OUT 2>&1
cat $OUT
