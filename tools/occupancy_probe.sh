set -e
mkdir -p gpurun_out/r3t
for k in NONE K1 K2 CORE; do
  echo "== one workgroup per CU: $k"
  env SGC_EXTRA_LDS_$k=45000 python3 tools/tune.py --variants 4 --rounds 2 --steps 3 --nocheck 2>&1 | grep -E "^variant|rror"
done
