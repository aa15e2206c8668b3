mkdir -p gpurun_out/r04
for wu in 1 3 4; do
  python tools/tune.py --lib-path sgcount_amd/libsgcount_hip_wu$wu.so --variants 4 --rounds 3 --steps 5 2>&1 | grep "^variant" | sed "s/^/WU=$wu /"
done
python tools/tune.py --variants 4 --rounds 3 --steps 5 2>&1 | grep "^variant" | sed "s/^/WU=2 /"
