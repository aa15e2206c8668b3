#!/bin/bash
# usage (GPU box, through gpurun): tools/box_sample.sh <tag>   — the 100M-read pass in three fresh processes on this box: one line per process
# (kernel times and ms per step), appended to gpurun_out/r04/boxes/<tag>.txt with the box's name and GPU id
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/r04/boxes; mkdir -p $D
{ echo "box $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -m1 -o '0x[0-9a-f]*')"; for i in 1 2 3; do python3 $R/tools/tune.py --variants 4 --rounds 2 --steps 5 2>/dev/null | grep "^variant"; done; } > $D/$1.txt
cat $D/$1.txt
