#!/bin/bash
# Per-kernel times of the training leg alone (run on the GPU box from the repository root):
#   tools/train_stats.sh <tag>   ->  gpurun_out/<tag>/train_kernel_stats.csv, train_bench.json
set -e
TAG=${1:-train}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 $R/bench.py --steps 1 --warmup 0 --rays 131072 --cpu-rays 0 --strong-leg 0 --train-steps 40 > $O/train_bench.json 2> $O/train_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --rays 131072 --cpu-rays 0 --strong-leg 0 --train-steps 20 > $O/train_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/train_kernel_stats.csv
python3 -c "
import json;d=json.load(open('$O/train_bench.json'))['train'];print(d)"
head -12 $O/train_kernel_stats.csv | cut -c1-160
