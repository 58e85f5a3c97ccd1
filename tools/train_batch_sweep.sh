#!/bin/bash
# The training step at several batch sizes (run on the GPU box from the repository root):
#   tools/train_batch_sweep.sh  ->  gpurun_out/train_batch_sweep.txt (N_rand, ms per step, rays/s, fraction of the MFMA peak)
set -e
O=gpurun_out/train_batch_sweep.txt
: > $O
for n in 1024 2048 4096 8192 16384 32768; do
  timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --rays 131072 --cpu-rays 0 --strong-leg 0 --train-steps 40 --n-rand $n > gpurun_out/sweep_$n.json 2> gpurun_out/sweep_$n.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/sweep_$n.json'))['train']
print('N_rand %6d  %7.3f ms/step  %9.0f rays/s  %.3f of the MFMA peak  %.3f of the HBM peak' % ($n, d['ms_per_step'], d['rays_per_s'], d['frac_of_mfma_peak'], d['frac_of_hbm_peak']))" >> $O
done
cat $O
