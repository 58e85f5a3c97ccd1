#!/bin/bash
# What bounds the inference MLP kernel, measured on the GPU box in one call (variants built
# beforehand with tools/build_variant.sh; see the names below).  Writes gpurun_out/fwd_bounds/.
#   bare MFMA loops -> mfma_ceiling.txt;  inner-loop micro-benchmark -> mfma_mix.txt
#   in-kernel clock, per-layer cycles and timing-only knock-outs -> knockouts.txt
#   two against three point tiles per wave, bit equality and time -> ab.txt
set -e
O=gpurun_out/fwd_bounds
mkdir -p $O
timeout -k 10 200 tools/bin/mfma_ceiling > $O/mfma_ceiling.txt
timeout -k 10 200 tools/bin/mfma_mix > $O/mfma_mix.txt
: > $O/knockouts.txt
for v in n2s n2stale n2nolds n3s n3stale n3nolds; do
  echo "=== $v" >> $O/knockouts.txt
  timeout -k 10 120 python3 tools/clock_probe.py tools/bin/libnerfhip_$v.so ${v:1:1} 2>/dev/null >> $O/knockouts.txt
done
timeout -k 10 200 python3 tools/ab_fwd_builds.py tools/bin/libnerfhip_n2.so tools/bin/libnerfhip_n3.so 2>/dev/null > $O/ab.txt
tail -4 $O/ab.txt
