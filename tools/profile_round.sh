#!/bin/bash
# Profile set of a round (run on the GPU box from the repository root, e.g. through gpurun):
#   tools/profile_round.sh r02
# kernel stats of the default bench, three PMC passes over render + training kernels (each in a
# run of its own), the un-profiled line.  Results land in gpurun_out/<round>/; pmc_summary.txt
# there starts with the hash of the MLP kernel's sources, which bench.py checks before quoting
# the traffic figure.  A step that fails ends the script: nothing else is started on the GPU.
set -e
ROUND=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$ROUND
rm -rf $O && mkdir -p $O
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 > $O/bench_under_rocprof.json 2> $O/stats.err
B="python3 $R/bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 3 --rays 262144 --strong-leg 0"
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.out 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.out 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.out 2>&1
cd $R
{
  python3 -c "import bench; print('# source_hash', bench.kernel_source_hash())"
  for p in sq fetch write; do
    f=$(ls $O/pmc_$p/*/*counter_collection.csv | head -1)
    echo "== pass $p: $B"
    python3 tools/pmc_summary.py $f
  done
} > $O/pmc_summary.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
timeout -k 10 280 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
ls $O
