#!/bin/bash
# Kernel-trace statistics of a short bench run (development aid).  usage (on the GPU box): bash tools/kstats.sh <tag>
# -> gpurun_out/<tag>_kstats.csv
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-configs > $R/gpurun_out/${tag}_kbench.json 2> $R/gpurun_out/${tag}_rocprof.log
cp $R/gpurun_out/${tag}_trace/t_kernel_stats.csv $R/gpurun_out/${tag}_kstats.csv
rm -rf $R/gpurun_out/${tag}_trace
head -6 $R/gpurun_out/${tag}_kstats.csv | cut -c1-150
