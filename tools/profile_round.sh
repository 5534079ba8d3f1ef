#!/bin/bash
# One round of measurements for profiles/: the bench line, the same command under rocprofv3 --kernel-trace --stats,
# the counter passes and the configuration table.  usage (on the GPU box): bash tools/profile_round.sh <tag>
# -> gpurun_out/<tag>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, _config_table.md, pmc_<tag>.json
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python3 -u bench.py > gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-configs > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_rocprof.log
cp $R/gpurun_out/${tag}_trace/t_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
rm -f $R/gpurun_out/${tag}_trace/t_kernel_trace.csv
cd $R
timeout -k 10 300 python3 tools/config_table.py > gpurun_out/${tag}_config_table.md
bash tools/pmc_passes.sh $tag
