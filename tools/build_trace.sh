# kernel trace of repeated device builds of one scene: bash tools/build_trace.sh <scene> <reps>  -> gpurun_out/bt_<scene>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bt_$1 -o bt --output-format csv -- python3 $R/tools/build_probe.py $1 $2 > $R/gpurun_out/bt_$1.log 2>&1
grep build $R/gpurun_out/bt_$1.log | tail -2
