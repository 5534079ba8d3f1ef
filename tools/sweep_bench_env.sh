# usage: tools/sweep_bench_env.sh VAR v1 v2 ...   bench step time per value of a tuning variable (development aid)
var=$1; shift
for v in "$@"; do
export $var=$v
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-configs > gpurun_out/ab.json 2>/dev/null || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/ab.json')); print('$var=$v', round(b['ms_per_step'],3), round(b['roofline']['launch_ms_avg'],4))"
done
