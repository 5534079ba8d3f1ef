# usage: tools/ab_env.sh <lib> VAR=value ...   one bench line per setting of the environment, same library (development aid)
lib=$1; shift
for kv in "$@"; do
env $kv PTK_LIBRARY=$lib timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-configs > gpurun_out/ab.json 2>/dev/null || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/ab.json')); print('$lib $kv', round(b['value'],1), round(b['ms_per_step'],3), round(b['roofline']['launch_ms_avg'],4))"
done
