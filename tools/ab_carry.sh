# sweep of the carry-over knobs on the bench scene (development aid)
for lag in 2 3; do for below in 16 12 8 4; do
PTK_WF_MAX_LAG=$lag PTK_WF_PARK_BELOW=$below timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-configs > gpurun_out/ab.json 2>/dev/null || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/ab.json')); print('lag $lag park_below $below', round(b['value'],1), round(b['ms_per_step'],3), round(b['roofline']['launch_ms_avg'],4), b['roofline']['launches_per_step'])"
done; done
