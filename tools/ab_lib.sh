# usage: tools/ab_lib.sh <libA> <libB> ...   bench step and light-scene pipeline times per library, twice each
for round in 1 2; do
for lib in "$@"; do
export PTK_LIBRARY=$lib
timeout -k 10 200 python -u bench.py --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
b=$(python3 -c "
import json; b=json.load(open('gpurun_out/ab.json')); print(round(b['value'],1), round(b['ms_per_step'],3), round(b['roofline']['launch_ms_avg'],4))")
l=""
for s in fake_cornell_box benchmark_random_scene dragon_test; do
timeout -k 10 100 python tools/scene_profile.py $s 2 > gpurun_out/sp.txt || exit 1
l="$l $(head -1 gpurun_out/sp.txt | sed 's/.*dispatch //')"
done
echo "$lib bench $b | cornell random dragon:$l"
done
done
