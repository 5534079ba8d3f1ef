# usage: tools/sweep_env.sh VAR v1 v2 ...   bench step and light-scene pipeline times per value of a tuning variable
var=$1; shift
for v in "$@"; do
export $var=$v
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-configs > gpurun_out/ab.json 2>/dev/null || exit 1
b=$(python3 -c "
import json; b=json.load(open('gpurun_out/ab.json')); print(round(b['ms_per_step'],3))")
l=""
for s in sphere_scene1 fake_cornell_box sphere_scene3 benchmark_random_scene dragon_test; do
timeout -k 10 100 python tools/scene_profile.py $s 2 > gpurun_out/sp.txt || exit 1
l="$l $(head -1 gpurun_out/sp.txt | sed 's/.*dispatch //')"
done
echo "$var=$v bench $b | C1 cornell spheres3 random dragon:$l"
done
