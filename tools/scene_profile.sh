cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in fake_cornell_box sphere_scene1 dragon_test; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lp_$s -o lp --output-format csv -- python3 $R/tools/scene_profile.py $s 2 || exit 1
done
