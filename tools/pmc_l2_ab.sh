#!/bin/bash
# L2 hit rate and launch length of the traversal kernel under two settings of an environment variable (development aid).
# usage (on the GPU box): bash tools/pmc_l2_ab.sh VAR value_a value_b
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
var=$1
for v in $2 $3; do
  export $var=$v
  rm -rf $R/gpurun_out/pmc_ab_W $R/gpurun_out/pmc_ab_B
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_ab_W --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_ab_W.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_ab_B --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_ab_B.log 2>&1
  python3 $R/tools/pmc_summary.py ptk_wf_traverse $R/gpurun_out/pmc_ab.json W=$R/gpurun_out/pmc_ab_W B=$R/gpurun_out/pmc_ab_B > /dev/null
  python3 - <<PY
import json
r = json.load(open("$R/gpurun_out/pmc_ab.json"))
m = lambda g, k: r[g][k]["mean_per_dispatch"]
cyc = m("B", "GRBM_GUI_ACTIVE") / 8.0
print("$var=$v: L2 hit rate %.3f, cycles per launch %.0f, TA busy %.3f, TD busy %.3f, TCP pending stall %.3f of CU cycles, L2 misses per launch %.3g" % (
    m("W", "TCC_HIT_sum") / (m("W", "TCC_HIT_sum") + m("W", "TCC_MISS_sum")), cyc, m("B", "TA_TA_BUSY_sum") / (cyc * 256), m("B", "TD_TD_BUSY_sum") / (cyc * 256),
    m("B", "TCP_PENDING_STALL_CYCLES_sum") / (cyc * 256), m("W", "TCC_MISS_sum")))
PY
done
