#!/bin/bash
# rocprofv3 counter passes over a short bench run, one pass per counter group (development aid).
# usage: tools/pmc_passes.sh <tag>   -> gpurun_out/pmc_<tag>_{A,B,C,L,F,W}/ and gpurun_out/pmc_<tag>.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
run() {  # name, counters
  timeout -s KILL 240 rocprofv3 --pmc $2 -d $R/gpurun_out/pmc_${tag}_$1 --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_${tag}_$1.log 2>&1
}
run A "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES"
run B "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
run C "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run L "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run F "FETCH_SIZE"
run W "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
cd $R
python3 tools/pmc_summary.py ptk_wf_traverse gpurun_out/pmc_${tag}.json A=gpurun_out/pmc_${tag}_A B=gpurun_out/pmc_${tag}_B C=gpurun_out/pmc_${tag}_C L=gpurun_out/pmc_${tag}_L F=gpurun_out/pmc_${tag}_F W=gpurun_out/pmc_${tag}_W > /dev/null
python3 - <<PY
import json
r = json.load(open("gpurun_out/pmc_${tag}.json"))
for g, d in r.items():
    for k, v in d.items():
        print(g, k, v["dispatches"], "%.4g" % v["mean_per_dispatch"])
PY
