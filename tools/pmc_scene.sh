#!/bin/bash
# rocprofv3 counter passes over one fixture scene with one renderer pinned (development aid).
# usage: tools/pmc_scene.sh <tag> <scene> <renderer 1|2|3> <kernel-name substring>
#   -> gpurun_out/pmc_<tag>_{A,B,C,L,F,W}/ and gpurun_out/pmc_<tag>.json (mean per dispatch of that kernel) and
#      gpurun_out/pmc_<tag>.md (derived figures: VALU busy, share of working lanes, waves, memory-side bytes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; scene=$2; renderer=$3; kernel=$4
run() {  # name, counters
  timeout -s KILL 240 rocprofv3 --pmc $2 -d $R/gpurun_out/pmc_${tag}_$1 --output-format csv -- python3 $R/tools/scene_profile.py $scene $renderer > $R/gpurun_out/pmc_${tag}_$1.log 2>&1
}
run A "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES"
run B "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
run C "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run L "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run F "FETCH_SIZE"
run W "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
cd $R
python3 tools/pmc_summary.py "$kernel" gpurun_out/pmc_${tag}.json A=gpurun_out/pmc_${tag}_A B=gpurun_out/pmc_${tag}_B C=gpurun_out/pmc_${tag}_C L=gpurun_out/pmc_${tag}_L F=gpurun_out/pmc_${tag}_F W=gpurun_out/pmc_${tag}_W > /dev/null
python3 - <<PY
import json
r = json.load(open("gpurun_out/pmc_${tag}.json"))
m = lambda g, k: r[g][k]["mean_per_dispatch"]
cycles = m("B", "GRBM_GUI_ACTIVE") / 8.0   # summed over the 8 XCDs
cu = 256
lines = ["# ${tag}: ${scene}, renderer ${renderer}, kernel *${kernel}* (%d dispatches averaged)" % r["A"]["SQ_WAVES"]["dispatches"], ""]
lines.append("| figure | value |")
lines.append("|---|---|")
lines.append("| cycles per dispatch (GRBM_GUI_ACTIVE / 8) | %.0f (%.3f ms at 2.4 GHz) |" % (cycles, cycles / 2.4e6))
lines.append("| waves per dispatch | %.0f |" % m("A", "SQ_WAVES"))
lines.append("| VALU instructions per dispatch | %.3g |" % m("A", "SQ_INSTS_VALU"))
lines.append("| VALU busy (SQ_ACTIVE_INST_VALU / (cycles x CUs)) | %.3f |" % (m("A", "SQ_ACTIVE_INST_VALU") / (cycles * cu)))
lines.append("| share of working lanes (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)) | %.3f |" % (m("L", "SQ_THREAD_CYCLES_VALU") / (m("L", "SQ_ACTIVE_INST_VALU") * 64)))
lines.append("| waves resident per SIMD on average (SQ_WAVE_CYCLES / (cycles x 1024)) | %.2f |" % (m("A", "SQ_WAVE_CYCLES") / (cycles * 1024)))
lines.append("| wave time waiting for anything (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | %.3f |" % (m("A", "SQ_WAIT_ANY") / m("A", "SQ_WAVE_CYCLES")))
lines.append("| wave time waiting for an instruction to issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) | %.3f |" % (m("A", "SQ_WAIT_INST_ANY") / m("A", "SQ_WAVE_CYCLES")))
lines.append("| TA busy, TD busy | %.3f, %.3f |" % (m("B", "TA_TA_BUSY_sum") / (cycles * cu), m("B", "TD_TD_BUSY_sum") / (cycles * cu)))
lines.append("| SALU / LDS / VMEM-read instructions per dispatch | %.3g / %.3g / %.3g |" % (m("C", "SQ_INSTS_SALU"), m("C", "SQ_INSTS_LDS"), m("C", "SQ_INSTS_VMEM_RD")))
lines.append("| LDS bank-conflict cycles / LDS active cycles | %.3f |" % (m("C", "SQ_LDS_BANK_CONFLICT") / max(m("C", "SQ_LDS_IDX_ACTIVE"), 1)))
mem = (2.0 * m("F", "FETCH_SIZE") + m("W", "WRITE_SIZE")) * 1024.0
lines.append("| memory-side bytes per dispatch (2 x FETCH_SIZE + WRITE_SIZE, KiB units) | %.3g (%.2f TB/s) |" % (mem, mem / (cycles / 2.4e9) / 1e12))
lines.append("| L2 hit rate | %.3f |" % (m("W", "TCC_HIT_sum") / (m("W", "TCC_HIT_sum") + m("W", "TCC_MISS_sum"))))
open("gpurun_out/pmc_${tag}.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
