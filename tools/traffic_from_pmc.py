"""profiles/traffic.json from a counter summary of tools/pmc_passes.sh (development aid).
usage: python tools/traffic_from_pmc.py gpurun_out/pmc_<tag>.json profiles/<dir>/<tag>_pmc_traverse.json
Memory-side bytes per launch of the traversal kernel (FETCH_SIZE doubled, the gfx950 correction for 16-byte-per-lane
loads, plus WRITE_SIZE, both in KiB-like units of 1024 bytes), the L2 hit rate, and how busy the units were that bound
the kernel: busy cycles summed over the 256 compute units against the launch's cycles (GRBM_GUI_ACTIVE is summed
over the 8 XCDs)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # kernel_source_hash: the stamp bench.py checks before it reports these figures as its own
src, kept_as = sys.argv[1], sys.argv[2]
r = json.load(open(src))
m = lambda g, k: r[g][k]["mean_per_dispatch"]
cycles = m("B", "GRBM_GUI_ACTIVE") / 8.0
n_cu = 256
fetch_kb, write_kb = m("F", "FETCH_SIZE"), m("W", "WRITE_SIZE")
old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
out = {
    "workload": old["workload"],
    "source_hash": bench.kernel_source_hash(),
    "kernel": "ptk_wf_traverse",
    "command": "rocprofv3 --pmc <group> -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs, one pass per counter "
               "group (tools/pmc_passes.sh); the averages are over every launch of the kernel in those runs, the short last "
               "launches of a step included: FETCH_SIZE; WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; SQ_*; TA_/TD_/TCP_/GRBM_",
    "launches_averaged": r["F"]["FETCH_SIZE"]["dispatches"],
    "fetch_size_kb_per_launch": fetch_kb,
    "write_size_kb_per_launch": write_kb,
    "correction": old["correction"],
    "traverse_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "l2_hit_rate": m("W", "TCC_HIT_sum") / (m("W", "TCC_HIT_sum") + m("W", "TCC_MISS_sum")),
    "limiter": {
        "what": old["limiter"]["what"],
        "valu_busy": m("A", "SQ_ACTIVE_INST_VALU") / (cycles * n_cu),
        # share of the 64 lanes an issued VALU instruction works for (group L: both counters from one pass)
        "lane_fraction": (m("L", "SQ_THREAD_CYCLES_VALU") / (m("L", "SQ_ACTIVE_INST_VALU") * 64.0)) if "L" in r and "SQ_THREAD_CYCLES_VALU" in r["L"] else None,
        "valu_insts_per_launch": m("A", "SQ_INSTS_VALU"),
        "ta_busy": m("B", "TA_TA_BUSY_sum") / (cycles * n_cu),
        "td_busy": m("B", "TD_TD_BUSY_sum") / (cycles * n_cu),
        "cycles_per_launch": cycles,
        "source": kept_as,
    },
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
