#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per dispatch of one kernel.

  python tools/pmc_summary.py <kernel-substring> <out.json> name=dir [name=dir ...]

Every dir is the -d directory of one rocprofv3 pass (csv output); all *counter_collection.csv below it
are read.  Dispatches of the untraced render kernel only are kept when the substring says so.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarise(directory, needle):
    per = defaultdict(lambda: defaultdict(float))  # counter -> dispatch id -> value
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if needle not in row["Kernel_Name"]:
                    continue
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {c: {"dispatches": len(v), "mean_per_dispatch": sum(v.values()) / max(len(v), 1)} for c, v in sorted(per.items())}


def main():
    needle, out = sys.argv[1], sys.argv[2]
    res = {}
    for spec in sys.argv[3:]:
        name, directory = spec.split("=", 1)
        res[name] = summarise(directory, needle)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
