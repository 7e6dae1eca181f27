#!/usr/bin/env python3
"""Mean counter value per launch and kernel from rocprofv3 --pmc CSV output
(directories g0, g1, ... written by tools/prof_pmc.sh)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    return name[:70]


def main():
    root = sys.argv[1]
    values = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    durations = defaultdict(list)
    for path in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"),
                          recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                kernel = short(row.get("Kernel_Name", ""))
                if "scvae" not in kernel:
                    continue
                values[kernel][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(root, "g*", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                kernel = short(row.get("Kernel_Name", ""))
                if "scvae" in kernel:
                    durations[kernel].append(
                        (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
    counters = sorted({c for k in values.values() for c in k})
    print("kernel, launches, duration_us, " + ", ".join(counters))
    for kernel, per in sorted(values.items()):
        d = durations.get(kernel, [0.0])
        n = max(len(v) for v in per.values())
        cells = ["{:.4g}".format(sum(per[c]) / len(per[c])) if c in per else "" for c in counters]
        print("{}, {}, {:.1f}, {}".format(kernel, n, sum(d) / len(d), ", ".join(cells)))


if __name__ == "__main__":
    main()
