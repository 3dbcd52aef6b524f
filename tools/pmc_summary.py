"""Summarise rocprofv3 CSV output (kernel trace and/or --pmc counter collection) per kernel.

    python tools/pmc_summary.py <dir> [<dir> ...] > summary.txt

Walks the directories for *_counter_collection.csv and *_kernel_trace.csv, prints per kernel:
calls, mean duration (us), and the mean of every collected counter per dispatch.
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("attn16_kernel", "attn_kernel", "qkv_kernel", "linear_kernel", "layernorm_kernel", "im2col", "assemble", "dino_specials"):
        if key in name:
            i = name.find(key)
            return name[i:i + 60]
    return name[:60]


def main():
    counters = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for root in sys.argv[1:]:
        for d, _, files in os.walk(root):
            for f in files:
                p = os.path.join(d, f)
                if f.endswith("counter_collection.csv"):
                    for row in csv.DictReader(open(p)):
                        counters[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
                elif f.endswith("kernel_trace.csv"):
                    for row in csv.DictReader(open(p)):
                        durs[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k in sorted(set(counters) | set(durs)):
        line = "%-62s" % k
        if durs[k]:
            line += " calls=%d mean_us=%.1f" % (len(durs[k]), sum(durs[k]) / len(durs[k]))
        print(line)
        for c in sorted(counters[k]):
            v = counters[k][c]
            print("    %-32s n=%-5d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
