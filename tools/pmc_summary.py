"""Summarise rocprofv3 CSV output (kernel trace and/or --pmc counter collection) per (kernel, grid size).

    python tools/pmc_summary.py <dir> [<dir> ...] [--only SUBSTR] > summary.txt

Walks the directories for *_counter_collection.csv and *_kernel_trace.csv and prints, per kernel AND launch grid (so the
shapes of one kernel never mix): calls, mean duration (us), the mean of every collected counter per dispatch, and the
derived figures used in profiles/README.md (matrix-pipe busy, VALU issue share, MFMA/VALU co-execution, LDS conflict
rate, L2 hit rate, fabric bytes with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md).
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("attn16_kernel", "attn_kernel", "qkv256_kernel", "linear256_kernel", "qkv_kernel", "linear_kernel", "layernorm_kernel",
                "conv_kernel", "im2col", "assemble", "dino_specials", "attn_merge"):
        if key in name:
            i = name.find(key)
            return name[i:i + 64]
    return name[:64]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--only"]
    counters = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for root in args:
        for d, _, files in os.walk(root):
            for f in files:
                p = os.path.join(d, f)
                if f.endswith("counter_collection.csv"):
                    for row in csv.DictReader(open(p)):
                        key = (short(row["Kernel_Name"]), int(row["Grid_Size"]))
                        counters[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                elif f.endswith("kernel_trace.csv"):
                    for row in csv.DictReader(open(p)):
                        gs = int(row.get("Grid_Size", 0) or (int(row.get("Grid_Size_X", 0)) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1))))
                        durs[(short(row["Kernel_Name"]), gs)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k in sorted(set(counters) | set(durs)):
        if only and not any(o in k[0] for o in only):
            continue
        if not counters[k] and len(durs[k]) < 2:
            continue
        line = "%-66s grid=%-9d" % k
        if durs[k]:
            line += " calls=%d mean_us=%.1f min_us=%.1f" % (len(durs[k]), sum(durs[k]) / len(durs[k]), min(durs[k]))
        print(line)
        c = {n: sum(v) / len(v) for n, v in counters[k].items()}
        for n in sorted(c):
            print("    %-32s n=%-5d mean=%.4g" % (n, len(counters[k][n]), c[n]))
        d = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA_BUSY over all 1024 SIMDs (= 16 cycles x SQ_INSTS_MFMA for 16x16x32)
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            d.append("matrix pipe busy = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = %.1f %% of the cycles actually clocked" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
            if durs[k]:
                d.append("effective clock = GUI_ACTIVE / 8 / duration = %.2f GHz (2.4 GHz nominal: the roofline peak is quoted at 2.4)" % (cyc / (sum(durs[k]) / len(durs[k]) * 1e3)))
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
            d.append("VALU-active share of wave cycles = %.1f %%" % (100 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
            d.append("issue-stall (WAIT_INST_ANY) = %.1f %%, parked (WAIT_ANY) = %.1f %% of wave cycles" %
                     (100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]))
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in c and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d.append("MFMA/VALU co-execution = %.1f %% of MFMA-busy cycles" % (100 * c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]))
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            d.append("non-MFMA VALU per MFMA = %.2f" % ((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]))
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            d.append("LDS bank-conflict cycles = %.1f %% of LDS-active" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d.append("L2 hit rate = %.1f %%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d.append("fabric bytes per launch = FETCH_SIZE[KiB] x 1024 x 2 + WRITE_SIZE[KiB] x 1024 = %.4g" % (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024))
        for s in d:
            print("    => " + s)


if __name__ == "__main__":
    main()
