"""Write profiles/traffic.json (what bench.py prints as roofline.traffic) from rocprofv3 --pmc passes.

    python tools/traffic_json.py --views 8 <dir> [<dir> ...] --views 64 <dir> [<dir> ...] --out profiles/traffic.json --source "..."

Each directory group holds the counter_collection.csv files of the TWO traffic passes (`--pmc FETCH_SIZE TCC_HIT_sum` and
`--pmc WRITE_SIZE TCC_MISS_sum`, separate runs, MI355X_MICROARCH.md HBM section) of
`python tests/bench_kernels.py attn --modes global --views S --variants 0` for ONE view count. One global attention = one
dispatch of every distinct (attn16 kernel, grid) pair the launch plan issues (main launch, tail launch, split-KV merge), so
bytes per attention = sum over those pairs of the mean per-dispatch bytes:
    FETCH_SIZE [KiB] * 1024 * 2   (gfx950: FETCH_SIZE reports half of a wide coalesced read -- the guide's correction)
  + WRITE_SIZE [KiB] * 1024.
The record carries the sha256 of the attention sources it was measured on (bench.attention_source_digest): bench.py refuses
to print a traffic figure whose digest differs from the tree it runs from.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(dirs):
    vals = defaultdict(lambda: defaultdict(list))
    for root in dirs:
        for d, _, files in os.walk(root):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    for row in csv.DictReader(open(os.path.join(d, f))):
                        if "attn16_kernel" not in row["Kernel_Name"] and "attn_split_merge" not in row["Kernel_Name"]:
                            continue
                        key = (row["Kernel_Name"][:120], int(row["Grid_Size"]))
                        vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return vals


def main():
    a = sys.argv[1:]
    groups, out, source, cur = {}, os.path.join(ROOT, "profiles", "traffic.json"), "", None
    i = 0
    while i < len(a):
        if a[i] == "--views":
            cur = int(a[i + 1]); groups[cur] = []; i += 2
        elif a[i] == "--out":
            out = a[i + 1]; i += 2
        elif a[i] == "--source":
            source = a[i + 1]; i += 2
        else:
            groups[cur].append(a[i]); i += 1
    import bench
    rec = {"_how": __doc__.split("\n\n")[2].replace("\n", " ").strip(), "attention_source_digest": bench.attention_source_digest()}
    for S, dirs in sorted(groups.items()):
        vals = collect(dirs)
        total, hit, miss, parts = 0.0, 0.0, 0.0, []
        for key, c in sorted(vals.items()):
            m = {n: sum(v) / len(v) for n, v in c.items()}
            b = m.get("FETCH_SIZE", 0.0) * 1024 * 2 + m.get("WRITE_SIZE", 0.0) * 1024
            total += b
            hit += m.get("TCC_HIT_sum", 0.0); miss += m.get("TCC_MISS_sum", 0.0)
            i = key[0].find("attn")
            parts.append({"kernel": key[0][i:i + 64], "grid": key[1], "bytes": round(b)})
        n = S * 1374
        rec["global_attn_S%d_bytes_per_launch" % S] = round(total)
        rec["global_attn_S%d_algorithmic_bytes" % S] = 4 * n * 1024 * 2
        rec["global_attn_S%d_l2_hit_rate" % S] = round(hit / max(hit + miss, 1.0), 4)
        rec["global_attn_S%d_dispatches" % S] = parts
    rec["source"] = source or "profiles/traffic.json: rocprofv3 --pmc passes of these kernels on these shapes; PMC counters cannot be read from inside bench.py, so the figure is not re-measured in the bench run"
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
