"""Where the wall time of a short forward goes: kernel time vs the gaps between kernels, from a rocprofv3 kernel trace.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python bench.py --views 8 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-e2e
    python tools/probes/launch_gaps.py <kernel_trace.csv> [--forwards 3]

Takes the LAST `forwards` aggregator forwards of the trace (a forward = 24 launches of the global-attention kernel, the longest attention launches),
prints per forward: span, sum of kernel durations, number of launches, total gap, and the 12 largest gaps with the kernels on both sides."""
import argparse
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--tail-ms", type=float, default=150.0, help="analyse the kernels that start within this many ms of the end of the trace")
    args = ap.parse_args()
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(args.trace))]
    rows.sort()
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - args.tail_ms * 1e6]
    span = (rows[-1][1] - rows[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in rows) / 1e6
    gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)]
    tot_gap = sum(max(0, g[0]) for g in gaps) / 1e6
    print("last %.0f ms of the trace: %d launches, span %.2f ms, kernel time %.2f ms (%.1f %%), gaps %.2f ms; mean gap %.1f us, median %.1f us"
          % (args.tail_ms, len(rows), span, busy, 100 * busy / span, tot_gap, 1e-3 * sum(g[0] for g in gaps) / len(gaps), 1e-3 * sorted(g[0] for g in gaps)[len(gaps) // 2]))
    hist = [0] * 8
    edges = [2, 5, 10, 20, 50, 100, 1000]
    for g in gaps:
        us = g[0] / 1e3
        hist[sum(us > e for e in edges)] += 1
    print("gap histogram (us):", "  ".join("%s%s: %d" % ("<=" if i < len(edges) else ">", edges[min(i, len(edges) - 1)], h) for i, h in enumerate(hist)))
    short = lambda n: n.split("(")[0][-48:]
    for g in sorted(gaps, key=lambda g: -g[0])[:12]:
        print("  gap %8.1f us   after %-50s before %s" % (g[0] / 1e3, short(g[1]), short(g[2])))


if __name__ == "__main__":
    sys.exit(main())
