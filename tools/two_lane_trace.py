#!/usr/bin/env python3
"""Kernel durations of the one-lane and the two-lane block of one bench run (tools/two_lane_trace.sh): per kernel the
median duration in each block, and for the two-lane block how much of the wall time has 0 / 1 / 2+ kernels in flight.

    python tools/two_lane_trace.py gpurun_out/r04_2lane [out.md]"""
import collections
import csv
import glob
import os
import statistics
import sys


def main(d, out=None):
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f))]
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))) for r in rows]
    ev.sort()
    # the two-lane block = the time range in which two queues alternate; find it from the queue ids of the LSTM launches
    lstm = [e for e in ev if "lstm_persist" in e[2]]
    queues = collections.Counter(e[3] for e in lstm)
    main_q = lstm[0][3]
    first_other = next((e for e in lstm if e[3] != main_q), None)
    if first_other is None:
        print("no second queue in the trace", queues)
        return
    t_split = first_other[0] - 3_000_000          # the two-lane block starts a little before the first LSTM on the second queue
    blocks = {"one lane": [e for e in ev if e[1] < t_split], "two lanes": [e for e in ev if e[0] >= t_split]}
    lines = ["# One-lane against two-lane block of one bench run: kernel durations (rocprofv3 kernel trace; tools/two_lane_trace.py)", ""]
    names = ["lstm_persist", "resblock16_kernel<32", "resblock16_kernel<64", "gemm16s_kernel<128, 192, 4, 2, 3, 2, 1", "gemm16s_kernel<128, 192, 4, 2, 3, 3, 0",
             "gemm16s_kernel<128, 192, 4, 2, 3, 0, 0", "gemm16s_kernel<128, 128, 4, 2, 3, 0, 2", "gemm16s_kernel<128, 128, 4, 2, 3, 4, 1", "gemm16s_kernel<128, 192, 4, 2, 3, 6, 0",
             "dwconv_ln_kernel", "gn_tile_kernel<1>"]
    lines += ["| kernel | one lane: median us | two lanes: median us | ratio |", "|---|---|---|---|"]
    for nm in names:
        med = {}
        for b, es in blocks.items():
            ds = [(e[1] - e[0]) / 1e3 for e in es if nm in e[2]]
            ds = ds[len(ds) // 4:]                  # drop the warm-up quarter
            med[b] = statistics.median(ds) if ds else float("nan")
        lines.append(f"| `{nm}` | {med['one lane']:.1f} | {med['two lanes']:.1f} | {med['two lanes'] / med['one lane']:.2f} |")
    for b, es in blocks.items():
        es = es[len(es) // 4:]
        t0, t1 = es[0][0], max(e[1] for e in es)
        pts = sorted([(e[0], 1) for e in es] + [(e[1], -1) for e in es])
        depth, last, hist = 0, t0, collections.Counter()
        for t, dlt in pts:
            hist[min(depth, 3)] += t - last
            last, depth = t, depth + dlt
        tot = sum(hist.values())
        busy = sum(e[1] - e[0] for e in es)
        lines += ["", f"{b}: wall {1e-6 * (t1 - t0):.2f} ms for {len(es)} launches; sum of kernel durations {1e-6 * busy:.2f} ms; "
                  + ", ".join(f"{k if k < 3 else '3+'} kernels in flight {100.0 * v / tot:.1f} %" for k, v in sorted(hist.items()))]
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
