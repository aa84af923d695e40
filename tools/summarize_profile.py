#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/, scratch) into the small summaries committed under
profiles/.

  python tools/summarize_profile.py stats  <kernel-trace dir> <steps> profiles/rNN_kernel_stats.md "<command>"
  python tools/summarize_profile.py pmc    <FETCH_SIZE dir> <WRITE_SIZE dir> profiles/rNN_pmc_traffic.json

`pmc` applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): bytes read =
2 x FETCH_SIZE x 1024, bytes written = WRITE_SIZE x 1024 (checked on known byte counts:
profiles/r01_pmc_calibration.txt).
"""
import collections
import csv
import glob
import json
import os
import sys


def _one(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)      # the newest run (output directories are reused)
    if not fs:
        raise SystemExit(f"no file matches {pattern}")
    return fs[-1]


def stats(trace_dir, steps, out_md, command):
    f = _one(os.path.join(trace_dir, "**", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(f)))
    # model-load work (weight upload blits, weight splits) runs once, before the steps: listed apart
    load = [r for r in rows if "rocclr" in r["Name"] or "split_" in r["Name"]]
    rows = [r for r in rows if r not in load]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    steps = int(steps)
    with open(out_md, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{command}`\n\n")
        o.write(f"kernel time total {tot / 1e6:.2f} ms over {steps} steps (warm-up included) = {tot / 1e6 / steps:.2f} ms/step\n\n")
        o.write("| kernel | calls | avg us | ms/step | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            name = r["Name"].replace("wt::", "").replace("(wt::GemmArgs)", "")
            if len(name) > 90:
                name = name[:87] + "..."
            o.write("| `%s` | %s | %.1f | %.3f | %.1f |\n" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6 / steps,
                                                            100 * float(r["TotalDurationNs"]) / tot))
        o.write("\nnot per step (model load: weight upload blits and weight splits): %.2f ms in %d launches\n"
                % (sum(float(r["TotalDurationNs"]) for r in load) / 1e6, sum(int(r["Calls"]) for r in load)))
        o.write("\nGEMM template arguments: gemm16s_kernel<BM, BN, waves_m, waves_n, stages, EPI, OUT>, gemm16_kernel / gemm_kernel"
                "<BM, BN, waves_m, waves_n, PRO, EPI>; PRO 0 none, 1 ELU; EPI 0 bias, 1 bias+res, "
                "2 bias+GELU (pwconv1), 3 gamma*(.)+res (pwconv2), 4 ISTFT head, 6 VQ argmax, 7 scale, "
                "8 row bias, 9 bias+res+ELU, 10 bias+ELU; OUT 0 fp32, 1 S32, 2 S32 raw + S32 elu, 3 fp32 + S32.\n")
    print("wrote", out_md)


def pmc(fetch_dir, write_dir, out_json):
    def load(d):
        f = _one(os.path.join(d, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}
    fe, nfe = load(fetch_dir)
    wr, _ = load(write_dir)
    out = {}
    for k in fe:
        if "wt::" not in k:
            continue
        rd = 2.0 * fe[k] * 1024.0
        w = wr.get(k, 0.0) * 1024.0
        out[k] = {"launches_sampled": nfe[k], "fetch_size_kb_raw": round(fe[k], 1), "write_size_kb": round(wr.get(k, 0.0), 1),
                  "bytes_read_corrected": round(rd), "bytes_written": round(w), "traffic_bytes_per_launch": round(rd + w)}
    with open(out_json, "w") as o:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py "
                           "--steps 3 --warmup 1 --no-cpu-baseline`; read bytes = 2 x FETCH_SIZE x 1024 (gfx950), "
                           "written = WRITE_SIZE x 1024; Infinity-Cache hits are counted too (fabric-side counter)",
                   "kernels": out}, o, indent=1, sort_keys=True)
    print("wrote", out_json)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:6])
    elif sys.argv[1] == "pmc":
        pmc(*sys.argv[2:5])
    else:
        raise SystemExit(__doc__)
