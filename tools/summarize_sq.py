#!/usr/bin/env python3
"""SQ counters per launch from the passes of tools/pmc_kernel.sh (rocprofv3 --pmc, 4 counters per pass,
--kernel-trace only): mean over the launches of each kernel, summed over the device.

    python tools/summarize_sq.py gpurun_out profiles/rNN_sq_counters.md [tag]      (tag of tools/pmc_kernel.sh, default pmck)
"""
import collections
import csv
import glob
import os
import re
import sys


def main(root, out_md, tag="pmck"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
    meta = {}
    for d in sorted(glob.glob(os.path.join(root, tag + "_*"))):
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            continue
        for r in csv.DictReader(open(files[-1])):
            k = r["Kernel_Name"]
            if not any(t in k for t in ("gemm16s_kernel", "resblock16", "lstm_persist", "gn_tile", "dwconv_ln", "istft_ola")):
                continue
            k = re.sub(r"^void ", "", k).replace("wt::", "").split("(")[0]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    counters = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_VALU_MFMA_BUSY_CYCLES",
                "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"]
    with open(out_md, "w") as f:
        f.write("# SQ counters per launch (rocprofv3 --pmc, separate passes of 4 counters, --kernel-trace only; `tools/pmc_kernel.sh`)\n\n")
        f.write("command per pass: `rocprofv3 --pmc <4 counters> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                "--no-other-configs --repeats 1`; values are means over the launches of a kernel, summed over the device, in millions.\n"
                "Units on gfx950: `SQ_VALU_MFMA_BUSY_CYCLES` counts cycles; `SQ_*_CYCLES` / `SQ_ACTIVE_INST_*` / `SQ_WAIT_*` count quad-cycles.\n\n")
        f.write("| kernel | VGPR+AGPR | " + " | ".join(c.replace("SQ_", "") for c in counters) + " |\n")
        f.write("|---|---|" + "---|" * len(counters) + "\n")
        for k in sorted(vals):
            row = []
            for c in counters:
                v = vals[k].get(c)
                row.append("%.2f" % (sum(v) / len(v) / 1e6) if v else "-")
            f.write(f"| `{k}` | {meta[k][0]}+{meta[k][1]} | " + " | ".join(row) + " |\n")
    print("wrote", out_md)


if __name__ == "__main__":
    main(*sys.argv[1:4])
