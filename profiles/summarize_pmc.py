#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSVs (profiles/collect_pmc.sh) to per-kernel means per launch:
   python profiles/summarize_pmc.py gpurun_out/pmc_<tag>  ->  {"S": {...}, "W": {...}}
FETCH_SIZE / WRITE_SIZE: raw counter value plus *_GB with the unit the MI355X guide prescribes (see grep below)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        nm = name.replace(" ", "")
        if "fm_search_kernel<false>" in nm or "fm_search_kernel<false,false>" in nm:      # production first pass (not the touch-counting instantiation; two parameters until round 2's clean-up)
            k = "S"
        elif "fm_width_kernel<false>" in nm:
            k = "W"
        elif "fm_deep_kernel" in nm:                      # kernel D: one deep search per wavefront
            k = "D"
        else:
            continue
        d = acc.setdefault(k, {})
        c = d.setdefault(r["Counter_Name"], [])
        c.append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size"]),
                  r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
out = {}
for k, d in acc.items():
    o = out.setdefault(k, {})
    for cn, rows in d.items():
        # the first-pass launch is the one with the largest grid; second-pass launches (few reads) are dropped
        g = max(r[2] for r in rows)
        big = [r for r in rows if r[2] == g]
        o[cn] = sum(r[0] for r in big) / len(big)
        o["dur_ms"] = sum(r[1] for r in big) / len(big)
        o["launches_averaged"] = len(big)
        o["VGPR"], o["AGPR"], o["LDS_block"], o["scratch"] = big[0][3], big[0][4], big[0][5], big[0][6]
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        if cn in o:
            o[cn + "_GB"] = o[cn] * 1024 / 1e9
print(json.dumps(out, indent=1, sort_keys=True))
