#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel name: python tools/pmc_summary.py gpurun_out/pmc_TAG_*"""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"] + f)
for k, v in tot.items():
    if not any(s in k for s in ("conv_", "wino")):
        continue
    print(k, "dispatches/pass ~", len(cnt[k]))
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:.4g}")
