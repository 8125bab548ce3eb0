#!/bin/bash
# HBM traffic of every kernel of the headline bench: two separate --pmc passes over the same command
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE x2 on gfx950).
# Writes profiles-ready JSON to gpurun_out/pmc_bench.json: {kernel: {launches, fetch_bytes, write_bytes, hbm_bytes_per_launch}}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcb_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcb_$C.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/pmcb_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != C: continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][C] += float(r["Counter_Value"]) * 1024.0      # counter unit: KB
            n[k] += 1
    for k, v in n.items(): acc[k]["launches"] = v
out = {}
for k, v in acc.items():
    if v["launches"] == 0: continue
    hbm = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
    out[k] = {"launches": v["launches"], "fetch_bytes_x2": 2.0 * v["FETCH_SIZE"], "write_bytes": v["WRITE_SIZE"],
              "hbm_bytes_per_launch": hbm / v["launches"]}
json.dump({"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); counters in KB", "kernels": out},
          open("$R/gpurun_out/pmc_bench.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:60]:60s} launches {v['launches']:4d} HBM/launch {v['hbm_bytes_per_launch']/1e9:7.3f} GB")
PY
