#!/bin/bash
# The other BASELINE.json configurations, one bench line each -> gpurun_out/other_configs.jsonl
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/other_configs.jsonl; : > $out
for w in "config2" "config3 --batch 16" "config4" "complex_fullloss" "complex_eval" "config5" "config5_dehaze"; do
  python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $out || exit 1
done
python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l); print(d["metric"], round(d["value"], 2), round(d["ms_per_step"], 1))
PY
