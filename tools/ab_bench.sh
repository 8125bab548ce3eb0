#!/bin/bash
# Same-box A/B of two builds of the library on the whole headline step (fp32 headline and the bf16 x 3 opt-in leg), alternating:
#   bash tools/ab_bench.sh A.so B.so [reps]
R=$(cd "$(dirname "$0")/.." && pwd)
A=$1; B=$2; REPS=${3:-2}
for rep in $(seq 1 $REPS); do
for L in $A $B; do
  ADH_LIB_PATH=$L python3 $R/bench.py --no-cpu-baseline --no-forward-eval --steps 5 --warmup 2 2>/dev/null | tail -1 > /tmp/ab_bench.json
  python3 - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bench.json"))
o = d["roofline"]["other_kernels"]
s = d.get("split_bf16x3") or {}
f = s.get("families_ms_per_step", {})
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 2), "wino43", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 4),
      "wino32", round(o["adh_conv_wino32_forward"]["seconds"] / d["steps"] * 1e3, 2),
      "| bf16x3 ms/step", round(s.get("ms_per_step", 0), 2), "wino43", round(f.get("adh_conv_wino43_forward", 0), 2), "wino32", round(f.get("adh_conv_wino32_forward", 0), 2))
PY
done
done
