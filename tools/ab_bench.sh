#!/bin/bash
# Same-box A/B of two builds of the library on the whole headline step: bash tools/ab_bench.sh A.so B.so
R=$(cd "$(dirname "$0")/.." && pwd)
for L in "$@"; do
  ADH_LIB_PATH=$L python3 $R/bench.py --no-cpu-baseline --no-forward-eval --steps 5 --warmup 2 2>/dev/null | tail -1 > /tmp/ab_bench.json
  python3 - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bench.json"))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 2), "wino43", round(d["roofline"]["avg_launch_ms"], 4),
      " ".join(f'{h["kernel"].split()[0]}={h["ms_per_step"]:.2f}' for h in d["hbm_kernels"]))
PY
done
