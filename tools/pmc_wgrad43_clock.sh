cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ADH_WINO43_WGRAD=1
for L in default d8; do
  if [ $L = d8 ]; then export ADH_LIB_PATH=$R/tools/micro/ablate/libwg43_d8.so; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_wg43clk_$L -- python3 $R/tools/bench_kernels.py --only conv96 --pass wgrad --iters 3 > $R/gpurun_out/pmc_wg43clk_$L.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for L in ("default", "d8"):
    tot = collections.defaultdict(float); n = 0; dur = []
    for f in glob.glob("gpurun_out/pmc_wg43clk_%s/**/*counter_collection.csv" % L, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wgrad_wino43" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n += 1; dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    ms = sum(dur) / max(1, len(dur))
    print(L, "launches", n, "avg ms %.3f" % ms, {k: "%.4g" % (v / max(1, n)) for k, v in tot.items()}, "clock GHz %.3f" % (tot["GRBM_GUI_ACTIVE"] / max(1, n) / 8 / (ms * 1e6) if ms else 0))
PY
