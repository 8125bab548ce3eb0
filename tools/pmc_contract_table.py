#!/usr/bin/env python3
"""Markdown table from the raw SQ-counter sums `tools/final_profiles.sh pmc2` writes (rNN_pmc_contract_raw.txt): per contraction and
shape the block of the Winograd forward kernel -- cycles per launch and XCD, MFMA-busy, instruction mix per MFMA, LDS conflict share,
wait shares.   python tools/pmc_contract_table.py profiles/r04b_pmc_contract_raw.txt"""
import re
import sys

MFMA_PER_LAUNCH = {"fp32": {"conv_wino43": 5.308e7, "conv_wino32": None}, "bf16x3": {"conv_wino43": 3.982e7, "conv_wino32": None}}


def main(path):
    sec, blocks, cur, name = None, [], None, None
    for ln in open(path):
        ln = ln.rstrip("\n")
        m = re.match(r"=== (\S+) (\S+)", ln)
        if m:
            sec = (m.group(1), m.group(2))
            cur, name = {}, None
            continue
        m = re.match(r"(void )?(conv_wino\w+<[^>]*>).*dispatches", ln)
        if m:
            name = m.group(2)
            cur = {}
            blocks.append((sec, name, cur))
            continue
        m = re.match(r"\s+(\w+)\s+([0-9.e+]+)$", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    print("| contraction | shape | kernel | launches | cycles / launch | MFMA-busy | MFMAs / launch | other VALU per MFMA | LDS instr per MFMA | VMEM reads per MFMA | LDS conflict share | wave time waiting on an instruction | ... parked (barriers, `s_waitcnt`) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (c, shape), name, k in blocks:
        if not k.get("SQ_INSTS_MFMA"):
            continue
        fam = "conv_wino43" if "wino43" in name else "conv_wino32"
        per = MFMA_PER_LAUNCH[c][fam]
        if per is None:       # four launches per pass of the single-kernel shapes (1 warm-up + 3 timed)
            per = k["SQ_INSTS_MFMA"] / 4.0
        n = k["SQ_INSTS_MFMA"] / per
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0
        mf = k["SQ_INSTS_VALU_MFMA_MOPS_BF16"] if c == "bf16x3" and False else k["SQ_INSTS_MFMA"]
        print(f"| {c} | {shape} | `{name}` | {n:.2f} | {cyc / n:.3e} | **{100 * k['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} %** | {mf / n:.3e} | "
              f"{(k['SQ_INSTS_VALU'] - mf) / mf:.2f} | {k['SQ_INSTS_LDS'] / mf:.2f} | {k['SQ_INSTS_VMEM_RD'] / mf:.2f} | "
              f"{100 * k['SQ_LDS_BANK_CONFLICT'] / max(k['SQ_ACTIVE_INST_LDS'], 1):.0f} % | {100 * k['SQ_WAIT_INST_ANY'] / k['SQ_WAVE_CYCLES']:.0f} % | "
              f"{100 * k['SQ_WAIT_ANY'] / k['SQ_WAVE_CYCLES']:.0f} % |")


if __name__ == "__main__":
    main(sys.argv[1])
