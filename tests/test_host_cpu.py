"""CPU-side checks of the host logic and the C-ABI surface (no GPU needed, no compute calls)."""
import ctypes
import hashlib
import os
import sys
import re

import numpy as np
import pytest
import torch

import adam_dehaze_amd as A
from adam_dehaze_amd import _hip as H
from tests._util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "adam_dehaze_hip.h")).read()
    declared = set(re.findall(r"^int\s+(adh_\w+)\s*\(", header, flags=re.M))
    assert declared, "no prototypes parsed"
    lib = ctypes.CDLL(H.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(H.exported_symbols()), declared ^ set(H.exported_symbols())
    H.load()
    assert H.value("adh_version") >= 100


_CTYPE = {"void*": ctypes.c_void_p, "float*": ctypes.c_void_p, "int32_t*": ctypes.c_void_p, "int64_t*": ctypes.c_void_p,
          "int": ctypes.c_int32, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
          "double": ctypes.c_double, "double*": ctypes.c_void_p, "uint8_t*": ctypes.c_void_p, "adh_adam_tensor*": ctypes.c_void_p,
          "adh_fpn_levels*": ctypes.c_void_p}


def test_ctypes_signatures_match_header():
    header = open(os.path.join(ROOT, "include", "adam_dehaze_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"^int\s+(adh_\w+)\s*\((.*?)\)\s*;", header, flags=re.M | re.S)
    assert len(protos) == len(H.exported_symbols())
    for name, args in protos:
        args = " ".join(args.split())
        expect = []
        if args not in ("void", ""):
            for a in args.split(","):
                a = a.replace("const ", "").strip()
                m = re.match(r"(\w+)\s*(\*?)\s*\w+$", a)
                assert m, (name, a)
                ty = m.group(1) + m.group(2)
                if ty == "adh_conv_desc*":
                    expect.append(H.PD)
                elif ty == "adh_wlayout*":
                    expect.append(H.PL)
                else:
                    expect.append(_CTYPE[ty])
        got = H._SIGNATURES[name]
        assert len(got) == len(expect), (name, len(got), len(expect))
        for i, (g, e) in enumerate(zip(got, expect)):
            assert g is e, (name, i, g, e)


def test_struct_layouts_match_header_field_order():
    header = open(os.path.join(ROOT, "include", "adam_dehaze_hip.h")).read()
    body = re.search(r"typedef struct adh_conv_desc \{(.*?)\} adh_conv_desc;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(float\*|int32_t)\s*", "", decl)
        names += [n.strip() for n in decl.split(",")]
    got = [f[0].rstrip("_") for f in H.ConvDesc._fields_]
    assert names == got, (names, got)
    assert ctypes.sizeof(H.ConvDesc) == 7 * 8 + 26 * 4
    assert ctypes.sizeof(H.WLayout) == 9 * 4
    body = re.search(r"typedef struct adh_adam_tensor \{(.*?)\} adh_adam_tensor;", header, flags=re.S).group(1)
    names = [re.split(r"[\s*]+", d.strip())[-1] for d in body.split(";") if d.strip()]
    assert names == [f[0] for f in H.AdamTensor._fields_], names
    assert ctypes.sizeof(H.AdamTensor) == 4 * 8 + 8 + 2 * 4


def test_state_dict_keys_and_seeded_init_match_reference():
    rec = load_golden("fullwidth_summaries")
    ctors = {"light": lambda: A.LightweightDehazeModel(base_channels=32, n_blocks=3),
             "medium": lambda: A.MediumIntensityDehazeModel(base_channels=64),
             "high": lambda: A.HighIntensityDehazeModel(base_channels=96)}
    for name, ctor in ctors.items():
        torch.manual_seed(42)
        m = ctor()
        assert list(m.state_dict().keys()) == list(rec[name + ".state_keys"])
        assert sum(p.numel() for p in m.parameters()) == int(rec[name + ".n_params"])
        sha = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in m.parameters())).hexdigest()
        assert sha == str(rec[name + ".param_sha256"]), name


@pytest.mark.parametrize("fixture,ctor", [
    ("light_b8", lambda: A.LightweightDehazeModel(base_channels=8, n_blocks=3)),
    ("lowint_b8", lambda: A.LowIntensityDehazeModel(base_channels=8, n_blocks=3)),
    ("medium_b8", lambda: A.MediumIntensityDehazeModel(base_channels=8)),
    ("corun_b8", lambda: A.COrunInspiredModel(base_channels=8, n_blocks=2)),
    ("high_b16", lambda: A.HighIntensityDehazeModel(base_channels=16)),
    ("dual_b16", lambda: A.DualBranchAttentionModel(base_channels=16)),
])
def test_reference_state_dicts_load(fixture, ctor):
    rec = load_golden(fixture)
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in rec.items() if k.startswith("sd.")}
    m = ctor()
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert m.get_info()["model_type"] == str(rec["class_name"])


def test_cpu_input_fails_loudly_no_fallback():
    m = A.LightweightDehazeModel(base_channels=8, n_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 16, 16))
    with pytest.raises(NotImplementedError):
        A.BaseDehazeModel()(torch.rand(1, 3, 8, 8))
    with pytest.raises(NotImplementedError):
        A.EncoderDecoder()(torch.rand(1, 3, 8, 8))


def test_factories_follow_config_switches():
    cfg = {"dehazing": {"low": {"model_type": "lightweight", "channels": 8, "blocks": 2},
                        "medium": {"model_type": "standard", "channels": 8, "blocks": 6},
                        "high": {"model_type": "complex", "channels": 16, "blocks": 9}}}
    assert type(A.create_low_intensity_model(cfg)) is A.LightweightDehazeModel
    assert type(A.create_medium_intensity_model(cfg)) is A.MediumIntensityDehazeModel
    assert type(A.create_high_intensity_model(cfg)) is A.HighIntensityDehazeModel
    cfg["dehazing"]["low"]["model_type"] = "unet"
    cfg["dehazing"]["medium"]["model_type"] = "corun"
    cfg["dehazing"]["high"]["model_type"] = "dual_branch"
    assert type(A.create_low_intensity_model(cfg)) is A.LowIntensityDehazeModel
    assert type(A.create_medium_intensity_model(cfg)) is A.COrunInspiredModel
    assert type(A.create_high_intensity_model(cfg)) is A.DualBranchAttentionModel


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "adam-dehaze_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+[^\n]*oracle", src, flags=re.M), fn

def test_counted_wait_kernels_have_no_scratch_traffic(tmp_path):
    """conv_wino_kernel / conv_wino32_kernel / conv_wino43_kernel fetch their weights with inline-asm loads and
    hand-counted vmcnt waits (csrc/conv_wino.hip, w2_load_b).  A register spill inside the counted span would add scratch
    loads / stores that the counts do not know about (and compiler waits in the middle of the span): this compiles the
    files to ISA on the host and checks that no instantiation of the first two touches scratch memory at all, and that
    conv_wino43_kernel (whose NT = 3 build parks a few epilogue indices in scratch across the main loop) has none
    between its first and last contraction MFMA, i.e. anywhere in the loop that holds the counted loads."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    kernels = {}
    for fn in ("conv_wino.hip", "conv_wino43.hip"):
        src = os.path.join(ROOT, "adam-dehaze_amd", "csrc", fn)
        out = tmp_path / (fn + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only", src, "-o",
                        str(out)], check=True, timeout=900)
        cur = None
        for line in out.read_text().splitlines():
            if line.startswith("_Z") and ":" in line and any(k in line for k in ("conv_wino_kernel", "conv_wino32_kernel",
                                                                                 "conv_wino43_kernel")):
                cur = line.split(":")[0]
                kernels[cur] = []
            elif cur and line.strip().startswith("s_endpgm"):
                cur = None
            elif cur and ("scratch_" in line or "v_mfma" in line or "flat_load" in line or "flat_store" in line):
                if "v_mfma" in line and line.split(";")[0].rstrip().endswith(", 0"):
                    continue      # C = 0: the accumulator-zeroing MFMAs at the top of a region of conv_wino43_kernel, outside the counted span
                kernels[cur].append("s" if "scratch_" in line else ("m" if "v_mfma" in line else "f"))
    # NT = 1, 2, 3 of the three kernels + conv_wino43_kernel<NT, BNRED = true>, and the bf16 x 3 forms of conv_wino43_kernel (6) and
    # conv_wino32_kernel (3)
    assert len(kernels) == 21, sorted(kernels)
    for name, ops in kernels.items():
        seq = "".join(ops)
        assert "m" in seq, name
        # LDS operands must be ds_* instructions: a pointer that went through an opaque asm loses its address space and
        # hipcc falls back to flat loads, which count in vmcnt AND lgkmcnt (seen once in conv_wino43's transform)
        assert "f" not in seq, name
        if "conv_wino43_kernel" in name:
            assert "s" not in seq[seq.index("m"):seq.rindex("m")], name
        else:
            assert "s" not in seq, name
    # gfx950 store-data hazard (conv_wino43.hip, W4_STORE_NOPS): a 128-bit buffer store must be followed by at least two
    # wait states before any instruction overwrites its data registers -- also when its soffset is an SGPR, the form
    # LLVM's hazard recogniser exempts (measured: such an overwrite corrupts lanes 12-15 of every 16).  Walk the ISA of
    # both files and check every buffer_store_dwordx3/x4.
    import re
    nstores = 0
    for fn in ("conv_wino.hip", "conv_wino43.hip"):
        code = [ln.strip() for ln in (tmp_path / (fn + ".s")).read_text().splitlines()
                if ln.strip() and ln.startswith("\t") and not ln.strip().startswith((";", "."))]
        for k, ln in enumerate(code):
            m = re.match(r"buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\]", ln)
            if not m:
                continue
            nstores += 1
            data = set(range(int(m.group(1)), int(m.group(2)) + 1))
            waited = 0
            for nxt in code[k + 1:k + 4]:
                t = nxt.split()
                if t[0] == "s_nop":
                    waited += int(t[1]) + 1
                    continue
                w = set()
                mm = re.match(r"v\[(\d+):(\d+)\]", t[1]) if len(t) > 1 else None
                if mm:
                    w = set(range(int(mm.group(1)), int(mm.group(2)) + 1))
                mm = re.match(r"v(\d+),?$", t[1]) if len(t) > 1 else None
                if mm:
                    w = {int(mm.group(1))}
                writes = t[0].startswith(("v_", "ds_read", "buffer_load", "global_load", "scratch_load"))
                assert not (writes and (w & data) and waited < 2), (fn, ln, nxt)
                waited += 1
    assert nstores >= 100


_ISA_CACHE = {}


def _isa_lines(fn, tmpdir):
    """Instruction lines (mnemonic + operands, labels kept as 'label:') of csrc/<fn> compiled to gfx950 ISA on the host."""
    import shutil
    import subprocess
    if fn in _ISA_CACHE:
        return _ISA_CACHE[fn]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "adam-dehaze_amd", "csrc", fn)
    out = os.path.join(str(tmpdir), fn + ".hz.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, timeout=900)
    code = []
    for ln in open(out).read().splitlines():
        t = ln.strip()
        t = t.split(";")[0].split("//")[0].strip()
        if not t or (t.startswith(".") and not (t.startswith(".LBB") and t.endswith(":"))):   # directives go, block labels stay
            continue
        if ln.startswith("\t"):
            code.append(t)
        elif t.endswith(":"):
            code.append(t)
    _ISA_CACHE[fn] = code
    return code


def _regs(tok):
    """'v12' / 'v[4:7]' / 'a[0:15]' / 'a3' -> {('v', 12)} ... ; anything else -> empty set."""
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([va])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


MFMA_ASM_FILES = ("conv_rows.hip", "conv_wgrad.hip", "conv_wgrad43.hip", "conv_wgrad32.hip", "conv_wino.hip", "conv_wino43.hip",
                  "conv_igemm.hip")


@pytest.mark.parametrize("fn", MFMA_ASM_FILES)
def test_no_valu_write_within_two_wait_states_of_an_mfma_operand_read(fn, tmp_path_factory):
    """gfx950 VALU -> MFMA operand hazard (csrc/common.h adh_mfma_operand_fence): an MFMA
    that reads a VGPR as srcA / srcB which a VALU instruction wrote one or two instructions earlier gets the OLD value.
    hipcc pads this for MFMA instructions it emits itself; the inline-asm MFMAs of the six kernel files (twelve asm
    strings: the AGPR- and VGPR-accumulator form of `mfma_pinned` in each) are outside its hazard recogniser, and only
    two of them carry an `s_nop 1` of their own -- the others rely on their operands coming straight from LDS / global
    loads.  This walks the ISA of every kernel in those files: for each v_mfma_*, no VALU instruction among the
    instructions that precede it by fewer than two wait states (s_nop N = N + 1 wait states) may write a register of
    srcA or srcB; a VALU write of srcC (accumulator in VGPRs, or v_accvgpr_write into its AGPRs) is held to the same
    distance.  Branch targets inside the window are walked through as fall-through code (conservative for the loops
    these kernels have: the back edge's predecessor is the loop's last instruction, checked as a separate window
    below)."""
    code = _isa_lines(fn, tmp_path_factory.getbasetemp())
    nmfma = 0
    labels = {ln[:-1]: i for i, ln in enumerate(code) if ln.endswith(":")}

    def valu_writes(ln):
        t = ln.split(None, 1)
        op = t[0]
        if not op.startswith("v_") or op.startswith(("v_mfma", "v_smfmac", "v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
            return set()
        if len(t) < 2:
            return set()
        return _regs(t[1].split(",")[0])

    def check_window(k, preds, what):
        """preds: instruction indices in reverse program order that lead to the MFMA at k."""
        ops = [o.strip() for o in code[k].split(None, 1)[1].split(",")]
        srcs = _regs(ops[1]) | _regs(ops[2]) | _regs(ops[3].split()[0])
        waited = 0
        for j in preds:
            ln = code[j]
            if ln.endswith(":"):
                continue
            t = ln.split()
            if t[0] == "s_nop":
                waited += int(t[1]) + 1
            else:
                if waited >= 2:
                    break
                hit = valu_writes(ln) & srcs
                assert not hit, (fn, what, j, ln, k, code[k], sorted(hit))
                waited += 1
            if waited >= 2:
                break

    # back edges: for `s_cbranch* label` / `s_branch label` at index b, the instructions before b precede label's code too
    back = {}
    for b, ln in enumerate(code):
        t = ln.split()
        if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in labels:
            back.setdefault(labels[t[-1]], []).append(b)
    for k, ln in enumerate(code):
        if not ln.startswith("v_mfma"):
            continue
        nmfma += 1
        check_window(k, range(k - 1, max(k - 8, -1), -1), "fall-through")
        # if a label sits within the two instructions before the MFMA, also walk each branch that jumps to it
        for j in range(k - 1, max(k - 4, -1), -1):
            if code[j].endswith(":"):
                between = list(range(k - 1, j, -1))
                for b in back.get(j, []):
                    check_window(k, between + list(range(b - 1, max(b - 8, -1), -1)), "via branch at %d" % b)
    assert nmfma >= (8 if fn != "conv_igemm.hip" else 4), (fn, nmfma)


def _vm_inflight_violations(code):
    """Walk a linear instruction list: a register an inline-asm / compiler vector-memory LOAD is still filling (issued, not yet
    covered by an `s_waitcnt vmcnt(N)`: vmcnt retires in issue order, stores and LDS-DMA pieces count too) must not be read or
    written by any other instruction.  hipcc does not know that the asm statements of the counted-wait kernels are loads: it is
    free to COPY such a register (live-range split, loop-carried ring resolved at the back edge) right behind the load -- the
    copy then holds the old contents.  Seen in round 4 on conv_wino43_kernel<3, false, true> with weights requested across the
    chunk boundary (csrc/conv_wino43.hip, w4b_groups).  Loop bodies are walked twice (state carried over the back edge)."""
    labels = {ln[:-1]: i for i, ln in enumerate(code) if ln.endswith(":")}
    bad = []

    def all_regs(ln):
        t = ln.split(None, 1)
        out = set()
        if len(t) > 1:
            for tok in re.split(r"[ ,]+", t[1]):
                out |= _regs(tok)
        return out

    def walk(lo, hi, queue, depth):
        k = lo
        while k < hi:
            ln = code[k]
            if ln.endswith(":"):
                k += 1
                continue
            t = ln.split(None, 1)
            op = t[0]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1))
                    while len(queue) > n:
                        queue.pop(0)
            elif op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
                inflight = set().union(*[q for q in queue if q]) if queue else set()
                dst = _regs(t[1].split(",")[0]) if " lds" not in ln and not ln.rstrip().endswith("lds") else set()
                srcs = all_regs(ln) - dst
                if (srcs | dst) & inflight:
                    bad.append((k, ln))
                queue.append(dst)
            elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic")):
                inflight = set().union(*[q for q in queue if q]) if queue else set()
                if all_regs(ln) & inflight:
                    bad.append((k, ln))
                queue.append(set())
            else:
                inflight = set().union(*[q for q in queue if q]) if queue else set()
                if inflight and all_regs(ln) & inflight:
                    bad.append((k, ln))
                if op.startswith(("s_cbranch", "s_branch")) and t[-1].split()[-1] in labels and depth < 1:
                    tgt = labels[t[-1].split()[-1]]
                    if tgt < k:      # back edge: once more round the loop with what is in flight now
                        walk(tgt, k, list(queue), depth + 1)
            k += 1

    walk(0, len(code), [], 0)
    return bad


@pytest.mark.parametrize("fn", ["conv_wino43.hip"])
def test_no_register_is_touched_while_its_asm_load_is_in_flight(fn, tmp_path_factory):
    """See _vm_inflight_violations: every instantiation of conv_wino43_kernel, fp32 and bf16 x 3 forms.  (conv_wino.hip's main
    loops branch between edge and interior staging paths: a linear walk double-counts their pieces; its contract is covered by
    the bit-equal / direct-path parity tests on the GPU.)"""
    code = _isa_lines(fn, tmp_path_factory.getbasetemp())
    # split into functions at their labels (kernels start with _Z...:) so that state does not leak across kernels
    starts = [i for i, ln in enumerate(code) if ln.startswith("_Z") and ln.endswith(":")] + [len(code)]
    nk = 0
    for a, b in zip(starts[:-1], starts[1:]):
        if "conv_wino" not in code[a]:
            continue
        fcode = code[a:b]
        # (not the C = 0 MFMAs: conv_wino43_kernel zeroes its accumulators with them at the top of a region, outside the chunk loop)
        mf = [i for i, ln in enumerate(fcode) if ln.startswith("v_mfma") and not ln.split(";")[0].rstrip().endswith(", 0")]
        labels = {ln[:-1]: i for i, ln in enumerate(fcode) if ln.endswith(":")}
        # the main loop = the smallest backward branch span that contains every MFMA of the contraction (the epilogue's
        # `if (residual) load` blocks are branchy and compiler-waited: a linear walk over them would see phantom overlaps)
        loops = []
        for j, ln in enumerate(fcode):
            t = ln.split()
            if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in labels and labels[t[-1]] < j:
                if labels[t[-1]] <= mf[0] and j >= mf[-1]:
                    loops.append((j - labels[t[-1]], labels[t[-1]], j))
        assert loops, code[a]
        _, lo, hi = min(loops)
        nk += 1
        bad = _vm_inflight_violations(fcode[lo:hi + 1])
        assert not bad, (fn, code[a], bad[:4])
    assert nk >= 6


def test_inflight_walker_flags_a_planted_copy():
    ld = "global_load_dwordx4 v[4:7], v1, s[2:3]"
    assert _vm_inflight_violations([ld, "v_mov_b32_e32 v9, v5", "s_waitcnt vmcnt(0)"])
    assert not _vm_inflight_violations([ld, "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v9, v5"])
    assert not _vm_inflight_violations([ld, "global_load_dwordx4 v[8:11], v1, s[2:3]", "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v20, v5"])
    assert _vm_inflight_violations([ld, "global_load_dwordx4 v[8:11], v1, s[2:3]", "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v20, v9"])
    # a loop whose back edge carries a load into a copy at the top
    loop = ["top:", "v_mov_b32_e32 v20, v5", "s_waitcnt vmcnt(0)", ld, "s_cbranch_scc1 top"]
    assert _vm_inflight_violations(loop)


def test_mfma_hazard_walker_flags_a_planted_violation():
    """The walker above must actually see the pattern it guards against (and accept it once padded)."""
    ok = ["v_fmac_f32_e32 v5, v1, v2", "s_nop 1", "v_mfma_f32_32x32x2_f32 a[0:15], v5, v6, a[0:15]"]
    bad1 = ["v_fmac_f32_e32 v5, v1, v2", "v_mfma_f32_32x32x2_f32 a[0:15], v5, v6, a[0:15]"]
    bad2 = ["v_pk_fma_f32 v[6:7], v[0:1], v[2:3], v[8:9]", "ds_read_b32 v9, v10", "v_mfma_f32_32x32x2_f32 a[0:15], v5, v6, a[0:15]"]
    fine = ["v_pk_fma_f32 v[6:7], v[0:1], v[2:3], v[8:9]", "ds_read_b32 v9, v10", "s_nop 0",
            "v_mfma_f32_32x32x2_f32 a[0:15], v5, v6, a[0:15]"]
    accw = ["v_accvgpr_write_b32 a3, v1", "v_mfma_f32_32x32x2_f32 a[0:15], v5, v6, a[0:15]"]
    import tests.test_host_cpu as me
    for name, prog, want_ok in (("ok", ok, True), ("bad1", bad1, False), ("bad2", bad2, False), ("fine", fine, True),
                                ("accw", accw, False)):
        me._ISA_CACHE["__planted__.hip"] = prog + ["v_mfma_f32_32x32x2_f32 a[16:31], v50, v60, a[16:31]"] * 8
        try:
            _run_walker("__planted__.hip")
            passed = True
        except AssertionError:
            passed = False
        finally:
            del me._ISA_CACHE["__planted__.hip"]
        assert passed == want_ok, name


def _run_walker(fn):
    class _F:
        @staticmethod
        def getbasetemp():
            return "/tmp"
    test_no_valu_write_within_two_wait_states_of_an_mfma_operand_read(fn, _F)


def test_adam_state_dict_layout_matches_torch():
    """optim.Adam.state_dict() uses torch.optim.Adam's layout (duplicates under the index of their last occurrence,
    as recorded from torch in adam_dup_foreach.npz) and round-trips through torch.optim.Adam.load_state_dict."""
    from adam_dehaze_amd.optim import Adam
    rec = load_golden("adam_dup_foreach")
    a, b, c = (torch.nn.Parameter(torch.randn(s)) for s in ((5, 4), (7,), (33,)))
    listed = [a, b, a, c, c, c]
    opt = Adam(listed, lr=5e-5, weight_decay=1e-4)
    for p, n in ((a, 6), (b, 3), (c, 9)):
        opt.state[id(p)] = {"step": n, "m": torch.randn_like(p), "v": torch.rand_like(p)}
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == rec["state_params"].tolist()
    assert [float(sd["state"][k]["step"]) for k in sorted(sd["state"])] == rec["state_steps"].tolist()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        topt = torch.optim.Adam(listed, lr=1.0)
        topt.load_state_dict(sd)                      # torch accepts our layout ...
        back = topt.state_dict()
    opt2 = Adam(listed, lr=1.0)
    opt2.load_state_dict(back)                        # ... and we accept torch's
    assert opt2.param_groups[0]["lr"] == 5e-5
    for p in (a, b, c):
        assert opt2.state[id(p)]["step"] == opt.state[id(p)]["step"]
        assert torch.equal(opt2.state[id(p)]["m"], opt.state[id(p)]["m"])
    with pytest.raises(ValueError):
        Adam([a], duplicates="nope")


def test_cli_keeps_the_reference_flags():
    """main.py:29-56 of the reference: same flag names; --resume works bare (store_true there) or with a path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("adh_main", os.path.join(ROOT, "main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    old = sys.argv
    try:
        sys.argv = ["main.py", "--mode", "train_joint", "--resume", "--data_dir", "/d", "--device", "cuda:0", "--seed", "3",
                    "--exp_name", "e", "--config", "c.yaml"]
        a = mod.parse_args()
        assert a.resume is True and a.data_dir == "/d" and a.seed == 3
        sys.argv = ["main.py", "--mode", "train_dehazing", "--resume", "/x/ck.pth"]
        assert mod.parse_args().resume == "/x/ck.pth"
        sys.argv = ["main.py"]
        assert mod.parse_args().mode == "train_all" and mod.parse_args().resume is None
    finally:
        sys.argv = old


def test_resume_checkpoint_discovery(tmp_path):
    from adam_dehaze_amd.train import find_resume_checkpoint
    d = str(tmp_path)
    assert find_resume_checkpoint(d) is None
    torch.save({"epoch": 4}, os.path.join(d, "checkpoint_epoch_5.pth"))
    torch.save({"epoch": 9}, os.path.join(d, "checkpoint_epoch_10.pth"))
    assert find_resume_checkpoint(d).endswith("checkpoint_epoch_10.pth")
    torch.save({"epoch": 11, "val_psnr": 20.0}, os.path.join(d, "best_model.pth"))
    assert find_resume_checkpoint(d).endswith("best_model.pth")       # written later in training than epoch 10
    torch.save({"epoch": 7, "val_psnr": 20.0}, os.path.join(d, "best_model.pth"))
    assert find_resume_checkpoint(d).endswith("checkpoint_epoch_10.pth")


def test_explicit_resume_file_goes_to_the_stage_that_wrote_it(tmp_path):
    """`--resume <file>` in a multi-stage mode handed one checkpoint to all three branches and the joint stage
    (KeyError / cross-branch load).  train.checkpoint_stage names the owner by key set; mismatches raise ValueError."""
    from adam_dehaze_amd import train as T
    cfg = {"dehazing": {"low": {"model_type": "lightweight", "channels": 8, "blocks": 2},
                        "medium": {"model_type": "standard", "channels": 8, "blocks": 6},
                        "high": {"model_type": "complex", "channels": 16, "blocks": 9}}}
    for level, factory in (("low", A.create_low_intensity_model), ("medium", A.create_medium_intensity_model),
                           ("high", A.create_high_intensity_model)):
        p = str(tmp_path / f"{level}.pth")
        T.save_checkpoint_atomic({"epoch": 0, "model_state_dict": factory(cfg).state_dict()}, p)
        assert T.checkpoint_stage(p, cfg) == level
    pj = str(tmp_path / "joint.pth")
    torch.save({"epoch": 1, "router_state_dict": {}}, pj)
    assert T.checkpoint_stage(pj, cfg) == "joint"
    with pytest.raises(ValueError, match="not a joint-training checkpoint"):
        T.resume_joint({}, str(tmp_path / "low.pth"))
    torch.save({"epoch": 1}, str(tmp_path / "junk.pth"))
    with pytest.raises(ValueError):
        T.checkpoint_stage(str(tmp_path / "junk.pth"), cfg)


def test_hard_routing_with_sync_bn_and_mismatched_resume_are_refused(tmp_path, monkeypatch):
    """ADVICE r3.  (i) routing.type 'hard' + parallel.sync_bn at world_size > 1 would make ranks issue different sequences of
    BatchNorm collectives (a rank skips a branch whose sub-batch is empty): build_joint_system raises ValueError.  (ii) `--resume
    <file>` whose owner stage the selected --mode does not run used to print a note, train from scratch and overwrite
    best_model.pth: main.py now exits before anything is trained."""
    import importlib
    import yaml
    from adam_dehaze_amd import train as T
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "config.yaml")))
    cfg["device"] = "cpu"
    cfg["classifier"]["pretrained"] = False
    cfg["classifier"]["checkpoint_dir"] = cfg["dehazing"]["checkpoint_dir"] = str(tmp_path / "none")
    for lvl, ch in (("low", 8), ("medium", 8), ("high", 16)):
        cfg["dehazing"][lvl]["channels"] = ch
    cfg["routing"]["type"] = "hard"
    cfg["parallel"] = {"sync_bn": True}
    with pytest.raises(ValueError, match="sync_bn"):
        T.build_joint_system(cfg, world_size=2)
    # (ii)
    pj = str(tmp_path / "joint.pth")
    torch.save({"epoch": 1, "router_state_dict": {}}, pj)
    mod = importlib.import_module("main")
    cfgp = str(tmp_path / "cfg.yaml")
    cfg.pop("parallel")
    cfg["routing"]["type"] = "soft"
    yaml.safe_dump(cfg, open(cfgp, "w"))
    monkeypatch.setattr(sys, "argv", ["main.py", "--mode", "train_dehazing", "--config", cfgp, "--resume", pj, "--device", "cpu"])
    called = []
    monkeypatch.setattr(T, "train_dehazing_model", lambda *a, **k: called.append("trained"))
    with pytest.raises(SystemExit, match="joint"):
        mod.main()
    assert called == []


def test_detector_surface_and_torchvision_key_names():
    """`models.detection` keeps the reference's names (models/detection.py:7-140) and the detector's parameters carry
    torchvision's fasterrcnn_resnet50_fpn state_dict keys (v0.13 layout) under `model.`, so a real checkpoint loads where one
    exists.  Spot-checked key names and shapes; the unsupported variants of the reference's switch raise ValueError."""
    import warnings
    import models.detection as MD
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = MD.create_detection_model({"detection": {"model": "faster_rcnn_resnet50_fpn", "pretrained": True}})
    assert any("RANDOMLY" in str(x.message) for x in w)
    sd = m.state_dict()
    want = {"model.backbone.body.conv1.weight": (64, 3, 7, 7), "model.backbone.body.bn1.running_var": (64,),
            "model.backbone.body.layer1.0.downsample.0.weight": (256, 64, 1, 1), "model.backbone.body.layer2.0.conv2.weight": (128, 128, 3, 3),
            "model.backbone.body.layer4.2.bn3.weight": (2048,), "model.backbone.fpn.inner_blocks.3.0.weight": (256, 2048, 1, 1),
            "model.backbone.fpn.layer_blocks.0.0.bias": (256,), "model.rpn.head.conv.0.0.weight": (256, 256, 3, 3),
            "model.rpn.head.cls_logits.weight": (3, 256, 1, 1), "model.rpn.head.bbox_pred.bias": (12,),
            "model.roi_heads.box_head.fc6.weight": (1024, 12544), "model.roi_heads.box_predictor.cls_score.weight": (91, 1024),
            "model.roi_heads.box_predictor.bbox_pred.weight": (364, 1024)}
    for k, shp in want.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert not any("num_batches_tracked" in k for k in sd)            # FrozenBatchNorm2d has none
    assert len(sd) == 295 and all(not p.requires_grad for p in m.parameters())
    assert isinstance(MD.create_integrated_system(torch.nn.Identity(), m), MD.IntegratedDetectionSystem)
    for bad in ("faster_rcnn_mobilenet_v3_large_fpn", "mask_rcnn_resnet50_fpn", "yolov8n"):
        with pytest.raises(ValueError):
            MD.DetectionModel(model_name=bad, pretrained=False)
    from adam_dehaze_amd.detection import base_anchors, resized_size
    assert base_anchors(32).tolist() == [[-23.0, -11.0, 23.0, 11.0], [-16.0, -16.0, 16.0, 16.0], [-11.0, -23.0, 11.0, 23.0]]
    assert resized_size(512, 1024, 800, 1333) == (666, 1333) and resized_size(512, 512, 800, 1333) == (800, 800)


def test_loss_extractor_warnings_and_weight_loaders():
    """the VGG16 / LPIPS extractors start randomly initialised -- say so once; torchvision / lpips checkpoints
    load through key-remapping helpers (`features.{i}.*` -> `model.{i}.*`; `features.{i}.*` + `lin{k}.model.1.weight` ->
    `loss_fn.net.slice{k}.{i}.*`, `loss_fn.lin{k}` and `loss_fn.lins.{k}`)."""
    import warnings
    import adam_dehaze_amd.loss as L
    from tests._thirdparty_init import lpips_alex_sd
    L._WARNED.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c, pl = L.ContentLoss(), L.PerceptualLoss()
        L.ContentLoss()                                       # second construction: no second warning
    msgs = [str(x.message) for x in w]
    assert len(msgs) == 2 and all("RANDOMLY" in m for m in msgs)
    tv = {"features." + k[len("model."):]: v.clone() + 1 for k, v in c.state_dict().items()}
    tv["classifier.0.weight"] = torch.zeros(1)
    c.load_torchvision_vgg(tv)
    assert c.weights_loaded and torch.equal(c.state_dict()["model.0.weight"], tv["features.0.weight"])
    with pytest.raises(KeyError):
        L.ContentLoss().load_torchvision_vgg({"features.0.weight": torch.zeros(64, 3, 3, 3)})
    lsd = lpips_alex_sd(0)
    pl.load_lpips(lsd)
    alex = {f"features.{k.split('.')[3]}.{k.split('.')[4]}": v for k, v in lsd.items() if ".net.slice" in k}
    lin = {k[len("loss_fn."):]: v for k, v in lsd.items() if k.startswith("loss_fn.lin") and not k.startswith("loss_fn.lins")}
    p2 = L.PerceptualLoss().load_lpips(alex, lin)
    assert p2.weights_loaded and all(torch.equal(p2.state_dict()[k], pl.state_dict()[k]) for k in pl.state_dict())


def test_pixel_split_choice_balances_the_xcds():
    """engine._rows_nsplit: workgroups are dealt to 8 XCDs of 32 CUs and XCD x gets the splits = x mod 8 of every
    group, so 3 groups x 85 splits (255 workgroups) is TWO rounds -- the choice must keep groups * ceil(nsplit / 8) within whole
    rounds of 32, respect the slab budget and the tile count, and prefer fewer splits at equal cost."""
    import adam_dehaze_amd.engine as E

    def rounds(groups, ns):
        return -(-(groups * (-(-ns // 8))) // 32)
    # Conv2d k4 s2 96 -> 192 weight gradient: 3 groups, 15136 strips per class, four classes in one grid
    ns = E._rows_nsplit(3, 15136, launches=4, slab_bytes=4 * 16 * 96 * 192 * 4, tile_us=5.0, max_splits=227)
    assert ns == 80 and rounds(3, ns) == 1 and rounds(3, 85) == 2
    # the 3x3 layers of the headline model keep whole rounds
    for groups, tiles, kp in ((3, 65536, 96), (12, 16384, 192), (48, 4096, 384), (6, 65536, 192)):
        ns = E._rows_nsplit(groups, tiles, slab_bytes=36 * kp * 96 * 4, tile_us=3.6)
        blocks_per_xcd = groups * (-(-ns // 8))
        assert blocks_per_xcd / (rounds(groups, ns) * 32) >= 0.98, (groups, ns)
    assert E._rows_nsplit(1, 1) == 1
    assert E._rows_nsplit(48, 5) <= 5                                   # never more splits than tiles
    assert E._rows_nsplit(3, 100000, max_splits=10) <= 10               # slab budget
    assert E._rows_nsplit(500, 1000) >= 1
    # more splits at the same rounds * tiles product cost slab traffic: the smaller count wins
    a = E._rows_nsplit(3, 15136, launches=4, slab_bytes=4 * 16 * 96 * 192 * 4, tile_us=5.0, max_splits=400)
    assert a <= 168
