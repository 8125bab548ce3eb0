# dev probe (GPU box only, run by hand): which layers move when one ReLU flips under Winograd rounding in the tiny fixtures
import sys, os, torch
sys.path.insert(0, '.')
import adam_dehaze_amd as A
import adam_dehaze_amd.engine as E
from adam_dehaze_amd.loss import l1_loss
from tests._util import load_golden, sub_sd, t
from tests.test_gpu_parity import _load_into
dev = 'cuda:0'
rec = load_golden("medium_b8_odd")
logs = {}
orig_conv = E.Engine.conv
orig_wgrad = E.Engine._wgrad
orig_gather = E.Engine._run_gather
def run(mode):
    E.USE_WINOGRAD = mode != "off"
    E._WINO_ONLY = "fwd" if mode == "fwd" else ""
    log = []
    def conv(self, x, w, *a, **k):
        o = orig_conv(self, x, w, *a, **k)
        torch.cuda.synchronize()
        log.append(("conv_out", tuple(w.shape), o.t.detach().clone()))
        return o
    def wgrad(self, plans, x, g_y, gC, w):
        dw = orig_wgrad(self, plans, x, g_y, gC, w)
        torch.cuda.synchronize()
        log.append(("wgrad", tuple(w.shape), x.t.detach().clone(), g_y.detach().clone(), dw.detach().clone()))
        return dw
    E.Engine.conv = conv
    E.Engine._wgrad = wgrad
    m = _load_into(A.MediumIntensityDehazeModel(base_channels=8), rec)
    m.train()
    x = t(rec["x"]).to(dev)
    out = m(x)
    loss = l1_loss(out, t(rec["target"]).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    return log, grads, out.detach().clone()
l0, g0, o0 = run("off")
l1, g1, o1 = run("fwd")
print("out diff", (o0 - o1).abs().max().item(), "len", len(l0), len(l1))
tgt = t(rec["target"]).to(dev)
print("min |out-target| off:", (o0 - tgt).abs().min().item(), " sign flips between runs:", int(((o0 - tgt).sign() != (o1 - tgt).sign()).sum()))
for i, (a, b) in enumerate(zip(l0, l1)):
    if a[0] == "conv_out":
        d = (a[2] - b[2]).abs().max().item()
        print(i, a[0], a[1], "diff %.2e" % d, "scale %.2e" % a[2].abs().max().item())
    else:
        print(i, a[0], a[1], "x diff %.2e g diff %.2e (g scale %.2e) dw diff %.2e (dw scale %.2e)" % ((a[2] - b[2]).abs().max().item(), (a[3] - b[3]).abs().max().item(), a[3].abs().max().item(), (a[4] - b[4]).abs().max().item(), a[4].abs().max().item()))
