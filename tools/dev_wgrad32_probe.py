"""dev probe: F(3x3,2x2)-domain weight gradient, every delta position of dY against random x (finds per-position errors such
as the VALU -> inline-asm MFMA hazard: dY row 1, tile column 2, k-steps >= 1 lost their b = 3 product before the operand fence)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adam_dehaze_amd.engine as E
from adam_dehaze_amd.engine import Act, Engine
DEV = "cuda:0"
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
Ci = Co = 32
def wg(x, gy):
    w = torch.zeros(Co, Ci, 4, 4, device=DEV, requires_grad=True)
    E.USE_WINOGRAD = True
    eng = Engine(torch.device(DEV), record=False)
    plans = eng._launch_plan("conv", 4, 2, 1, w, "fwd")
    return eng._wgrad(plans, Act(x), gy, Co, w)
Hh, Ww = 12, 192
g = torch.Generator().manual_seed(0)
x = torch.randn(1, Hh, Ww, Ci, generator=g).to(DEV)
xp = torch.nn.functional.pad(x[0], (0, 0, 1, 1, 1, 1))
bad = []
for vy in range(6):
    for vx in range(96):
        gy = torch.zeros(1, Hh // 2, Ww // 2, Co, device=DEV)
        gy[0, vy, vx, 5] = 1.0
        got = wg(x, gy)
        want = torch.zeros_like(got)
        want[5] = torch.stack([torch.stack([xp[2 * vy + ky, 2 * vx + kx] for kx in range(4)], -1) for ky in range(4)], -2)
        e = float((got - want).abs().max())
        if e > 1e-4:
            bad.append((vy, vx, e))
print("bad positions:", len(bad), bad[:20])
