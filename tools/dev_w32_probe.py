# dev probe (GPU box only, run by hand): where do conv_wino32_kernel outputs differ from the direct kernels?
import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adam_dehaze_amd.engine as E
from adam_dehaze_amd.engine import Act, Engine
dev = "cuda:0"
for (kind, N, Ci, Co, Hh, Ww) in (("conv", 1, 32, 32, 24, 96), ("conv", 1, 32, 96, 24, 96), ("convT", 1, 32, 32, 12, 48)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Hh, Ww, Ci, generator=g).to(dev)
    if kind == "conv":
        w = (torch.randn(Co, Ci, 4, 4, generator=g) / (Ci * 16) ** 0.5).to(dev); OH, OW = Hh // 2, Ww // 2
    else:
        w = (torch.randn(Ci, Co, 4, 4, generator=g) / (Ci * 4) ** 0.5).to(dev); OH, OW = Hh * 2, Ww * 2
    outs = {}
    for wino in (False, True):
        E.USE_WINOGRAD = wino
        eng = Engine(torch.device(dev), record=False)
        y = torch.zeros(N, OH, OW, Co, device=dev)
        eng._run_gather(eng._launch_plan(kind, 4, 2, 1, w, "fwd"), Act(x), y, Co, w)
        torch.cuda.synchronize()
        outs[wino] = y.cpu()
    d = (outs[True] - outs[False]).abs()
    bad = (d > 1e-3).nonzero()
    print(f"case {kind, N, Ci, Co, Hh, Ww}: max {float(d.max()):.3e} nbad {len(bad)} of {d.numel()}")
    if len(bad):
        print("  y%3", collections.Counter((bad[:, 1] % 3).tolist()), "x%3", collections.Counter((bad[:, 2] % 3).tolist()))
        print("  ch%4", collections.Counter((bad[:, 3] % 4).tolist()), "ch//4%8", collections.Counter(((bad[:, 3] // 4) % 8).tolist()))
        print("  y", sorted(collections.Counter(bad[:, 1].tolist()).items())[:16])
        print("  x", sorted(collections.Counter(bad[:, 2].tolist()).items())[:24])
        a, b_ = outs[True][0], outs[False][0]
        for (n_, y_, x_, c_) in bad[:3].tolist():
            got, want = float(a[y_, x_, c_]), float(b_[y_, x_, c_])
            near = ((b_ - got).abs() < 1e-4).nonzero()[:4].tolist()
            print(f"  at (y {y_}, x {x_}, c {c_}): got {got:.5f} want {want:.5f}; got == want at {near}")
