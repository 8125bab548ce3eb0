# dev probe (GPU box only, run by hand): where do conv_wino43_kernel outputs differ from the direct kernel?
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adam_dehaze_amd.engine as E
from adam_dehaze_amd.engine import Act, Engine
dev = "cuda:0"
for (N, Ci, Co, Hh, Ww) in ((1, 96, 96, 16, 32), (1, 32, 32, 16, 32), (1, 96, 96, 32, 64), (2, 16, 16, 15, 23)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Hh, Ww, Ci, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(dev)
    outs = {}
    for wino in (False, True):
        E.USE_WINOGRAD = wino
        eng = Engine(torch.device(dev), record=False)
        y = torch.zeros(N, Hh, Ww, Co, device=dev)
        eng._run_gather(eng._launch_plan("conv", 3, 1, 1, w, "fwd"), Act(x), y, Co, w)
        torch.cuda.synchronize()
        outs[wino] = y.cpu()
    d = (outs[True] - outs[False]).abs()
    bad = (d > 1e-3).nonzero()
    print(f"case {N,Ci,Co,Hh,Ww}: max {float(d.max()):.3e} nbad {len(bad)} of {d.numel()}")
    if len(bad):
        import collections
        print("  rows%4", collections.Counter((bad[:, 1] % 4).tolist()), "cols%4", collections.Counter((bad[:, 2] % 4).tolist()))
        print("  ch//4 %8", collections.Counter(((bad[:, 3] // 4) % 8).tolist()), "ch//32", collections.Counter((bad[:, 3] // 32).tolist()))
        print("  tiles (y//4,x//4)", collections.Counter(zip((bad[:, 1] // 4).tolist(), (bad[:, 2] // 4).tolist())).most_common(8))
        print("  first", bad[:6].tolist())
    if len(bad):
        a, b_ = outs[True][0], outs[False][0]
        for (n_, y_, x_, c_) in bad[:4].tolist():
            got, want = float(a[y_, x_, c_]), float(b_[y_, x_, c_])
            print(f"  at (y {y_}, x {x_}, c {c_}): got {got:.6f} want {want:.6f} diff {got - want:.6f}")
            # does `got` equal the expected value somewhere nearby?
            near = (b_ - got).abs() < 1e-4
            print("    got == want at", near.nonzero()[:5].tolist())
            near2 = (b_ - (got - want)).abs() < 1e-4
            print("    diff == want at", near2.nonzero()[:5].tolist())
            print("    got quad", a[y_, x_, c_ - 1:c_ + 3].tolist(), "want quad", b_[y_, x_, c_ - 1:c_ + 3].tolist())
