import sys, torch
sys.path.insert(0, '/root/repo')
import adam_dehaze_amd.engine as E
from adam_dehaze_amd import _hip as H
from adam_dehaze_amd.engine import Act, Engine
dev = torch.device('cuda:0')
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
calls = []
real = H.call
def counting(name, *a, **k):
    calls.append(name); return real(name, *a, **k)
H.call = counting
for Cin, Cout, alloc in [(3, 16, 8), (16, 16, 16), (16, 1, 16), (48, 3, 48)]:
    for wino in (True, False):
        E.USE_WINOGRAD = wino
        eng = Engine(dev, record=False)
        x = Act(torch.randn(8, 512, 1024, alloc, device=dev), Cin)
        k = 1 if (Cin, Cout) == (16, 1) else 3
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.1).requires_grad_(True)
        plans = eng._launch_plan("conv", k, 1, k // 2, w, "fwd")
        out = torch.empty(8, 512, 1024, max(4, (Cout + 3) // 4 * 4), device=dev)
        calls.clear()
        ms = timeit(lambda: eng._run_gather(plans, x, out, Cout, w))
        print(f"{Cin}->{Cout} k{k} wino={wino}: {ms:.3f} ms via {sorted(set(c for c in calls if 'conv' in c and 'pack' not in c))}")
# data gradient of the 48 -> 3 head: 3 (8-channel tensor) -> 48
for few in (True, False):
    E.USE_FEWOUT = few
    eng = Engine(dev, record=False)
    wh = (torch.randn(3, 48, 3, 3, device=dev) * 0.1).requires_grad_(True)
    gy = torch.zeros(8, 512, 1024, 8, device=dev); gy[..., :3] = torch.randn(8, 512, 1024, 3, device=dev)
    gx = torch.empty(8, 512, 1024, 48, device=dev)
    plans = eng._launch_plan("conv", 3, 1, 1, wh, "dgrad")
    calls.clear()
    ms = timeit(lambda: eng._run_gather(plans, Act(gy, 3), gx, 48, wh))
    print(f"dgrad of 48->3 (3->48) fewin={few}: {ms:.3f} ms via {sorted(set(c for c in calls if 'conv' in c and 'pack' not in c))}")
