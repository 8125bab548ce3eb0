# dev probe (GPU box only, run by hand): does the weight-gradient kernel overlap with BN passes on a second stream?
import sys, torch
sys.path.insert(0, '/root/repo')
from adam_dehaze_amd import _hip as H
from adam_dehaze_amd.engine import Act, Engine
dev = torch.device('cuda:0')
N, Hh, Ww, Cc = 8, 512, 1024, 96
eng = Engine(dev, False)
x = Act(torch.randn(N, Hh, Ww, Cc, device=dev))
w = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05).requires_grad_(True)
g = torch.randn(N, Hh, Ww, Cc, device=dev)
y = torch.randn(N, Hh, Ww, Cc, device=dev)
out = torch.relu(y)
mean = torch.zeros(Cc, device=dev); invstd = torch.ones(Cc, device=dev)
P = N * Hh * Ww
nblk = H.value("adh_bn_bwd_num_blocks", P, Cc)
partial = torch.empty(nblk, 2, Cc, device=dev)
coef = torch.zeros(3, Cc, device=dev); coef[0] = 1
gy = torch.empty_like(g)
plans = eng._launch_plan("conv", 3, 1, 1, w, "fwd")
dplans = eng._launch_plan("conv", 3, 1, 1, w, "dgrad")
gx = torch.empty_like(g)
def wgrad(): eng._wgrad(plans, x, g, Cc, w)
def dgrad(): eng._run_gather(dplans, Act(g, Cc), gx, Cc, w)
def bn():
    H.call("adh_bn_bwd_reduce", g.data_ptr(), Cc, out.data_ptr(), Cc, 1, y.data_ptr(), Cc, mean.data_ptr(), invstd.data_ptr(), partial.data_ptr(), P, Cc, None, None)
    H.call("adh_bn_bwd_apply", g.data_ptr(), Cc, out.data_ptr(), Cc, 1, y.data_ptr(), Cc, mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), 1, gy.data_ptr(), Cc, None, 0, P, Cc, None, None)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
side = torch.cuda.Stream()
def both(a, b):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a()
        b()
        torch.cuda.current_stream().wait_stream(side)
    return f
print("wgrad", t(wgrad), "dgrad", t(dgrad), "bn(reduce+apply)", t(bn))
print("wgrad || bn", t(both(wgrad, bn)))
print("dgrad || bn", t(both(dgrad, bn)))
print("wgrad || dgrad", t(both(wgrad, dgrad)))
