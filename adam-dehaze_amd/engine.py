"""Forward/backward engine over the HIP library.

A branch forward is a sequence of fused ops on fp32 NHWC activations (`Act`); every op launches
kernels through the C ABI (`_hip.call`) and, when gradients are needed, appends a backward closure to
the tape.  `Engine.backward()` replays the tape in reverse.  torch supplies device memory, the HIP
stream and the outer autograd hook (`functional.BranchFunction`); no torch arithmetic runs here.

Reference semantics cited per op (paths under /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _hip as H
from ._hip import ConvDesc, WLayout

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# Winograd F(2x2,3x3) for eligible 3x3 stride-1 convolutions and their data gradients (ADH_WINOGRAD=0 disables)
USE_WINOGRAD = os.environ.get("ADH_WINOGRAD", "1") != "0"
# F(4x4,3x3) where it applies, else F(2x2,3x3): True / False, or "fwd" / "dgrad" to restrict it to one direction
USE_WINO43 = {"0": False, "1": True}.get(os.environ.get("ADH_WINO43", "1"), os.environ.get("ADH_WINO43", "1"))
# F(4x4,3x3)-domain weight gradient (conv_wgrad43.hip, round-3 redesign: 32 x 96 channels x all 36 frequencies per workgroup):
# 2.75 / 2.50 / 2.52 ms on the 96 / 192 / 384-channel layers against 3.10 / 2.78 / 2.79 for the F(2x2,3x3)-domain kernel
#  -- the default where it applies (Cin % 32 == 0, Cout % 96 == 0); ADH_WINO43_WGRAD=0 falls back
USE_WINO43_WGRAD = os.environ.get("ADH_WINO43_WGRAD", "1") != "0"
# Contraction arithmetic of the F(4x4,3x3) forward / data-gradient launches: "fp32" = v_mfma_f32_32x32x2_f32 (default, the
# headline), "bf16x3" = v_mfma_f32_32x32x16_bf16 over exact three-plane bf16 splits of both operands (opt-in)
CONTRACT = os.environ.get("ADH_CONTRACT", "fp32")
if CONTRACT not in ("fp32", "bf16x3"):
    raise ValueError(f"ADH_CONTRACT must be 'fp32' or 'bf16x3', got {CONTRACT!r}")
W43_WGRAD_ROUNDS = int(os.environ.get("ADH_W43_WGRAD_ROUNDS", "4"))   # dev: rounds of workgroups the pixel splits may form
USE_SMALL_WGRAD = os.environ.get("ADH_SMALL_WGRAD", "1") != "0"
USE_FEWOUT = os.environ.get("ADH_FEWOUT", "1") != "0"               # conv_fewout.hip for the <= 4-output-channel 3x3 heads
# BatchNorm-backward sums of a ConvBlock taken in the epilogue of its single consumer's data-gradient launch
# (adh_conv_wino43_dgrad_bnred) instead of a bn_bwd_reduce pass over the same tensor (A/B switch)
USE_BN_FUSED_REDUCE = os.environ.get("ADH_BN_FUSED_REDUCE", "1") != "0"
# the output-parity class launches of one transposed layer on separate streams (their partial last rounds overlap): measured
# -0.2 .. -0.4 ms per layer in isolation, +1.5 ms on the whole step -- opt-in
CLASS_STREAMS = os.environ.get("ADH_CLASS_STREAMS", "0") != "0"
# ... and as one grid (adh_conv_wino32_forward_multi): what the streams were after, without their events
MERGE_CLASSES = os.environ.get("ADH_MERGE_CLASSES", "1") != "0"
_SIDE_STREAMS: Dict[tuple, list] = {}


def _side_streams(device: torch.device, n: int):
    key = (device.type, device.index)
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]
# Bit-packed ReLU mask for the residual BN layers (1 bit per element written by bn_apply, read by the two backward passes
# instead of `out`): correct and tested, but measured SLOWER on MI355X (bench, ms/step: bn_apply 8.19 -> 8.47,
# bn_bwd_reduce 8.79 -> 9.85, bn_bwd_apply 12.78 -> 12.52; +1.1 ms in all): the byte loads double the number of
# vector-memory instructions of passes that were already running at 5+ TB/s.  Opt-in (ADH_RELU_BITS=1).
USE_RELU_BITS = os.environ.get("ADH_RELU_BITS", "0") != "0"      # conv_wgrad_small.hip for the few-channel 3x3 layers
_WINO_ONLY = os.environ.get("ADH_WINOGRAD_ONLY", "")   # dev: "fwd" or "dgrad" restricts the Winograd path to one direction


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# ---------------------------------------------------------------------------------------------------------------------
# Packed / Winograd-transformed weights are a function of the parameter only: cache them per parameter *version* so that
# the forward pass and the data gradient of one step (and every pass of an eval loop) share one pack launch.
# Key: (id, data_ptr, torch version counter, kind, layout).  torch bumps the version counter on every in-place op
# (optimizers, load_state_dict's copy_); writers that bypass it -- the HIP Adam kernel, or `p.data.op_()` -- must call
# `invalidate_weight_cache()` (optim.Adam.step does).  ADH_PACK_CACHE=0 disables the cache.
# ---------------------------------------------------------------------------------------------------------------------
USE_PACK_CACHE = os.environ.get("ADH_PACK_CACHE", "1") != "0"
_PACK_CACHE: Dict[tuple, Tuple["weakref.ref", torch.Tensor]] = {}
_PACK_CACHE_MAX = 4096


def invalidate_weight_cache() -> None:
    _PACK_CACHE.clear()


def _layout_key(L: WLayout) -> tuple:
    return (L.K, L.Nc, L.KHt, L.KWt, L.tap_off0, L.tap_off_sy, L.tap_off_sx, L.stride_k, L.stride_n)


# ---------------------------------------------------------------------------------------------------------------------
# Gradient sinks and the grad-ready hook (data-parallel training, parallel.GradientSynchronizer).
# GRAD_SINK(param) -> a preallocated contiguous tensor of the parameter's shape (a view of a flat bucket buffer) that the
# engine writes the parameter's gradient into, or None; GRAD_READY(param, grad) is called from inside Engine.backward()
# the moment the gradient is final (all of the parameter's uses in this engine have been processed), so a bucket's
# all-reduce can start while the rest of the backward pass is still running.
# ---------------------------------------------------------------------------------------------------------------------
# Test hook (tests/_util.py: kink_matched): when a dict, every fused conv(+BN)+ReLU op stores its post-activation output
# under id(weight), so a test can replay the ReLU masks this implementation actually used in the float64 oracle.
RELU_CAPTURE: Optional[Dict[int, torch.Tensor]] = None
# test hook (tests/_util.py kink_matched): {id(AttentionBlock fc.0.weight): (global max-pool arg-max [N, C], per-pixel channel
# arg-max [N, H*W])} of every attention block run while it is a dict -- the other two kinks of the network besides the ReLUs
CBAM_CAPTURE: Optional[Dict[int, Tuple[torch.Tensor, torch.Tensor]]] = None
# Synchronised BatchNorm (parallel.GradientSynchronizer(sync_bn=True)): SYNC_BN(sums) all-reduces (SUM) the fp64 vector
# [sum(C) | second row (C) | count] of one BatchNorm layer over the ranks, in place, ordered after the kernels enqueued so far
# on the current stream; None = per-replica statistics.
SYNC_BN: Optional[Callable[[torch.Tensor], None]] = None
GRAD_SINK: Optional[Callable[[torch.Tensor], Optional[torch.Tensor]]] = None
GRAD_READY: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None


class Act:
    """An NHWC activation: `t` is a [N,H,W,C] float32 cuda tensor whose last-dim stride is 1 and whose
    pixel stride (t.stride(2)) may exceed C (channel slice of a wider buffer).  `C` is the logical
    channel count; `t.shape[3]` may be larger for the small padded tensors (3->8, 1->4 channels)."""

    __slots__ = ("t", "C", "grad", "needs_grad", "bn_src", "bn_partial")

    def __init__(self, t: torch.Tensor, C: Optional[int] = None, needs_grad: bool = True):
        assert t.dim() == 4 and t.stride(3) == 1
        self.t = t
        self.C = t.shape[3] if C is None else C
        self.grad: Optional[torch.Tensor] = None
        self.bn_src = None       # (y, {scale, shift}, mean) when this is the output of a train-mode Conv+BN+ReLU without residual
        self.bn_partial = None   # (rows, nblk, pitch, g): that layer's BN-backward sums, taken by the launch that wrote `grad`
        self.needs_grad = needs_grad

    @property
    def N(self): return self.t.shape[0]
    @property
    def Hh(self): return self.t.shape[1]
    @property
    def Ww(self): return self.t.shape[2]
    @property
    def cs(self): return self.t.stride(2)
    @property
    def pixels(self): return self.t.shape[0] * self.t.shape[1] * self.t.shape[2]


def _check_dense_pixels(t: torch.Tensor):
    # rows and images must be dense in units of the pixel stride
    cs = t.stride(2)
    assert t.stride(1) == t.shape[2] * cs and t.stride(0) == t.shape[1] * t.shape[2] * cs, "unsupported activation strides"


def new_act(N, Hh, Ww, C, device, alloc_C: Optional[int] = None, zero: bool = False) -> Act:
    ac = C if alloc_C is None else alloc_C
    t = (torch.zeros if zero else torch.empty)((N, Hh, Ww, ac), device=device, dtype=torch.float32)
    return Act(t, C)


class BNState:
    """Parameters/buffers of one BatchNorm2d, by reference to the owning module's tensors."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")

    def __init__(self, weight, bias, running_mean, running_var, num_batches_tracked):
        self.weight, self.bias = weight, bias
        self.running_mean, self.running_var = running_mean, running_var
        self.num_batches_tracked = num_batches_tracked


_SLAB_BUDGET = 1 << 30      # bytes of split-accumulation slab a weight-gradient launch may write (and its reduce kernel read)


def _rows_nsplit(groups: int, ntiles: int, cus: int = 256, max_rounds: int = 4, max_splits: Optional[int] = None,
                 slab_bytes: int = 0, tile_us: float = 5.0, launches: int = 1) -> int:
    """Pixel splits for the split-accumulating weight-gradient kernels.  One workgroup is resident per CU and the hardware deals
    workgroups to the 8 XCDs round-robin (block b runs on XCD b % 8, 32 CUs each); every such kernel maps block b to
    (group, split) = ((b >> 3) % groups, (b >> 3) / groups * 8 + (b & 7)), i.e. XCD x gets the splits = x mod 8 of every
    group.  A launch therefore takes rounds = ceil(groups * ceil(nsplit / 8) / 32) workgroup times of ceil(ntiles / nsplit)
    tiles each -- NOT ceil(groups * nsplit / 256): groups = 3, nsplit = 85 is 255 workgroups but 33 on each of XCDs 0-4, two
    rounds (measured: Conv2d k4 s2 96 -> 192 weight gradient 5.92 ms with 85 splits, 3.69 with 80).  Every split also writes
    `slab_bytes` of partial sums that the reduce kernel reads back (80 -> 160 splits on the same layer: 3.69 -> 3.78 ms although
    the rounds * tiles product is the same).  Returns the nsplit (up to `max_rounds` rounds, at most `max_splits`) that
    minimises  launches * rounds * tiles per split * tile_us  +  nsplit * 2 * slab_bytes / (4 TB/s),  smallest on ties."""
    if os.environ.get("ADH_NSPLIT"):          # development: force the split count
        return max(1, min(ntiles, int(os.environ["ADH_NSPLIT"])))
    per_xcd = max(1, cus // 8)
    limit = max(1, min(ntiles, max_rounds * cus // max(1, groups) + 1))
    if max_splits is not None:
        limit = max(1, min(limit, max_splits))
    best, best_cost = 1, None
    for ns in range(1, limit + 1):
        rounds = -(-(groups * (-(-ns // 8))) // per_xcd)
        cost = launches * rounds * (-(-ntiles // ns)) * tile_us + ns * 2.0 * slab_bytes / 4.0e6
        if best_cost is None or cost < best_cost * (1.0 - 5e-3):
            best, best_cost = ns, cost
    return best


class Engine:
    def __init__(self, device: torch.device, record: bool):
        self.device = device
        self.record = record
        self.tape: List[Callable[[], None]] = []
        self.param_grads: Dict[int, torch.Tensor] = {}   # id(param) -> grad
        self.params: Dict[int, torch.Tensor] = {}
        self.alias: Dict[int, int] = {}                  # id(reshaped view of a param) -> id(param)
        self.uses: Dict[int, int] = {}                   # id(param) -> uses recorded on the tape and not yet differentiated
        self.upstream: Dict[str, torch.Tensor] = {}      # 'g': device scalar cotangent of a scalar loss
        # None: follow USE_WINO43.  The loss networks set "dgrad": they difference the features of two nearly equal
        # images behind max-pools, so their forward pass keeps F(2x2,3x3) (whose rounding is below the direct kernel's)
        self.wino43: Optional[str] = None

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape, zero=False):
        return (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=torch.float32)

    def use_param(self, *ps: Optional[torch.Tensor]):
        """Forward-side bookkeeping: each recorded op declares the parameters its backward closure will add a gradient
        for, so that `add_param_grad` knows when a gradient is final (GRAD_READY)."""
        for p in ps:
            if p is not None:
                k = self.alias.get(id(p), id(p))
                self.uses[k] = self.uses.get(k, 0) + 1

    def grad_buffer(self, p: torch.Tensor, zero: bool = False) -> torch.Tensor:
        """Where a backward closure should write the gradient of `p`: the registered sink (a view of a flat bucket
        buffer) when there is one, it has not been used by an earlier op of this engine, and `p.grad` is not already
        populated (gradient accumulation across backward passes must not alias); else a fresh tensor."""
        k = self.alias.get(id(p), id(p))
        if GRAD_SINK is not None and k == id(p) and k not in self.param_grads and getattr(p, "grad", None) is None:
            buf = GRAD_SINK(p)
            if buf is not None and buf.numel() == p.numel():
                buf = buf.view(p.shape)
                if zero:
                    buf.zero_()
                return buf
        return self._f(*p.shape, zero=zero)

    def add_param_grad(self, p: torch.Tensor, g: torch.Tensor):
        k = self.alias.get(id(p), id(p))
        if k in self.param_grads:
            H.call("adh_add_inplace", self.param_grads[k].data_ptr(), g.data_ptr(), g.numel())
        else:
            self.param_grads[k] = g
            self.params[k] = p
        left = self.uses.get(k, 1) - 1
        self.uses[k] = left
        if left == 0 and GRAD_READY is not None:
            GRAD_READY(self.params[k], self.param_grads[k])

    def accum(self, act: Act, g: torch.Tensor):
        """act.grad (+)= g ; g is [N,H,W,>=C] possibly strided."""
        if not act.needs_grad:
            return
        if act.grad is None:
            act.grad = g
        else:
            act.bn_partial = None      # the fused sums covered the first contribution only
            H.call("adh_axpby_strided", act.grad.data_ptr(), act.grad.stride(2), g.data_ptr(), g.stride(2),
                   act.pixels, _round_up(act.C, 4), 1.0, 1.0)

    def backward(self):
        while self.tape:
            self.tape.pop()()

    # ------------------------------------------------------------------ layout at the boundary
    def image_to_nhwc8(self, x: torch.Tensor) -> Act:
        N, Cc, Hh, Ww = x.shape
        assert Cc == 3
        out = self._f(N, Hh, Ww, 8)
        H.call("adh_image_to_nhwc8", x.data_ptr(), N, Hh, Ww, out.data_ptr())
        return Act(out, 8, needs_grad=False)

    def image_normalize_to_nhwc8(self, x: torch.Tensor, mean, std, holder: dict) -> Act:
        """(x - mean[c]) / std[c] -> NHWC8 (loss.py:60-66); the gradient wrt the NCHW image is left in
        holder['gx'] by the backward closure."""
        N, Cc, Hh, Ww = x.shape
        out = self._f(N, Hh, Ww, 8)
        inv = [1.0 / s_ for s_ in std]
        H.call("adh_image_normalize_to_nhwc8", x.data_ptr(), N, Hh, Ww, mean[0], mean[1], mean[2], inv[0], inv[1], inv[2],
               out.data_ptr())
        a = Act(out, 8, needs_grad=self.record)
        if self.record:
            def bwd():
                g = a.grad
                a.grad = None
                if g is None:
                    return
                gx = torch.empty_like(x)
                H.call("adh_image_normalize_bwd", g.data_ptr(), g.stride(2), N, Hh, Ww, inv[0], inv[1], inv[2], gx.data_ptr())
                holder["gx"] = gx
            self.tape.append(bwd)
        return a

    def mse(self, a: Act, b: Act, scale: float, sink: list):
        """sink.append(device scalar mean((a-b)^2) * scale) (F.mse_loss, loss.py:81); gradient flows to `a` only."""
        assert a.t.is_contiguous() and b.t.is_contiguous() and a.t.shape == b.t.shape
        n = a.t.numel()
        nblk = H.value("adh_reduce_num_blocks", n)
        partial = self._f(nblk)
        val = self._f(1)
        H.call("adh_mse_partial", a.t.data_ptr(), b.t.data_ptr(), n, partial.data_ptr())
        H.call("adh_sum_partials", partial.data_ptr(), nblk, scale / n, val.data_ptr())
        sink.append(val)
        if self.record:
            def bwd():
                up = sink_grad.get("g")
                if up is None or not a.needs_grad:
                    return
                ga = self._f(*a.t.shape)
                H.call("adh_mse_bwd", a.t.data_ptr(), b.t.data_ptr(), n, scale / n, up.data_ptr(), ga.data_ptr())
                self.accum(a, ga)
            sink_grad = self.upstream
            self.tape.append(bwd)

    # ------------------------------------------------------------------ convolution family
    @staticmethod
    def _conv_desc(x: Act, Cin: int, out_t: torch.Tensor, Cout: int, NcP: int, VH, VW, KH, KW, in_s, out_s, out_o,
                   dy0, dx0, dstep) -> ConvDesc:
        d = ConvDesc()
        d.in_ = x.t.data_ptr()
        d.N, d.IH, d.IW, d.Cin, d.in_cstride = x.N, x.Hh, x.Ww, Cin, x.cs
        d.out = out_t.data_ptr()
        d.OH, d.OW, d.Cout, d.out_cstride = out_t.shape[1], out_t.shape[2], Cout, out_t.stride(2)
        d.VH, d.VW = VH, VW
        d.in_sy = d.in_sx = in_s
        d.out_sy = d.out_sx = out_s
        d.out_oy, d.out_ox = out_o
        d.KH, d.KW = KH, KW
        d.dy0, d.dx0 = dy0, dx0
        d.dstep_y = d.dstep_x = dstep
        d.act = H.ACT_NONE
        d.NcP = NcP
        return d

    def _packed(self, fn: str, w: torch.Tensor, L: WLayout, nfloats: int) -> torch.Tensor:
        """Run the pack kernel `fn` (adh_pack_weights*) for (w, L), or return the cached result for this version of w."""
        key = None
        if USE_PACK_CACHE:
            key = (id(w), w.data_ptr(), w._version, fn, _layout_key(L), nfloats, str(w.device))
            hit = _PACK_CACHE.get(key)
            if hit is not None and hit[0]() is w:
                return hit[1]
        wp = self._f(nfloats)
        H.call(fn, w.data_ptr(), C.byref(L), wp.data_ptr())
        if key is not None:
            if len(_PACK_CACHE) >= _PACK_CACHE_MAX:
                _PACK_CACHE.clear()
            try:
                _PACK_CACHE[key] = (weakref.ref(w), wp)
            except TypeError:
                pass
        return wp

    def _pack(self, w: torch.Tensor, L: WLayout) -> torch.Tensor:
        KQ = _round_up(L.K, 8) // 4
        NcP = _round_up(L.Nc, 32)
        return self._packed("adh_pack_weights", w, L, L.KHt * L.KWt * KQ * NcP * 4)

    @staticmethod
    def _launch_plan(kind: str, k: int, stride: int, pad: int, w: torch.Tensor, direction: str):
        """Return a list of launches [(WLayout, geometry dict)] realising one conv layer's forward
        ('fwd'), data gradient ('dgrad') in the gather form.  Weight gradients reuse the 'fwd' plan.

        conv  : Conv2d weight [Cout,Cin,k,k], stride 1 or 2      (base_model.py:11-13)
        convT : ConvTranspose2d weight [Cin,Cout,4,4], s2 p1      (medium_intensity.py:53,63)
        geometry: in_s, out_s, out_o, dy0, dx0, dstep, KH, KW, vgrid ('out'|'in'|'out_half')
        """
        plans = []
        if kind == "conv":
            Cout, Cin = w.shape[0], w.shape[1]
            kk = k * k
            if direction == "fwd":
                L = WLayout(Cin, Cout, k, k, 0, k, 1, kk, Cin * kk)
                plans.append((L, dict(in_s=stride, out_s=1, out_o=(0, 0), dy0=-pad, dx0=-pad, dstep=1, KH=k, KW=k,
                                      vgrid="out")))
            elif stride == 1:   # dgrad of a stride-1 conv: correlation with taps walked backwards
                L = WLayout(Cout, Cin, k, k, 0, k, 1, Cin * kk, kk)
                plans.append((L, dict(in_s=1, out_s=1, out_o=(0, 0), dy0=pad, dx0=pad, dstep=-1, KH=k, KW=k,
                                      vgrid="out")))
            else:               # dgrad of a stride-2 conv = transposed-conv form, one launch per output parity
                assert stride == 2
                for py in range(2):
                    for px in range(2):
                        ky0, kx0 = (py + pad) % 2, (px + pad) % 2
                        Ty, Tx = (k - ky0 + 1) // 2, (k - kx0 + 1) // 2
                        if Ty <= 0 or Tx <= 0:
                            continue    # no tap reaches this parity class (e.g. 1x1 stride-2): gradient is zero there
                        L = WLayout(Cout, Cin, Ty, Tx, ky0 * k + kx0, 2 * k, 2, Cin * kk, kk)
                        plans.append((L, dict(in_s=1, out_s=2, out_o=(py, px), dy0=(py + pad - ky0) // 2,
                                              dx0=(px + pad - kx0) // 2, dstep=-1, KH=Ty, KW=Tx, vgrid="in")))
        else:
            assert kind == "convT" and k == 4 and stride == 2 and pad == 1
            Cin, Cout = w.shape[0], w.shape[1]
            if direction == "fwd":
                for py in range(2):
                    for px in range(2):
                        L = WLayout(Cin, Cout, 2, 2, (1 - py) * 4 + (1 - px), 8, 2, Cout * 16, 16)
                        plans.append((L, dict(in_s=1, out_s=2, out_o=(py, px), dy0=py, dx0=px, dstep=-1, KH=2, KW=2,
                                              vgrid="in")))
            else:               # dgrad of convT = k4 s2 p1 conv of the output gradient
                L = WLayout(Cout, Cin, 4, 4, 0, 4, 1, 16, Cout * 16)
                plans.append((L, dict(in_s=2, out_s=1, out_o=(0, 0), dy0=-1, dx0=-1, dstep=1, KH=4, KW=4,
                                      vgrid="out")))
        return plans

    def _run_gather(self, plans, src: Act, dst_t: torch.Tensor, dstC: int, w: torch.Tensor, scale=None, shift=None,
                    residual: Optional[torch.Tensor] = None, act=H.ACT_NONE, want_stats=False, bnred=None):
        """Launch every plan of one layer; returns (stats partials or None, number of stat rows).
        `bnred` = (y, scale_shift[2][C4], mean) of the train-mode ConvBlock that produced the tensor this data gradient is
        the gradient of: when the layer runs as ONE F(4x4,3x3) launch the producer's BatchNorm-backward sums are taken in
        its epilogue (adh_conv_wino43_dgrad_bnred) and returned as the stats rows; otherwise (None, 0) comes back and
        nothing was fused."""
        descs = []
        total_blocks = 0
        for L, gm in plans:
            NcP = _round_up(L.Nc, 32)
            Kp = _round_up(L.K, 8)
            assert src.t.shape[3] >= Kp or src.cs >= Kp, "input activation narrower than the padded contraction"
            if gm["vgrid"] == "in":   # one output-parity class of a stride-2 transposed form
                VH = (dst_t.shape[1] - gm["out_o"][0] + 1) // 2
                VW = (dst_t.shape[2] - gm["out_o"][1] + 1) // 2
            else:
                VH, VW = dst_t.shape[1], dst_t.shape[2]
            d = self._conv_desc(src, Kp, dst_t, dstC, NcP, VH, VW, gm["KH"], gm["KW"], gm["in_s"], gm["out_s"],
                                gm["out_o"], gm["dy0"], gm["dx0"], gm["dstep"])
            wino = False
            if L.Nc <= 4 and gm["KH"] == 3 and gm["KW"] == 3 and residual is None and not want_stats and bnred is None and \
                    USE_FEWOUT and gm["dstep"] == 1 and H.value("adh_conv_fewout_supported", C.byref(d)):
                # at most four output channels (the reconstruction heads): one pixel per thread instead of a 32-wide MFMA tile
                wino = "fewout"
                wp = self._packed("adh_pack_weights_fewout", w, L, 9 * Kp * 4)
            if not wino and L.K <= 4 and Kp == 8 and 4 <= L.Nc <= 64 and L.Nc % 4 == 0 and gm["KH"] == 3 and gm["KW"] == 3 and \
                    gm["in_s"] == 1 and gm["out_s"] == 1 and residual is None and (not want_stats or L.Nc <= 16) and USE_FEWOUT and \
                    (gm["dy0"], gm["dx0"], gm["dstep"]) in ((-1, -1, 1), (1, 1, -1)):
                # at most four input channels (data gradient of the reconstruction head, 3 -> 48): a per-pixel kernel as well
                Lw = L
                dsave = (d.dy0, d.dx0, d.dstep_y, d.dstep_x)
                if gm["dstep"] == -1:   # data gradient: the same correlation with the filter flipped in both axes
                    Lw = WLayout(L.K, L.Nc, 3, 3, L.tap_off0 + 2 * L.tap_off_sy + 2 * L.tap_off_sx, -L.tap_off_sy,
                                 -L.tap_off_sx, L.stride_k, L.stride_n)
                    d.dy0 = d.dx0 = -1
                    d.dstep_y = d.dstep_x = 1
                if H.value("adh_conv_fewin_supported", C.byref(d)):
                    wino = "fewin"
                    wp = self._packed("adh_pack_weights_fewin", w, Lw, 9 * 4 * 64)
                else:
                    d.dy0, d.dx0, d.dstep_y, d.dstep_x = dsave
            if not wino and USE_WINOGRAD and (not _WINO_ONLY or _WINO_ONLY == ("dgrad" if gm["dstep"] == -1 else "fwd")) \
                    and gm["KH"] == 3 and gm["KW"] == 3 and gm["in_s"] == 1 and gm["out_s"] == 1 and Kp % 16 == 0 \
                    and (gm["dy0"], gm["dx0"], gm["dstep"]) in ((-1, -1, 1), (1, 1, -1)):
                Lw = L
                if gm["dstep"] == -1:   # data gradient: the same correlation with the filter flipped in both axes
                    Lw = WLayout(L.K, L.Nc, 3, 3, L.tap_off0 + 2 * L.tap_off_sy + 2 * L.tap_off_sx, -L.tap_off_sy,
                                 -L.tap_off_sx, L.stride_k, L.stride_n)
                    d.dy0 = d.dx0 = -1
                    d.dstep_y = d.dstep_x = 1
                w43 = USE_WINO43 if (self.wino43 is None or USE_WINO43 is not True) else self.wino43
                if (w43 is True or w43 == ("dgrad" if gm["dstep"] == -1 else "fwd")) and \
                        H.value("adh_conv_wino43_supported", C.byref(d)):
                    wino = 43
                    if CONTRACT == "bf16x3":   # three bf16 planes of U: 6 bytes per weight and frequency
                        wp = self._packed("adh_pack_weights_wino43_bf16x3", w, Lw, 36 * Kp * NcP * 3 // 2)
                    else:
                        wp = self._packed("adh_pack_weights_wino43", w, Lw, 36 * (Kp // 4) * NcP * 4)
                else:
                    wino = bool(H.value("adh_conv_wino_supported", C.byref(d)))
                if wino == 43:
                    pass
                elif wino:
                    KQ = Kp // 4
                    wp = self._packed("adh_pack_weights_wino", w, Lw, 16 * KQ * NcP * 4)
                else:
                    d.dy0 = d.dx0 = gm["dy0"]
                    d.dstep_y = d.dstep_x = gm["dstep"]
            if not wino and USE_WINOGRAD and Kp % 16 == 0 and \
                    ((gm["KH"] == 2 and gm["KW"] == 2 and gm["in_s"] == 1 and gm["dstep"] in (1, -1)) or
                     (gm["KH"] == 4 and gm["KW"] == 4 and gm["in_s"] == 2 and gm["dstep"] == 1)):
                # F(3x3,2x2): 2x2-tap forms (one parity class of a transposed conv / of a k4 s2 data gradient) and the
                # k4 s2 forms as four input-parity classes.  Backward-walking taps = forward-walking with the filter flipped.
                Lw = L
                if gm["dstep"] == -1:
                    Lw = WLayout(L.K, L.Nc, 2, 2, L.tap_off0 + L.tap_off_sy + L.tap_off_sx, -L.tap_off_sy, -L.tap_off_sx,
                                 L.stride_k, L.stride_n)
                    d.dy0, d.dx0 = gm["dy0"] - 1, gm["dx0"] - 1
                    d.dstep_y = d.dstep_x = 1
                if H.value("adh_conv_wino32_supported", C.byref(d)):
                    wino = 32
                    ncls = 4 if gm["KH"] == 4 else 1
                    if CONTRACT == "bf16x3":
                        wp = self._packed("adh_pack_weights_wino32_bf16x3", w, Lw, ncls * 16 * Kp * NcP * 3 // 2)
                    else:
                        wp = self._packed("adh_pack_weights_wino32", w, Lw, ncls * 16 * (Kp // 4) * NcP * 4)
                else:
                    d.dy0 = d.dx0 = gm["dy0"]
                    d.dstep_y = d.dstep_x = gm["dstep"]
            if not wino and gm["KH"] == 7 and L.K <= 3 and residual is None and USE_SMALL_WGRAD and \
                    H.value("adh_conv_stem_num_blocks", C.byref(d)):
                wino = "stem"   # 7x7 stem on the NHWC8 image: (kx, c)-packed 16x16x4 tiles (conv_stem.hip)
                wp = self._packed("adh_pack_weights_stem", w, L, 7 * 24 * NcP)
            if not wino:
                wp = self._pack(w, L)   # keep alive until the launch below is enqueued
            d.wp = wp.data_ptr()
            d.scale = H.ptr(scale)
            d.shift = H.ptr(shift)
            if residual is not None:
                d.residual = residual.data_ptr()
                d.res_cstride = residual.stride(2)
            d.act = act
            if wino == "fewout":
                nb = 0
            else:
                nb = H.value({32: "adh_conv_wino32_num_blocks", 43: "adh_conv_wino43_num_blocks", True: "adh_conv_wino_num_blocks",
                              "stem": "adh_conv_stem_num_blocks", "fewin": "adh_conv_fewin_num_blocks",
                              False: "adh_conv_num_blocks"}[wino], C.byref(d))
            descs.append((d, nb, wp, wino))
            total_blocks += nb
        if bnred is not None:
            if len(descs) != 1 or descs[0][3] != 43 or residual is not None or scale is not None or shift is not None:
                bnred = None
            else:
                y_p, ss_p, mean_p = bnred
                d0 = descs[0][0]
                d0.residual, d0.res_cstride = y_p.data_ptr(), y_p.stride(2)
                d0.scale, d0.shift = ss_p[0].data_ptr(), ss_p[1].data_ptr()
                want_stats = True
        _B3 = "_bf16x3" if CONTRACT == "bf16x3" else ""      # entry-point suffix of the opt-in contraction
        stats = None
        if want_stats:
            NcP = descs[0][0].NcP
            stats = self._f(total_blocks, 2, NcP)
        row = 0
        row_i = 0
        flops_kn = [L.K * L.Nc for L, _ in plans]
        # The output-parity classes of a transposed form are independent launches whose grids are not multiples of the CU count
        # (e.g. 1056 workgroups = 4.125 rounds of one workgroup per CU: the last round runs on 1/8 of the chip).  On separate
        # streams the next class fills the CUs the previous one's tail leaves idle.
        if MERGE_CLASSES and 2 <= len(descs) <= 4 and all(w == 32 and dd.KH == 2 for dd, _, _, w in descs):
            # the output-parity classes of a transposed form as ONE grid: one partial last round of workgroups instead of four
            for dd, nb, _wp, _w in descs:
                if stats is not None:
                    dd.stats = stats.data_ptr() + row * 2 * dd.NcP * 4
                row += nb
            arr = (H.ConvDesc * len(descs))(*[dd for dd, _, _, _ in descs])
            work = sum(2.0 * dd.N * dd.VH * dd.VW * dd.KH * dd.KW * kn for (dd, _, _, _), kn in zip(descs, flops_kn))
            try:
                H.call("adh_conv_wino32_forward_multi" + _B3, arr, len(descs), work=work, work_exec=work * 4.0 / 9.0,
                       family="adh_conv_wino32_forward")
                return stats, total_blocks
            except RuntimeError as e:
                if "unsupported" not in str(e).lower():
                    raise
                row = 0           # descriptors that differ in more than the class fields: one launch each, below
        fork = None
        if CLASS_STREAMS and len(descs) > 1 and all(w == 32 for _, _, _, w in descs):
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            side = _side_streams(self.device, min(4, len(descs)))
        for d, nb, _wp, wino in descs:
            if stats is not None:
                d.stats = stats.data_ptr() + row * 2 * d.NcP * 4
            if fork is not None:
                st = side[row_i % len(side)]
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    work = 2.0 * d.N * d.VH * d.VW * d.KH * d.KW * flops_kn[row_i]
                    H.call("adh_conv_wino32_forward" + _B3, C.byref(d), work=work, work_exec=work * 4.0 / 9.0,
                           family="adh_conv_wino32_forward")
                row += nb
                row_i += 1
                continue
            # algorithmic FLOPs of this launch: 2 * virtual pixels * taps * real K * real Nc (Winograd executes 4/9)
            work = 2.0 * d.N * d.VH * d.VW * d.KH * d.KW * flops_kn[row_i]
            if wino == "fewout":
                H.call("adh_conv_fewout_forward", C.byref(d), work=work, family="adh_conv_forward")
            elif wino == "fewin":
                H.call("adh_conv_fewin_forward", C.byref(d), work=work, family="adh_conv_forward")
            elif wino == "stem":
                H.call("adh_conv_stem_forward", C.byref(d), work=work)
            elif wino == 43 and bnred is not None:
                H.call("adh_conv_wino43_dgrad_bnred" + ("_bf16x3" if CONTRACT == "bf16x3" else ""), C.byref(d), bnred[2].data_ptr(),
                       work=work, work_exec=work * 0.25, family="adh_conv_wino43_forward")
            elif wino == 43:
                H.call("adh_conv_wino43_forward" + ("_bf16x3" if CONTRACT == "bf16x3" else ""), C.byref(d), work=work,
                       work_exec=work * 0.25, family="adh_conv_wino43_forward")
            elif wino == 32:
                H.call("adh_conv_wino32_forward" + _B3, C.byref(d), work=work, work_exec=work * 4.0 / 9.0,
                       family="adh_conv_wino32_forward")
            elif wino:
                H.call("adh_conv_wino_forward", C.byref(d), work=work, work_exec=work * 4.0 / 9.0)
            else:
                H.call("adh_conv_forward", C.byref(d), work=work)
            row += nb
            row_i += 1
        if fork is not None:
            for st in side:
                join = torch.cuda.Event()
                join.record(st)
                main.wait_event(join)
        return stats, total_blocks

    def _wgrad(self, plans, x: Act, g_y: torch.Tensor, gC: int, w: torch.Tensor) -> torch.Tensor:
        """Weight gradient in the parameter's own layout (OIHW / IOHW)."""
        dw = self.grad_buffer(w)
        if x.C == 8 and x.cs == 8 and w.dim() == 4 and w.shape[1] <= 8 and w.shape[2] == 7 and len(plans) == 1 \
                and plans[0][1]["in_s"] == 1:
            gm = plans[0][1]
            if USE_SMALL_WGRAD and w.shape[1] <= 3:
                L = plans[0][0]
                NcP = _round_up(L.Nc, 32)
                VH, VW = g_y.shape[1], g_y.shape[2]
                d = self._conv_desc(x, 4, g_y, _round_up(gC, 4), NcP, VH, VW, 7, 7, 1, 1, (0, 0), gm["dy0"], gm["dx0"], 1)
                nslabs = H.value("adh_conv_wgrad_stem_slabs", C.byref(d))
                if nslabs:   # (kx, c)-packed 16x16x4 tiles (conv_stem.hip)
                    slab = self._f(nslabs * 49 * 8 * NcP)
                    H.call("adh_conv_wgrad_stem", C.byref(d), slab.data_ptr(), NcP,
                           work=2.0 * d.N * VH * VW * 49 * L.K * L.Nc)
                    H.call("adh_wgrad_reduce_small", slab.data_ptr(), nslabs, 8, NcP, C.byref(L), dw.data_ptr(), 0)
                    return dw
            return self._wgrad_packed_stem(gm, x, g_y, gC, w, dw)
        if MERGE_CLASSES and USE_WINOGRAD and 2 <= len(plans) <= 4 and all(gm["KH"] == 2 and gm["KW"] == 2 for _, gm in plans) \
                and self._wgrad_merged_classes(plans, x, g_y, gC, dw):
            return dw
        for L, gm in plans:
            NcP = _round_up(L.Nc, 32)
            KP = _round_up(L.K, 32)
            Cin4 = _round_up(L.K, 4)
            if gm["vgrid"] == "in":
                VH = (g_y.shape[1] - gm["out_o"][0] + 1) // 2
                VW = (g_y.shape[2] - gm["out_o"][1] + 1) // 2
            else:
                VH, VW = g_y.shape[1], g_y.shape[2]
            d = self._conv_desc(x, Cin4, g_y, _round_up(gC, 4), NcP, VH, VW, gm["KH"], gm["KW"], gm["in_s"],
                                gm["out_s"], gm["out_o"], gm["dy0"], gm["dx0"], gm["dstep"])
            ntiles_est = x.N * ((VH + 3) // 4) * ((VW + 31) // 32)
            T_all = gm["KH"] * gm["KW"]
            zgroups = 7 if T_all == 49 else 1
            groups = (KP // 32) * max(1, NcP // 96) * zgroups
            # one 512-thread workgroup per CU is resident: aim at ~4 rounds of 256 workgroups
            T = gm["KH"] * gm["KW"]
            small = H.value("adh_conv_wgrad_small_slabs", C.byref(d)) if USE_SMALL_WGRAD else 0
            if small:
                # 3x3 stride-1 with few channels (guidance branch, output convolution): 16x16x4 MFMA tiles
                slab = self._f(small * 9 * KP * NcP)
                H.call("adh_conv_wgrad_small", C.byref(d), slab.data_ptr(), KP, NcP,
                       work=2.0 * d.N * d.VH * d.VW * T * L.K * L.Nc)
                H.call("adh_wgrad_reduce_small", slab.data_ptr(), small, KP, NcP, C.byref(L), dw.data_ptr(), 0)
                continue
            w43_groups = H.value("adh_conv_wgrad_wino43_groups", C.byref(d)) if (USE_WINOGRAD and USE_WINO43_WGRAD) else 0
            if w43_groups:
                # 3x3 stride-1, Cin % 32 == 0, Cout % 96 == 0: accumulate in the F(4x4,3x3) domain (36 frequency slabs)
                nstrips = H.value("adh_conv_wgrad_wino43_strips", C.byref(d))
                nsplit = _rows_nsplit(w43_groups, nstrips, max_rounds=W43_WGRAD_ROUNDS, slab_bytes=36 * KP * NcP * 4,
                                      max_splits=max(1, _SLAB_BUDGET // (36 * KP * NcP * 4)), tile_us=3.6)
                while nsplit * 36 * KP * NcP * 4 > (1 << 30) and nsplit > 1:
                    nsplit //= 2
                slab = self._f(nsplit * 36 * KP * NcP)
                H.call("adh_conv_wgrad_wino43", C.byref(d), slab.data_ptr(), nsplit,
                       work=2.0 * d.N * d.VH * d.VW * T * L.K * L.Nc, work_exec=2.0 * d.N * d.VH * d.VW * 2.25 * L.K * L.Nc)
                H.call("adh_wgrad_reduce_wino43", slab.data_ptr(), nsplit, KP, NcP, C.byref(L), dw.data_ptr(), 0)
                continue
            wino_groups = H.value("adh_conv_wgrad_wino_groups", C.byref(d)) if USE_WINOGRAD else 0
            if wino_groups:
                # 3x3 stride-1: accumulate in the Winograd domain (16 frequency slabs), G^T(.)G in the reduce
                nsplit = _rows_nsplit(wino_groups, ntiles_est, slab_bytes=16 * KP * NcP * 4,
                                      max_splits=max(1, _SLAB_BUDGET // (16 * KP * NcP * 4)), tile_us=8.6)
                while nsplit * 16 * KP * NcP * 4 > (1 << 30) and nsplit > 1:
                    nsplit //= 2
                slab = self._f(nsplit * 16 * KP * NcP)
                H.call("adh_conv_wgrad_wino", C.byref(d), slab.data_ptr(), nsplit,
                       work=2.0 * d.N * d.VH * d.VW * T * L.K * L.Nc, work_exec=2.0 * d.N * d.VH * d.VW * 4 * L.K * L.Nc)
                H.call("adh_wgrad_reduce_wino", slab.data_ptr(), nsplit, KP, NcP, C.byref(L), dw.data_ptr(), 0)
                continue
            w32_groups = H.value("adh_conv_wgrad_wino32_groups", C.byref(d)) if USE_WINOGRAD else 0
            if w32_groups:
                # the 2x2-tap forms (k4 s2 / transposed layers): accumulate in the F(3x3,2x2) domain, 16 frequency slabs per
                # class, A^T(.)A in the reduce
                ncls = H.value("adh_conv_wgrad_wino32_classes", C.byref(d))
                # (conv_wgrad32v2_kernel runs the classes of a k4 s2 form in ONE grid: classes x groups workgroup groups)
                launches = max(1, H.value("adh_conv_wgrad_wino32_launches", C.byref(d)))
                nsplit = _rows_nsplit(w32_groups * ncls // launches, H.value("adh_conv_wgrad_wino32_tiles", C.byref(d)),
                                      launches=launches, slab_bytes=ncls * 16 * KP * NcP * 4, tile_us=5.0,
                                      max_splits=max(1, _SLAB_BUDGET // (ncls * 16 * KP * NcP * 4)))
                while nsplit * ncls * 16 * KP * NcP * 4 > (1 << 30) and nsplit > 1:
                    nsplit //= 2
                slab = self._f(nsplit * ncls * 16 * KP * NcP)
                H.call("adh_conv_wgrad_wino32", C.byref(d), slab.data_ptr(), nsplit,
                       work=2.0 * d.N * d.VH * d.VW * T * L.K * L.Nc, work_exec=2.0 * d.N * d.VH * d.VW * T * (4.0 / 9.0) * L.K * L.Nc)
                H.call("adh_wgrad_reduce_wino32", slab.data_ptr(), nsplit, C.byref(d), KP, NcP, C.byref(L), dw.data_ptr(), 0)
                continue
            nsplit = max(1, min(ntiles_est, max(1, 1024 // groups), 512))
            rows_groups = H.value("adh_conv_wgrad_groups", C.byref(d))
            if rows_groups:
                nsplit = _rows_nsplit(rows_groups, ntiles_est)
            T = gm["KH"] * gm["KW"]
            nslabs = H.value("adh_conv_wgrad_slabs", C.byref(d), nsplit)
            # cap the slab at 1 GiB
            while nslabs * T * KP * NcP * 4 > (1 << 30) and nsplit > 1:
                nsplit //= 2
                nslabs = H.value("adh_conv_wgrad_slabs", C.byref(d), nsplit)
            slab = self._f(nslabs * T * KP * NcP)
            H.call("adh_conv_wgrad", C.byref(d), slab.data_ptr(), nsplit,
                   work=2.0 * d.N * d.VH * d.VW * T * L.K * L.Nc)
            H.call("adh_wgrad_reduce", slab.data_ptr(), nslabs, KP, NcP, C.byref(L), dw.data_ptr(), 0)
        return dw

    def _wgrad_merged_classes(self, plans, x: Act, g_y: torch.Tensor, gC: int, dw: torch.Tensor) -> bool:
        """The output-parity classes of a transposed layer through ONE launch of conv_wgrad32v2_kernel
        (adh_conv_wgrad_wino32_multi): classes x groups workgroup groups pack into whole rounds of the chip.  False: not its
        shapes -- the caller launches the classes one by one."""
        descs = []
        for L, gm in plans:
            NcP = _round_up(L.Nc, 32)
            if gm["vgrid"] == "in":
                VH = (g_y.shape[1] - gm["out_o"][0] + 1) // 2
                VW = (g_y.shape[2] - gm["out_o"][1] + 1) // 2
            else:
                VH, VW = g_y.shape[1], g_y.shape[2]
            descs.append(self._conv_desc(x, _round_up(L.K, 4), g_y, _round_up(gC, 4), NcP, VH, VW, gm["KH"], gm["KW"], gm["in_s"],
                                         gm["out_s"], gm["out_o"], gm["dy0"], gm["dx0"], gm["dstep"]))
        L0 = plans[0][0]
        d0, n = descs[0], len(descs)
        KP, NcP = d0.Cin, d0.NcP
        if any(L.K != L0.K or L.Nc != L0.Nc for L, _ in plans) or KP != _round_up(L0.K, 32):
            return False
        groups = H.value("adh_conv_wgrad_wino32_groups", C.byref(d0))
        if not groups or H.value("adh_conv_wgrad_wino32_classes", C.byref(d0)) != 1:
            return False
        plane = 16 * KP * NcP
        nsplit = _rows_nsplit(groups * n, H.value("adh_conv_wgrad_wino32_tiles", C.byref(d0)), slab_bytes=n * plane * 4,
                              tile_us=5.0, max_splits=max(1, _SLAB_BUDGET // (n * plane * 4)))
        slab = self._f(nsplit * n * plane)
        arr = (H.ConvDesc * n)(*descs)
        work = sum(2.0 * dd.N * dd.VH * dd.VW * 4 * L.K * L.Nc for dd, (L, _) in zip(descs, plans))
        try:
            H.call("adh_conv_wgrad_wino32_multi", arr, n, slab.data_ptr(), nsplit, work=work, work_exec=work * 4.0 / 9.0,
                   family="adh_conv_wgrad_wino32")
        except RuntimeError as e:
            if "unsupported" not in str(e).lower():
                raise
            return False
        for m, (L, _) in enumerate(plans):     # the splits are summed: one class plane each
            H.call("adh_wgrad_reduce_wino32", slab.data_ptr() + m * plane * 4, 1, C.byref(descs[m]), KP, NcP, C.byref(L),
                   dw.data_ptr(), 0)
        return True

    def _wgrad_packed_stem(self, gm, x: Act, g_y: torch.Tensor, gC: int, w: torch.Tensor, dw: torch.Tensor):
        """7x7 stem (Cin 3 stored as NHWC8): 4 adjacent pixels x 8 channels fill one 32-wide MFMA row tile, so a
        launch handles taps (ky, group of 4 kx) -- 14 packed taps instead of 49 mostly-empty ones."""
        Cout, Cin, KH, KW = w.shape
        NcP = _round_up(Cout, 32)
        KWg = (KW + 3) // 4
        VH, VW = g_y.shape[1], g_y.shape[2]
        d = self._conv_desc(x, 8, g_y, _round_up(gC, 4), NcP, VH, VW, KH, KWg, 1, 1, (0, 0), gm["dy0"], gm["dx0"], 1)
        d.dstep_x = 4
        ntiles_est = x.N * ((VH + 3) // 4) * ((VW + 31) // 32)
        groups = max(1, NcP // 96) * KWg
        nsplit = max(1, min(ntiles_est, max(1, 1024 // groups), 512))
        slab = self._f(nsplit * KH * KWg * 32 * NcP)
        H.call("adh_conv_wgrad", C.byref(d), slab.data_ptr(), nsplit, work=2.0 * d.N * VH * VW * KH * KW * Cin * Cout)
        H.call("adh_wgrad_reduce_packed", slab.data_ptr(), nsplit, NcP, Cin, KH, KW, Cout, dw.data_ptr(), 0)
        return dw

    def _channel_sum(self, g: torch.Tensor, Cc: int) -> torch.Tensor:
        """sum over pixels of g[..., :Cc] (bias gradient) using the BN-backward reduction kernels."""
        P = g.shape[0] * g.shape[1] * g.shape[2]
        C4 = _round_up(Cc, 4)
        nblk = H.value("adh_bn_bwd_num_blocks", P, C4)
        partial = self._f(nblk, 2, C4)
        zeros = self._f(C4, zero=True)
        H.call("adh_bn_bwd_reduce", g.data_ptr(), g.stride(2), None, 0, H.ACT_NONE, g.data_ptr(), g.stride(2),
               zeros.data_ptr(), zeros.data_ptr(), partial.data_ptr(), P, C4, None, None)
        dbeta = self._f(C4)
        coef = self._f(3, C4)
        H.call("adh_bn_bwd_finalize", partial.data_ptr(), nblk, C4, float(P), None, zeros.data_ptr(), None,
               dbeta.data_ptr(), 0, coef.data_ptr())
        return dbeta[:Cc]

    def conv(self, x: Act, w: torch.Tensor, b: Optional[torch.Tensor], bn: Optional[BNState], *, kind: str = "conv",
             k: int = 3, stride: int = 1, pad: int = 1, relu: bool = True, residual: Optional[Act] = None,
             training: bool = False, out: Optional[torch.Tensor] = None, out_alloc_C: Optional[int] = None) -> Act:
        """ConvBlock / ConvTranspose+BN+ReLU / bare Conv2d as one fused op
        (base_model.py:4-24,26-41; medium_intensity.py:52-56).  `residual` is added after BN and before the
        ReLU (ResidualBlock tail).  `out`: optional preallocated [N,OH,OW,>=Cout] view to write into."""
        if kind == "conv":
            Cout = w.shape[0]
            OH = (x.Hh + 2 * pad - k) // stride + 1
            OW = (x.Ww + 2 * pad - k) // stride + 1
        else:
            Cout = w.shape[1]
            OH, OW = x.Hh * 2, x.Ww * 2
        N = x.N
        act_code = H.ACT_RELU if relu else H.ACT_NONE
        if out is None:
            # channel allocation is a multiple of 8 (zero padded) so the tensor can feed the MFMA kernels
            ac = out_alloc_C if out_alloc_C is not None else _round_up(Cout, 8)
            out = self._f(N, OH, OW, ac, zero=(ac != Cout))
        _check_dense_pixels(out)
        plans = self._launch_plan(kind, k, stride, pad, w, "fwd")
        res_t = residual.t if residual is not None else None
        P = N * OH * OW

        if bn is not None and training:
            # raw conv output + per-block statistics, then normalise (+residual, ReLU) in one streaming pass
            y = self._f(N, OH, OW, _round_up(Cout, 4))
            stats, nblk = self._run_gather(plans, x, y, Cout, w, shift=b, want_stats=True)
            ss = self._f(2, _round_up(Cout, 4), zero=True)   # {scale, shift}: kept for the backward ReLU mask
            scale, shift = ss[0], ss[1]
            mean, invstd = self._f(Cout), self._f(Cout)
            NcP = _round_up(Cout, 32)
            if SYNC_BN is not None:
                # statistics over the GLOBAL batch: local sums in fp64 -> all-reduce of 2 C + 1 doubles -> finalize
                sums = torch.empty(2 * Cout + 1, device=self.device, dtype=torch.float64)
                H.call("adh_bn_partial_sums", stats.data_ptr(), nblk, NcP, Cout, float(P), sums.data_ptr())
                SYNC_BN(sums)
                H.call("adh_bn_finalize_sums", sums.data_ptr(), Cout, bn.weight.data_ptr(), bn.bias.data_ptr(), BN_EPS,
                       BN_MOMENTUM, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                       mean.data_ptr(), invstd.data_ptr(), H.ptr(bn.num_batches_tracked))
            else:
                H.call("adh_bn_finalize", stats.data_ptr(), nblk, NcP, Cout, float(P), bn.weight.data_ptr(),
                       bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                       scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), H.ptr(bn.num_batches_tracked))
            # residual + ReLU (ResidualBlock tail): the backward ReLU mask cannot be recomputed from y alone; keep it as one
            # bit per element (1/32 of `out`) written by this pass instead of reading `out` twice in the backward pass
            mbits = None
            if USE_RELU_BITS and relu and res_t is not None and self.record and Cout % 8 == 0:
                mbits = torch.empty((P * Cout + 7) // 8, device=self.device, dtype=torch.uint8)
            H.call("adh_bn_apply", y.data_ptr(), y.stride(2), scale.data_ptr(), shift.data_ptr(), H.ptr(res_t),
                   res_t.stride(2) if res_t is not None else 0, act_code, out.data_ptr(), out.stride(2), P, Cout, H.ptr(mbits),
                   work=4.0 * P * Cout * (3 if res_t is not None else 2))     # bytes: read y (+ residual), write out
            saved = ("train", y, mean, invstd, ss, mbits)
        elif bn is not None:
            scale, shift = self._f(Cout), self._f(Cout)
            H.call("adh_bn_fold_eval", Cout, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                   bn.running_var.data_ptr(), BN_EPS, H.ptr(b), scale.data_ptr(), shift.data_ptr())
            self._run_gather(plans, x, out, Cout, w, scale=scale, shift=shift, residual=res_t, act=act_code)
            saved = ("eval", scale)
        else:
            self._run_gather(plans, x, out, Cout, w, shift=b, residual=res_t, act=act_code)
            saved = ("plain",)

        o = Act(out, Cout)
        if self.record and saved[0] == "train" and relu and residual is None and USE_BN_FUSED_REDUCE and SYNC_BN is None:
            o.bn_src = (saved[1], saved[4], saved[2])
        if RELU_CAPTURE is not None and relu:
            RELU_CAPTURE[id(w)] = out
        if self.record:
            # the parameters _conv_backward will produce a gradient for (must mirror its add_param_grad calls)
            bn_grads = bn is not None and (training or bn.weight.requires_grad or bn.bias.requires_grad)
            self.use_param(w if (w.requires_grad or self.alias.get(id(w)) is not None) else None, b,
                           bn.weight if bn_grads else None, bn.bias if bn_grads else None)
            self.tape.append(lambda: self._conv_backward(x, w, b, bn, kind, k, stride, pad, relu, residual, o, saved))
        return o

    def _conv_backward(self, x: Act, w, b, bn, kind, k, stride, pad, relu, residual, o: Act, saved):
        g = o.grad
        o.grad = None
        if g is None:
            return
        Cout = o.C
        C4 = _round_up(Cout, 4)
        N, OH, OW = o.N, o.Hh, o.Ww
        P = N * OH * OW
        act_code = H.ACT_RELU if relu else H.ACT_NONE
        mode = saved[0]
        C8 = _round_up(Cout, 8)
        g_y = self._f(N, OH, OW, C8, zero=(C8 != C4))   # padded channels must be finite zeros (dgrad reads them)
        g_res = None
        if residual is not None and residual.needs_grad:
            g_res = self._f(N, OH, OW, C4)
        if mode == "train":
            _, y, mean, invstd, ss, mbits = saved
            # without a residual the ReLU mask is recomputed from y (fma(y, scale, shift) > 0, the forward expression):
            # the two backward passes then read two tensors each instead of three
            mask_ss = ss.data_ptr() if (relu and residual is None) else None
            fused = o.bn_partial if (o.bn_partial is not None and o.bn_partial[3] is g and SYNC_BN is None) else None
            o.bn_partial = None
            if fused is None:
                nblk = H.value("adh_bn_bwd_num_blocks", P, C4)
                partial = self._f(nblk, 2, C4)
                H.call("adh_bn_bwd_reduce", g.data_ptr(), g.stride(2), o.t.data_ptr(), o.cs, act_code, y.data_ptr(),
                       y.stride(2), mean.data_ptr(), invstd.data_ptr(), partial.data_ptr(), P, C4, mask_ss, H.ptr(mbits),
                       work=4.0 * P * Cout * (2 if (mask_ss is not None or mbits is not None or not relu) else 3))   # g, y (+ out)
            if C4 == Cout:
                dgamma, dbeta = self.grad_buffer(bn.weight), self.grad_buffer(bn.bias)
            else:
                dgamma, dbeta = self._f(C4), self._f(C4)
            coef = self._f(3, C4)
            if SYNC_BN is not None:
                # d-gamma / d-beta: local sums (averaged with the other gradients); the means inside the input gradient:
                # global sums (torch.nn.SyncBatchNorm's backward)
                loc = torch.empty(2 * C4 + 1, device=self.device, dtype=torch.float64)
                H.call("adh_bn_partial_sums", partial.data_ptr(), nblk, C4, C4, float(P), loc.data_ptr())
                glob = loc.clone()
                SYNC_BN(glob)
                H.call("adh_bn_bwd_finalize_sums", loc.data_ptr(), glob.data_ptr(), C4, bn.weight.data_ptr(), invstd.data_ptr(),
                       dgamma.data_ptr(), dbeta.data_ptr(), 0, coef.data_ptr())
            elif fused is not None:
                # the sums came with g: rows of (sum g m, sum g m (y - mean)) from the consumer's data-gradient epilogue
                H.call("adh_bn_bwd_finalize_centered", fused[0].data_ptr(), fused[1], fused[2], C4, float(P), bn.weight.data_ptr(),
                       invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0, coef.data_ptr())
            else:
                H.call("adh_bn_bwd_finalize", partial.data_ptr(), nblk, C4, float(P), bn.weight.data_ptr(),
                       invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0, coef.data_ptr())
            H.call("adh_bn_bwd_apply", g.data_ptr(), g.stride(2), o.t.data_ptr(), o.cs, act_code, y.data_ptr(),
                   y.stride(2), mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), 1, g_y.data_ptr(), g_y.stride(2),
                   H.ptr(g_res), g_res.stride(2) if g_res is not None else 0, P, C4, mask_ss, H.ptr(mbits),
                   work=4.0 * P * Cout * ((2 if (mask_ss is not None or mbits is not None or not relu) else 3) + 1 +
                                          (1 if g_res is not None else 0)))
            self.add_param_grad(bn.weight, dgamma[:Cout])
            self.add_param_grad(bn.bias, dbeta[:Cout])
            if b is not None:   # a bias feeding train-mode BN has an exactly zero gradient
                self.add_param_grad(b, self._f(Cout, zero=True))
        else:
            coef = self._f(3, C4, zero=True)
            if mode == "eval":
                coef[0, :Cout].copy_(saved[1])
            else:
                coef[0].fill_(1.0)
            H.call("adh_bn_bwd_apply", g.data_ptr(), g.stride(2), o.t.data_ptr(), o.cs, act_code, None, 0, None, None,
                   coef.data_ptr(), 0, g_y.data_ptr(), g_y.stride(2), H.ptr(g_res),
                   g_res.stride(2) if g_res is not None else 0, P, C4, None, None)
            if mode == "eval" and (bn.weight.requires_grad or bn.bias.requires_grad):
                # frozen-statistics BN (fine-tuning under module.eval()): dbeta = sum(g'), dgamma = sum(g' * xhat) with
                # xhat = (out_pre - beta) / gamma recovered from the block output (where the ReLU mask is off g' is 0 and
                # xhat does not matter); the train-mode reduction kernels do the sums
                pre = o.t
                if residual is not None:     # out_pre = out - residual (rare path: one extra pass)
                    pre = o.t[..., :C4].clone() if o.t.shape[3] >= C4 and o.t[..., :C4].is_contiguous() else None
                    if pre is None:
                        pre = self._f(N, OH, OW, C4)
                        H.call("adh_axpby_strided", pre.data_ptr(), C4, o.t.data_ptr(), o.cs, P, C4, 0.0, 1.0)
                    H.call("adh_axpby_strided", pre.data_ptr(), pre.stride(2), residual.t.data_ptr(), residual.cs, P, C4,
                           1.0, -1.0)
                vmean, vinv = self._f(C4), self._f(C4)
                H.call("adh_bn_eval_bwd_vectors", Cout, C4, bn.weight.data_ptr(), bn.bias.data_ptr(), vmean.data_ptr(),
                       vinv.data_ptr())
                nblk = H.value("adh_bn_bwd_num_blocks", P, C4)
                partial = self._f(nblk, 2, C4)
                H.call("adh_bn_bwd_reduce", g.data_ptr(), g.stride(2), o.t.data_ptr(), o.cs, act_code, pre.data_ptr(),
                       pre.stride(2), vmean.data_ptr(), vinv.data_ptr(), partial.data_ptr(), P, C4, None, None)
                dgamma, dbeta, scratch = self._f(C4), self._f(C4), self._f(3, C4)
                ones = self._f(C4)
                ones.fill_(1.0)
                H.call("adh_bn_bwd_finalize", partial.data_ptr(), nblk, C4, float(P), ones.data_ptr(), ones.data_ptr(),
                       dgamma.data_ptr(), dbeta.data_ptr(), 0, scratch.data_ptr())
                self.add_param_grad(bn.weight, dgamma[:Cout])
                self.add_param_grad(bn.bias, dbeta[:Cout])
            if b is not None:
                self.add_param_grad(b, self._channel_sum(g_y, Cout))
        if g_res is not None:
            self.accum(residual, g_res)
        # weight gradient (skipped for frozen weights, e.g. the VGG16 feature extractor of the content loss)
        if w.requires_grad or self.alias.get(id(w)) is not None:
            self.add_param_grad(w, self._wgrad(self._launch_plan(kind, k, stride, pad, w, "fwd"), x, g_y, Cout, w))
        # data gradient
        if x.needs_grad:
            plans = self._launch_plan(kind, k, stride, pad, w, "dgrad")
            gsrc = Act(g_y, Cout)
            if x.grad is None:
                sparse = (kind == "conv" and stride == 2 and len(plans) < 4)
                gx = self._f(x.N, x.Hh, x.Ww, _round_up(x.C, 4), zero=sparse)
                # x = ReLU(BN(conv)) of a train-mode ConvBlock and this is the first gradient to reach it: take that BN's
                # backward sums in this launch's epilogue.  They are used only if no other gradient is added to x.grad
                # afterwards (accum / the in-place branch below drop them; the producer checks `grad is gx`).
                rows, nrows = self._run_gather(plans, gsrc, gx, x.C, w, bnred=x.bn_src if SYNC_BN is None else None)
                x.grad = gx
                x.bn_partial = (rows, nrows, _round_up(x.C, 32), gx) if rows is not None else None
            else:   # accumulate in place through the epilogue's residual input
                x.bn_partial = None
                self._run_gather(plans, gsrc, x.grad, x.C, w, residual=x.grad)

    # ------------------------------------------------------------------ attention (base_model.py:43-78)
    def attention(self, x: Act, w1: torch.Tensor, w2: torch.Tensor, wsp: torch.Tensor,
                  out: Optional[torch.Tensor] = None) -> Act:
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        HW = Hh * Ww
        Ch = w1.shape[0]
        nblk = H.value("adh_cbam_pool_num_blocks", HW)
        partial = self._f(N, nblk, 2, Cc)
        partial_idx = torch.empty((N, nblk, Cc), device=self.device, dtype=torch.int32)
        pooled = self._f(N, 2, Cc)
        amax_idx = torch.empty((N, Cc), device=self.device, dtype=torch.int32)
        xbytes = 4.0 * N * HW * Cc
        H.call("adh_cbam_pool", x.t.data_ptr(), x.cs, N, HW, Cc, partial.data_ptr(), partial_idx.data_ptr(), nblk,
               pooled.data_ptr(), amax_idx.data_ptr(), work=xbytes)
        ca = self._f(N, Cc)
        hidden = self._f(N, 2, Ch)
        H.call("adh_cbam_mlp", pooled.data_ptr(), w1.data_ptr(), w2.data_ptr(), N, Cc, Ch, ca.data_ptr(),
               hidden.data_ptr())
        smap = self._f(N, HW, 2)
        cidx = torch.empty((N, HW), device=self.device, dtype=torch.int32)
        H.call("adh_cbam_spatial_stats", x.t.data_ptr(), x.cs, ca.data_ptr(), N, HW, Cc, smap.data_ptr(),
               cidx.data_ptr(), work=xbytes)
        sa = self._f(N, HW)
        if out is None:
            out = self._f(N, Hh, Ww, Cc)
        _check_dense_pixels(out)
        H.call("adh_cbam_apply", x.t.data_ptr(), x.cs, ca.data_ptr(), smap.data_ptr(), wsp.data_ptr(), N, Hh, Ww, Cc,
               sa.data_ptr(), out.data_ptr(), out.stride(2), work=2 * xbytes)
        o = Act(out, Cc)
        if CBAM_CAPTURE is not None:
            CBAM_CAPTURE[id(w1)] = (amax_idx, cidx)
        if self.record:
            self.use_param(wsp, w1, w2)

            def bwd():
                g = o.grad
                o.grad = None
                if g is None:
                    return
                gsa_pre = self._f(N, HW)
                H.call("adh_cbam_bwd_a", g.data_ptr(), g.stride(2), x.t.data_ptr(), x.cs, ca.data_ptr(), sa.data_ptr(),
                       N, HW, Cc, gsa_pre.data_ptr(), work=2 * xbytes)
                nb = H.value("adh_cbam_bwd_b_num_blocks", N, Hh, Ww)
                gsmap = self._f(N, HW, 2)
                dwsp_partial = self._f(nb, 98)
                dwsp = self.grad_buffer(wsp)
                H.call("adh_cbam_bwd_b", gsa_pre.data_ptr(), smap.data_ptr(), wsp.data_ptr(), N, Hh, Ww,
                       gsmap.data_ptr(), dwsp_partial.data_ptr(), nb, dwsp.data_ptr(), 0)
                gca_partial = self._f(N, nblk, Cc)
                H.call("adh_cbam_bwd_c", g.data_ptr(), g.stride(2), x.t.data_ptr(), x.cs, sa.data_ptr(),
                       gsmap.data_ptr(), cidx.data_ptr(), N, HW, Cc, gca_partial.data_ptr(), nblk, work=2 * xbytes)
                gpool = self._f(N, 2, Cc)
                dw1, dw2 = self.grad_buffer(w1), self.grad_buffer(w2)
                scratch = self._f(H.value("adh_cbam_bwd_d_scratch_floats", N, Cc, Ch))
                H.call("adh_cbam_bwd_d", gca_partial.data_ptr(), nblk, ca.data_ptr(), pooled.data_ptr(),
                       hidden.data_ptr(), w1.data_ptr(), w2.data_ptr(), N, Cc, Ch, gpool.data_ptr(), dw1.data_ptr(),
                       dw2.data_ptr(), 0, scratch.data_ptr())
                self.add_param_grad(wsp, dwsp)
                self.add_param_grad(w1, dw1)
                self.add_param_grad(w2, dw2)
                if x.needs_grad:
                    gx = self._f(N, Hh, Ww, Cc)
                    H.call("adh_cbam_bwd_e", g.data_ptr(), g.stride(2), x.t.data_ptr(), x.cs, ca.data_ptr(),
                           sa.data_ptr(), gsmap.data_ptr(), cidx.data_ptr(), gpool.data_ptr(), amax_idx.data_ptr(), N,
                           HW, Cc, gx.data_ptr(), gx.stride(2), work=2 * xbytes)   # reads g, writes gx (x itself is not read)
                    self.accum(x, gx)
            self.tape.append(bwd)
        return o

    # ------------------------------------------------------------------ zero-copy concat
    def concat_buffer(self, N, Hh, Ww, channels: Tuple[int, ...]):
        """Allocate [N,H,W,sum(C)] and return (buffer, [slice views]) so producers write in place
        (replaces torch.cat at medium_intensity.py:100,114 / high_intensity.py:117,129)."""
        buf = self._f(N, Hh, Ww, sum(channels))
        views, o = [], 0
        for c in channels:
            views.append(buf[..., o:o + c])
            o += c
        return buf, views

    def concat(self, buf: torch.Tensor, parts: List[Act]) -> Act:
        o = Act(buf)
        offs, off = [], 0
        for p in parts:
            assert p.t.data_ptr() == buf.data_ptr() + off * 4 and p.cs == buf.shape[3], "part is not a slice of buf"
            offs.append(off)
            off += p.C
        assert off == buf.shape[3]
        if self.record:
            def bwd():
                g = o.grad
                o.grad = None
                if g is None:
                    return
                for p, of in zip(parts, offs):
                    self.accum(p, g[..., of:of + p.C])
            self.tape.append(bwd)
        return o

    # ------------------------------------------------------------------ pooling / resize (alt models, odd sizes)
    def maxpool(self, x: Act, k: int, stride: Optional[int] = None, pad: int = 0) -> Act:
        stride = k if stride is None else stride
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        OH, OW = (Hh + 2 * pad - k) // stride + 1, (Ww + 2 * pad - k) // stride + 1
        out = self._f(N, OH, OW, Cc)
        idx = torch.empty((N, OH, OW, Cc), device=self.device, dtype=torch.int32) if self.record else None
        H.call("adh_maxpool", x.t.data_ptr(), x.cs, N, Hh, Ww, Cc, k, stride, pad, out.data_ptr(), Cc, H.ptr(idx))
        o = Act(out, Cc)
        if self.record:
            def bwd():
                g = o.grad
                o.grad = None
                if g is None or not x.needs_grad:
                    return
                gx = self._f(N, Hh, Ww, Cc)
                H.call("adh_maxpool_bwd", g.data_ptr(), g.stride(2), idx.data_ptr(), N, OH, OW, Cc, k, stride, pad, Hh, Ww,
                       gx.data_ptr(), Cc)
                self.accum(x, gx)
            self.tape.append(bwd)
        return o

    def bilinear(self, x: Act, OH: int, OW: int, align_corners: bool, out: Optional[torch.Tensor] = None) -> Act:
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        if out is None:
            out = self._f(N, OH, OW, Cc)
        H.call("adh_bilinear", x.t.data_ptr(), x.cs, N, Hh, Ww, Cc, OH, OW, int(align_corners), out.data_ptr(),
               out.stride(2))
        o = Act(out, Cc)
        if self.record:
            def bwd():
                g = o.grad
                o.grad = None
                if g is None or not x.needs_grad:
                    return
                gx = self._f(N, Hh, Ww, Cc, zero=True)
                H.call("adh_bilinear_bwd", g.data_ptr(), g.stride(2), N, Hh, Ww, Cc, OH, OW, int(align_corners),
                       gx.data_ptr(), Cc)
                self.accum(x, gx)
            self.tape.append(bwd)
        return o

    # ------------------------------------------------------------------ classifier helpers
    def global_avgpool(self, x: Act) -> Act:
        """AdaptiveAvgPool2d(1) -> [N,1,1,C] (torchvision resnet/densenet); reuses the CBAM pooling kernels."""
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        HW = Hh * Ww
        nblk = H.value("adh_cbam_pool_num_blocks", HW)
        means = self._f(N, Cc)
        for c0 in range(0, Cc, 1024):      # the pooling kernel takes <= 1024 channels per launch (resnet50: 2048): channel slices of x
            cw = min(1024, Cc - c0)
            partial = self._f(N, nblk, 2, cw)
            partial_idx = torch.empty((N, nblk, cw), device=self.device, dtype=torch.int32)
            pooled = self._f(N, 2, cw)
            amax_idx = torch.empty((N, cw), device=self.device, dtype=torch.int32)
            H.call("adh_cbam_pool", x.t.data_ptr() + 4 * c0, x.cs, N, HW, cw, partial.data_ptr(), partial_idx.data_ptr(), nblk,
                   pooled.data_ptr(), amax_idx.data_ptr())
            means[:, c0:c0 + cw] = pooled[:, 0, :]
        o = Act(means.view(N, 1, 1, Cc), Cc)   # [N,1,1,C] means
        if self.record:
            def bwd():
                g = o.grad
                o.grad = None
                if g is None or not x.needs_grad:
                    return
                gc = g.reshape(N, -1)[:, :Cc].contiguous()
                gx = self._f(N, Hh, Ww, Cc)
                H.call("adh_global_avgpool_bwd", gc.data_ptr(), N, HW, Cc, gx.data_ptr(), Cc)
                self.accum(x, gx)
            self.tape.append(bwd)
        return o

    def avgpool(self, x: Act, k: int) -> Act:
        assert not self.record, "avgpool backward is not implemented (DenseNet121 runs forward-only)"
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        out = self._f(N, Hh // k, Ww // k, Cc)
        H.call("adh_avgpool", x.t.data_ptr(), x.cs, N, Hh, Ww, Cc, k, out.data_ptr(), Cc)
        return Act(out, Cc)

    def bn_relu_eval(self, x: Act, bn: BNState, out: Optional[torch.Tensor] = None) -> Act:
        """Stand-alone eval-mode BatchNorm + ReLU (DenseNet's pre-activation norm layers)."""
        assert not self.record, "pre-activation BN backward is not implemented (DenseNet121 runs forward-only)"
        N, Hh, Ww, Cc = x.N, x.Hh, x.Ww, x.C
        scale, shift = self._f(Cc), self._f(Cc)
        H.call("adh_bn_fold_eval", Cc, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
               bn.running_var.data_ptr(), BN_EPS, None, scale.data_ptr(), shift.data_ptr())
        if out is None:
            out = self._f(N, Hh, Ww, Cc)
        H.call("adh_bn_apply", x.t.data_ptr(), x.cs, scale.data_ptr(), shift.data_ptr(), None, 0, H.ACT_RELU,
               out.data_ptr(), out.stride(2), x.pixels, Cc, None)
        return Act(out, Cc)

    def mul_mask(self, x: Act, mask: torch.Tensor) -> Act:
        """x * mask (dropout with a pre-drawn, pre-scaled mask of x's shape and strides)."""
        assert x.t.is_contiguous() and mask.shape == x.t.shape
        out = self._f(*x.t.shape)
        H.call("adh_mul", out.data_ptr(), x.t.data_ptr(), mask.data_ptr(), out.numel())
        o = Act(out, x.C)
        if self.record:
            def bwd():
                g = o.grad
                o.grad = None
                if g is None or not x.needs_grad:
                    return
                gx = self._f(*x.t.shape)
                H.call("adh_mul", gx.data_ptr(), g.contiguous().data_ptr(), mask.data_ptr(), gx.numel())
                self.accum(x, gx)
            self.tape.append(bwd)
        return o

    # ------------------------------------------------------------------ branch heads
    def head_blend(self, mode: int, x_img: torch.Tensor, r: Act, gd: Optional[Act], alpha: Optional[torch.Tensor]):
        """Final blend producing the NCHW output (low_intensity.py:41-45,116; medium_intensity.py:117;
        high_intensity.py:135-138,214)."""
        N, _, Hh, Ww = x_img.shape
        out = torch.empty_like(x_img)
        H.call("adh_head_blend", mode, x_img.data_ptr(), r.t.data_ptr(), r.cs, H.ptr(gd.t if gd else None),
               gd.cs if gd else 0, H.ptr(alpha), N, Hh, Ww, out.data_ptr())
        holder = {"g": None}
        if self.record:
            if mode == 0:
                self.use_param(alpha)

            def bwd():
                g = holder["g"]
                if g is None:
                    return
                nb = H.value("adh_head_blend_bwd_num_blocks", N, Hh, Ww)
                g_r = self._f(N, Hh, Ww, r.t.shape[3])
                g_gd = self._f(N, Hh, Ww, gd.t.shape[3]) if gd else None
                ga_partial = self._f(nb) if mode == 0 else None
                H.call("adh_head_blend_bwd", mode, g.data_ptr(), x_img.data_ptr(), r.t.data_ptr(), r.cs,
                       H.ptr(gd.t if gd else None), gd.cs if gd else 0, H.ptr(alpha), N, Hh, Ww, g_r.data_ptr(),
                       H.ptr(g_gd), H.ptr(ga_partial), nb)
                self.accum(r, g_r)
                if gd:
                    self.accum(gd, g_gd)
                if mode == 0:
                    ga = self.grad_buffer(alpha).reshape(1)
                    H.call("adh_sum_partials", ga_partial.data_ptr(), nb, 1.0, ga.data_ptr())
                    self.add_param_grad(alpha, ga.reshape(alpha.shape))
            self.tape.append(bwd)
        return out, holder
