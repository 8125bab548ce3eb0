"""The dehazing branches ("CORUN-Light / Medium / Complex" and the three alternates) behind the
reference's class names, constructor arguments, forward signatures and state_dict keys
(/root/reference models/dehazing/{base_model,low_intensity,medium_intensity,high_intensity}.py).

forward(x): x float32 [N,3,H,W] on a cuda device -> float32 [N,3,H,W].  All arithmetic runs in the
HIP library through `engine.Engine`; there is no CPU path (a CPU tensor raises RuntimeError).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import _hip as H
from .engine import Act, Engine
from .layers import AttentionBlock, BNParams, ConvBlock, ConvParams, ResidualBlock, Seq, run_up

# head_blend modes (adam_dehaze_hip.h)
BLEND_LIGHT, BLEND_RESIDUAL, BLEND_GUIDED, BLEND_LOWINT, BLEND_DUAL = 0, 1, 2, 3, 4


class BranchFunction(torch.autograd.Function):
    """Outer autograd hook: one node per branch forward; the tape inside the Engine does the rest."""

    @staticmethod
    def forward(ctx, module, record, x, *params):
        eng = Engine(x.device, record)
        out, holder = module._run(eng, x)
        ctx.eng, ctx.holder, ctx.params = eng, holder, params
        return out

    @staticmethod
    def backward(ctx, g):
        eng = ctx.eng
        ctx.holder["g"] = g.contiguous()
        eng.backward()
        grads = []
        for p in ctx.params:
            gp = eng.param_grads.get(id(p))
            grads.append(gp.reshape(p.shape) if gp is not None else None)
        ctx.eng = None
        return (None, None, None, *grads)


class BaseDehazeModel(nn.Module):
    """Base class for all dehazing models (base_model.py:80-96)."""

    def forward(self, x):
        if type(self) is BaseDehazeModel:
            raise NotImplementedError
        H.require_cuda(x, "input image batch")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected [N,3,H,W] input, got {tuple(x.shape)}")
        x = x.contiguous()
        params = [p for p in self.parameters()]
        for p in params:
            H.require_cuda(p, "model parameter")
        record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return BranchFunction.apply(self, record, x, *params)

    def _run(self, eng: Engine, x: torch.Tensor):
        raise NotImplementedError

    def get_info(self):
        return {
            "model_type": self.__class__.__name__,
            "params": sum(p.numel() for p in self.parameters()),
            "trainable_params": sum(p.numel() for p in self.parameters() if p.requires_grad),
        }

    def _info(self):
        info = BaseDehazeModel.get_info(self)
        info.update({"model_type": type(self).__name__, "base_channels": self.base_channels, "n_blocks": self.n_blocks})
        return info


class EncoderDecoder(BaseDehazeModel):
    """The reference class of this name (base_model.py:98-230) is dead code that raises on its first
    forward; the name stays importable."""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        raise NotImplementedError("EncoderDecoder is unusable in the reference (channel bookkeeping does not add up)")


def _bare_conv(eng: Engine, p: ConvParams, x: Act, k: int, pad: int, alloc_C: int) -> Act:
    """nn.Conv2d with bias and no BN/activation (the head's last conv); output padded to alloc_C channels."""
    return eng.conv(x, p.weight, p.bias, None, kind="conv", k=k, stride=1, pad=pad, relu=False, training=False,
                    out_alloc_C=alloc_C)


# ------------------------------------------------------------------------------------------------
# low intensity
# ------------------------------------------------------------------------------------------------
class LightweightDehazeModel(BaseDehazeModel):
    """ "CORUN-Light": stem + n ResidualBlocks + 2-conv head, (1-a)*x + a*sigmoid(.)  (low_intensity.py:5-45)."""

    def __init__(self, in_channels=3, base_channels=32, n_blocks=3):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        self.init_conv = ConvBlock(in_channels, base_channels, 3, padding=1)
        self.residual_blocks = Seq([(i, ResidualBlock(base_channels)) for i in range(n_blocks)])
        self.output_conv = Seq([(0, ConvBlock(base_channels, base_channels, 3, padding=1)),
                                (1, ConvParams(base_channels, in_channels, 3, bias=True))])
        self.skip_alpha = nn.Parameter(torch.tensor(0.1))

    def _run(self, eng: Engine, x: torch.Tensor):
        tr = self.training
        f = self.init_conv.run(eng, eng.image_to_nhwc8(x), tr)
        for i in range(self.n_blocks):
            f = self.residual_blocks.at(i).run(eng, f, tr)
        f = self.output_conv.at(0).run(eng, f, tr)
        r = _bare_conv(eng, self.output_conv.at(1), f, 3, 1, 8)
        return eng.head_blend(BLEND_LIGHT, x, r, None, self.skip_alpha)

    def get_info(self):
        return self._info()


class LowIntensityDehazeModel(BaseDehazeModel):
    """One-level U-Net variant, clamp(x + (sigmoid(.)-0.5)*2)  (low_intensity.py:56-116)."""

    def __init__(self, in_channels=3, base_channels=32, n_blocks=3):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        c = base_channels
        self.init_conv = ConvBlock(in_channels, c, 3, padding=1)
        self.down1 = Seq([(0, ConvBlock(c, 2 * c, 4, stride=2, padding=1)), (1, ResidualBlock(2 * c))])
        self.bottleneck = Seq([(i, ResidualBlock(2 * c)) for i in range(n_blocks - 1)])
        self.up1 = Seq([(0, ConvParams(2 * c, c, 4, bias=True, transposed=True)), (1, BNParams(c))])
        self.output_conv = Seq([(0, ConvBlock(2 * c, c, 3, padding=1)), (1, ConvBlock(c, c, 3, padding=1)),
                                (2, ConvParams(c, in_channels, 3, bias=True))])

    def _run(self, eng: Engine, x: torch.Tensor):
        tr = self.training
        N, _, Hh, Ww = x.shape
        c = self.base_channels
        buf, (v_up, v_init) = eng.concat_buffer(N, Hh, Ww, (c, c))
        init = self.init_conv.run(eng, eng.image_to_nhwc8(x), tr, out=v_init)
        d = self.down1.at(0).run(eng, init, tr)
        d = self.down1.at(1).run(eng, d, tr)
        for i in range(self.n_blocks - 1):
            d = self.bottleneck.at(i).run(eng, d, tr)
        up = run_up(eng, self.up1.at(0), self.up1.at(1), d, tr, out=v_up)
        cat = eng.concat(buf, [up, init])
        o = self.output_conv.at(0).run(eng, cat, tr)
        o = self.output_conv.at(1).run(eng, o, tr)
        r = _bare_conv(eng, self.output_conv.at(2), o, 3, 1, 8)
        return eng.head_blend(BLEND_LOWINT, x, r, None, None)

    def get_info(self):
        return self._info()


def create_low_intensity_model(config):
    """low_intensity.py:127-140."""
    cfg = config["dehazing"]["low"]
    cls = LightweightDehazeModel if cfg["model_type"] == "lightweight" else LowIntensityDehazeModel
    return cls(base_channels=cfg["channels"], n_blocks=cfg["blocks"])


# ------------------------------------------------------------------------------------------------
# medium / high: two-level U-Net trunk (medium_intensity.py:5-117, high_intensity.py:6-138)
# ------------------------------------------------------------------------------------------------
class _UNetTrunk(BaseDehazeModel):
    attention = False

    def _build_trunk(self, in_channels, base_channels):
        c0, c1, c2 = base_channels, base_channels * 2, base_channels * 4
        att = self.attention
        self.init_conv = ConvBlock(in_channels, c0, 7, padding=3)

        def enc(cin, cout):
            items = [(0, ConvBlock(cin, cout, 4, stride=2, padding=1)), (1, ResidualBlock(cout)), (2, ResidualBlock(cout))]
            if att:
                items.append((3, AttentionBlock(cout)))
            return Seq(items)

        self.encoder = Seq([(0, enc(c0, c1)), (1, enc(c1, c2))])
        if att:
            self.bottleneck = Seq([(0, ResidualBlock(c2)), (1, AttentionBlock(c2)), (2, ResidualBlock(c2)),
                                   (3, AttentionBlock(c2))])
        else:
            self.bottleneck = Seq([(0, ResidualBlock(c2)), (1, ResidualBlock(c2))])

        def dec(cin, cout):
            items = [(0, ConvParams(cin, cout, 4, bias=True, transposed=True)), (1, BNParams(cout)), (3, ResidualBlock(cout))]
            if att:
                items.append((4, AttentionBlock(cout)))
            return Seq(items)

        self.decoder = Seq([(0, dec(c2, c1)), (1, dec(c1 * 2, c0))])
        self.output_conv = Seq([(0, ConvBlock(c0 * 2, c0, 3, padding=1)), (1, ConvBlock(c0, c0 // 2, 3, padding=1)),
                                (2, ConvParams(c0 // 2, in_channels, 3, bias=True))])

    def _trunk(self, eng: Engine, x: torch.Tensor, x8: Act) -> Act:
        """Returns the pre-tanh residual (NHWC, 8-channel padded)."""
        tr, att = self.training, self.attention
        N, _, Hh, Ww = x.shape
        c0, c1 = self.base_channels, self.base_channels * 2
        H1, W1 = (Hh + 2 - 4) // 2 + 1, (Ww + 2 - 4) // 2 + 1
        # skip concatenations are zero-copy: producers write straight into their channel slice
        buf2, (v_dec1, v_f0) = eng.concat_buffer(N, Hh, Ww, (c0, c0))
        buf1, (v_dec0, v_f1) = eng.concat_buffer(N, H1, W1, (c1, c1))

        f0 = self.init_conv.run(eng, x8, tr, out=v_f0)
        feats: List[Act] = [f0]
        for e in range(2):
            stage = self.encoder.at(e)
            last_out = v_f1 if e == 0 else None
            h = stage.at(0).run(eng, feats[-1], tr)
            h = stage.at(1).run(eng, h, tr)
            h = stage.at(2).run(eng, h, tr, out=None if att else last_out)
            if att:
                h = stage.at(3).run(eng, h, out=last_out)
            feats.append(h)
        b = feats[-1]
        if att:
            b = self.bottleneck.at(0).run(eng, b, tr)
            b = self.bottleneck.at(1).run(eng, b)
            b = self.bottleneck.at(2).run(eng, b, tr)
            b = self.bottleneck.at(3).run(eng, b)
        else:
            b = self.bottleneck.at(0).run(eng, b, tr)
            b = self.bottleneck.at(1).run(eng, b, tr)

        h = b
        for dlev, (buf, v_dst, skip) in enumerate(((buf1, v_dec0, feats[1]), (buf2, v_dec1, feats[0]))):
            stage = self.decoder.at(dlev)
            up_h, up_w = h.Hh * 2, h.Ww * 2
            same = (up_h == skip.Hh and up_w == skip.Ww)
            h = run_up(eng, stage.at(0), stage.at(1), h, tr)
            if att:
                h = stage.at(3).run(eng, h, tr)
                h = stage.at(4).run(eng, h, out=v_dst if same else None)
            else:
                h = stage.at(3).run(eng, h, tr, out=v_dst if same else None)
            if not same:   # F.interpolate(..., mode='bilinear', align_corners=False) (medium_intensity.py:93-99)
                h = eng.bilinear(h, skip.Hh, skip.Ww, False, out=v_dst)
            h = eng.concat(buf, [h, skip])
        o = self.output_conv.at(0).run(eng, h, tr)
        o = self.output_conv.at(1).run(eng, o, tr)
        return _bare_conv(eng, self.output_conv.at(2), o, 3, 1, 8)

    def get_info(self):
        return self._info()


class MediumIntensityDehazeModel(_UNetTrunk):
    """ "CORUN-Medium": clamp(x + tanh(trunk(x)), 0, 1)   (medium_intensity.py:5-117)."""
    attention = False

    def __init__(self, in_channels=3, base_channels=64, n_blocks=6):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        self._build_trunk(in_channels, base_channels)

    def _run(self, eng: Engine, x: torch.Tensor):
        r = self._trunk(eng, x, eng.image_to_nhwc8(x))
        return eng.head_blend(BLEND_RESIDUAL, x, r, None, None)


class HighIntensityDehazeModel(_UNetTrunk):
    """ "CORUN-Complex": attention U-Net + guidance map, clamp(x + tanh(trunk)*sigmoid(detail), 0, 1)
    (high_intensity.py:6-138)."""
    attention = True

    def __init__(self, in_channels=3, base_channels=96, n_blocks=9):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        self._build_trunk(in_channels, base_channels)
        self.detail_branch = Seq([(0, ConvBlock(in_channels, 16, 3, padding=1)), (1, ConvBlock(16, 16, 3, padding=1)),
                                  (2, ConvParams(16, 1, 1, bias=True))])

    def _run(self, eng: Engine, x: torch.Tensor):
        tr = self.training
        x8 = eng.image_to_nhwc8(x)
        g = self.detail_branch.at(0).run(eng, x8, tr)
        g = self.detail_branch.at(1).run(eng, g, tr)
        gd = _bare_conv(eng, self.detail_branch.at(2), g, 1, 0, 8)
        r = self._trunk(eng, x, x8)
        return eng.head_blend(BLEND_GUIDED, x, r, gd, None)


class COrunInspiredModel(BaseDehazeModel):
    """Multi-scale pooling + fusion + ResidualBlocks (medium_intensity.py:128-190)."""

    def __init__(self, in_channels=3, base_channels=64, n_blocks=6):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        c = base_channels
        self.init_conv = ConvBlock(in_channels, c, 7, padding=3)
        self.scale1_conv = ConvBlock(c, c, 3, padding=1)
        self.scale2_conv = Seq([(1, ConvBlock(c, 2 * c, 3, padding=1))])
        self.scale3_conv = Seq([(1, ConvBlock(c, 4 * c, 3, padding=1))])
        self.fusion_conv = ConvBlock(7 * c, 2 * c, 1, padding=0)
        self.residual_blocks = Seq([(i, ResidualBlock(2 * c)) for i in range(n_blocks)])
        self.output_conv = Seq([(0, ConvBlock(2 * c, c, 3, padding=1)), (1, ConvParams(c, in_channels, 3, bias=True))])

    def _run(self, eng: Engine, x: torch.Tensor):
        tr = self.training
        N, _, Hh, Ww = x.shape
        c = self.base_channels
        init = self.init_conv.run(eng, eng.image_to_nhwc8(x), tr)
        buf, (v1, v2, v3) = eng.concat_buffer(N, Hh, Ww, (c, 2 * c, 4 * c))
        s1 = self.scale1_conv.run(eng, init, tr, out=v1)
        s2 = self.scale2_conv.at(1).run(eng, eng.maxpool(init, 2), tr)
        s2 = eng.bilinear(s2, s2.Hh * 2, s2.Ww * 2, True, out=v2)      # nn.UpsamplingBilinear2d: align_corners=True
        s3 = self.scale3_conv.at(1).run(eng, eng.maxpool(init, 4), tr)
        s3 = eng.bilinear(s3, s3.Hh * 4, s3.Ww * 4, True, out=v3)
        f = self.fusion_conv.run(eng, eng.concat(buf, [s1, s2, s3]), tr)
        for i in range(self.n_blocks):
            f = self.residual_blocks.at(i).run(eng, f, tr)
        o = self.output_conv.at(0).run(eng, f, tr)
        r = _bare_conv(eng, self.output_conv.at(1), o, 3, 1, 8)
        return eng.head_blend(BLEND_RESIDUAL, x, r, None, None)

    def get_info(self):
        return self._info()


def create_medium_intensity_model(config):
    """medium_intensity.py:201-215."""
    cfg = config["dehazing"]["medium"]
    cls = COrunInspiredModel if cfg["model_type"] == "corun" else MediumIntensityDehazeModel
    return cls(base_channels=cfg["channels"], n_blocks=cfg["blocks"])


class DualBranchAttentionModel(BaseDehazeModel):
    """Global + local branches with a transmission map, clamp(x + (1-t)*residual)  (high_intensity.py:149-214)."""

    def __init__(self, in_channels=3, base_channels=96, n_blocks=9):
        super().__init__()
        self.in_channels, self.base_channels, self.n_blocks = in_channels, base_channels, n_blocks
        c = base_channels
        self.global_branch = Seq([
            (0, ConvBlock(in_channels, c, 7, padding=3)), (2, ResidualBlock(c)), (3, AttentionBlock(c)),
            (5, ResidualBlock(c)), (6, AttentionBlock(c)), (7, ResidualBlock(c)), (9, ResidualBlock(c)),
            (11, ConvBlock(c, c // 2, 3, padding=1))])
        self.local_branch = Seq([(0, ConvBlock(in_channels, c // 2, 3, padding=1)), (1, ResidualBlock(c // 2)),
                                 (2, ResidualBlock(c // 2)), (3, ConvBlock(c // 2, c // 2, 3, padding=1))])
        self.transmission_branch = Seq([(0, ConvBlock(c, c // 2, 3, padding=1)), (1, ConvBlock(c // 2, c // 4, 3, padding=1)),
                                        (2, ConvParams(c // 4, 1, 1, bias=True))])
        self.fusion_conv = Seq([(0, ConvBlock(c, c // 2, 3, padding=1)), (1, ConvParams(c // 2, in_channels, 3, bias=True))])

    def _run(self, eng: Engine, x: torch.Tensor):
        tr = self.training
        N, _, Hh, Ww = x.shape
        c = self.base_channels
        x8 = eng.image_to_nhwc8(x)
        gb = self.global_branch
        buf, (v_g, v_l) = eng.concat_buffer(N, Hh, Ww, (c // 2, c // 2))
        g = gb.at(0).run(eng, x8, tr)
        g = eng.maxpool(g, 2)
        g = gb.at(2).run(eng, g, tr)
        g = gb.at(3).run(eng, g)
        g = eng.maxpool(g, 2)
        g = gb.at(5).run(eng, g, tr)
        g = gb.at(6).run(eng, g)
        g = gb.at(7).run(eng, g, tr)
        g = eng.bilinear(g, g.Hh * 2, g.Ww * 2, True)
        g = gb.at(9).run(eng, g, tr)
        g = eng.bilinear(g, g.Hh * 2, g.Ww * 2, True)
        g = gb.at(11).run(eng, g, tr, out=v_g)
        l = self.local_branch.at(0).run(eng, x8, tr)
        l = self.local_branch.at(1).run(eng, l, tr)
        l = self.local_branch.at(2).run(eng, l, tr)
        l = self.local_branch.at(3).run(eng, l, tr, out=v_l)
        cat = eng.concat(buf, [g, l])
        t = self.transmission_branch.at(0).run(eng, cat, tr)
        t = self.transmission_branch.at(1).run(eng, t, tr)
        t = _bare_conv(eng, self.transmission_branch.at(2), t, 1, 0, 8)
        r = self.fusion_conv.at(0).run(eng, cat, tr)
        r = _bare_conv(eng, self.fusion_conv.at(1), r, 3, 1, 8)
        return eng.head_blend(BLEND_DUAL, x, r, t, None)

    def get_info(self):
        return self._info()


def create_high_intensity_model(config):
    """high_intensity.py:225-239."""
    cfg = config["dehazing"]["high"]
    cls = DualBranchAttentionModel if cfg["model_type"] == "dual_branch" else HighIntensityDehazeModel
    return cls(base_channels=cfg["channels"], n_blocks=cfg["blocks"])
