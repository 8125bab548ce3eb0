#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference's torch-only modules on CPU.

Dev-only: runs in the build container where /root/reference exists; never on the GPU box and never
imported by the product.  Nothing from the reference is copied -- only inputs, seeded weights
(state_dicts produced by the reference's own constructors) and the outputs/gradients the reference
computes are stored.  models/classifier.py and training/loss.py cannot be imported here
(torchvision / timm / lpips are not installed), so fixtures cover: the building blocks, the six
branch classes, the three routers (with a stub classifier), and the duplicated-parameter Adam step.

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import hashlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    # the reference must shadow the repo's own `models` package
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != root]
    sys.path.insert(0, REF)
    from models.dehazing import base_model, low_intensity, medium_intensity, high_intensity  # noqa
    from models import routing  # noqa
    return base_model, low_intensity, medium_intensity, high_intensity, routing


def sd_np(module, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def randomize_bn(module, g):
    """Give BN layers non-trivial affine params and running stats so eval mode is not the identity."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))


def run_case(module, x, g, train_modes=(False, True)):
    """eval output; train output + updated BN buffers + grads of sum(out * gout)."""
    rec = {}
    rec.update(sd_np(module, "sd."))
    rec["x"] = x.numpy().copy()
    module.eval()
    with torch.no_grad():
        out_shape = module(x).shape
    gout = torch.randn(out_shape, generator=g)
    rec["gout"] = gout.numpy().copy()
    if False in train_modes:
        module.eval()
        with torch.no_grad():
            rec["out_eval"] = module(x).numpy().copy()
        # eval-mode gradients (BN as a fixed affine)
        xe = x.clone().requires_grad_(True)
        module.zero_grad(set_to_none=True)
        out = module(xe)
        (out * gout).sum().backward()
        rec["gx_eval"] = xe.grad.numpy().copy()
        for k, p in module.named_parameters():
            if p.grad is not None:
                rec["gp_eval." + k] = p.grad.numpy().copy()
    if True in train_modes:
        module.train()
        xt = x.clone().requires_grad_(True)
        module.zero_grad(set_to_none=True)
        out = module(xt)
        rec["out_train"] = out.detach().numpy().copy()
        (out * gout).sum().backward()
        rec["gx_train"] = xt.grad.numpy().copy()
        for k, p in module.named_parameters():
            if p.grad is not None:
                rec["gp_train." + k] = p.grad.numpy().copy()
        rec.update(sd_np(module, "sd_after_train."))
    return rec


def save(name, rec):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {len(rec)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


class StubClassifier(torch.nn.Module):
    """Tiny (logits, features) producer so routers can be exercised without torchvision."""

    def __init__(self, fd=512):
        super().__init__()
        self.proj = torch.nn.Linear(3, fd)
        self.head = torch.nn.Linear(fd, 3)

    def forward(self, x):
        f = self.proj(x.mean(dim=(2, 3)))
        return self.head(f), f


def main():
    base_model, low_intensity, medium_intensity, high_intensity, routing = _import_reference()
    torch.set_num_threads(4)

    # ---------------- (i) blocks -------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(1234)
    for C in (16, 32):
        x = torch.randn(2, C, 24, 40, generator=g)
        cases = {
            f"convblock_k3_c{C}": base_model.ConvBlock(C, C, 3, 1, 1),
            f"convblock_k4s2_c{C}": base_model.ConvBlock(C, 2 * C, 4, 2, 1),
            f"convblock_k1_c{C}": base_model.ConvBlock(C, C // 2, 1, 1, 0),
            f"convblock_nobn_noact_c{C}": base_model.ConvBlock(C, C, 3, 1, 1, use_bn=False, activation=None),
            f"resblock_c{C}": base_model.ResidualBlock(C),
            f"attention_c{C}": base_model.AttentionBlock(C),
        }
        for name, mod in cases.items():
            randomize_bn(mod, g)
            save(name, run_case(mod, x, g))
    x3 = torch.rand(2, 3, 24, 40, generator=g)
    stem = base_model.ConvBlock(3, 16, 7, 1, 3)
    randomize_bn(stem, g)
    save("convblock_k7_stem", run_case(stem, x3, g))
    stem3 = base_model.ConvBlock(3, 16, 3, 1, 1)
    randomize_bn(stem3, g)
    save("convblock_k3_stem", run_case(stem3, x3, g))

    # ---------------- (ii) branches at reduced width ------------------------------------------
    torch.manual_seed(42)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(2, 3, 32, 48, generator=g)
    x_odd = torch.rand(2, 3, 30, 46, generator=g)
    branches = {
        "light_b8": (low_intensity.LightweightDehazeModel(base_channels=8, n_blocks=3), [x]),
        "lowint_b8": (low_intensity.LowIntensityDehazeModel(base_channels=8, n_blocks=3), [x]),
        "medium_b8": (medium_intensity.MediumIntensityDehazeModel(base_channels=8), [x, x_odd]),
        "corun_b8": (medium_intensity.COrunInspiredModel(base_channels=8, n_blocks=2), [x]),
        "high_b16": (high_intensity.HighIntensityDehazeModel(base_channels=16), [x, x_odd]),
        "dual_b16": (high_intensity.DualBranchAttentionModel(base_channels=16), [x]),
    }
    for name, (mod, inputs) in branches.items():
        randomize_bn(mod, g)
        for i, xin in enumerate(inputs):
            rec = {"class_name": np.array(type(mod).__name__)}
            target = torch.rand(xin.shape, generator=g)
            rec["target"] = target.numpy().copy()
            rec.update(sd_np(mod, "sd."))
            rec["x"] = xin.numpy().copy()
            mod.eval()
            with torch.no_grad():
                rec["out_eval"] = mod(xin).numpy().copy()
            mod.train()
            mod.zero_grad(set_to_none=True)
            out = mod(xin)
            rec["out_train"] = out.detach().numpy().copy()
            loss = torch.nn.functional.l1_loss(out, target)
            rec["l1"] = loss.detach().numpy().copy()
            loss.backward()
            for k, p in mod.named_parameters():
                rec["gp_train." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            rec.update(sd_np(mod, "sd_after_train."))
            save(name + ("" if i == 0 else "_odd"), rec)
            # restore pre-train buffers so the odd-size case starts from the same state
            mod.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("sd.")})

    # ---------------- (iii) routers ------------------------------------------------------------
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(7)
    models = {
        "low": low_intensity.LightweightDehazeModel(base_channels=8, n_blocks=1),
        "medium": medium_intensity.MediumIntensityDehazeModel(base_channels=8),
        "high": high_intensity.HighIntensityDehazeModel(base_channels=16),
    }
    for m in models.values():
        randomize_bn(m, g)
    clf = StubClassifier()
    x = torch.rand(6, 3, 16, 24, generator=g)
    logits = torch.tensor([[2.0, 0.5, -1.0], [0.1, 0.3, 0.2], [-1.0, -2.0, 3.0],
                           [0.5, 0.5 + 1e-6, 0.4], [1.0, 1.0, 1.0], [0.0, 2.0, 2.0 - 1e-7]])
    rec = {"x": x.numpy().copy(), "logits": logits.numpy().copy()}
    for name, m in models.items():
        rec.update(sd_np(m, f"sd.models.{name}."))
    rec.update(sd_np(clf, "sd.classifier."))
    soft = routing.SoftRouter(models, classifier=clf, temperature=0.5, device="cpu").eval()
    with torch.no_grad():
        out, aux = soft(x, logits)
        rec["soft_out"] = out.numpy().copy()
        rec["soft_weights"] = aux["weights"].numpy().copy()
        for name, o in aux["individual_outputs"].items():
            rec["soft_ind." + name] = o.numpy().copy()
        out2, aux2 = soft(x)  # logits from the stub classifier
        rec["soft_out_clf"] = out2.numpy().copy()
        rec["soft_weights_clf"] = aux2["weights"].numpy().copy()
    # gradient wrt logits through the blend (eval-mode branches)
    lg = logits.clone().requires_grad_(True)
    out, _ = soft(x, lg)
    gout = torch.randn(out.shape, generator=g)
    (out * gout).sum().backward()
    rec["soft_gout"] = gout.numpy().copy()
    rec["soft_glogits"] = lg.grad.numpy().copy()
    hard = routing.HardRouter(models, classifier=clf, device="cpu").eval()
    with torch.no_grad():
        idx = torch.argmax(logits, dim=1)
        rec["hard_idx_from_logits"] = idx.numpy().copy()
        out, aux = hard(x, idx)
        rec["hard_out"] = out.numpy().copy()
        out_c, aux_c = hard(x)  # classifier path
        rec["hard_out_clf"] = out_c.numpy().copy()
        rec["hard_idx_clf"] = aux_c["intensity"].numpy().copy()
        clf_logits, clf_feats = clf(x)
        rec["clf_logits"] = clf_logits.numpy().copy()
        rec["clf_feats"] = clf_feats.numpy().copy()
    gated = routing.GatedRouter(models, classifier=clf, device="cpu").eval()
    rec.update({"sd.gate." + k: v.detach().numpy().copy() for k, v in gated.gate_network.state_dict().items()})
    with torch.no_grad():
        out, aux = gated(x)
        rec["gated_out"] = out.numpy().copy()
        rec["gated_weights"] = aux["gate_weights"].numpy().copy()
    save("routers", rec)

    # ---------------- (iv) Adam with duplicated params (train_joint.py:81-89) ------------------
    torch.manual_seed(3)
    w_dup = torch.nn.Parameter(torch.randn(5, 4))
    w_single = torch.nn.Parameter(torch.randn(7))
    opt = torch.optim.Adam([w_dup, w_single, w_dup], lr=5e-5, weight_decay=1e-4)
    rec = {"w_dup0": w_dup.detach().numpy().copy(), "w_single0": w_single.detach().numpy().copy()}
    for step in range(3):
        opt.zero_grad()
        gd = torch.randn(5, 4)
        gs = torch.randn(7)
        w_dup.grad = gd.clone()
        w_single.grad = gs.clone()
        rec[f"g_dup{step}"] = gd.numpy().copy()
        rec[f"g_single{step}"] = gs.numpy().copy()
        opt.step()
        rec[f"w_dup{step + 1}"] = w_dup.detach().numpy().copy()
        rec[f"w_single{step + 1}"] = w_single.detach().numpy().copy()
    save("adam_dup", rec)

    # ---------------- (v) full-width default-config summaries (config 1) -----------------------
    rec = {}
    for name, ctor in (("light", lambda: low_intensity.LightweightDehazeModel(base_channels=32, n_blocks=3)),
                       ("medium", lambda: medium_intensity.MediumIntensityDehazeModel(base_channels=64)),
                       ("high", lambda: high_intensity.HighIntensityDehazeModel(base_channels=96))):
        torch.manual_seed(42)
        mod = ctor().eval()
        g = torch.Generator().manual_seed(42)
        x = torch.rand(1, 3, 256, 256, generator=g)
        with torch.no_grad():
            out = mod(x)
        rec[name + ".n_params"] = np.array(sum(p.numel() for p in mod.parameters()))
        rec[name + ".param_sha256"] = np.array(hashlib.sha256(
            b"".join(p.detach().numpy().tobytes() for p in mod.parameters())).hexdigest())
        rec[name + ".out_mean"] = np.array(out.double().mean().item())
        rec[name + ".out_absmax"] = np.array(out.abs().max().item())
        rec[name + ".out_patch"] = out[0, :, 100:108, 100:108].numpy().copy()
        rec[name + ".state_keys"] = np.array(list(mod.state_dict().keys()))
    save("fullwidth_summaries", rec)


if __name__ == "__main__":
    main()
