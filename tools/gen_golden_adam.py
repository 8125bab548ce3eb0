#!/usr/bin/env python3
"""Golden vectors for the duplicated-parameter Adam step in torch's *foreach* form (tests/golden/adam_dup_foreach.npz).

/root/reference training/train_joint.py:81-89 lists every branch parameter twice.  tools/gen_golden.py pinned the
single-tensor loop (adam_dup.npz: torch's CPU default and every torch < 2.0); on CUDA torch >= 2.0 picks
`_multi_tensor_adam`, where the duplicate entries alias inside the _foreach_ ops and the result differs by ~8 % of an
update.  This script runs torch.optim.Adam(foreach=True) on CPU tensors -- same code path, same arithmetic --
on the scenario of adam_dup.npz (same seed, same gradients) and stores the trajectories.  torch only; no reference import.
"""
import os
import warnings

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    warnings.simplefilter("ignore")
    torch.manual_seed(3)
    w_dup = torch.nn.Parameter(torch.randn(5, 4))
    w_single = torch.nn.Parameter(torch.randn(7))
    g3 = torch.Generator().manual_seed(33)               # separate stream: w_dup / w_single see adam_dup.npz's numbers
    w_tri = torch.nn.Parameter(torch.randn(33, generator=g3))   # listed three times: the general `repeats` form
    opt = torch.optim.Adam([w_dup, w_single, w_dup, w_tri, w_tri, w_tri], lr=5e-5, weight_decay=1e-4, foreach=True)
    rec = {"w_dup0": w_dup.detach().numpy().copy(), "w_single0": w_single.detach().numpy().copy(),
           "w_tri0": w_tri.detach().numpy().copy(), "torch_version": np.array(torch.__version__)}
    for step in range(3):
        opt.zero_grad()
        gd, gs, gt = torch.randn(5, 4), torch.randn(7), torch.randn(33, generator=g3)
        w_dup.grad, w_single.grad, w_tri.grad = gd.clone(), gs.clone(), gt.clone()
        rec[f"g_dup{step}"], rec[f"g_single{step}"], rec[f"g_tri{step}"] = gd.numpy().copy(), gs.numpy().copy(), gt.numpy().copy()
        opt.step()
        rec[f"w_dup{step + 1}"] = w_dup.detach().numpy().copy()
        rec[f"w_single{step + 1}"] = w_single.detach().numpy().copy()
        rec[f"w_tri{step + 1}"] = w_tri.detach().numpy().copy()
    sd = opt.state_dict()
    rec["state_params"] = np.array(sd["param_groups"][0]["params"])     # how torch indexes duplicates in state_dict()
    rec["state_steps"] = np.array([float(sd["state"][k]["step"]) for k in sorted(sd["state"])])
    np.savez_compressed(os.path.join(OUT, "adam_dup_foreach.npz"), **rec)
    print("wrote adam_dup_foreach.npz", {k: v.shape for k, v in rec.items() if k.startswith("w_") and k.endswith("3")},
          rec["state_params"], rec["state_steps"])


if __name__ == "__main__":
    main()
