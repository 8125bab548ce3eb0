"""Input side of the hot path on device: the reference's synthetic fog model as one HIP kernel and a synthetic
foggy-frame loader that never leaves the GPU.

/root/reference utils/helpers.py:201-265 (`apply_random_fog`) converts every image to numpy, draws (beta, A) with
`np.random.uniform`, builds the depth map and the transmission in float64 and loops over the channels on the host.
Here the draws stay on the host in the reference's order (so `np.random.seed(s)` reproduces its parameters), and
`I = clip(J*t + A*(1-t))` runs in `adh_apply_fog` for the whole batch.  Dataset files (cv2 image folders,
data/dataset.py) are out of scope; `synthetic_loader` stands in for the DataLoader with the same batch dict keys.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Sequence, Union

import numpy as np
import torch

from . import _hip as H

# helpers.py:222-234
FOG_RANGES = {"low": ((0.1, 0.4), (0.5, 0.7)), "medium": ((0.4, 0.7), (0.7, 0.9)), "high": ((0.7, 1.0), (0.8, 1.0)),
              "random": ((0.1, 1.0), (0.5, 1.0))}
LEVEL_NAMES = ("low", "medium", "high")


def apply_fog(clear: torch.Tensor, beta: torch.Tensor, airlight: torch.Tensor) -> torch.Tensor:
    """hazy[n] = clip(clear[n]*t_n + A_n*(1 - t_n), 0, 1), t_n = exp(-beta_n * depth) (helpers.py:241-258);
    clear [N,3,H,W] float32 in [0,1] on the GPU, beta / airlight [N]."""
    H.require_cuda(clear, "clear image batch")
    if clear.dim() != 4 or clear.shape[1] != 3:
        raise RuntimeError(f"expected [N,3,H,W], got {tuple(clear.shape)}")
    clear = clear.contiguous()
    N, _, Hh, Ww = clear.shape
    beta = beta.to(device=clear.device, dtype=torch.float32).contiguous()
    airlight = airlight.to(device=clear.device, dtype=torch.float32).contiguous()
    if beta.numel() != N or airlight.numel() != N:
        raise RuntimeError("beta / airlight must have one entry per image")
    hazy = torch.empty_like(clear)
    H.call("adh_apply_fog", clear.data_ptr(), beta.data_ptr(), airlight.data_ptr(), N, Hh, Ww, hazy.data_ptr())
    return hazy


def draw_fog_params(intensities: Sequence[str], rng=None):
    """(beta, A) per image, drawn like helpers.py:237-238: np.random.uniform(*beta_range) then
    np.random.uniform(*A_range), image after image.  `rng`: anything with .uniform (default: the global np.random)."""
    rng = np.random if rng is None else rng
    betas, As = [], []
    for name in intensities:
        (b0, b1), (a0, a1) = FOG_RANGES.get(name, FOG_RANGES["random"])
        betas.append(rng.uniform(b0, b1))
        As.append(rng.uniform(a0, a1))
    return torch.tensor(betas, dtype=torch.float64), torch.tensor(As, dtype=torch.float64)


def apply_random_fog(clear_img: torch.Tensor, intensity: Union[str, Sequence[str]] = "random", rng=None) -> torch.Tensor:
    """helpers.py:201-265 for GPU tensors: [N,3,H,W] or [3,H,W] in [0,1] (values > 1 are taken as 0..255 and scaled,
    as the reference does); `intensity` one name for all images or one per image."""
    single = clear_img.dim() == 3
    x = clear_img.unsqueeze(0) if single else clear_img
    if float(x.max()) > 1.0:
        x = x / 255.0
    names = [intensity] * x.shape[0] if isinstance(intensity, str) else list(intensity)
    beta, A = draw_fog_params(names, rng)
    out = apply_fog(x.float(), beta, A)
    return out[0] if single else out


def draw_augment_params(n: int, rng=None) -> torch.Tensor:
    """[n, 5] float32 {flip_h, flip_v, brightness_first, b, c}: the random choices of the reference's training transform
    (data/dataset.py:59-64: RandomHorizontalFlip, RandomVerticalFlip, ColorJitter(brightness=0.1, contrast=0.1)) for n
    samples, drawn the way data/dataset.py:100-116 does: `seed = np.random.randint(2147483647)` per sample, then
    `torch.manual_seed(seed)` before the transform of EACH of the sample's images -- so hazy / clear / dehazed get the same
    choices, which is why one parameter row per sample suffices.  Draw order inside the transform (torchvision):
    `torch.rand(1) < 0.5` (horizontal), `torch.rand(1) < 0.5` (vertical), `torch.randperm(4)` (order of brightness /
    contrast / saturation / hue; the last two are off), `uniform(0.9, 1.1)` for brightness, then for contrast."""
    rng = np.random if rng is None else rng
    rows = []
    for _ in range(n):
        g = torch.Generator().manual_seed(int(rng.randint(2147483647)))
        fh = float(torch.rand(1, generator=g) < 0.5)
        fv = float(torch.rand(1, generator=g) < 0.5)
        perm = torch.randperm(4, generator=g).tolist()
        b = float(torch.empty(1).uniform_(0.9, 1.1, generator=g))
        c = float(torch.empty(1).uniform_(0.9, 1.1, generator=g))
        rows.append([fh, fv, float(perm.index(0) < perm.index(1)), b, c])
    return torch.tensor(rows, dtype=torch.float32)


def paired_augment(images: Sequence[torch.Tensor], params: Optional[torch.Tensor] = None, rng=None):
    """Apply ONE set of per-sample choices (`draw_augment_params`) to every tensor of `images` ([N,3,H,W] float32 in
    [0,1] on the GPU: hazy, clear, dehazed ...), on device.  Returns (list of augmented tensors, params)."""
    x0 = images[0]
    H.require_cuda(x0, "image batch")
    N, Cc, Hh, Ww = x0.shape
    if Cc != 3:
        raise RuntimeError(f"expected [N,3,H,W], got {tuple(x0.shape)}")
    if params is None:
        params = draw_augment_params(N, rng)
    pd = params.to(device=x0.device, dtype=torch.float32).contiguous()
    nblk = H.value("adh_augment_num_blocks", Hh * Ww)
    outs = []
    for x in images:
        if x.shape != x0.shape:
            raise RuntimeError("paired images must have one shape")
        H.require_cuda(x, "image batch")
        x = x.contiguous()
        partial = torch.empty(N * nblk, device=x.device, dtype=torch.float64)
        out = torch.empty_like(x)
        H.call("adh_paired_augment", x.data_ptr(), pd.data_ptr(), N, Hh, Ww, partial.data_ptr(), nblk, out.data_ptr())
        outs.append(out)
    return outs, params


def synthetic_loader(batch_size: int, size, steps: int, seed: int = 42, rank: int = 0,
                     device: Optional[torch.device] = None, augment: bool = False) -> Iterator[Dict]:
    """Foggy / clear pairs with the reference's fog model, generated on `device` (default cuda): low-pass random clear
    frames, labels uniform in {0,1,2}, (beta, A) from the label's range.  Batch dict keys as data/dataset.py:118-124
    ('dehazed' omitted: nothing on the path reads it)."""
    import torch.nn.functional as F
    h, w = (size, size) if isinstance(size, int) else size
    device = torch.device("cuda") if device is None else torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed + 1000 * rank)
    host = np.random.RandomState(seed + 1000 * rank)
    for _ in range(steps):
        clear = torch.rand(batch_size, 3, h, w, generator=g, device=device)
        clear = F.avg_pool2d(F.pad(clear, (2, 2, 2, 2), mode="reflect"), 5, 1)
        labels = torch.from_numpy(host.randint(0, 3, size=batch_size).astype(np.int64))
        beta, A = draw_fog_params([LEVEL_NAMES[int(i)] for i in labels], host)
        hazy = apply_fog(clear, beta, A)
        if augment:     # the reference's train-split transform, same choices for both images of a pair
            (hazy, clear), _ = paired_augment([hazy, clear], rng=host)
        yield {"hazy": hazy, "clear": clear, "intensity": labels.to(device),
               "name": [f"synthetic_{i}" for i in range(batch_size)]}
