"""The training steps of the hot path (/root/reference training/train_joint.py:29-166,
training/train_dehazing.py:16-106) on the HIP engine, plus the data-parallel wrapper.

`build_joint_system(config)` reproduces the reference's construction order (classifier, low, medium,
high -> router -> optimiser -> scheduler -> criterion, train_joint.py:36-98) including the duplicated
branch parameters in the optimiser's parameter list.  `joint_train_step` is the body of the hot loop
(train_joint.py:129-166); `dehazing_train_step` the one of train_dehazing.py:71-106.
Dataset I/O (cv2 image folders) is out of scope: `synthetic_loader` produces foggy frames with the
reference's own fog model when no loader is supplied.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Optional

import torch

from .classifier import create_classifier
from .dehazing import create_high_intensity_model, create_low_intensity_model, create_medium_intensity_model
from .loss import get_dehazing_loss, get_joint_loss
from .optim import Adam
from .parallel import GradientSynchronizer
from .routing import create_router


class ReduceLROnPlateau:
    """optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience) host logic (train_joint.py:92-94)."""

    def __init__(self, optimizer, mode="min", factor=0.5, patience=3, threshold=1e-4, verbose=False):
        assert mode == "min"
        self.opt, self.factor, self.patience, self.threshold = optimizer, factor, patience, threshold
        self.best, self.bad = float("inf"), 0

    def step(self, metric: float):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            for gr in self.opt.param_groups:
                gr["lr"] *= self.factor
            self.bad = 0


def load_pretrained_model(model, checkpoint_path):
    """train_joint.py:18-27: missing checkpoints are not errors."""
    if os.path.exists(checkpoint_path):
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
        model.load_state_dict(checkpoint["model_state_dict"])
        print(f"Loaded pretrained weights from {checkpoint_path}")
        return True
    print(f"Checkpoint {checkpoint_path} not found. Starting with random weights.")
    return False


def build_joint_system(config, world_size: int = 1) -> Dict:
    device = torch.device(config["device"])
    classifier = create_classifier(config)
    low = create_low_intensity_model(config)
    medium = create_medium_intensity_model(config)
    high = create_high_intensity_model(config)
    load_pretrained_model(classifier, os.path.join(config["classifier"]["checkpoint_dir"], "best_model.pth"))
    for name, m in (("low", low), ("medium", medium), ("high", high)):
        load_pretrained_model(m, os.path.join(config["dehazing"]["checkpoint_dir"], name, "best_model.pth"))
    models = {"low": low, "medium": medium, "high": high}
    router = create_router(models, classifier, config).to(device)
    # router.parameters() already holds classifier + branches; the reference appends the branches AGAIN
    params = list(router.parameters())
    for m in models.values():
        params.extend(list(m.parameters()))
    optimizer = Adam(params, lr=config["joint_training"]["learning_rate"], weight_decay=0.0001)
    scheduler = ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=3)
    criterion = get_joint_loss(config).to(device)
    sync = GradientSynchronizer(list(router.parameters()), world_size) if world_size > 1 else None
    return {"classifier": classifier, "models": models, "router": router, "optimizer": optimizer,
            "scheduler": scheduler, "criterion": criterion, "sync": sync, "device": device}


def joint_train_step(system: Dict, batch: Dict) -> Dict:
    """One iteration of train_joint.py:129-166.  Losses stay on the device (no .item() sync)."""
    dev = system["device"]
    hazy, clear, labels = batch["hazy"].to(dev), batch["clear"].to(dev), batch["intensity"].to(dev)
    system["optimizer"].zero_grad()
    logits, _ = system["classifier"](hazy)
    dehazed, _ = system["router"](hazy, logits)
    loss, comps = system["criterion"](dehazed, clear, logits, labels)
    loss.backward()
    if system["sync"] is not None:
        system["sync"].all_reduce()
    system["optimizer"].step()
    return {"loss": loss.detach(), "dehazing": comps["dehazing"].detach(),
            "classification": comps["classification"].detach()}


def dehazing_train_step(model, criterion, optimizer, batch: Dict, level: Optional[int], device, sync=None):
    """One iteration of train_dehazing.py:71-106: keep only the images of this branch's fog level."""
    hazy, clear, labels = batch["hazy"], batch["clear"], batch["intensity"]
    if level is not None:
        keep = labels == level
        if int(keep.sum()) == 0:
            return None
        hazy, clear = hazy[keep], clear[keep]
    hazy, clear = hazy.to(device), clear.to(device)
    optimizer.zero_grad()
    out = model(hazy)
    loss, comps = criterion(out, clear)
    loss.backward()
    if sync is not None:
        sync.all_reduce()
    optimizer.step()
    return {"loss": loss.detach(), "l1": comps["l1"].detach()}


def synthetic_loader(batch_size: int, size, steps: int, seed: int = 42, rank: int = 0) -> Iterator[Dict]:
    """Foggy/clear pairs from the reference's fog model I = J*t + A*(1-t) (utils/helpers.py:241-258)."""
    import torch.nn.functional as F
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(seed + 1000 * rank)
    xs = torch.linspace(0, 1, w).view(1, w)
    ys = torch.linspace(0, 1, h).view(h, 1)
    depth = 0.3 + 0.7 * torch.sqrt((xs - 0.5) ** 2 + (ys - 0.2) ** 2)
    ranges = {0: ((0.1, 0.4), (0.5, 0.7)), 1: ((0.4, 0.7), (0.7, 0.9)), 2: ((0.7, 1.0), (0.8, 1.0))}
    for _ in range(steps):
        clear = torch.rand(batch_size, 3, h, w, generator=g)
        clear = F.avg_pool2d(F.pad(clear, (2, 2, 2, 2), mode="reflect"), 5, 1)
        labels = torch.randint(0, 3, (batch_size,), generator=g)
        u = torch.rand(batch_size, 2, generator=g)
        hazy = torch.empty_like(clear)
        for i in range(batch_size):
            (b0, b1), (a0, a1) = ranges[int(labels[i])]
            t = torch.exp(-(b0 + (b1 - b0) * float(u[i, 0])) * depth)
            hazy[i] = (clear[i] * t + (a0 + (a1 - a0) * float(u[i, 1])) * (1 - t)).clamp(0, 1)
        yield {"hazy": hazy, "clear": clear, "intensity": labels, "name": [f"synthetic_{i}" for i in range(batch_size)]}


def train_joint_model(config, train_loader=None, steps_per_epoch: int = 4, epochs: Optional[int] = None):
    """train_joint.py:29 entry point (training part); returns the system and the per-epoch mean losses."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    system = build_joint_system(config, world)
    system["classifier"].train()
    for m in system["models"].values():
        m.train()
    system["router"].train()
    epochs = config["joint_training"]["epochs"] if epochs is None else epochs
    history = []
    for epoch in range(epochs):
        loader = train_loader if train_loader is not None else synthetic_loader(
            config["dataset"]["batch_size"], config["dataset"]["img_size"], steps_per_epoch, seed=config["seed"] + epoch,
            rank=rank)
        total, n = None, 0
        for batch in loader:
            stats = joint_train_step(system, batch)
            total = stats["loss"] if total is None else total + stats["loss"]
            n += 1
        mean = float(total) / max(1, n)     # one host read-back per epoch
        history.append(mean)
        system["scheduler"].step(mean)
        if rank == 0:
            print(f"Epoch {epoch + 1}/{epochs} train_loss={mean:.4f}")
    if rank == 0:
        ck = config["joint_training"]["checkpoint_dir"]
        os.makedirs(ck, exist_ok=True)
        torch.save({"epoch": epochs, "router_state_dict": system["router"].state_dict(),
                    "low_model_state_dict": system["models"]["low"].state_dict(),
                    "medium_model_state_dict": system["models"]["medium"].state_dict(),
                    "high_model_state_dict": system["models"]["high"].state_dict(),
                    "classifier_state_dict": system["classifier"].state_dict()}, os.path.join(ck, "last_model.pth"))
    return system, history


def train_dehazing_model(config, intensity_level: str, train_loader=None, steps: int = 4):
    """train_dehazing.py:16 entry point (training part) for one branch: Adam(lr, wd 1e-4) + DehazingLoss."""
    device = torch.device(config["device"])
    factory = {"low": create_low_intensity_model, "medium": create_medium_intensity_model,
               "high": create_high_intensity_model}[intensity_level]
    model = factory(config).to(device).train()
    criterion = get_dehazing_loss(config).to(device)
    optimizer = Adam(model.parameters(), lr=config["dehazing"][intensity_level]["learning_rate"], weight_decay=1e-4)
    level = {"low": 0, "medium": 1, "high": 2}[intensity_level]
    loader = train_loader if train_loader is not None else synthetic_loader(
        config["dataset"]["batch_size"], config["dataset"]["img_size"], steps, seed=config["seed"])
    losses = []
    for batch in loader:
        st = dehazing_train_step(model, criterion, optimizer, batch, level, device)
        if st is not None:
            losses.append(st["loss"])
    return model, [float(x) for x in losses]
