#!/usr/bin/env python3
"""Entry point with the reference's command line (/root/reference main.py:29-56): the modes that lie on
the hot path run on the HIP engine; the others (dataset preprocessing, classifier training, detection
evaluation) are outside this build's scope and say so.

    python main.py --mode train_joint --config config/config.yaml
    torchrun --nproc-per-node 8 main.py --mode train_joint        # data-parallel over RCCL
"""
import argparse
import os
import random

import numpy as np
import torch
import yaml


def parse_args():
    p = argparse.ArgumentParser(description="Adaptive fog intensity dehazing framework (MI355X build)")
    p.add_argument("--config", type=str, default="config/config.yaml")
    p.add_argument("--mode", type=str, required=True,
                   choices=["preprocess", "train_classifier", "train_dehazing", "train_joint", "train_all", "evaluate", "demo"])
    p.add_argument("--data_dir", type=str, default=None)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--exp_name", type=str, default="default")
    p.add_argument("--epochs", type=int, default=None, help="override the configured number of epochs")
    return p.parse_args()


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def main():
    args = parse_args()
    with open(args.config) as f:
        config = yaml.safe_load(f)
    if args.data_dir:
        for k in ("train_path", "val_path", "test_path"):
            config["dataset"][k] = args.data_dir
    if args.device:
        config["device"] = args.device
    if args.seed:
        config["seed"] = args.seed
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        config["device"] = f"cuda:{local}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    seed_everything(config["seed"])
    from adam_dehaze_amd import train as T
    if args.mode == "train_joint":
        T.train_joint_model(config, epochs=args.epochs)
    elif args.mode == "train_dehazing":
        for level in ("low", "medium", "high"):
            T.train_dehazing_model(config, level)
    elif args.mode == "demo":
        from adam_dehaze_amd.routing import create_router
        system = T.build_joint_system(config)
        system["router"].eval()
        batch = next(T.synthetic_loader(2, config["dataset"]["img_size"], 1, seed=config["seed"]))
        with torch.no_grad():
            out, aux = system["router"](batch["hazy"].to(config["device"]))
        print("dehazed", tuple(out.shape), "weights", aux["weights"].cpu().tolist())
    else:
        raise SystemExit(f"mode '{args.mode}' is outside the hot path this build covers (see DESIGN.md section 7)")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
