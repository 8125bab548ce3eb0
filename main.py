#!/usr/bin/env python3
"""Entry point with the reference's command line (/root/reference main.py:29-56): the modes that lie on
the hot path run on the HIP engine; the others (dataset preprocessing, classifier training, detection
evaluation) are outside this build's scope and say so.

    python main.py --mode train_joint --config config/config.yaml
    python main.py --mode train_joint --resume                     # continue from the latest checkpoint
    torchrun --nproc-per-node 8 main.py --mode train_joint        # data-parallel over RCCL

`--resume` is the reference's flag (main.py:50-51 parses it and never reads it); here it restores model, optimiser,
scheduler, epoch and best-PSNR from the latest checkpoint of the mode's checkpoint directory (or from the file given
as its optional value).  Image folders are not read by this build (data/dataset.py needs cv2): `--data_dir` is recorded
in the config like the reference does, and the drivers then say loudly that they train on synthetic frames.
"""
import argparse
import os
import random
import warnings

import numpy as np
import torch
import yaml


def parse_args():
    p = argparse.ArgumentParser(description="Adaptive fog intensity dehazing framework (MI355X build)")
    p.add_argument("--config", type=str, default="config/config.yaml")
    p.add_argument("--mode", type=str, default="train_all",
                   choices=["preprocess", "train_classifier", "train_dehazing", "train_joint", "train_all", "evaluate", "demo"])
    p.add_argument("--exp_name", type=str, default=None)
    p.add_argument("--data_dir", type=str, default=None)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--resume", nargs="?", const=True, default=None,
                   help="resume training from the latest checkpoint (or from the given checkpoint file)")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--epochs", type=int, default=None, help="override the configured number of epochs")
    return p.parse_args()


def seed_everything(seed):
    """utils/helpers.py:10-19."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def main():
    args = parse_args()
    with open(args.config) as f:
        config = yaml.safe_load(f)
    if args.data_dir:   # main.py:66-69
        config["dataset"]["train_path"] = os.path.join(args.data_dir, "train")
        config["dataset"]["val_path"] = os.path.join(args.data_dir, "val")
        config["dataset"]["test_path"] = os.path.join(args.data_dir, "test")
        warnings.warn(f"--data_dir {args.data_dir}: this build does not read image folders (cv2 dataset code is out of "
                      "scope); the training modes fall back to synthetic foggy frames and say so.")
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
        backend = os.environ.get("ADH_DIST_BACKEND", "nccl")   # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    seed_everything(config["seed"])
    from adam_dehaze_amd import train as T

    def resume_for(stage):
        """Bare `--resume`: every stage discovers the latest checkpoint in its own directory.  `--resume <file>`: the file
        goes to the ONE stage that wrote it (a branch checkpoint loaded into another branch, or into the joint stage, used
        to fail with a KeyError deep in the loader); the other stages of a multi-stage mode start fresh."""
        if args.resume is None or args.resume is True:
            return args.resume
        if resume_owner != stage:
            print(f"--resume {args.resume}: a '{resume_owner}' checkpoint; stage '{stage}' does not resume from it")
            return None
        return args.resume
    # `--resume <file>`: resolve the stage that wrote the file ONCE (checkpoint_stage reads the file and compares key sets), and
    # refuse a mode that runs no such stage -- it used to print a note per stage, train from scratch and overwrite best_model.pth
    # although the user asked to resume
    resume_owner = None
    if isinstance(args.resume, str):
        resume_owner = T.checkpoint_stage(args.resume, config)
        stages = {"train_joint": ("joint",), "train_dehazing": ("low", "medium", "high"),
                  "train_all": ("low", "medium", "high", "joint")}.get(args.mode, ())
        if stages and resume_owner not in stages:
            raise SystemExit(f"--resume {args.resume} is a '{resume_owner}' checkpoint, but --mode {args.mode} runs only the "
                             f"stage(s) {', '.join(stages)}: nothing would resume from it (and training from scratch would "
                             "overwrite best_model.pth).  Pick the matching --mode, or drop --resume.")
    if args.mode == "train_joint":
        T.train_joint_model(config, epochs=args.epochs, resume=resume_for("joint"))
    elif args.mode == "train_dehazing":
        for level in ("low", "medium", "high"):   # train_dehazing.py:216-232
            T.train_dehazing_model(config, level, epochs=args.epochs or 30, resume=resume_for(level))
    elif args.mode == "train_all":
        # main.py:121-139 of the reference: classifier -> dehazing branches -> joint -> evaluation.  Step 1 (stand-alone
        # classifier training) is outside this build's scope: the joint step fine-tunes the classifier
        # from whatever checkpoint `classifier.checkpoint_dir` holds, exactly as train_joint.py:18-27 does when it is missing
        print("\n===== Step 1: fog-intensity classifier training is outside this build's scope: skipped =====")
        print("\n===== Step 2: Training Dehazing Models =====")
        for level in ("low", "medium", "high"):
            T.train_dehazing_model(config, level, epochs=args.epochs or 30, resume=resume_for(level))
        print("\n===== Step 3: Training Joint Model =====")
        T.train_joint_model(config, epochs=args.epochs, resume=resume_for("joint"))
        print("\n===== Step 4: Comprehensive Evaluation =====")
        T.run_comprehensive_evaluation(config)
    elif args.mode == "evaluate":
        # evaluate.py:464-540 (the reference first rewrites the checkpoint paths to a hard-coded experiment directory, main.py:143-145:
        # here the paths of the config are used as they are)
        T.run_comprehensive_evaluation(config)   # baseline branches, routed system, detector on hazy vs dehazed frames
    elif args.mode == "demo":
        system = T.build_joint_system(config)
        system["router"].eval()
        batch = next(T.synthetic_loader(2, config["dataset"]["img_size"], 1, seed=config["seed"], device=config["device"]))
        with torch.no_grad():
            out, aux = system["router"](batch["hazy"])
        print("dehazed", tuple(out.shape), "weights", aux["weights"].cpu().tolist())
    else:
        raise SystemExit(f"mode '{args.mode}' is outside the hot path this build covers (see DESIGN.md section 7)")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
