"""training/train_joint.py of the reference -> adam-dehaze_amd.train (HIP engine)."""
from adam_dehaze_amd.train import load_pretrained_model, train_joint_model, joint_train_step, build_joint_system  # noqa: F401
