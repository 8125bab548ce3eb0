"""training/train_dehazing.py of the reference -> adam-dehaze_amd.train (HIP engine)."""
from adam_dehaze_amd.train import train_dehazing_model, dehazing_train_step  # noqa: F401
