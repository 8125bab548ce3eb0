"""training/loss.py of the reference -> adam-dehaze_amd (HIP kernels)."""
from adam_dehaze_amd.loss import (  # noqa: F401
    ContentLoss, DehazingLoss, JointLoss, get_dehazing_loss, get_joint_loss)
