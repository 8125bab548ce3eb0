"""models/dehazing/high_intensity.py of the reference -> adam-dehaze_amd (HIP engine)."""
from adam_dehaze_amd.dehazing import (  # noqa: F401
    HighIntensityDehazeModel, DualBranchAttentionModel, create_high_intensity_model)
