"""models/dehazing/low_intensity.py of the reference -> adam-dehaze_amd (HIP engine)."""
from adam_dehaze_amd.dehazing import (  # noqa: F401
    LightweightDehazeModel, LowIntensityDehazeModel, create_low_intensity_model)
