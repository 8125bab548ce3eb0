"""models/dehazing/medium_intensity.py of the reference -> adam-dehaze_amd (HIP engine)."""
from adam_dehaze_amd.dehazing import (  # noqa: F401
    MediumIntensityDehazeModel, COrunInspiredModel, create_medium_intensity_model)
