"""models/routing.py of the reference -> adam-dehaze_amd (HIP kernels)."""
from adam_dehaze_amd.routing import HardRouter, SoftRouter, GatedRouter, create_router  # noqa: F401
