"""Import shim: the product package lives in the directory `adam-dehaze_amd/` (a name Python cannot
import directly because of the hyphen); this package re-points its search path there so that
`import adam_dehaze_amd.dehazing` etc. resolve to `adam-dehaze_amd/*.py`."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "adam-dehaze_amd")
if not _os.path.isdir(_pkg_dir):
    raise ImportError(f"product directory {_pkg_dir} is missing")
__path__ = [_pkg_dir]

from . import _hip  # noqa: E402,F401
from .dehazing import (  # noqa: E402,F401
    BaseDehazeModel, EncoderDecoder, LightweightDehazeModel, LowIntensityDehazeModel,
    MediumIntensityDehazeModel, COrunInspiredModel, HighIntensityDehazeModel, DualBranchAttentionModel,
    create_low_intensity_model, create_medium_intensity_model, create_high_intensity_model,
)
from .layers import ConvBlock, ResidualBlock, AttentionBlock  # noqa: E402,F401
