"""models/dehazing/base_model.py of the reference -> adam-dehaze_amd (HIP engine)."""
from adam_dehaze_amd.layers import ConvBlock, ResidualBlock, AttentionBlock  # noqa: F401
from adam_dehaze_amd.dehazing import BaseDehazeModel, EncoderDecoder  # noqa: F401
