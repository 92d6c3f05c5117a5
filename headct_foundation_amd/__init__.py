"""headct_foundation_amd: MI355X-native (gfx950) MAE pre-training hot path behind the reference's interface."""
from .mae import MaskedAutoencoderViT, build_sincos_position_embedding  # noqa: F401
from ._lib import HctError  # noqa: F401
from .pos_embed import interpolate_pos_embed  # noqa: F401
from .vit import ViT  # noqa: F401
from .classifier import AttentionClassifier, LinearClassifier, cross_entropy  # noqa: F401
