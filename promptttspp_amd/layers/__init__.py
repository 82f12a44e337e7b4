from .activations import AntiAliasActivation, Snake  # noqa: F401
from .embedding import PhonemeEmbedding  # noqa: F401
from .norm import LayerNorm  # noqa: F401
