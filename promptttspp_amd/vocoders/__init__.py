from .bigvgan import BigVGAN  # noqa: F401
from .bigvgan_f0 import F0AwareBigVGAN  # noqa: F401
