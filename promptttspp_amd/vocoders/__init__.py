from .bigvgan import BigVGAN  # noqa: F401
