from .activations import AntiAliasActivation, Snake  # noqa: F401
