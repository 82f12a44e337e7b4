"""Drop-in alias: ``promptttspp.*`` resolves to the MI355X implementation in
``promptttspp_amd.*`` (the SAME module objects, registered under both names), so
the reference's Hydra ``_target_`` paths (``promptttspp.models...``,
``promptttspp.vocoders.BigVGAN`` ...), ``egs/proposed/bin/train.py`` and ``app.py``
work against this repository unchanged."""
import importlib
import pkgutil
import sys

import promptttspp_amd as _impl

__version__ = _impl.__version__
__path__ = []  # every submodule is provided through the alias table below


def _alias_all():
    names = ["promptttspp_amd"]
    for m in pkgutil.walk_packages(_impl.__path__, prefix="promptttspp_amd."):
        if ".csrc" in m.name:
            continue
        names.append(m.name)
    for real in names:
        alias = "promptttspp" + real[len("promptttspp_amd"):]
        if real == "promptttspp_amd":
            continue
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        if parent == "promptttspp":
            globals()[leaf] = mod


_alias_all()
