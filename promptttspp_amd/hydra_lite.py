"""Minimal Hydra/OmegaConf-compatible config loader.

The reference's entry points are ``@hydra.main`` scripts whose plug-in boundary is
``hydra.utils.instantiate(cfg.<node>)`` on ``_target_`` dotted paths
(egs/proposed/bin/train.py:20-23, app.py:136-146).  hydra-core / omegaconf are not
installed on the build or GPU boxes, so this module implements the subset the
reference's config tree uses:

* ``defaults`` lists with ``_self_`` and ``group: name`` entries (config groups are
  sub-directories), later entries overriding earlier ones;
* interpolation ``${a.b}`` (absolute) and ``${..a.b}`` / ``${...a.b}`` (relative);
* command-line overrides ``group=name``, ``a.b=value``, ``+a.b=value``;
* ``instantiate(node, **overrides)``: recursive ``_target_`` construction.

When real hydra is importable the entry points use it instead (see
egs/proposed/bin/train.py); this loader is the offline fallback, and what the
tests use to instantiate the model from the YAML tree.
"""
import copy
import importlib
import os

import yaml


class Cfg(dict):
    """dict with attribute access (the part of DictConfig the trainer uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _parse_value(s):
    try:
        return yaml.safe_load(s)
    except yaml.YAMLError:
        return s


def _set_path(cfg, dotted, value):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = value


def _lookup(root, path_keys):
    node = root
    for k in path_keys:
        node = node[int(k)] if isinstance(node, list) else node[k]
    return node


def _resolve(root):
    """Resolve ${...} interpolations in place (string-valued leaves only).  A referenced value that
    itself holds interpolations is resolved relative to ITS OWN location first."""

    def target(expr, here):
        # here: key path of the node CONTAINING the leaf; returns the absolute key path of the target
        ndots = len(expr) - len(expr.lstrip("."))
        rest = expr.lstrip(".")
        if ndots == 0:
            base = []
        else:
            base = here[: len(here) - (ndots - 1)] if ndots - 1 <= len(here) else []
        return base + (rest.split(".") if rest else [])

    def fetch(expr, here, depth):
        if depth > 32:
            raise ValueError(f"interpolation cycle at ${{{expr}}}")
        tp = target(expr, here)
        val = _lookup(root, tp)
        if isinstance(val, str) and "${" in val:
            val = resolve_str(val, tp[:-1], depth + 1)
        elif isinstance(val, (dict, list)):
            walk(val, tp)
        return val

    def resolve_str(v, here, depth=0):
        s = v.strip()
        if s.startswith("${") and s.endswith("}") and s.count("${") == 1:
            return copy.deepcopy(fetch(s[2:-1], here, depth))
        out = v
        while "${" in out:
            a = out.index("${")
            b = out.index("}", a)
            out = out[:a] + str(fetch(out[a + 2 : b], here, depth)) + out[b + 1 :]
        return out

    def walk(node, path):
        items = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(items):
            if isinstance(v, (dict, list)):
                walk(v, path + [k])
            elif isinstance(v, str) and "${" in v:
                node[k] = resolve_str(v, path)

    walk(root, [])
    return root


def compose(config_dir, config_name, overrides=()):
    """Load ``config_dir/config_name.yaml`` with its defaults list and overrides."""
    primary = _load_yaml(os.path.join(config_dir, config_name + ".yaml"))
    defaults = primary.pop("defaults", [])
    group_choice, dotted = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if os.path.isdir(os.path.join(config_dir, key)) and "." not in key:
            group_choice[key] = val
        else:
            dotted.append((key, _parse_value(val)))
    cfg = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(cfg, primary)
            self_done = True
        elif isinstance(d, dict):
            for group, name in d.items():
                if group == "override hydra/job_logging" or group.startswith("override "):
                    continue
                name = group_choice.pop(group, name)
                if name is None:
                    continue
                path = os.path.join(config_dir, group, str(name) + ".yaml")
                if not os.path.exists(path):
                    raise FileNotFoundError(f"config group '{group}' has no option '{name}' ({path})")
                _merge(cfg, {group: _load_yaml(path)})
    for group, name in group_choice.items():  # groups selected only on the command line
        _merge(cfg, {group: _load_yaml(os.path.join(config_dir, group, str(name) + ".yaml"))})
    if not self_done:
        _merge(cfg, primary)
    cfg.pop("hydra", None)
    for k, v in dotted:
        _set_path(cfg, k, v)
    return _wrap(_resolve(cfg))


def load_node(path):
    """Load a single YAML file (e.g. conf/model/xyz.yaml) and resolve interpolations."""
    return _wrap(_resolve(_load_yaml(path)))


def get_class(dotted):
    mod, _, name = dotted.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, *args, **kwargs):
    """hydra.utils.instantiate: build ``node['_target_'](**children)`` recursively."""
    if node is None:
        return None
    if isinstance(node, list):
        return [instantiate(v) if isinstance(v, (dict, list)) else v for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return _wrap({k: instantiate(v) if isinstance(v, dict) and "_target_" in v else v for k, v in node.items()})
    params = {}
    for k, v in node.items():
        if k in ("_target_", "_recursive_", "_convert_", "_partial_"):
            continue
        params[k] = instantiate(v) if isinstance(v, (dict, list)) else v
    params.update(kwargs)
    return get_class(node["_target_"])(*args, **params)


def to_yaml(cfg):
    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, list):
            return [plain(v) for v in o]
        return o

    return yaml.safe_dump(plain(cfg), sort_keys=False)
