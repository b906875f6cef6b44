"""Loader for the reference's Hydra-style YAML configs (conf/conf.yaml + conf/log/*.yaml) without
Hydra/OmegaConf (neither is a dependency): ``defaults`` list, ``${a.b}`` interpolation, ``${now:fmt}``
and ``key=value`` command-line overrides (README.md:30-34).  Values are read with ``.get()`` exactly
as train.py:206-251 does."""
from __future__ import annotations

import datetime
import os
import re
from typing import Any, Iterable, Optional

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


class _Loader(yaml.SafeLoader):
    """SafeLoader whose floats follow YAML 1.2 / OmegaConf: ``1e-2`` is a float, not a string."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+]?[0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _yaml(text: str):
    return yaml.load(text, Loader=_Loader)


class Config(dict):
    """dict with attribute access and lazy ``${...}`` interpolation against the root config."""

    def __init__(self, data=None, root: Optional["Config"] = None):
        super().__init__()
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            dict.__setitem__(self, k, self._wrap(v))

    def _wrap(self, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v, self._root)
        if isinstance(v, list):
            return [self._wrap(x) for x in v]
        return v

    def _resolve(self, v):
        if isinstance(v, str) and "${" in v:
            def sub(m):
                key = m.group(1).strip()
                if key.startswith("now:"):
                    return datetime.datetime.now().strftime(key[4:])
                cur: Any = self._root
                for part in key.split("."):
                    cur = cur[part]
                return str(cur)
            whole = _INTERP.fullmatch(v)
            if whole and not whole.group(1).startswith("now:"):
                cur: Any = self._root
                for part in whole.group(1).strip().split("."):
                    cur = cur[part]
                return cur
            return _INTERP.sub(sub, v)
        return v

    def __getitem__(self, k):
        return self._resolve(dict.__getitem__(self, k))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def set_path(self, dotted: str, value) -> None:
        cur = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in cur or not isinstance(dict.__getitem__(cur, p), dict):
                dict.__setitem__(cur, p, Config({}, self._root))
            cur = dict.__getitem__(cur, p)
        dict.__setitem__(cur, parts[-1], cur._wrap(value))

    def to_dict(self):
        def conv(v):
            if isinstance(v, Config):
                return {k: conv(v[k]) for k in v}
            if isinstance(v, list):
                return [conv(x) for x in v]
            return v
        return conv(self)


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(config_path: str = "conf", config_name: str = "conf", overrides: Iterable[str] = ()) -> Config:
    path = os.path.join(config_path, config_name + ("" if config_name.endswith((".yaml", ".yml")) else ".yaml"))
    with open(path, encoding="utf-8") as f:
        raw = _yaml(f.read()) or {}
    merged: dict = {}
    for item in raw.pop("defaults", None) or []:
        items = item.items() if isinstance(item, dict) else [(None, item)]
        for group, name in items:
            sub = os.path.join(config_path, group or "", str(name) + ".yaml")
            if not os.path.exists(sub):
                continue
            text = open(sub, encoding="utf-8").read()
            data = _yaml(text) or {}
            first = text.lstrip().splitlines()[0] if text.strip() else ""
            if "@package _global_" in first or group is None:
                _merge(merged, data)
            else:
                _merge(merged.setdefault(group, {}), data)
    _merge(merged, raw)
    cfg = Config(merged)
    for ov in overrides:
        if "=" not in ov:
            raise ValueError("override must be key=value, got %r" % ov)
        k, v = ov.split("=", 1)
        cfg.set_path(k.lstrip("+"), _yaml(v))
    return cfg
