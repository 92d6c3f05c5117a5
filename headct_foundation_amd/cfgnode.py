"""Minimal yacs-compatible CfgNode (yacs is not installed here and must not be assumed on the GPU box).

Implements the subset config.py uses (config.py:6-273 of the reference): attribute access, clone, defrost/freeze,
merge_from_file (yaml), merge_from_list (["A.B", value, ...] with literal decoding), dump.
"""
from __future__ import annotations

import copy
from ast import literal_eval

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    @staticmethod
    def _decode(v):
        """yacs decoding: strings that parse as Python literals become those literals ('None' -> None)."""
        if not isinstance(v, str):
            return v
        try:
            return literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _check_type(new, old, key):
        if old is None or new is None or type(new) is type(old):
            return new
        for a, b in ((list, tuple), (tuple, list), (int, float)):
            if isinstance(new, a) and isinstance(old, b):
                return b(new)
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")

    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise ValueError(f"config key {full} is not a node")
                self[k]._merge(v, full)
            else:
                v = self._decode(v)
                dict.__setitem__(self, k, self._check_type(v, self[k], full))

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2:
            raise ValueError(f"Override list has odd length: {cfg_list}; it must be a list of pairs")
        for full, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            keys = full.split(".")
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError(f"Non-existent key: {full}")
                node = node[k]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent key: {full}")
            v = self._decode(v)
            dict.__setitem__(node, keys[-1], self._check_type(v, node[keys[-1]], full))

    def _to_dict(self):
        return {k: (v._to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        return yaml.safe_dump(self._to_dict(), **kwargs)

    def __repr__(self):
        return f"CfgNode({dict.__repr__(self)})"
