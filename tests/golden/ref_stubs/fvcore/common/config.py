import ast
import copy
import os
import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    """Minimal yacs/fvcore-compatible CfgNode (attribute access, _BASE_ inheritance, merge)."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__["_frozen"] = False
        self.__dict__["_new_allowed"] = new_allowed
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get("_frozen"):
            raise AttributeError("frozen")
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def is_frozen(self):
        return self.__dict__["_frozen"]

    def is_new_allowed(self):
        return self.__dict__["_new_allowed"]

    def dump(self, **kw):
        def conv(n):
            return {k: conv(v) for k, v in n.items()} if isinstance(n, dict) else n
        return yaml.safe_dump(conv(self), **kw)

    @staticmethod
    def load_yaml_with_base(filename, allow_unsafe=False):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and k in b and isinstance(b[k], dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not base.startswith("/"):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode.load_yaml_with_base(base)
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, cfg_filename, allow_unsafe=False):
        loaded = CfgNode.load_yaml_with_base(cfg_filename)
        self.merge_from_other_cfg(CfgNode(loaded))

    def merge_from_other_cfg(self, other):
        _merge(other, self, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            parts = k.split(".")
            for s in parts[:-1]:
                d = d[s]
            d[parts[-1]] = _coerce(_decode(v), d.get(parts[-1]))


def _decode(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode(v)
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old):
    if old is None:
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    return new


def _merge(a, b, path):
    for k, v in a.items():
        v = _decode(v)
        if k not in b:
            b[k] = v  # permissive (reference uses new_allowed for some nodes)
            continue
        if isinstance(v, CfgNode) and isinstance(b[k], CfgNode):
            _merge(v, b[k], path + [k])
        else:
            b[k] = _coerce(v, b[k])
