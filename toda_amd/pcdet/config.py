"""YAML experiment configs as attribute dictionaries (reference pcdet/config.py:16-85):
`_BASE_CONFIG_` includes, `--set KEY VALUE` overrides with type coercion, a global `cfg`."""
import ast
from pathlib import Path

import yaml


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def update(self, other=None, **kw):
        for k, v in {**(other or {}), **kw}.items():
            self[k] = v

    def __deepcopy__(self, memo):
        import copy

        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


EasyDict = AttrDict  # the reference's name for the same thing


def log_config_to_file(cfg, pre="cfg", logger=None):
    emit = logger.info if logger is not None else print
    for key, val in cfg.items():
        if isinstance(val, AttrDict):
            emit(f"\n{pre}.{key} = edict()")
            log_config_to_file(val, pre=f"{pre}.{key}", logger=logger)
        else:
            emit(f"{pre}.{key}: {val}")


def _coerce(text, like):
    """Parse a CLI string with the type of the value it replaces."""
    try:
        val = ast.literal_eval(text)
    except (ValueError, SyntaxError):
        val = text
    if like is None or isinstance(like, AttrDict):
        return val
    if isinstance(like, bool):
        return val if isinstance(val, bool) else str(text).lower() in ("1", "true", "yes")
    if isinstance(like, (list, tuple)):
        if isinstance(val, (list, tuple)):
            return type(like)(val)
        parts = str(text).split(",")
        kind = type(like[0]) if len(like) else str
        return type(like)(kind(p) for p in parts)
    if isinstance(val, type(like)) or (isinstance(like, float) and isinstance(val, int)):
        return type(like)(val)
    raise TypeError(f"cannot set a {type(like).__name__} config entry from {text!r}")


def cfg_from_list(cfg_list, config):
    """`--set A.B.C value ...` (reference config.py:16-48)."""
    assert len(cfg_list) % 2 == 0, "--set expects KEY VALUE pairs"
    for key, text in zip(cfg_list[0::2], cfg_list[1::2]):
        node = config
        *parents, leaf = key.split(".")
        for p in parents:
            assert p in node, f"unknown config key: {key}"
            node = node[p]
        assert leaf in node, f"unknown config key: {key}"
        node[leaf] = _coerce(text, node[leaf])


def merge_new_config(config, new_config, search_dirs=()):
    """Recursive merge honouring `_BASE_CONFIG_` (reference config.py:51-68)."""
    if "_BASE_CONFIG_" in new_config:
        base = Path(new_config["_BASE_CONFIG_"])
        for d in (Path.cwd(), *search_dirs):
            if (Path(d) / base).exists():
                base = Path(d) / base
                break
        with open(base) as f:
            merge_new_config(config, yaml.safe_load(f), search_dirs)   # a base may name its own base
    for key, val in new_config.items():
        if key == "_BASE_CONFIG_":
            continue
        if isinstance(val, dict):
            if key not in config or not isinstance(config[key], AttrDict):
                config[key] = AttrDict()
            merge_new_config(config[key], val, search_dirs)
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file, config):
    cfg_file = Path(cfg_file)
    with open(cfg_file) as f:
        new_config = yaml.safe_load(f)
    # `_BASE_CONFIG_: cfgs/...` is written relative to tools/ in the reference's YAMLs
    search = [cfg_file.parent, *cfg_file.parents]
    merge_new_config(config, new_config, search)
    return config


cfg = AttrDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / "../../").resolve()
cfg.LOCAL_RANK = 0
