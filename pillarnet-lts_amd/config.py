"""Python-file configs with attribute access, as the reference's `Config.fromfile`
(det3d/torchie/utils/config.py:12-162) provides on top of `addict` (not installed here, so the
attribute dict is our own).  Live configs load unchanged: they only need `itertools`/`logging`
and `__file__`."""
import importlib.util
import os.path as osp
import sys


class ConfigDict(dict):
    """dict with attribute access; nested dicts (also inside lists/tuples) convert on the way in.
    A missing attribute raises AttributeError, a missing key KeyError (config.py:12-29)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def copy(self):
        return ConfigDict(self)

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]


def get_downsample_factor(model_config):
    """det3d/utils/config_tool.py:39-53 (the module imports protobuf for its other, KITTI-era helpers):
    product of the neck's down-sampling strides, over its last up-sampling stride, times the backbone's."""
    try:
        neck_cfg = model_config["neck"]
    except (KeyError, TypeError):
        model_config = model_config["first_stage_cfg"]
        neck_cfg = model_config["neck"]
    factor = 1.0
    for s in neck_cfg.get("ds_layer_strides", [1]):
        factor *= s
    if len(neck_cfg.get("us_layer_strides", [])) > 0:
        factor /= neck_cfg.get("us_layer_strides", [])[-1]
    factor = int(factor * model_config["backbone"]["ds_factor"])
    assert factor > 0
    return factor


def _ensure_det3d_config_tool():
    """The legacy configs (configs/nusc/**, configs/waymo/**) start with
    `from det3d.utils.config_tool import get_downsample_factor`.  Where the reference package is not
    importable, that name resolves to the function above."""
    import types
    try:
        import det3d.utils.config_tool  # noqa: F401
        return
    except Exception:
        pass
    for name in ("det3d", "det3d.utils", "det3d.utils.config_tool"):
        if name not in sys.modules or not hasattr(sys.modules[name], "__path__") and name != "det3d.utils.config_tool":
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    tool = sys.modules["det3d.utils.config_tool"]
    tool.get_downsample_factor = get_downsample_factor
    sys.modules["det3d"].utils = sys.modules["det3d.utils"]
    sys.modules["det3d.utils"].config_tool = tool


class Config:
    @staticmethod
    def fromfile(filename):
        filename = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(filename):
            raise FileNotFoundError(f'file "{filename}" does not exist')
        if filename.endswith(".py"):
            name = osp.basename(filename)[:-3]
            if "." in name:
                raise ValueError("Dots are not allowed in config file path.")
            spec = importlib.util.spec_from_file_location(name, filename)
            mod = importlib.util.module_from_spec(spec)
            sys.path.insert(0, osp.dirname(filename))
            _ensure_det3d_config_tool()
            try:
                spec.loader.exec_module(mod)
            finally:
                sys.path.pop(0)
            cfg_dict = {k: v for k, v in vars(mod).items()
                        if not k.startswith("__") and not isinstance(v, type(sys))}
        elif filename.endswith((".yml", ".yaml")):
            import yaml
            with open(filename) as f:
                cfg_dict = yaml.safe_load(f)
        elif filename.endswith(".json"):
            import json
            with open(filename) as f:
                cfg_dict = json.load(f)
        else:
            raise IOError("Only py/yml/yaml/json type are supported now!")
        return Config(cfg_dict, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        elif not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        text = ""
        if filename:
            with open(filename) as f:
                text = f.read()
        object.__setattr__(self, "_text", text)

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    def __repr__(self):
        return f"Config (path: {self._filename}): {self._cfg_dict!r}"

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)
