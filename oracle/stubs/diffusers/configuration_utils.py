import functools
import inspect


class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        if not hasattr(self, "_internal_dict"):
            object.__setattr__(self, "_internal_dict", _Config())
        self._internal_dict.update(kwargs)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    """Captures every constructor argument (incl. defaults) into `self.config`."""

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.items())[1:]
        cfg = {name: p.default for name, p in params if p.default is not inspect.Parameter.empty}
        for (name, _), a in zip(params, args):
            cfg[name] = a
        cfg.update(kwargs)
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return inner
