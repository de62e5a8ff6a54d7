import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields

import torch


class BaseOutput(OrderedDict):
    """dataclass-style output whose fields are also attributes (diffusers.utils.BaseOutput)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]


def deprecate(*args, **kwargs):
    pass


def maybe_allow_in_graph(cls):
    return cls


def is_torch_version(op, version):
    from packaging import version as V
    cur = V.parse(torch.__version__.split("+")[0])
    ref = V.parse(version)
    return {">=": cur >= ref, ">": cur > ref, "<": cur < ref, "<=": cur <= ref, "==": cur == ref}[op]


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


logging = _Logging()
