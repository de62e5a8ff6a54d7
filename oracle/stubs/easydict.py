"""ORACLE TEST INFRASTRUCTURE — stand-in for the `easydict` package (attribute-access dict)."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in {**(d or {}), **kwargs}.items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
