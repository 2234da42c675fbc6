"""Minimal recursive attribute-dict.

The reference reads its options and batches through ``easydict.EasyDict``
(/root/reference/options.py:7, models/matchnerf.py:2).  That package is not
part of this image, so the host side ships a ~40 line equivalent with the
same observable behaviour on the hot path: attribute access == item access,
nested dicts are wrapped on assignment, ``update``/``pop`` work.
If the real ``easydict`` is importable it is used instead so that objects
created by a caller's own code are the same type.
"""
try:  # pragma: no cover - not installed in the build image
    from easydict import EasyDict  # type: ignore
except Exception:  # noqa: BLE001

    class EasyDict(dict):
        def __init__(self, d=None, **kwargs):
            super().__init__()
            if d is None:
                d = {}
            if kwargs:
                d = dict(d, **kwargs)
            for k, v in d.items():
                setattr(self, k, v)

        @classmethod
        def _wrap(cls, value):
            if isinstance(value, dict) and not isinstance(value, EasyDict):
                return cls(value)
            if isinstance(value, (list, tuple)):
                return type(value)(cls._wrap(x) for x in value)
            return value

        def __setattr__(self, name, value):
            value = self._wrap(value)
            super().__setattr__(name, value)
            super().__setitem__(name, value)

        __setitem__ = __setattr__

        def __getattr__(self, name):  # only called when normal lookup fails
            try:
                return self[name]
            except KeyError as e:
                raise AttributeError(name) from e

        def update(self, e=None, **f):
            d = dict(e or {})
            d.update(f)
            for k, v in d.items():
                setattr(self, k, v)

        def pop(self, k, *args):
            if hasattr(self, k) and k in self.__dict__:
                delattr(self, k)
            return super().pop(k, *args)


def to_plain_dict(d):
    """Recursively turn an EasyDict (or dict) into plain dicts (cf. misc/utils.py:147-152)."""
    out = dict(d)
    for k, v in out.items():
        if isinstance(v, dict):
            out[k] = to_plain_dict(v)
    return out
