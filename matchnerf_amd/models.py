"""Model registry of the hot path.

The reference builds its network as ``models_dict[opts.model](opts)`` (coach.py:77, registry in
models/__init__.py); the same lookup works here, and ``build_model`` adds the usual
construct-and-move step for callers that do not go through a Coach."""
from .matchnerf import MatchNeRF


def _registry():
    return {"matchnerf": MatchNeRF}


models_dict = _registry()


def build_model(opts):
    """``models_dict[opts.model](opts)`` placed on ``opts.device``."""
    try:
        cls = models_dict[opts.model]
    except KeyError as e:
        raise KeyError(f"unknown model {opts.model!r}; available: {sorted(models_dict)}") from e
    return cls(opts).to(opts.device)
