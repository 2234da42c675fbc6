"""Model registry: the reference builds its network as ``models_dict[opts.model](opts)``
(coach.py:77, models/__init__.py)."""
from .matchnerf import MatchNeRF

models_dict = {"matchnerf": MatchNeRF}
