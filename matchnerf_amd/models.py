"""Model registry with the reference's name (/root/reference/models/__init__.py:3-5):
``models_dict[opts.model](opts)`` is how coach.py:77 builds the network."""
from .matchnerf import MatchNeRF

models_dict = {
    "matchnerf": MatchNeRF,
}
