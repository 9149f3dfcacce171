"""Drop-in for the reference's Python package ``eetq`` (python/eetq/__init__.py): re-exports the hot-path
surface from eetq_amd, and ``AutoEETQForCausalLM`` (python/eetq/__init__.py:1-2 -> models/auto.py) lazily -- the reference
imports transformers / accelerate / huggingface_hub eagerly there (models/base.py:5-22); here ``import eetq`` stays light and
the export classes are resolved on first use."""
from eetq_amd.modules.qlinear import *  # noqa: F401,F403
from eetq_amd.utils.quantizer import *  # noqa: F401,F403
from eetq_amd.utils.accelerator import eet_accelerator  # noqa: F401


def __getattr__(name):
    if name in ("AutoEETQForCausalLM", "EETQForCausalLM", "EETQConfig"):
        from eetq_amd import models
        return getattr(models, name)
    raise AttributeError("module 'eetq' has no attribute %r" % name)
