"""Drop-in for the reference's Python package ``eetq`` (python/eetq/__init__.py): re-exports the hot-path
surface from eetq_amd.  The offline export layer (AutoEETQForCausalLM) is out of scope."""
from eetq_amd.modules.qlinear import *  # noqa: F401,F403
from eetq_amd.utils.quantizer import *  # noqa: F401,F403
from eetq_amd.utils.accelerator import eet_accelerator  # noqa: F401
