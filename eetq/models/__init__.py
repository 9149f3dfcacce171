"""``eetq.models`` of the reference (python/eetq/models/__init__.py): the export classes live in eetq_amd/models.py."""
from eetq_amd.models import AutoEETQForCausalLM, EETQConfig, EETQForCausalLM  # noqa: F401
from eetq_amd.models import EETQForCausalLM as BaseEETQForCausalLM  # noqa: F401
