from eetq_amd.utils.quantizer import *  # noqa: F401,F403
from eetq_amd.utils.accelerator import *  # noqa: F401,F403
