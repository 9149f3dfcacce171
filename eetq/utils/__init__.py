from eetq_amd.utils.quantizer import *  # noqa: F401,F403
