from eetq_amd.modules.qlinear import *  # noqa: F401,F403
from eetq_amd.modules.qlinear import quantize_and_preprocess_weights  # noqa: F401
