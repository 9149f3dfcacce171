from eetq_amd.modules.qlinear import *  # noqa: F401,F403
