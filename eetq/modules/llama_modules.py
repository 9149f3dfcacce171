from eetq_amd.modules.llama_modules import *  # noqa: F401,F403
