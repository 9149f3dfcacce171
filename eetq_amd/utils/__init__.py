from .quantizer import eet_quantize, find_layers, get_named_linears, set_op_by_name  # noqa: F401
from .replicas import ReplicaGroup  # noqa: F401,E402
