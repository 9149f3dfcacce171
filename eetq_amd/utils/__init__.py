from .quantizer import eet_quantize, find_layers, get_named_linears, set_op_by_name  # noqa: F401
from .replicas import ReplicaGroup  # noqa: F401,E402
from .fuse import FusedW8A16Linear, fuse_w8a16_linears  # noqa: F401,E402
