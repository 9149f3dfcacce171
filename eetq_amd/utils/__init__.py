from .quantizer import eet_quantize, find_layers, get_named_linears, set_op_by_name  # noqa: F401
from .replicas import ReplicaGroup  # noqa: F401,E402
from .fuse import FusedW8A16Linear, fuse_w8a16_linears  # noqa: F401,E402
from .accelerator import (eet_accelerator, replace_with_eet_fp16_fused_attn, replace_with_eet_fused_mlp,  # noqa: F401,E402
                          replace_with_eet_fused_residual, replace_with_eet_qlinear, replace_with_eet_quant_fused_attn,
                          replace_with_eet_rmsnorm)
from .graph_decoder import GraphDecoder  # noqa: F401,E402
from ..checkpoint import (checkpoint_layout, convert_checkpoint, convert_model_layout_, get_wire_layout,  # noqa: F401,E402
                         quantization_config, set_wire_layout, wire_layout)
from .hf import use_with_transformers  # noqa: F401,E402
