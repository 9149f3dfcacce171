"""Drop-in for the reference's native extension module ``EETQ`` (csrc/eetpy.cpp:7-19):
``from EETQ import quant_weights, preprocess_weights, w8_a16_gemm`` keeps working, served by eetq_amd."""
from eetq_amd.ops import (layernorm_forward, preprocess_weights, quant_weights,  # noqa: F401
                          rotary_embedding_neox, w8_a16_gemm, w8_a16_gemm_)
