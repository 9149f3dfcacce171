from .qlinear import (EetqLinear, EetqLinearMMFunction, W8A16Linear,  # noqa: F401
                      quantize_and_preprocess_weights)
