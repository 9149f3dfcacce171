"""Whole-model quantisation entry point.

Host-side mirror of /root/reference/python/eetq/utils/quantizer.py:40-61 (``eet_quantize``) and the helpers
it needs from python/eetq/utils/base.py (``find_layers`` :280-285, ``set_op_by_name`` :25-38,
``get_named_linears`` :273-274).  The checkpoint-export helpers of base.py (fuse/split/"tp") are offline
weight surgery outside the GEMM path and are out of scope.
"""
import gc

import torch
import torch.nn as nn

from ..modules.qlinear import W8A16Linear

__all__ = ["eet_quantize", "find_layers", "set_op_by_name", "get_named_linears"]


def find_layers(module, include=(nn.Linear,), exclude=("lm_head",)):
    """name -> module for every submodule whose exact type is in ``include`` and whose qualified name
    contains none of the ``exclude`` substrings."""
    include = tuple(include)
    found = {}
    for name, sub in module.named_modules():
        if type(sub) in include and not any(tag in name for tag in exclude):
            found[name] = sub
    return found


def get_named_linears(module):
    return {name: m for name, m in module.named_modules() if isinstance(m, nn.Linear) and "lm_head" not in name}


def set_op_by_name(root, name, new_module):
    """Replace the submodule at dotted path ``name`` (numeric components index containers)."""
    *parents, leaf = name.split(".")
    node = root
    for part in parents:
        node = node[int(part)] if part.isdigit() else getattr(node, part)
    # plain assignment keeps the child's position in ordered containers (the reference's delattr + setattr moves
    # it to the end, which reorders an nn.Sequential)
    setattr(node, leaf, new_module)


def _progress(items, desc):
    try:
        from tqdm import tqdm
        return tqdm(items, desc=desc)
    except Exception:  # tqdm is cosmetic
        return items


def eet_quantize(model, init_only=False, include=[nn.Linear], exclude=["lm_head"], device="cuda:0"):
    """Swap every matching ``nn.Linear`` of ``model`` for a :class:`W8A16Linear` (in place).

    fp16 weights are quantised by the HIP quantiser; int8 weights (bitsandbytes ``Linear8bitLt``) reuse their
    ``SCB / 127`` scales; other dtypes raise ValueError.  ``device`` is accepted for signature
    compatibility; like the reference, each layer stays on the device its weight is on.
    """
    targets = find_layers(model, include=include, exclude=exclude)
    desc = "[EET][INFO] quantization preprocessing..." + ("(init only)" if init_only else "")
    for name in _progress(list(targets), desc):
        linear = targets.pop(name)  # the model and this loop hold the only references
        wdtype = linear.weight.dtype
        if wdtype == torch.float16:
            qlinear = W8A16Linear.from_torch(linear, scales=None, init_only=init_only)
        elif wdtype == torch.int8:
            scales = torch.div(linear.state_dict()["SCB"], 127.0)
            qlinear = W8A16Linear.from_torch(linear, scales=scales, init_only=init_only)
        else:
            raise ValueError("Unsupported data type: {}".format(wdtype))
        set_op_by_name(model, name, qlinear)
        # the reference moves the replaced nn.Linear to the CPU before dropping it (quantizer.py:53-57): a 26 GB copy
        # over PCIe for a 13B model that nothing reads; dropping the last reference frees the HBM just the same
        del linear
        if not init_only and torch.cuda.is_available():
            torch.cuda.empty_cache()
    gc.collect()
    return model
