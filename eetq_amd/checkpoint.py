"""Checkpoint / wire-format compatibility of the quantised weights.

The reference stores the *processed* int8 bytes in its checkpoints: ``save_pretrained`` / ``save_quantized``
(python/eetq/models/base.py:108-146, README.md:62-68) write ``qweight`` exactly as ``quant_weights`` returned it, i.e. in
the layout of the CUDA architecture that quantised the model (cutlass_preprocessors.cc:113-128; sm75..sm89 share one
layout, called ``sm80`` here), tagged only by ``quantization_config = {"quant_method": "eetq", "zero_point": false,
"bits": 8}`` (python/eetq/models/_config.py:80-85).  This library computes on its own ``gfx950`` layout, so:

* **on disk the weights are kept in the reference's layout** ("wire layout", default ``sm80``): every quantised module
  carries a ``state_dict`` hook that re-encodes its int8 buffer gfx950 -> wire on the way out and a ``load_state_dict``
  pre-hook that re-encodes wire -> gfx950 on the way in.  A checkpoint written here loads in CUDA-EETQ and vice versa,
  with no extra keys.  In memory (and in ``torch.save(model)`` pickles, which keep working: the hooks are module-level
  callables) the buffers stay gfx950.  The layout of the bytes is recorded IN BAND, in the state dict's ``_metadata``
  entry of the module (next to nn.Module's ``version``; it survives ``torch.save(state_dict)`` and adds no key), and the
  load hook trusts that tag first: a state dict saved under ``wire_layout("gfx950")`` loads correctly whatever the
  process-wide setting is at load time.  Untagged dicts (safetensors files, reference-written checkpoints) are read as the
  module's ``checkpoint_layout`` pin or the process-wide wire layout -- "sm80" unless told otherwise.
* shapes the reference's layout cannot hold (it needs K % 64 == 0 and N % 64 == 0; the reference cannot quantise such
  layers at all) pass through both hooks unchanged.
* :func:`set_wire_layout` / :func:`wire_layout` choose the layout for a process or a block (``"gfx950"`` = store native
  bytes, e.g. for checkpoints that only this library will read and that should load without a re-encode);
  :func:`quantization_config` is the config dict with a ``layout`` tag naming what was written, and
  :func:`checkpoint_layout` reads the tag back (absent = a reference-written checkpoint = ``sm80``).
* :func:`convert_checkpoint` rewrites a safetensors checkpoint directory from one layout to the other offline, and
  :func:`convert_model_layout_` re-encodes the int8 buffers of a live model in place -- for loaders that fill module
  buffers directly instead of calling ``load_state_dict`` (transformers' own EETQ integration does).
"""
import contextlib
import json
import os

import torch

__all__ = ["set_wire_layout", "get_wire_layout", "wire_layout", "set_state_dict_device", "install_layout_hooks",
           "convert_model_layout_",
           "convert_checkpoint", "quantization_config", "checkpoint_layout", "QUANTIZED_WEIGHT_NAMES"]

_LAYOUTS = ("sm80", "gfx950")
_wire = ["sm80"]
# buffer names of the int8 weight in the two module shapes (W8A16Linear / EetqLinear)
QUANTIZED_WEIGHT_NAMES = ("qweight", "weight")


def _check(layout):
    if layout not in _LAYOUTS:
        raise ValueError("unknown checkpoint layout %r (expected one of %s)" % (layout, ", ".join(_LAYOUTS)))
    return layout


def set_wire_layout(layout):
    """Layout of the int8 weights inside state dicts / checkpoints from now on: "sm80" (the reference's, default) or
    "gfx950" (native bytes)."""
    _wire[0] = _check(layout)


def get_wire_layout():
    return _wire[0]


@contextlib.contextmanager
def wire_layout(layout):
    old = _wire[0]
    set_wire_layout(layout)
    try:
        yield
    finally:
        _wire[0] = old


def _wire_holds(shape):
    """the reference's processed layout exists for K % 64 == 0 and N % 64 == 0 only (cutlass_preprocessors.cc:139-142 via
    fpA_intB_gemm_template.h, :455)"""
    return len(shape) == 2 and shape[0] % 64 == 0 and shape[1] % 64 == 0


def _reencode(tensor, src, dst):
    if src == dst or tensor.dtype != torch.int8 or not _wire_holds(tensor.shape):
        return tensor
    from .ops import convert_layout
    return convert_layout(tensor.contiguous(), src, dst)


def _offload(t):
    """state_dict() entries re-encoded to the wire layout are NEW tensors (the live buffer stays gfx950); with
    ``set_state_dict_device("cpu")`` they are moved to host memory one by one, so taking the state dict of a large model
    does not hold a second device copy of every int8 weight."""
    return t.to(_sd_device[0]) if _sd_device[0] is not None else t


_sd_device = [None]
_META_KEY = "eetq_layout"   # in-band layout tag: state_dict()._metadata[<module path>]["eetq_layout"]


def set_state_dict_device(device):
    """Where the re-encoded int8 tensors of ``state_dict()`` live: None (default) = next to the module's buffers,
    "cpu" = offloaded tensor by tensor (bounds device memory while saving)."""
    _sd_device[0] = device


class _SaveHook:
    """state_dict hook (a module-level class, so modules that carry it stay picklable: ``torch.save(model)`` works).
    Re-encodes the int8 buffer gfx950 -> wire and records, in the state dict's ``_metadata`` entry of the module (the
    dict nn.Module keeps its ``version`` in; it travels with ``torch.save(state_dict)`` and adds no key), which layout
    the bytes are in -- "gfx950" for shapes the reference's layout cannot hold."""

    def __init__(self, weight_name):
        self.weight_name = weight_name

    def __call__(self, mod, state_dict, prefix, local_metadata):
        key = prefix + self.weight_name
        t = state_dict.get(key)
        if t is None or t.dtype != torch.int8 or not t.numel() or t.is_meta:   # meta: a skeleton being loaded, no bytes yet
            return
        dst = _wire[0] if _wire_holds(t.shape) else "gfx950"
        if dst != "gfx950":
            state_dict[key] = _offload(_reencode(t, "gfx950", dst))
        if local_metadata is not None:   # the very dict stored at state_dict._metadata[prefix[:-1]]
            local_metadata[_META_KEY] = dst


class _LoadHook:
    """load_state_dict pre-hook (picklable, see _SaveHook).  Source layout, in this order: the in-band tag written by
    _SaveHook (present in ``torch.save``d state dicts of this library), the module's ``checkpoint_layout`` pin, the
    process-wide wire layout (default "sm80" = what the reference writes; safetensors files and reference-written state
    dicts carry no tag)."""

    def __init__(self, weight_name):
        self.weight_name = weight_name

    def __call__(self, mod, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + self.weight_name
        t = state_dict.get(key)
        if t is None or not isinstance(t, torch.Tensor) or t.dtype != torch.int8 or not t.numel() or t.is_meta:
            return
        src = (local_metadata or {}).get(_META_KEY) or getattr(mod, "checkpoint_layout", None) or _wire[0]
        state_dict[key] = _reencode(t, _check(src), "gfx950")


def install_layout_hooks(module, weight_name):
    """Give ``module`` (whose int8 buffer is called ``weight_name``) the save / load re-encoding hooks described in the
    module docstring.  Idempotent; the hooks are picklable."""
    if getattr(module, "_eetq_layout_hooks", False):
        return module
    module._eetq_layout_hooks = True
    module._register_state_dict_hook(_SaveHook(weight_name))
    module._register_load_state_dict_pre_hook(_LoadHook(weight_name), with_module=True)
    return module


def _quantized_buffers(model):
    for mod in model.modules():
        scales = getattr(mod, "weight_scales", None)
        if scales is None:
            continue
        for name in QUANTIZED_WEIGHT_NAMES:
            t = getattr(mod, name, None)
            if isinstance(t, torch.Tensor) and t.dtype == torch.int8 and t.dim() == 2:
                yield mod, name, t
                break


def convert_model_layout_(model, src, dst="gfx950"):
    """Re-encode, in place, the int8 weight of every quantised module of a live model (anything with an int8 ``qweight``
    or ``weight`` next to ``weight_scales``).  For loaders that copy checkpoint tensors straight into module buffers."""
    _check(src), _check(dst)
    n = 0
    for mod, name, t in _quantized_buffers(model):
        new = _reencode(t.data, src, dst)
        if new is not t.data:
            t.data.copy_(new.to(t.device))
            n += 1
    return n


def quantization_config(layout=None):
    """The reference's ``quantization_config`` (python/eetq/models/_config.py:80-85) plus the layout the bytes are in."""
    return {"quant_method": "eetq", "zero_point": False, "bits": 8, "layout": _check(layout or _wire[0])}


def checkpoint_layout(config):
    """Layout named by a config (dict, or an object with ``quantization_config``); a reference-written checkpoint has no
    tag and is ``sm80``."""
    qc = config if isinstance(config, dict) else getattr(config, "quantization_config", None)
    if qc is not None and not isinstance(qc, dict):
        qc = qc.to_dict() if hasattr(qc, "to_dict") else dict(vars(qc))
    if isinstance(qc, dict) and "quantization_config" in qc and "quant_method" not in qc:
        qc = qc["quantization_config"]
    return _check((qc or {}).get("layout", "sm80"))


def _is_quantized_weight_key(key, tensors):
    if not key.endswith(tuple("." + n for n in QUANTIZED_WEIGHT_NAMES)) and key not in QUANTIZED_WEIGHT_NAMES:
        return False
    stem = key.rsplit(".", 1)[0] + "." if "." in key else ""
    return tensors[key].dtype == torch.int8 and (stem + "weight_scales") in tensors


def convert_checkpoint(src_dir, dst_dir=None, src=None, dst="gfx950"):
    """Rewrite every ``*.safetensors`` file of a checkpoint directory with its EETQ int8 weights re-encoded from layout
    ``src`` (default: what ``config.json`` says, i.e. ``sm80`` for a reference-written checkpoint) to ``dst``; every
    other tensor and file is copied as it is and ``config.json``'s ``quantization_config`` gets ``"layout": dst``.
    ``dst_dir=None`` converts in place.  Returns the number of re-encoded tensors."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    import shutil
    dst_dir = dst_dir or src_dir
    cfg_path = os.path.join(src_dir, "config.json")
    cfg = json.load(open(cfg_path)) if os.path.exists(cfg_path) else None
    if src is None:
        src = checkpoint_layout(cfg or {})
    _check(src), _check(dst)
    os.makedirs(dst_dir, exist_ok=True)
    count = 0
    for name in sorted(os.listdir(src_dir)):
        path = os.path.join(src_dir, name)
        out = os.path.join(dst_dir, name)
        if name.endswith(".safetensors"):
            with safe_open(path, framework="pt") as f:
                meta = f.metadata()
                tensors = {k: f.get_tensor(k) for k in f.keys()}
            for k in list(tensors):
                if _is_quantized_weight_key(k, tensors):
                    new = _reencode(tensors[k], src, dst)
                    if new is not tensors[k]:
                        tensors[k] = new.cpu().contiguous()
                        count += 1
            save_file(tensors, out + ".tmp" if out == path else out, metadata=meta)
            if out == path:
                os.replace(out + ".tmp", out)
        elif name == "config.json":
            continue
        elif out != path and os.path.isfile(path):
            shutil.copy2(path, out)
    if cfg is not None:
        qc = dict(cfg.get("quantization_config") or {"quant_method": "eetq", "zero_point": False, "bits": 8})
        qc["layout"] = dst
        cfg["quantization_config"] = qc
        with open(os.path.join(dst_dir, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
    return count
