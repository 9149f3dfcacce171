"""transformers' own EETQ integration on top of this repo's ``EETQ`` operator module.

``transformers.integrations.eetq`` (transformers >= 5) is the caller the reference's README sends users to
(/root/reference/README.md:55-102): ``from_pretrained(..., quantization_config=EetqConfig("int8"))`` swaps every
``nn.Linear`` for transformers' ``EetqLinear`` (int8 ``weight`` and fp16 ``weight_scales`` as ``nn.Parameter``s),
quantises with ``quant_weights(w.t().contiguous().cpu(), torch.int8, False)`` and runs ``w8_a16_gemm(x, weight, scales)``
through a kernel module it fetches from the kernels hub (``get_kernel("kernels-community/quantization-eetq")``: CUDA
binaries, needs the ``kernels`` package and a network).  :func:`use_with_transformers` points those three places at the
compiled ``EETQ`` module of this repo instead -- nothing else of transformers is touched, its ``EetqLinear``,
``EetqLinearMMFunction`` (forward AND backward) and ``EetqQuantize.convert`` run as shipped.

Bytes on disk.  transformers' loader assigns checkpoint tensors straight to the parameters (no ``load_state_dict``), and
its ``EetqLinear`` knows nothing about layouts; the int8 bytes this library computes on are the ``gfx950`` layout, the
bytes EETQ checkpoints hold are the reference's ``sm80`` layout (eetq_amd/checkpoint.py).  With ``wire=True`` (default)
the helper keeps the disk format the reference's:
  * every ``EetqLinear`` it sees gets the ``state_dict`` hook of eetq_amd/checkpoint.py, so ``save_pretrained`` writes
    sm80 bytes (``get_wire_layout()``), readable by CUDA-EETQ / TGI;
  * after a PRE-QUANTISED checkpoint has been loaded, the int8 parameters are re-encoded to gfx950 in place
    (``convert_model_layout_``) FROM THE LAYOUT THE CHECKPOINT NAMES: ``quantization_config["layout"]`` in its config.json
    (``convert_checkpoint(dst="gfx950")``, ``save_quantized`` and ``save_pretrained`` under a non-sm80 wire layout write
    it; the same tag ``models.from_quantized`` honours), ``sm80`` when there is none -- so NVIDIA-written EETQ checkpoints
    load as they are and gfx950-tagged ones are not re-encoded a second time;
  * ``save_pretrained`` under ``wire_layout("gfx950")`` writes that tag (transformers' ``EetqConfig`` would drop it).
Quantise-on-load (fp16 checkpoint + ``EetqConfig``) produces gfx950 bytes directly and needs no re-encode.
"""
import functools

__all__ = ["use_with_transformers", "HUB_KERNEL_NAME"]

HUB_KERNEL_NAME = "kernels-community/quantization-eetq"
_installed = [False]


def use_with_transformers(wire=True):
    """Idempotent.  Returns the ``EETQ`` module transformers will call."""
    import EETQ
    import transformers.integrations.eetq as hf_eetq
    import transformers.integrations.hub_kernels as hub
    import transformers.quantizers.quantizer_eetq as hf_quantizer

    hf_eetq.eetq_kernels_hub = EETQ   # the handle EetqQuantize / EetqLinearMMFunction read at call time
    if _installed[0]:
        return EETQ
    _installed[0] = True

    # 1. replace_with_eetq_linear() re-fetches the handle with get_kernel(<hub name>) on every model load
    hub_get_kernel = hub.get_kernel

    @functools.wraps(hub_get_kernel)
    def get_kernel(kernel_name, *args, **kwargs):
        if kernel_name == HUB_KERNEL_NAME:
            return EETQ
        return hub_get_kernel(kernel_name, *args, **kwargs)
    hub.get_kernel = get_kernel

    # 2. the quantizer's environment check asks for the `kernels` package, which only serves to fetch that module
    hf_quantizer.is_kernels_available = lambda: True

    if wire:
        from ..checkpoint import convert_model_layout_, get_wire_layout, install_layout_hooks
        replace = hf_eetq.replace_with_eetq_linear

        @functools.wraps(replace)
        def replace_with_eetq_linear(model, *args, **kwargs):
            model = replace(model, *args, **kwargs)
            for mod in model.modules():
                if isinstance(mod, hf_eetq.EetqLinear):
                    install_layout_hooks(mod, "weight")   # save hook: gfx950 -> wire bytes in state_dict()
            return model
        hf_eetq.replace_with_eetq_linear = replace_with_eetq_linear
        import transformers.integrations as hf_integrations
        if getattr(hf_integrations, "replace_with_eetq_linear", None) is not None:
            try:
                hf_integrations.replace_with_eetq_linear = replace_with_eetq_linear
            except Exception:  # noqa: BLE001  lazy module that refuses attribute writes: the quantizer patch below covers it
                pass
        cls = hf_quantizer.EetqHfQuantizer
        before = cls._process_model_before_weight_loading
        after = cls._process_model_after_weight_loading

        def _process_model_before_weight_loading(self, model, **kwargs):
            out = before(self, model, **kwargs)
            for mod in model.modules():
                if isinstance(mod, hf_eetq.EetqLinear):
                    install_layout_hooks(mod, "weight")
            return out

        def _process_model_after_weight_loading(self, model, **kwargs):
            out = after(self, model, **kwargs)
            if self.pre_quantized:
                # the layout of the bytes just assigned to the parameters: the checkpoint's own tag (config.json's
                # quantization_config["layout"], kept on the EetqConfig by the __init__ wrapper below; written by
                # convert_checkpoint / models.save_quantized / save_pretrained under a non-sm80 wire layout) -- a
                # reference-written checkpoint has none and IS sm80, whatever the process-wide wire layout says (the same
                # default as checkpoint.read_layout / models.from_quantized: an NVIDIA-written checkpoint loaded under
                # wire_layout("gfx950") was left un-converted before -- garbage weights, silently; round-5 ADVICE)
                src = getattr(self.quantization_config, "layout", None) or "sm80"
                if src != "gfx950":
                    convert_model_layout_(model, src, "gfx950")
            return out
        # transformers' EetqConfig swallows unknown keys (``**kwargs`` dropped), so a ``layout`` tag in config.json would be
        # lost on load and never written on save: keep it on the object, and let to_dict() -- what save_pretrained serialises
        # into config.json -- name the layout the state_dict hook writes (absent for sm80: the reference's config, byte for byte)
        from transformers.utils.quantization_config import EetqConfig
        from ..checkpoint import _check
        cfg_init, cfg_to_dict = EetqConfig.__init__, EetqConfig.to_dict

        @functools.wraps(cfg_init)
        def __init__(self, *args, layout=None, **kwargs):
            cfg_init(self, *args, **kwargs)
            if layout is not None:
                self.layout = _check(layout)

        @functools.wraps(cfg_to_dict)
        def to_dict(self):
            d = cfg_to_dict(self)
            d.pop("layout", None)
            if get_wire_layout() != "sm80":
                d["layout"] = get_wire_layout()
            return d
        EetqConfig.__init__ = __init__
        EetqConfig.to_dict = to_dict
        cls._process_model_before_weight_loading = _process_model_before_weight_loading
        cls._process_model_after_weight_loading = _process_model_after_weight_loading
    return EETQ
