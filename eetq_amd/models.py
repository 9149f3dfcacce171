"""``from eetq import AutoEETQForCausalLM``: the reference's offline quantise-and-save entry (README.md:55-68).

Counterpart of /root/reference/python/eetq/models/{auto,base}.py (``AutoEETQForCausalLM`` auto.py:19-38,
``BaseEETQForCausalLM.from_pretrained / quantize / save_quantized`` base.py:40-205, ``EETQConfig`` _config.py:8-85), reduced
to what the hot path needs: load an fp16 causal LM with transformers, swap its linears for ``W8A16Linear`` (quantised on the
GPU by the HIP quantiser), write a checkpoint whose int8 tensors are in the reference's processed layout (``sm80`` -- the
modules' ``state_dict`` hooks re-encode, eetq_amd/checkpoint.py) next to ``quantization_config = {"quant_method": "eetq",
"zero_point": false, "bits": 8}`` (_config.py:80-85), i.e. the files CUDA-EETQ / TGI read.

Differences, on purpose:
  * the reference fuses q|k|v and gate|up before quantising and splits them again afterwards (llama.py:15-77): per-channel
    symmetric scales belong to output columns, so fuse -> quantise -> split gives exactly the per-projection result; the
    detour is skipped and every projection is quantised as it is (same keys, same bytes);
  * ``tp > 1`` (offline weight slicing for tensor-parallel servers, utils/base.py:132-250) is refused: this port keeps the
    model replicated per GPU (BASELINE.json north_star);
  * ``from_quantized`` WORKS here (the reference's is ``pass``, auto.py:34-38): skeleton from the config, ``W8A16Linear``
    shells (``eet_quantize(init_only=True)``), tensors through ``load_state_dict`` (layout re-encoded on the way in).
"""
import json
import os

import torch
import torch.nn as nn

from .checkpoint import checkpoint_layout, wire_layout
from .utils.quantizer import eet_quantize

__all__ = ["AutoEETQForCausalLM", "EETQForCausalLM", "EETQConfig", "EETQ_CAUSAL_LM_MODEL_TYPES"]

# model types the reference maps (auto.py:6-10); all of them load through AutoModelForCausalLM (base.py:32-36).  Any other
# decoder whose linears are plain nn.Linear works too (allow_any_model_type=True).
EETQ_CAUSAL_LM_MODEL_TYPES = ("llama", "baichuan", "gemma")


class EETQConfig:
    """quant_config.json / config.json["quantization_config"] of an EETQ checkpoint (_config.py:8-85)."""
    config_file_name = "quant_config.json"

    def __init__(self, quant_method="eetq", zero_point=False, w_bit=8, **ignored):
        self.quant_method, self.zero_point, self.w_bit = quant_method, bool(zero_point), int(w_bit)

    @classmethod
    def from_pretrained(cls, save_dir, **kwargs):
        path = os.path.join(save_dir, cls.config_file_name)
        if os.path.isdir(save_dir) and os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                return cls(**json.load(f))
        return cls()

    def to_transformers_dict(self):
        return {"quant_method": self.quant_method, "zero_point": self.zero_point, "bits": self.w_bit}


def _model_type(model_dir, trust_remote_code, allow_any):
    from transformers import AutoConfig
    config = AutoConfig.from_pretrained(model_dir, trust_remote_code=trust_remote_code)
    if config.model_type not in EETQ_CAUSAL_LM_MODEL_TYPES and not allow_any:
        raise TypeError("%s isn't supported yet." % config.model_type)        # auto.py:14-16
    return config


class EETQForCausalLM(nn.Module):
    """The object ``AutoEETQForCausalLM.from_pretrained`` returns (BaseEETQForCausalLM, base.py:40-66)."""

    def __init__(self, model, model_type, is_quantized, config, quant_config):
        super().__init__()
        self.model = model
        self.model_type = model_type
        self.is_quantized = is_quantized
        self.config = config
        self.quant_config = quant_config

    def to(self, device):
        return self.model.to(device)

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def generate(self, *args, **kwargs):
        with torch.inference_mode():
            return self.model.generate(*args, **kwargs)

    @torch.no_grad()
    def quantize(self, save_dir, tp=1):
        """Quantise every nn.Linear but lm_head (``eet_quantize``) and save the result (base.py:68-98)."""
        if tp != 1:
            raise ValueError("tensor-parallel export (tp > 1) is not part of this port: the model stays replicated per GPU")
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":     # the quantiser and the kernels are GPU-only; the reference quantises wherever the weights are
            self.model.to("cuda:0")
        eet_quantize(self.model)
        self.is_quantized = True
        print("[EET][INFO] saving model ...")
        self.save_quantized(save_dir)

    def save_quantized(self, save_dir, safetensors=True, shard_size="5GB"):
        """config.json with the EETQ ``quantization_config`` + the state dict in the reference's layout (base.py:101-146)."""
        save_dir = save_dir.rstrip("/") or save_dir
        self.model.config.quantization_config = self.quant_config.to_transformers_dict()
        with wire_layout("sm80"):    # what CUDA-EETQ writes; the modules' state_dict hooks do the re-encoding
            self.model.save_pretrained(save_dir, safe_serialization=safetensors, max_shard_size=shard_size)
        return save_dir

    @classmethod
    def from_pretrained(cls, model_path, model_type, torch_dtype=torch.float16, trust_remote_code=True, safetensors=True,
                        device_map=None, **model_init_kwargs):
        """An fp16 model, ready for ``quantize`` (base.py:149-205)."""
        import transformers
        config = transformers.AutoConfig.from_pretrained(model_path, trust_remote_code=trust_remote_code)
        model = transformers.AutoModelForCausalLM.from_pretrained(
            model_path, trust_remote_code=trust_remote_code, dtype=torch_dtype, use_safetensors=safetensors,
            device_map=device_map, **model_init_kwargs)
        model.eval()
        return cls(model, model_type, is_quantized=False, config=config, quant_config=EETQConfig.from_pretrained(model_path))

    @classmethod
    def from_quantized(cls, quant_path, model_type, torch_dtype=torch.float16, trust_remote_code=True, device="cuda:0"):
        """A checkpoint written by ``quantize`` / ``save_quantized`` here OR by CUDA-EETQ (int8 tensors in the reference's
        layout unless config.json tags another one): skeleton from the config, W8A16Linear shells, tensors through
        load_state_dict -- every quantised module re-encodes its weight to the native layout on the way in."""
        import transformers
        from safetensors.torch import load_file
        config = transformers.AutoConfig.from_pretrained(quant_path, trust_remote_code=trust_remote_code)
        layout = checkpoint_layout(config)
        qc = getattr(config, "quantization_config", None)
        if hasattr(config, "quantization_config"):
            del config.quantization_config      # the skeleton is a plain fp16 model; transformers must not pick a quantizer
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch_dtype)
        try:
            with torch.device(device):
                model = transformers.AutoModelForCausalLM.from_config(config, trust_remote_code=trust_remote_code)
        finally:
            torch.set_default_dtype(old)
        model.eval()
        eet_quantize(model, init_only=True)
        files = sorted(f for f in os.listdir(quant_path) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError("no .safetensors files in %s" % quant_path)
        expected, seen = set(model.state_dict().keys()), set()
        with wire_layout(layout):
            for fn in files:                      # shard by shard: bounded host memory
                sd = load_file(os.path.join(quant_path, fn))
                seen.update(sd)
                model.load_state_dict(sd, strict=False)
        missing = sorted(k for k in expected - seen if "rotary_emb.inv_freq" not in k)
        tied = getattr(config, "tie_word_embeddings", False)
        missing = [k for k in missing if not (tied and k.startswith("lm_head."))]
        if missing:
            raise KeyError("checkpoint %s lacks %d tensors, e.g. %s" % (quant_path, len(missing), missing[:3]))
        if qc is not None:
            model.config.quantization_config = qc
        return cls(model, model_type, is_quantized=True, config=model.config, quant_config=EETQConfig.from_pretrained(quant_path))


class AutoEETQForCausalLM:
    def __init__(self):
        raise EnvironmentError("You must instantiate AutoEETQForCausalLM with\n"
                               "AutoEETQForCausalLM.from_quantized or AutoEETQForCausalLM.from_pretrained")

    @classmethod
    def from_pretrained(cls, model_path, trust_remote_code=True, safetensors=True, device_map=None,
                        allow_any_model_type=False, **model_init_kwargs):
        config = _model_type(model_path, trust_remote_code, allow_any_model_type)
        return EETQForCausalLM.from_pretrained(model_path, config.model_type, trust_remote_code=trust_remote_code,
                                               safetensors=safetensors, device_map=device_map, **model_init_kwargs)

    @classmethod
    def from_quantized(cls, quant_path, quant_filename="", max_new_tokens=None, trust_remote_code=True, fuse_layers=True,
                       safetensors=True, device_map="balanced", offload_folder=None, allow_any_model_type=False,
                       **config_kwargs):
        """Signature of auto.py:33-37 (whose body is ``pass``); arguments this port has no use for are accepted and ignored
        (single-device replicas: ``device_map`` only names the GPU when it is a "cuda:N" string)."""
        config = _model_type(quant_path, trust_remote_code, allow_any_model_type)
        device = device_map if isinstance(device_map, str) and device_map.startswith("cuda") else "cuda:0"
        return EETQForCausalLM.from_quantized(quant_path, config.model_type, trust_remote_code=trust_remote_code, device=device)
