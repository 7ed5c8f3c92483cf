"""Engine plugin `hf_transformers`: the model is whatever `AutoModelForCausalLM.from_pretrained` builds from a local
checkpoint directory, on the device of this rank (CPU when there is no GPU) — the engine behind the padded `CausalLM`
path and BASELINE config 1 (mirrors inference_engine/hf_transformers.py:11-78 of the reference).

Not offered here: `bitsandbytes` / AutoGPTQ loading (neither library is in the image; int4 GPTQ is served by the
`tgis_native` engine on the HIP kernels) and the `hf_accelerate` device maps."""
import os
from typing import Any, Optional

import torch

from tgis_amd.inference_engine.engine import BaseInferenceEngine


class InferenceEngine(BaseInferenceEngine):
    def __init__(self, model_path: Optional[str], model_class, dtype: torch.dtype, quantize: Optional[str],
                 model_config: Optional[Any], max_sequence_length: Optional[int] = None, *, preloaded=None,
                 tokenizer=None) -> None:
        """`preloaded` (an already constructed HF model) replaces the checkpoint read: tests and the cfg1 benchmark
        build seeded random models in memory, there are no checkpoint files offline."""
        if preloaded is not None and model_config is None:
            model_config = preloaded.config
        super().__init__(model_path, model_config, tokenizer=tokenizer)
        if quantize is None and getattr(self._config, "quantization_config", None):
            qc = self._config.quantization_config
            quantize = qc.get("quant_method") if isinstance(qc, dict) else getattr(qc, "quant_method", None)
        if quantize is not None:
            raise ValueError(f"{quantize} quantization not supported by hf_transformers engine in this build "
                             "(GPTQ checkpoints are served by the tgis_native engine)")
        if preloaded is not None:
            model = preloaded
        else:
            import transformers

            dtype_kw = "dtype" if int(transformers.__version__.split(".")[0]) >= 5 else "torch_dtype"
            kwargs = {"pretrained_model_name_or_path": model_path, "local_files_only": True, dtype_kw: dtype}
            if attn_impl := os.getenv("TRANSFORMERS_ATTN_IMPL"):
                kwargs["attn_implementation"] = attn_impl
            model = model_class.from_pretrained(**kwargs)
        self.model = model.to(device=self.device, dtype=dtype).requires_grad_(False).eval()
        self.process_group = None
