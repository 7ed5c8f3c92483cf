"""Model dispatch (mirrors models/__init__.py:31-166 of the reference for the flash path)."""
from typing import Optional

import torch

from tgis_amd.models.model import Model

# grads are never needed in a serving shard (reference models/__init__.py:28)
torch.set_grad_enabled(False)


def get_model(model_name: str, revision: Optional[str], deployment_framework: str, dtype_str: Optional[str],
              quantize: Optional[str], max_sequence_length: Optional[int]) -> Model:
    from tgis_amd.models.flash_causal_lm import FlashCausalLM
    from tgis_amd.utils.dist import get_torch_dtype

    if not torch.cuda.is_available():
        raise NotImplementedError("this build serves the GPU flash path only; there is no CPU fallback")
    if quantize is not None and quantize != "gptq":
        raise ValueError(f"{quantize} quantization is not supported")
    dtype = get_torch_dtype(dtype_str) if dtype_str else torch.float16  # fp16 default on GPU (server.py:287-288)
    if quantize == "gptq" and dtype != torch.float16:
        raise ValueError("GPTQ kernels are fp16-only (utils/gptq/exllamav2.py:18)")
    # FLASH_ATTENTION forces the tgis_native engine in the reference (models/__init__.py:81-114)
    return FlashCausalLM(model_name, revision, "tgis_native", dtype, quantize,
                         max_sequence_length=max_sequence_length)
