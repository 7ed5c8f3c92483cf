"""Model dispatch: which Model class and which engine plugin serve a checkpoint (the decisions of the reference's
models/__init__.py:31-166 that concern this build's two paths).

* flash path (`FlashCausalLM`, the MI355X kernels): taken when FLASH_ATTENTION is on — by default whenever a GPU is
  visible and the model family has a flash implementation.  As in the reference (:81-114) it always runs on the
  `tgis_native` engine; another `deployment_framework` is overridden with a warning.
* padded path (`CausalLM` on the engine plugin named by `deployment_framework`, e.g. `hf_transformers`): everything
  else that HF can load as a causal LM — BASELINE config 1 and any CPU run."""
import json
import os
from typing import Optional

import torch

from tgis_amd.models.model import Model

# grads are never needed in a serving shard (reference models/__init__.py:28)
torch.set_grad_enabled(False)

FLASH_MODEL_TYPES = ("llama", "gpt_bigcode")


def _flash_requested(model_type: str) -> bool:
    env = os.getenv("FLASH_ATTENTION")
    if env is not None:
        return env.lower() == "true"
    return torch.cuda.is_available() and model_type in FLASH_MODEL_TYPES


def get_model(model_name: str, revision: Optional[str], deployment_framework: str, dtype_str: Optional[str],
              quantize: Optional[str], max_sequence_length: Optional[int]) -> Model:
    from tgis_amd.utils.dist import get_torch_dtype, print_rank_n
    from tgis_amd.utils.hub import get_model_path

    model_path = get_model_path(model_name, revision)
    with open(os.path.join(model_path, "config.json")) as f:
        model_type = json.load(f).get("model_type")
    on_gpu = torch.cuda.is_available()
    # fp16 on GPU, fp32 on CPU unless told otherwise (server.py:287-288)
    dtype = get_torch_dtype(dtype_str) if dtype_str else (torch.float16 if on_gpu else torch.float32)
    if quantize is not None and not on_gpu:
        raise ValueError("Quantization requires CUDA")  # server.py:290-291
    if quantize is not None and quantize != "gptq":
        raise ValueError(f"{quantize} quantization is not supported")

    if _flash_requested(model_type):
        if not on_gpu:
            raise NotImplementedError("FLASH_ATTENTION is set but no GPU is visible: the flash path has no CPU fallback")
        if model_type not in FLASH_MODEL_TYPES:
            raise NotImplementedError(f"Flash attention currently only supported by the following model types: "
                                      f"{list(FLASH_MODEL_TYPES)}")
        if deployment_framework != "tgis_native":
            print_rank_n(f"WARNING: Using deployment engine tgis_native rather than {deployment_framework} "
                         "because FLASH_ATTENTION is enabled")
        if quantize == "gptq" and dtype != torch.float16:
            raise ValueError("GPTQ kernels are fp16-only (utils/gptq/exllamav2.py:18)")
        from tgis_amd.models.flash_causal_lm import FlashCausalLM

        return FlashCausalLM(model_name, revision, "tgis_native", dtype, quantize,
                             max_sequence_length=max_sequence_length)

    if quantize is not None:
        raise ValueError("GPTQ checkpoints are served by the flash path (FLASH_ATTENTION=true, tgis_native engine)")
    if int(os.getenv("WORLD_SIZE", "1")) > 1:
        raise NotImplementedError("more than one shard needs the tgis_native engine, which serves the flash model "
                                  f"families {list(FLASH_MODEL_TYPES)} on GPUs")
    from transformers import AutoConfig
    from transformers.models.auto import modeling_auto

    model_config = AutoConfig.from_pretrained(model_path)
    if model_type not in modeling_auto.MODEL_FOR_CAUSAL_LM_MAPPING_NAMES:
        raise NotImplementedError(f"Unsupported model type {model_type}")
    from tgis_amd.models.causal_lm import CausalLM

    if deployment_framework == "tgis_native":
        # the native engine only builds the flash model classes; the padded batch type runs the library model
        print_rank_n("WARNING: Using deployment engine hf_transformers rather than tgis_native because FLASH_ATTENTION "
                     "is disabled (the tgis_native engine serves the flash path only)")
        deployment_framework = "hf_transformers"
    return CausalLM(model_name, revision, deployment_framework, dtype, quantize, model_config, max_sequence_length)
