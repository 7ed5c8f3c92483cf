"""The native engine: model type -> MI355X model class, process group, safetensors `Weights`, GPTQ params
(mirrors inference_engine/tgis_native.py:23-139 of the reference; model families of the configs in
BASELINE.json: llama, and gpt_bigcode when flash_santacoder_modeling is present)."""
import os
from typing import Any, Optional

import torch
import torch.distributed

from tgis_amd.inference_engine.engine import BaseInferenceEngine
from tgis_amd.utils.dist import initialize_torch_distributed
from tgis_amd.utils.hub import local_weight_files
from tgis_amd.utils.weights import Weights

FLASH_TYPES = ["llama", "gpt_bigcode"]


def _barrier(group):
    if hasattr(group, "barrier") and not isinstance(group, torch.distributed.ProcessGroup):
        group.barrier()
    else:
        torch.distributed.barrier(group=group)


def model_class_for(config):
    model_type = config.model_type
    aliases = None
    if model_type == "llama":
        if getattr(config, "tie_word_embeddings", False):
            aliases = {"lm_head.weight": ["model.embed_tokens.weight"]}
        from tgis_amd.models.custom_modeling.flash_llama_modeling import FlashLlamaForCausalLM

        return FlashLlamaForCausalLM, aliases
    if model_type == "gpt_bigcode":
        try:
            from tgis_amd.models.custom_modeling.flash_santacoder_modeling import FlashSantacoderForCausalLM
        except ImportError as e:
            raise NotImplementedError("gpt_bigcode (Santacoder/Starcoder) is not built yet in this round") from e
        config.transpose = config.architectures[0].startswith("GPT2")
        return FlashSantacoderForCausalLM, {"transformer.wte.weight": ["lm_head.weight"]}
    raise NotImplementedError(f"Flash attention currently only supported by the following model types: {FLASH_TYPES}")


class InferenceEngine(BaseInferenceEngine):
    def __init__(self, model_path: str, model_class, dtype: torch.dtype, quantize: Optional[str],
                 model_config: Optional[Any], max_sequence_length: Optional[int]) -> None:
        super().__init__(model_path, model_config)
        model_class, aliases = model_class_for(self._config)
        self._config.quantize = quantize
        self.process_group = initialize_torch_distributed(self.world_size, self.rank)
        self.master = self.rank == 0
        _barrier(self.process_group)
        filenames = local_weight_files(model_path, extension=".safetensors")
        if not filenames:
            raise ValueError("No safetensors weights found - required for tgis_native engine")
        weights = Weights(filenames, device=self.device, dtype=dtype, process_group=self.process_group,
                          aliases=aliases)
        if quantize == "gptq":
            weights._set_gptq_params(self._config, model_path)
        model = model_class(self._config, weights)
        _barrier(self.process_group)
        if not hasattr(model, "config"):
            model.config = self._config
        self.model = model
