"""Base of every engine plugin (mirrors inference_engine/engine.py:11-37 of the reference): loads the config
and a left-padding / left-truncating tokenizer, maps RANK -> device, exposes get_components()/get_device()."""
import os
from typing import Any, Optional

import torch


class BaseInferenceEngine:
    def __init__(self, model_path: Optional[str], model_config: Optional[Any], tokenizer=None) -> None:
        if model_config is None:
            from transformers import AutoConfig

            model_config = AutoConfig.from_pretrained(model_path)
        self._config = model_config
        if tokenizer is None and model_path is not None:
            from transformers import AutoTokenizer

            tokenizer = AutoTokenizer.from_pretrained(model_path, padding_side="left", truncation_side="left")
        self.tokenizer = tokenizer
        self.model = None
        self.rank = int(os.getenv("RANK", "0"))
        self.world_size = int(os.getenv("WORLD_SIZE", "1"))
        if torch.cuda.is_available():
            gpu_count = torch.cuda.device_count()
            # TGIS_ALLOW_SHARED_GPU=1: several ranks on one device (TP tests on a single-GPU box, gloo collectives)
            assert self.world_size <= gpu_count or os.getenv("TGIS_ALLOW_SHARED_GPU") == "1", \
                f"{self.world_size} shards configured but only {gpu_count} GPUs detected"
            device_index = self.rank % gpu_count
            torch.cuda.set_device(device_index)
            self.device = torch.device("cuda", device_index)
        else:
            self.device = torch.device("cpu")

    def get_components(self):
        return self.model.config, self.tokenizer, self.model

    def get_device(self) -> torch.device:
        return self.device
