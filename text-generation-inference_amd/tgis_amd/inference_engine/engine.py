"""What every engine plugin shares: the model config, the tokenizer, and the rank -> device map.

An engine plugin is a module `tgis_amd.inference_engine.<DEPLOYMENT_FRAMEWORK>` exporting `InferenceEngine`; the model
classes use its `.model`, `.tokenizer`, `.device`, `.rank`, `.world_size`, `.process_group` and the two getters below
(drop-in boundary #3 of SURVEY.md §8b; the base the reference provides is inference_engine/engine.py:11-37).
Tokenizers pad and truncate on the LEFT: decoder-only batches are right-aligned."""
import os
from typing import Any, Optional, Tuple

import torch


def _shard_env() -> Tuple[int, int]:
    """(rank, world size) of this shard process, as the launcher exports them."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def _device_for(rank: int, world_size: int) -> torch.device:
    if not torch.cuda.is_available():
        return torch.device("cpu")
    visible = torch.cuda.device_count()
    shared_ok = os.environ.get("TGIS_ALLOW_SHARED_GPU") == "1"  # several ranks on one device: TP tests, gloo collectives
    if world_size > visible and not shared_ok:
        raise AssertionError(f"{world_size} shards configured but only {visible} GPUs detected")
    index = rank % visible
    torch.cuda.set_device(index)
    return torch.device("cuda", index)


class BaseInferenceEngine:
    def __init__(self, model_path: Optional[str], model_config: Optional[Any], tokenizer=None) -> None:
        self.model = None  # set by the plugin once the weights are loaded
        self.rank, self.world_size = _shard_env()
        self.device = _device_for(self.rank, self.world_size)
        self._config = model_config if model_config is not None else self._load_config(model_path)
        if tokenizer is None and model_path is not None:
            tokenizer = self._load_tokenizer(model_path)
        self.tokenizer = tokenizer

    @staticmethod
    def _load_config(model_path: str):
        from transformers import AutoConfig

        return AutoConfig.from_pretrained(model_path)

    @staticmethod
    def _load_tokenizer(model_path: str):
        from transformers import AutoTokenizer

        return AutoTokenizer.from_pretrained(model_path, padding_side="left", truncation_side="left")

    def get_device(self) -> torch.device:
        return self.device

    def get_components(self):
        """(config, tokenizer, model) — the triple `Model.__init__` unpacks."""
        return self.model.config, self.tokenizer, self.model
