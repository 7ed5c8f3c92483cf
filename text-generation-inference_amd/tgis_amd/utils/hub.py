"""Local checkpoint lookup (the offline subset of utils/hub.py of the reference: get_model_path accepts a
directory holding config.json, utils/hub.py:91-92; weight download/convert CLIs are out of scope)."""
import os
from pathlib import Path
from typing import List, Optional


def get_model_path(model_name: str, revision: Optional[str] = None) -> str:
    if os.path.isfile(os.path.join(model_name, "config.json")):
        return model_name
    raise ValueError(f"Weights not found in local cache for model {model_name} (no network in this build)")


def local_weight_files(model_path: str, extension: str = ".safetensors") -> List[Path]:
    return sorted(Path(model_path).glob(f"*{extension}"))
