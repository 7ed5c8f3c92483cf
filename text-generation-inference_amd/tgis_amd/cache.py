"""Shard-side store of live batches between RPCs (mirrors cache.py:8-33 of the reference)."""
from typing import Dict, List, Optional

from tgis_amd.models.types import Batch


class Cache:
    def __init__(self):
        self.cache: Dict[int, Batch] = {}

    def pop(self, batch_id: int) -> Optional[Batch]:
        return self.cache.pop(batch_id, None)

    def set(self, entry: Optional[Batch]):
        if entry is not None:
            self.cache[entry.batch_id] = entry

    def delete(self, batch_id: int):
        batch = self.cache.pop(batch_id)
        _release(batch)

    def clear(self):
        for batch in self.cache.values():
            _release(batch)
        self.cache.clear()

    def keys(self) -> List[int]:
        return list(self.cache)

    def __len__(self) -> int:
        return len(self.cache)

    def compact(self):
        for batch in self.cache.values():
            batch.compact()


def _release(batch):
    # paged-KV batches hand their pages back explicitly (a contiguous-KV batch just drops its tensors)
    rel = getattr(batch, "release", None)
    if rel is not None:
        rel()
