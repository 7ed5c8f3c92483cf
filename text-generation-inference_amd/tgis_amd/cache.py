"""Shard-side store of the batches that stay alive between two RPCs.

The servicer pops the batches a `NextToken` request names, works on them and puts the survivor back; `ClearCache`
drops one or all of them (the contract of the reference's cache.py:8-33).  Batches of this engine own pages of the KV
pool, so dropping one hands its pages back explicitly instead of waiting for the garbage collector."""
from collections import OrderedDict
from typing import List, Optional

from tgis_amd.models.types import Batch


def _give_back(batch) -> None:
    release = getattr(batch, "release", None)  # contiguous-KV batch types have nothing to hand back
    if callable(release):
        release()


class Cache:
    def __init__(self):
        self.cache: "OrderedDict[int, Batch]" = OrderedDict()  # insertion order = age

    def __len__(self) -> int:
        return len(self.cache)

    def keys(self) -> List[int]:
        return [bid for bid in self.cache]

    def set(self, entry: Optional[Batch]) -> None:
        if entry is None:
            return
        self.cache[entry.batch_id] = entry

    def pop(self, batch_id: int) -> Optional[Batch]:
        """The batch, now owned by the caller (None if the id is unknown)."""
        return self.cache.pop(batch_id, None)

    def delete(self, batch_id: int) -> None:
        _give_back(self.cache.pop(batch_id))  # KeyError for an unknown id, as a dict would

    def clear(self) -> None:
        while self.cache:
            _, batch = self.cache.popitem()
            _give_back(batch)

    def compact(self) -> None:
        for batch in list(self.cache.values()):
            batch.compact()
