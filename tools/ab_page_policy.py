"""A/B of the page allocation policy on one box: OLD=1 restores rounds 1-4 (fixed pseudo-random order, a sequence's pages
taken together) by patching the two methods, then runs bench.py with the remaining arguments."""
import os
import runpy
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "text-generation-inference_amd"))
if os.getenv("OLD") == "1":
    import torch
    from tgis_amd.models import flash_causal_lm as fcl
    from tgis_amd.utils import kv_cache as kvc

    _init = kvc.PagedKVCache.__init__

    def init(self, *a, **k):
        _init(self, *a, **k)
        self._free = torch.randperm(self.num_pages, generator=torch.Generator().manual_seed(0x5eed)).tolist()

    def alloc(self, n):
        if n > len(self._free):
            raise kvc.OutOfPages("KV cache exhausted")
        out = self._free[-n:][::-1] if n else []
        del self._free[len(self._free) - n:]
        return out

    def free(self, pages):
        self._free.extend(reversed(pages))

    def allocate_pages(self, kv_cache):
        need = [kvc.PagedKVCache.pages_for(n + 1) for n in self.input_lengths]
        flat = kv_cache.alloc(sum(need))
        self.kv_cache = kv_cache
        self.pages, o = [], 0
        for n in need:
            self.pages.append(flat[o:o + n])
            o += n
        self._rebuild_block_tables()

    kvc.PagedKVCache.__init__, kvc.PagedKVCache.alloc, kvc.PagedKVCache.free = init, alloc, free
    fcl.FlashCausalLMBatch.allocate_pages = allocate_pages
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
