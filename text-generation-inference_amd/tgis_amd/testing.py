"""Helpers shared by bench.py, smoke() and the tests: a deterministic tokenizer that needs no files and a
request-batch builder following the reference's own synthetic generator
(utils/memory_characterizer.py:219-240: fixed input_length, truncate=True, greedy, max_output_length=N)."""
import zlib
from typing import List

import numpy as np

from tgis_amd.pb import generate_pb2


class SyntheticTokenizer:
    """Whitespace tokenizer with seeded ids: word j of a text maps to an id in [3, vocab) drawn from a generator
    seeded by crc32(text).  Implements the slice of the HF tokenizer interface FlashCausalLMBatch.from_pb uses."""

    def __init__(self, vocab_size: int, eos_token_id: int = 2, pad_token_id: int = 0, bos_token_id: int = 1):
        self.vocab_size = vocab_size
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.add_bos_token = False
        self.padding_side = "left"
        self.truncation_side = "left"

    def __call__(self, texts: List[str], truncation=True, max_length=None, return_token_type_ids=False, **kw):
        out = []
        for t in texts:
            n = len(t.split())
            rng = np.random.default_rng(zlib.crc32(t.encode()))
            ids = rng.integers(3, self.vocab_size, size=n).tolist()
            if truncation and max_length is not None and n > max_length:
                ids = ids[-max_length:]  # truncation_side = left
            out.append(ids)
        return {"input_ids": out}


def make_batch_pb(input_lengths: List[int], max_new: int, batch_id: int = 0, first_request_id: int = 0,
                  logprobs: bool = False, seed_text: str = "w") -> generate_pb2.Batch:
    reqs = []
    for i, l in enumerate(input_lengths):
        text = " ".join(f"{seed_text}{first_request_id + i}_{j}" for j in range(l))
        r = generate_pb2.Request(id=first_request_id + i, inputs=text, input_length=l, truncate=True,
                                 max_output_length=max_new)
        r.details.logprobs = logprobs
        reqs.append(r)
    return generate_pb2.Batch(id=batch_id, requests=reqs, total_tokens=sum(input_lengths))
