"""Shared helpers for the golden-fixture tests (fixtures: tests/golden/*.npz, made by tests/golden/make_fixtures.py
from the reference's CPU causal_lm path)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# The fixtures are drawn until the reference decides every token by >= 0.8 logits (Llama) / 0.2 (GPT-BigCode): more
# than twice the fp16 logit tolerance of the GPU tests, so fp16 runs must reproduce every id exactly.  bf16 runs
# (tolerance 2.5 / 0.6) cannot be held to that: there a row may take another id only if the reference itself prefers its
# own by less than 2 x the tolerance, and the tests report how often that happened (TIE_USES).
TIE_USES = {}


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    steps = []
    i = 0
    while f"s{i}_ids" in z.files:
        steps.append({
            "ids": z[f"s{i}_ids"], "request_ids": z[f"s{i}_request_ids"], "logprobs": z[f"s{i}_logprobs"],
            "ranks": z[f"s{i}_ranks"], "logits": z[f"s{i}_logits"],
            "top": json.loads(bytes(z[f"s{i}_top"]).decode()),
        })
        i += 1
    return meta, steps


class FixtureTokenizer:
    """'t17 t203 ...' -> [17, 203, ...]: the WordLevel tokenizer the fixtures were generated with."""

    def __init__(self, vocab_size):
        self.vocab_size = vocab_size
        self.pad_token_id = 0
        self.bos_token_id = 1
        self.eos_token_id = 2
        self.add_bos_token = False
        self.probe_word = "t5"  # a word of this vocabulary, for the shard's synthetic start-up prefills

    def __call__(self, texts, truncation=True, max_length=None, return_token_type_ids=False, **kw):
        out = []
        for t in texts:
            ids = [int(w[1:]) for w in t.split()]
            if truncation and max_length is not None and len(ids) > max_length:
                ids = ids[-max_length:]
            out.append(ids)
        return {"input_ids": out}


def prompt_text(ids):
    return " ".join(f"t{i}" for i in ids)


def check_ids(got_ids, step, what, tie_margin=None):
    """Token ids must equal the reference's.  tie_margin (bf16 runs only): a row may differ if the reference prefers its
    own id over the one produced by less than tie_margin; returns how many rows used that, and records it."""
    lg = step["logits"]
    used = 0
    for r, (g, w) in enumerate(zip(got_ids, step["ids"])):
        if int(g) == int(w):
            continue
        gap = float(lg[r, int(w)] - lg[r, int(g)])
        assert tie_margin is not None and gap < tie_margin, \
            f"{what}: row {r} token {int(g)} != reference {int(w)} (the reference prefers its own by {gap:.3f})"
        used += 1
    if used:
        TIE_USES[what] = used
    return used
