"""Shared helpers for the golden-fixture tests (fixtures: tests/golden/*.npz, made by tests/golden/make_fixtures.py
from the reference's CPU causal_lm path)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# rows whose reference top-2 logit margin is below this are only required to pick one of the top two ids:
# fp16 kernels cannot be asked to break a near-tie the way an fp32 run happened to
TIE_MARGIN = 0.75


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


def check_ids(got_ids, step, what):
    """Token ids must equal the reference's; rows the reference itself decided by < TIE_MARGIN may take its runner-up."""
    lg = step["logits"]
    order = np.argsort(-lg, axis=1)
    margin = lg[np.arange(len(lg)), order[:, 0]] - lg[np.arange(len(lg)), order[:, 1]]
    for r, (g, w) in enumerate(zip(got_ids, step["ids"])):
        if int(g) == int(w):
            continue
        assert margin[r] < TIE_MARGIN and int(g) == int(order[r, 1]), \
            f"{what}: row {r} token {int(g)} != reference {int(w)} (reference margin {margin[r]:.3f})"
