#!/usr/bin/env python
"""Golden vectors for the next-token chooser, produced by the REFERENCE's own
`text_generation_server.utils.tokens.HeterogeneousNextTokenChooser` (with its Heterogeneous* processors) run
unmodified on CPU in this container.  Needs /root/reference; the output (tests/golden/chooser_reference.npz) is
committed, this script is how it was made.  Same harness-side shims as make_fixtures.py, no edits to reference files.

    python tests/golden/make_chooser_fixture.py

Per case: the request parameters (as the generate.v1.NextTokenChooserParameters fields), fp32 logits [B, V], the ids
seen so far [B, L], and for each of two consecutive calls the reference's outputs: warped scores [B, V] (-inf where
filtered), next ids of the GREEDY rows (sampled rows depend on the reference's torch generator and are not compared)
and log_softmax of the warped scores."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures  # noqa: E402  (sets sys.path for the repo; install_shims gives access to the reference)

CASES = {
    # name: list of per-request parameter dicts
    "warpers": [dict(temperature=0.8, top_k=50, top_p=0.9, seed=1), dict(temperature=1.3, typical_p=0.8, seed=2),
                dict(temperature=0.5, top_p=0.5, seed=3), dict(temperature=1.0, top_k=5, seed=4)],
    "mixed": [dict(temperature=0.7, top_k=40, seed=5, repetition_penalty=1.2), dict(repetition_penalty=1.3),
              dict(), dict(temperature=1.1, top_p=0.95, typical_p=0.9, seed=6, repetition_penalty=1.05)],
    "greedy_eos": [dict(min_new_tokens=2), dict(length_penalty=(0, 1.5)), dict(repetition_penalty=2.0, min_new_tokens=1),
                   dict(length_penalty=(1, 1.2), repetition_penalty=1.1)],
}
V, L = 512, 24
EOS, PAD = 2, 2  # eos doubles as pad: the repetition penalty must leave it alone (tokens.py:185-190)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        make_fixtures.install_shims(tmp)
        from text_generation_server.pb import generate_pb2 as pb
        from text_generation_server.utils.tokens import HeterogeneousNextTokenChooser

        out = {}
        for name, rows in CASES.items():
            params = []
            for r in rows:
                p = pb.NextTokenChooserParameters()
                for k, v in r.items():
                    if k == "length_penalty":
                        p.length_penalty.start_index, p.length_penalty.decay_factor = v
                    else:
                        setattr(p, k, v)
                params.append(p)
            g = torch.Generator().manual_seed(len(name))
            B = len(rows)
            ids = torch.randint(3, V, (B, L), generator=g)
            ids[:, :4] = PAD  # left padding, as all_input_ids_tensor has it
            chooser = HeterogeneousNextTokenChooser.from_pb(params, EOS, PAD, [True] * B, torch.float32, "cpu")
            out[f"{name}.ids"] = ids.numpy()
            for i, p in enumerate(params):  # serialized generate.v1.NextTokenChooserParameters, one per request
                out[f"{name}.params.{i}"] = np.frombuffer(p.SerializeToString(), dtype=np.uint8)
            for step in range(2):
                logits = torch.randn(B, V, generator=g) * 3
                next_ids, scores, logprobs = chooser(ids, logits.clone())
                out[f"{name}.{step}.logits"] = logits.numpy()
                out[f"{name}.{step}.scores"] = scores.numpy()
                out[f"{name}.{step}.next_ids"] = next_ids.numpy()
                out[f"{name}.{step}.logprobs"] = logprobs.numpy()
                out[f"{name}.{step}.greedy_rows"] = np.array([r.get("temperature", 0.0) == 0.0 for r in rows])
        path = os.path.join(HERE, "chooser_reference.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
