"""CPU: the protoc-free generate_pb2 speaks the reference's wire format (proto/generate.proto)."""
import os
import re

import pytest

from tgis_amd.pb import generate_pb2 as pb

REF_PROTO = "/root/reference/proto/generate.proto"


def test_roundtrip_and_presence():
    r = pb.Request(id=3, inputs="hi", input_length=2, max_output_length=5, truncate=True)
    r.parameters.seed = 7
    r.parameters.length_penalty.start_index = 4
    r.details.logprobs = True
    b = pb.Batch(id=1, requests=[r], total_tokens=2)
    b2 = pb.Batch.FromString(b.SerializeToString())
    assert b2.requests[0].parameters.HasField("seed") and b2.requests[0].parameters.seed == 7
    assert not b2.requests[0].parameters.HasField("repetition_penalty")
    assert b2.requests[0].parameters.HasField("length_penalty")
    assert not pb.NextTokenResponse().HasField("result")
    assert not pb.CachedBatch(batch_id=1).HasField("status")
    cb = pb.CachedBatch(batch_id=1)
    cb.status.completed_ids.extend([1, 5])
    assert pb.CachedBatch.FromString(cb.SerializeToString()).status.completed_ids == [1, 5]
    # hand-checked wire bytes: field 100 (min_new_tokens) varint tag = (100<<3)|0 = 800 -> 0xa0 0x06
    assert pb.NextTokenChooserParameters(min_new_tokens=3).SerializeToString() == b"\xa0\x06\x03"
    # healthcheck batch id 2**64-1 must be representable (server.py:37)
    assert pb.Batch.FromString(pb.Batch(id=2**64 - 1).SerializeToString()).id == 2**64 - 1


@pytest.mark.skipif(not os.path.exists(REF_PROTO), reason="reference checkout not present (GPU box)")
def test_field_numbers_match_reference_proto():
    """Every message/field/number/label of the reference .proto equals our descriptor table."""
    text = re.sub(r"//.*", "", open(REF_PROTO).read())
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9.]*|\d+|[{}=;()]", text)
    ref = {}
    stack = []
    i = 0
    while i < len(toks):
        t = toks[i]
        if t in ("message", "enum") and toks[i + 2] == "{":
            stack.append((t, ".".join([s[1] for s in stack if s[0] == "message"] + [toks[i + 1]])
                          if t == "message" else toks[i + 1]))
            if t == "message":
                ref.setdefault(stack[-1][1], [])
            i += 3
            continue
        if t == "service":
            depth = 0
            while True:  # skip the service block
                if toks[i] == "{":
                    depth += 1
                if toks[i] == "}":
                    depth -= 1
                    if depth == 0:
                        break
                i += 1
            i += 1
            continue
        if t == "}":
            stack.pop()
            i += 1
            continue
        if stack and stack[-1][0] == "message" and t not in ("message", "enum", "{", ";"):
            label = ""
            if t in ("repeated", "optional"):
                label = t
                i += 1
            ftype, fname, _eq, num = toks[i], toks[i + 1], toks[i + 2], toks[i + 3]
            ref[stack[-1][1]].append((fname, int(num), ftype.split(".")[-1], label))
            i += 5
            continue
        i += 1
    assert set(ref) == set(pb.SCHEMA), set(ref) ^ set(pb.SCHEMA)
    for msg, fields in ref.items():
        ours = {(n, num, t.split(":")[-1].split(".")[-1], lab) for n, num, t, lab in pb.SCHEMA[msg]}
        assert ours == set(fields), (msg, ours ^ set(fields))
    rpcs = re.findall(r"rpc\s+(\w+)\s*\((\w+)\)\s*returns\s*\((\w+)\)", text)
    assert {r[0]: (r[1], r[2]) for r in rpcs} == pb.RPCS
