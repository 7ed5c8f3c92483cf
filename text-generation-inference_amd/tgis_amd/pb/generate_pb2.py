"""`generate.v1` protobuf messages of the shard RPC surface, built at import time without protoc.

The wire contract is the reference's internal router<->shard protocol (proto/generate.proto:1-224):
package `generate.v1`, service `TextGenerationService` with ServiceDiscovery / ClearCache / ModelInfo /
Prefill / NextToken / PruneBatch / PrefixLookup / Health.  This image has neither `protoc` nor
`grpc_tools`, so the FileDescriptorProto is assembled here from a compact field table (name, number,
type) — field numbers and types are what make the Rust router interoperate, and they are asserted against
the reference's .proto in tests/test_pb_contract.py when /root/reference is present.
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto

_SCALAR = {
    "float": _F.TYPE_FLOAT, "uint32": _F.TYPE_UINT32, "uint64": _F.TYPE_UINT64, "bool": _F.TYPE_BOOL,
    "string": _F.TYPE_STRING,
}

# message -> list of (field name, number, type, label) ; label: "" | "repeated" | "optional"
# nested messages/enums are written "Outer.Inner".
SCHEMA = {
    "HealthRequest": [],
    "HealthResponse": [],
    "ServiceDiscoveryRequest": [],
    "ServiceDiscoveryResponse": [("urls", 1, "string", "repeated")],
    "ClearCacheRequest": [],
    "ClearCacheResponse": [],
    "ModelInfoRequest": [],
    "MemoryScalingModel": [
        ("prefill_linear_coef0", 1, "float", ""), ("prefill_quadratic_coef0", 2, "float", ""),
        ("prefill_quadratic_coef1", 3, "float", ""), ("nexttoken_linear_coef0", 4, "float", ""),
        ("nexttoken_linear_coef1", 5, "float", ""), ("weight_limit", 6, "uint64", ""),
    ],
    "ModelInfoResponse": [
        ("model_type", 1, "enum:ModelInfoResponse.ModelType", ""), ("eos_token", 2, "uint32", ""),
        ("batch_padding", 3, "bool", ""), ("memory_scaling_model", 4, "msg:MemoryScalingModel", ""),
    ],
    "NextTokenChooserParameters.LengthPenalty": [("start_index", 1, "uint32", ""), ("decay_factor", 2, "float", "")],
    "NextTokenChooserParameters": [
        ("temperature", 1, "float", ""), ("top_k", 2, "uint32", ""), ("top_p", 3, "float", ""),
        ("typical_p", 4, "float", ""), ("min_new_tokens", 100, "uint32", ""), ("seed", 101, "uint64", "optional"),
        ("repetition_penalty", 102, "float", "optional"),
        ("length_penalty", 103, "msg:NextTokenChooserParameters.LengthPenalty", "optional"),
    ],
    "RequestedDetails": [
        ("input_toks", 1, "bool", ""), ("logprobs", 2, "bool", ""), ("ranks", 3, "bool", ""),
        ("top_n_toks", 4, "uint32", ""),
    ],
    "Request": [
        ("id", 1, "uint64", ""), ("prefix_id", 2, "string", ""), ("inputs", 3, "string", ""),
        ("input_length", 4, "uint32", ""), ("truncate", 5, "bool", ""), ("max_output_length", 6, "uint32", ""),
        ("parameters", 7, "msg:NextTokenChooserParameters", ""), ("stream_response", 100, "bool", ""),
        ("details", 101, "msg:RequestedDetails", ""),
    ],
    "StopSequence": [("tokens", 1, "uint32", "repeated")],
    "Batch": [("id", 1, "uint64", ""), ("requests", 2, "msg:Request", "repeated"), ("total_tokens", 3, "uint32", "")],
    "TopToken": [("token_id", 1, "uint32", ""), ("logprob", 2, "float", "")],
    "Token": [
        ("request_id", 1, "uint64", ""), ("token_id", 2, "uint32", ""), ("logprob", 3, "float", ""),
        ("rank", 4, "uint32", ""), ("top_tokens", 5, "msg:TopToken", "repeated"),
    ],
    "GenerateError": [("request_id", 1, "uint64", ""), ("message", 2, "string", "")],
    "InputTokens": [("request_id", 1, "uint64", ""), ("tokens", 2, "msg:Token", "repeated")],
    "PrefillRequest": [("batch", 1, "msg:Batch", ""), ("to_prune", 2, "msg:CachedBatch", "repeated")],
    "GenerateResult": [
        ("output_tokens", 1, "msg:Token", "repeated"), ("errors", 2, "msg:GenerateError", "repeated"),
        ("batch_id", 3, "uint64", ""), ("forward_time_ns", 4, "uint64", ""),
    ],
    "PrefillResponse": [("result", 1, "msg:GenerateResult", ""), ("input_tokens", 2, "msg:InputTokens", "repeated")],
    "RequestsStatus": [("completed_ids", 3, "uint64", "repeated")],
    "CachedBatch": [("batch_id", 1, "uint64", ""), ("status", 2, "msg:RequestsStatus", "optional")],
    "NextTokenRequest": [("batches", 1, "msg:CachedBatch", "repeated")],
    "NextTokenResponse": [("result", 1, "msg:GenerateResult", "optional")],
    "PruneBatchRequest": [("batch", 1, "msg:CachedBatch", "")],
    "PruneBatchResponse": [("batch_id", 1, "uint64", "optional")],
    "PrefixLookupRequest": [("prefix_id", 1, "string", "")],
    "PrefixLookupResponse": [("prefix_length", 1, "uint32", "")],
}
ENUMS = {"ModelInfoResponse.ModelType": [("CAUSAL_LM", 0), ("SEQ2SEQ_LM", 1)]}

# rpc name -> (request message, response message)
SERVICE_NAME = "TextGenerationService"
RPCS = {
    "ServiceDiscovery": ("ServiceDiscoveryRequest", "ServiceDiscoveryResponse"),
    "ClearCache": ("ClearCacheRequest", "ClearCacheResponse"),
    "ModelInfo": ("ModelInfoRequest", "ModelInfoResponse"),
    "Prefill": ("PrefillRequest", "PrefillResponse"),
    "NextToken": ("NextTokenRequest", "NextTokenResponse"),
    "PruneBatch": ("PruneBatchRequest", "PruneBatchResponse"),
    "PrefixLookup": ("PrefixLookupRequest", "PrefixLookupResponse"),
    "Health": ("HealthRequest", "HealthResponse"),
}
PACKAGE = "generate.v1"


def _build_file() -> descriptor_pb2.FileDescriptorProto:
    fdp = descriptor_pb2.FileDescriptorProto(name="generate.proto", package=PACKAGE, syntax="proto3")
    protos = {}

    def get_msg(full: str):
        if full in protos:
            return protos[full]
        if "." in full:
            outer, inner = full.rsplit(".", 1)
            m = get_msg(outer).nested_type.add(name=inner)
        else:
            m = fdp.message_type.add(name=full)
        protos[full] = m
        return m

    for full in SCHEMA:
        get_msg(full)
    for full, values in ENUMS.items():
        outer, inner = full.rsplit(".", 1)
        e = get_msg(outer).enum_type.add(name=inner)
        for vname, num in values:
            e.value.add(name=vname, number=num)
    for full, fields in SCHEMA.items():
        m = protos[full]
        for fname, num, ftype, label in fields:
            f = m.field.add(name=fname, number=num)
            f.label = _F.LABEL_REPEATED if label == "repeated" else _F.LABEL_OPTIONAL
            if ftype.startswith("msg:"):
                f.type = _F.TYPE_MESSAGE
                f.type_name = f".{PACKAGE}.{ftype[4:]}"
            elif ftype.startswith("enum:"):
                f.type = _F.TYPE_ENUM
                f.type_name = f".{PACKAGE}.{ftype[5:]}"
            else:
                f.type = _SCALAR[ftype]
            if label == "optional":  # proto3 explicit presence = synthetic oneof
                f.proto3_optional = True
                f.oneof_index = len(m.oneof_decl)
                m.oneof_decl.add(name=f"_{fname}")
    svc = fdp.service.add(name=SERVICE_NAME)
    for rpc, (req, resp) in RPCS.items():
        svc.method.add(name=rpc, input_type=f".{PACKAGE}.{req}", output_type=f".{PACKAGE}.{resp}")
    return fdp


_pool = descriptor_pool.DescriptorPool()
FILE_DESCRIPTOR_PROTO = _build_file()
DESCRIPTOR = _pool.Add(FILE_DESCRIPTOR_PROTO) if hasattr(_pool, "Add") and False else None
if DESCRIPTOR is None:
    _pool.AddSerializedFile(FILE_DESCRIPTOR_PROTO.SerializeToString())
    DESCRIPTOR = _pool.FindFileByName("generate.proto")

for _name in SCHEMA:
    if "." not in _name:
        globals()[_name] = message_factory.GetMessageClass(_pool.FindMessageTypeByName(f"{PACKAGE}.{_name}"))

__all__ = [n for n in SCHEMA if "." not in n] + ["DESCRIPTOR", "SERVICE_NAME", "RPCS", "PACKAGE"]
