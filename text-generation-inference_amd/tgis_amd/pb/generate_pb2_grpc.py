"""gRPC glue for generate.v1.TextGenerationService without grpc_tools: generic method handlers built from the RPC
table of generate_pb2 (server side) and a matching stub (client side, used by tests and tools)."""
import grpc

from tgis_amd.pb import generate_pb2 as pb

_FULL = f"{pb.PACKAGE}.{pb.SERVICE_NAME}"


class TextGenerationServiceServicer:
    """Base class; override the RPCs you implement (PruneBatch is declared but unimplemented in the reference too)."""

    async def _unimplemented(self, request, context):
        await context.abort(grpc.StatusCode.UNIMPLEMENTED, "Method not implemented!")


for _rpc in pb.RPCS:
    setattr(TextGenerationServiceServicer, _rpc, TextGenerationServiceServicer._unimplemented)


def add_TextGenerationServiceServicer_to_server(servicer, server):
    handlers = {}
    for rpc, (req, resp) in pb.RPCS.items():
        handlers[rpc] = grpc.unary_unary_rpc_method_handler(
            getattr(servicer, rpc),
            request_deserializer=getattr(pb, req).FromString,
            response_serializer=getattr(pb, resp).SerializeToString)
    server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler(_FULL, handlers),))


class TextGenerationServiceStub:
    def __init__(self, channel):
        for rpc, (req, resp) in pb.RPCS.items():
            setattr(self, rpc, channel.unary_unary(
                f"/{_FULL}/{rpc}", request_serializer=getattr(pb, req).SerializeToString,
                response_deserializer=getattr(pb, resp).FromString))
