"""Per-token result records and their protobuf form (mirrors utils/token_types.py:1-56 of the reference)."""
from dataclasses import dataclass, field
from functools import total_ordering
from typing import List, Optional

from tgis_amd.pb import generate_pb2


@total_ordering
@dataclass(eq=True)
class TopToken:
    token_id: int
    logprob: float = 0.0

    def __gt__(self, other: "TopToken") -> bool:
        # equal logprobs: the LOWER token id ranks higher, the same tie-break as greedy argmax
        if self.logprob != other.logprob:
            return self.logprob > other.logprob
        return self.token_id < other.token_id

    def to_pb(self) -> generate_pb2.TopToken:
        return generate_pb2.TopToken(token_id=self.token_id, logprob=self.logprob)


@dataclass
class TokenInfo:
    token_id: int
    request_id: int = 0  # unset for input tokens
    logprob: float = 0.0
    rank: int = 0
    top_tokens: Optional[List[TopToken]] = None

    def to_pb(self) -> generate_pb2.Token:
        tops = [t.to_pb() for t in self.top_tokens] if self.top_tokens is not None else None
        return generate_pb2.Token(request_id=self.request_id, token_id=self.token_id, logprob=self.logprob,
                                  rank=self.rank, top_tokens=tops)


@dataclass
class InputTokens:
    request_id: int
    tokens: List[TokenInfo] = field(default_factory=list)

    def to_pb(self) -> generate_pb2.InputTokens:
        return generate_pb2.InputTokens(request_id=self.request_id, tokens=[t.to_pb() for t in self.tokens])
