"""Per-token result records and their protobuf form.

The same three records the reference's servicer consumes (utils/token_types.py:1-56): a candidate token with its
log-probability (`TopToken`), one generated or prompt token with optional details (`TokenInfo`), and the per-request
list of prompt tokens (`InputTokens`); each knows how to turn itself into its generate.v1 message.  Plain slotted
classes: thousands are created per second on the host side of the decode loop."""
from typing import List, Optional

from tgis_amd.pb import generate_pb2 as pb2


class TopToken:
    """Sorts by log-probability; on equal log-probabilities the LOWER token id ranks higher, the tie-break of a greedy
    argmax (`sorted(tops, reverse=True)` puts the best candidate first)."""
    __slots__ = ("token_id", "logprob")

    def __init__(self, token_id: int, logprob: float = 0.0):
        self.token_id = token_id
        self.logprob = logprob

    def _rank_key(self):
        return (self.logprob, -self.token_id)

    def __eq__(self, other) -> bool:
        return isinstance(other, TopToken) and self._rank_key() == other._rank_key()

    def __lt__(self, other: "TopToken") -> bool:
        return self._rank_key() < other._rank_key()

    def __gt__(self, other: "TopToken") -> bool:
        return self._rank_key() > other._rank_key()

    def __le__(self, other: "TopToken") -> bool:
        return self._rank_key() <= other._rank_key()

    def __ge__(self, other: "TopToken") -> bool:
        return self._rank_key() >= other._rank_key()

    def __hash__(self):
        return hash(self._rank_key())

    def __repr__(self) -> str:
        return f"TopToken(token_id={self.token_id}, logprob={self.logprob})"

    def to_pb(self) -> pb2.TopToken:
        return pb2.TopToken(token_id=self.token_id, logprob=self.logprob)


class TokenInfo:
    __slots__ = ("token_id", "request_id", "logprob", "rank", "top_tokens")

    def __init__(self, token_id: int = 0, request_id: int = 0, logprob: float = 0.0, rank: int = 0,
                 top_tokens: Optional[List[TopToken]] = None):
        self.token_id = token_id
        self.request_id = request_id  # stays 0 for prompt tokens
        self.logprob = logprob
        self.rank = rank
        self.top_tokens = top_tokens

    def __eq__(self, other) -> bool:
        return isinstance(other, TokenInfo) and all(getattr(self, f) == getattr(other, f) for f in self.__slots__)

    def __repr__(self) -> str:
        return "TokenInfo(" + ", ".join(f"{f}={getattr(self, f)!r}" for f in self.__slots__) + ")"

    def to_pb(self) -> pb2.Token:
        msg = pb2.Token(request_id=self.request_id, token_id=self.token_id, logprob=self.logprob, rank=self.rank)
        if self.top_tokens:
            msg.top_tokens.extend(t.to_pb() for t in self.top_tokens)
        return msg


class InputTokens:
    __slots__ = ("request_id", "tokens")

    def __init__(self, request_id: int, tokens: Optional[List[TokenInfo]] = None):
        self.request_id = request_id
        self.tokens = tokens if tokens is not None else []

    def __repr__(self) -> str:
        return f"InputTokens(request_id={self.request_id}, tokens={self.tokens!r})"

    def to_pb(self) -> pb2.InputTokens:
        msg = pb2.InputTokens(request_id=self.request_id)
        msg.tokens.extend(t.to_pb() for t in self.tokens)
        return msg
