"""Linear / embedding / head building blocks on the HIP kernels, with Megatron-style tensor parallelism.

Mirrors utils/layers.py of the reference: `FastLinear` (:104-111), the `get_linear` quant registry
(:172-203), `TensorParallelHead` (:215-277), `TensorParallelColumnLinear` (:280-297),
`TensorParallelRowLinear` (all-reduce, :300-322), `TensorParallelEmbedding` (:325-357) and
`PositionRotaryEmbedding` (:406-490); `Ex4bitLinearV2` mirrors utils/gptq/exllamav2.py:100-144.
Every forward runs a kernel of libtgis_hip.so (decode-sized M) or, for prefill-sized M, a library GEMM
on the same device; there is no CPU path."""
import math
from typing import List, Optional

import torch
import torch.distributed
from torch.nn import functional as F

from tgis_amd import native
from tgis_amd.utils.graph_segments import collective

import os

# rows up to which the dense weight-streaming MFMA kernel is used (one pass over the weights per 32 / 64 rows); above,
# hipBLASLt.  GPTQ linears ask the library how far its fused kernels reach (GptqWeight.fused_rows: 64-row streaming passes
# to 256 rows, then the tall fused kernel to ~3k rows; beyond that dequantise once + hipBLASLt, exllamav2.py:87).
SKINNY_MAX_M = int(os.getenv("TGIS_SKINNY_MAX_M", "256"))
# fuse the split-K reduce of decode-sized GPTQ GEMMs into the consumer kernel (rmsnorm / rope+KV write)
DEFER_REDUCE = os.getenv("TGIS_DEFER_REDUCE", "true").lower() not in ("0", "false")
# decode batches of up to 32 rows: rotary embedding + cache write in the epilogue of the int4 qkv GEMM (one launch
# instead of GEMM + tgis_rope_kv_write, no split-K slabs).  Costs a second image of the qkv weights.
FUSED_ROPE_GEMM = os.getenv("TGIS_FUSED_ROPE_GEMM", "true").lower() not in ("0", "false")
# GELU applied by the dense decode GEMM that produces its operand (TGIS_FUSED_GELU_GEMM=false: a launch of its own)
FUSED_GELU_GEMM = os.getenv("TGIS_FUSED_GELU_GEMM", "true").lower() not in ("0", "false")

_WORKSPACES = {}


def workspace(device) -> native.Workspace:
    key = str(device)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = native.Workspace(64 << 20, device)
    return ws


class FastLinear:
    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        self.weight = weight
        self.bias = bias
        self.prepared = native.DenseWeight(weight)
        self.out_features, self.in_features = weight.shape
        self._gate_up = False
        self._rope_heads = None
        self.rope_handle: Optional[native.DenseWeight] = None

    @property
    def gate_up(self) -> bool:
        return self._gate_up

    @gate_up.setter
    def gate_up(self, on: bool):
        """Set by LlamaMLP on the fused [gate | up] projection: SiLU(gate)*up then runs in the GEMM epilogue (the image
        is rebuilt with interleaved pairs; the checkpoint-layout weight stays for the large-M library path)."""
        if bool(on) != self._gate_up:
            self._gate_up = bool(on)
            self.prepared = native.DenseWeight(self.weight, gate_up=self._gate_up)

    @property
    def rope_heads(self):
        return self._rope_heads

    @rope_heads.setter
    def rope_heads(self, heads):
        """Set by FlashLlamaAttention on the fused qkv projection: (H, Hkv, D) of this shard.  Builds the rope image (a
        second copy of the weight with rotation pairs inside each tile) for native.dense_gemm_rope."""
        self._rope_heads = heads
        self.rope_handle = None
        if heads is not None and FUSED_ROPE_GEMM and heads[2] % 32 == 0:
            H, Hkv, D = heads
            # the image is a second copy of the weight: only build it where the fused launch will serve some decode batch
            # (tgis_dense_rope_ok asks for >= TGIS_ROPE_MIN_BLOCKS workgroups in the unsplit plan — TinyLlama has 40, a
            # 70B shard at TP = 8 has 20), as Ex4bitLinearV2.post_init does for the int4 image
            if any(native.rope_gemm_ok(m, self.prepared, D) for m in (1, 32, 64)):
                self.rope_handle = native.DenseWeight(self.weight, rope=(D, H + Hkv))

    def forward(self, x: torch.Tensor, act: int = 0, out_f32: bool = False, partial: bool = False,
                gelu: Optional[bool] = None) -> torch.Tensor:
        """`gelu` (None / False: exact erf form / True: tanh approximation): the activation behind this projection
        (santacoder's `c_fc`, flash_santacoder_modeling.py:303-305) — applied where the decode GEMM finishes its output
        instead of by a launch of its own; same bits as the two steps."""
        if gelu is not None:
            assert act == 0 and not out_f32 and not partial and not self._gate_up
            if x.shape[0] <= SKINNY_MAX_M and FUSED_GELU_GEMM:
                return native.dense_gemm(x, self.prepared, workspace(x.device), bias=self.bias, act=5 if gelu else 4)
            return native.gelu(self.forward(x), gelu)
        if self._gate_up:
            # output is the activated [M, I] tensor
            if x.shape[0] <= SKINNY_MAX_M:
                return native.dense_gemm(x, self.prepared, workspace(x.device), bias=self.bias, act=2)
            return native.act_mul(F.linear(x, self.weight, self.bias), self.out_features // 2)
        if partial and x.shape[0] <= native.PARTIAL_MAX_M and DEFER_REDUCE and not out_f32:
            # the split-K sum (and the bias) is finished by the consumer kernel (add+norm / rope): no reduce launch
            return native.dense_gemm_partial(x, self.prepared, bias=self.bias, act=act)
        if x.shape[0] <= SKINNY_MAX_M:
            return native.dense_gemm(x, self.prepared, workspace(x.device), bias=self.bias, out_f32=out_f32, act=act)
        if act:
            x = native.act_mul(x, self.in_features)
        out = F.linear(x, self.weight, self.bias)
        return out.float() if out_f32 else out

    __call__ = forward


class Ex4bitLinearV2:
    """4-bit GPTQ linear.  Construction keeps the checkpoint tensors; `post_init` repacks them for the
    kernels (the reference defers this the same way, server.py:347-354)."""
    _temp_dq = {}

    def __init__(self, qweight, qzeros, scales, g_idx, bias, bits, groupsize):
        assert bits == 4
        self.device = qweight.device
        self.qweight, self.qzeros, self.scales = qweight, qzeros, scales
        # ("perm", gather index, x columns): an explicit activation permutation (act-order row shards, utils/weights.py)
        self.g_idx = g_idx if isinstance(g_idx, tuple) else (g_idx.cpu() if g_idx is not None else None)
        self.bias = bias
        self.bits, self.groupsize = bits, groupsize
        self.height = qweight.shape[0] * 8
        self.width = qweight.shape[1]
        # (no device check here: native.GptqWeight refuses tensors that are not on the GPU — server.py:290-291)
        assert self.height % 32 == 0 and self.width % 32 == 0
        self.q_handle: Optional[native.GptqWeight] = None
        # set by LlamaMLP on the fused [gate | up] projection: SiLU(gate)*up runs in the GEMM epilogue
        self.gate_up = False
        # set by FlashLlamaAttention on the fused qkv projection: (H, Hkv, D) of this shard -> post_init also builds the
        # rope image (a second copy of the int4 matrix with rotation pairs inside each tile, +K*N/2 bytes) for
        # native.gptq_gemm_rope, the decode launch that rotates q / k and writes the cache in its epilogue
        self.rope_heads = None
        self.rope_handle: Optional[native.GptqWeight] = None
        self._fused = {}  # act -> rows up to which the library's fused kernels are used

    def post_init(self):
        self.q_handle = native.GptqWeight(self.qweight, self.qzeros, self.scales, self.g_idx, self.bits,
                                          self.groupsize, gate_up=self.gate_up)
        if self.rope_heads is not None and FUSED_ROPE_GEMM:
            H, Hkv, D = self.rope_heads
            if self.q_handle.perm is None and any(native.gptq_rope_ok(m, self.q_handle, D) for m in (1, 32, 64)):
                self.rope_handle = native.GptqWeight(self.qweight, self.qzeros, self.scales, None, self.bits,
                                                     self.groupsize, rope=(D, H + Hkv))
        self.qweight = self.qzeros = self.scales = None  # the prepared image replaces them

    def _fused_rows(self, act: int) -> int:
        got = self._fused.get(act)
        if got is None:
            got = self._fused[act] = self.q_handle.fused_rows(act)
        return got

    def _dequant_scratch(self) -> torch.Tensor:
        key = (str(self.device), self.height, self.width)
        buf = Ex4bitLinearV2._temp_dq.get(key)
        if buf is None:
            Ex4bitLinearV2._temp_dq.clear()  # one scratch at a time, like exllama's temp_dq
            buf = Ex4bitLinearV2._temp_dq[key] = torch.empty((self.height, self.width), dtype=torch.float16,
                                                             device=self.device)
        return buf

    def wants_fragments(self, rows: int, act: int = 0) -> bool:
        """Should the producer of this linear's operand write it in fragment order (native.FragAct) for `rows` decode
        rows?  act: 0 plain / partial, 2 the SiLU * up image, 3 the rope image (the fused qkv + rotary launch)."""
        if self.q_handle is None:
            self.post_init()
        if act == 2 and not self.gate_up:
            act = 0
        w = self.rope_handle if act == 3 else self.q_handle
        return w is not None and rows <= 64 and native.gptq_fragments_ok(rows, w, act)

    def forward(self, x, act: int = 0, partial: bool = False, out_frag: bool = False):
        """partial=True (decode-sized M only): return native.Partial — the consumer kernel finishes the split-K sum.
        x may be a native.FragAct (decode, <= 32 rows; see wants_fragments); out_frag: the SiLU * up output leaves as one."""
        if self.q_handle is None:
            self.post_init()
        if isinstance(x, native.FragAct):
            if self.gate_up:
                return native.gptq_gemm(x, self.q_handle, workspace(x.device), bias=self.bias, act=2, out_frag=out_frag)
            assert act == 0
            if partial and DEFER_REDUCE:
                return native.gptq_gemm_partial(x, self.q_handle, bias=self.bias)
            return native.gptq_gemm(x, self.q_handle, workspace(x.device), bias=self.bias)
        if self.gate_up:
            # output is the activated [M, I] tensor
            if x.shape[0] <= self._fused_rows(2):
                return native.gptq_gemm(x, self.q_handle, workspace(x.device), bias=self.bias, act=2)
            return native.act_mul(self._large_m(x), self.width // 2)
        if partial and x.shape[0] <= native.PARTIAL_MAX_M and DEFER_REDUCE:
            return native.gptq_gemm_partial(x, self.q_handle, bias=self.bias, act=act)
        if x.shape[0] <= self._fused_rows(act):
            return native.gptq_gemm(x, self.q_handle, workspace(x.device), bias=self.bias, act=act)
        if act:
            x = native.act_mul(x, self.height)
        return self._large_m(x)

    def _large_m(self, x: torch.Tensor) -> torch.Tensor:
        """prefill-sized M: dequantise once into scratch, then a library GEMM (exllamav2.py:87 "M > 50")."""
        if self.q_handle.perm is not None:
            perm = self.q_handle.perm.long()
            if self.q_handle.in_features != self.q_handle.K:  # padded shard: index -1 reads a zero column
                x = torch.nn.functional.pad(x, (0, 1))
                perm = torch.where(perm < 0, torch.full_like(perm, x.shape[1] - 1), perm)
            x = x.index_select(1, perm)
        w = self._dequant_scratch()
        native._check(native.load_library().tgis_gptq_dequant_f16(
            self.q_handle.image.data_ptr(), w.data_ptr(), self.height, self.width, self.q_handle.groups,
            self.q_handle.flags, native._stream()), "tgis_gptq_dequant_f16")
        out = torch.matmul(x, w)
        if self.bias is not None:
            out.add_(self.bias)
        return out

    __call__ = forward


def get_linear(weight, bias, quantize):
    if quantize is None:
        return FastLinear(weight, bias)
    if quantize == "gptq":
        try:
            qweight, qzeros, scales, g_idx, bits, groupsize, _use = weight
        except Exception:
            raise NotImplementedError("The passed weight is not `gptq` compatible, loader needs to be updated.")
        return Ex4bitLinearV2(qweight, qzeros, scales, g_idx, bias, bits, groupsize)
    raise NotImplementedError(f"Quantization `{quantize}` is not implemented yet.")


class SuperLayer:
    def __init__(self, linear):
        self.linear = linear

    def forward(self, x, **kw):
        return self.linear.forward(x, **kw)

    __call__ = forward


class TensorParallelHead(SuperLayer):
    def __init__(self, linear, process_group, should_gather: bool):
        super().__init__(linear)
        self.process_group = process_group
        self.should_gather = should_gather

    @staticmethod
    def load(config, prefix: str, weights):
        if weights.process_group.size() > 1:
            try:
                weight = weights.get_sharded(f"{prefix}.weight", dim=0)
                should_gather = True
            except AssertionError:
                # vocab not divisible by the number of shards: every rank keeps the full head
                weight = weights.get_tensor(f"{prefix}.weight")
                should_gather = False
        else:
            weight = weights.get_tensor(f"{prefix}.weight")
            should_gather = False
        # GPTQ doesn't quantize heads (nor embeddings)
        return TensorParallelHead(get_linear(weight, bias=None, quantize=None), weights.process_group, should_gather)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 logits [T, V]; vocab shards are gathered so every rank holds identical logits."""
        local = self.linear.forward(x, out_f32=True)
        if not self.should_gather:
            return local
        world = self.process_group.size()
        # Every rank contributes its [T, V/tp] block; the gathered [tp, T, V/tp] buffer is re-laid out ONCE into the
        # row-major [T, V] the fused argmax / warper kernels read.  (The reference gathers transposed [V/tp, T] blocks
        # and returns a transposed view, layers.py:244-269 — two transposing copies on this side of the gather.)
        local = local.contiguous()
        T, vp = local.shape
        gathered = torch.empty((world, T, vp), dtype=local.dtype, device=local.device)
        pg = self.process_group
        collective(lambda o=gathered, i=local: torch.distributed.all_gather_into_tensor(o.view(world * T, vp), i, group=pg))
        return gathered.permute(1, 0, 2).reshape(T, world * vp)

    __call__ = forward


class TensorParallelColumnLinear(SuperLayer):
    @classmethod
    def load(cls, config, prefix: str, weights, bias: bool):
        return cls.load_multi(config, [prefix], weights, bias, dim=0)

    @classmethod
    def load_multi(cls, config, prefixes: List[str], weights, bias: bool, dim: int):
        weight = weights.get_multi_weights_col(prefixes, quantize=config.quantize, dim=dim)
        if bias:
            b = [weights.get_sharded(f"{p}.bias", dim=0) for p in prefixes]
            bias = torch.cat(b, dim=dim)
        else:
            bias = None
        return cls(get_linear(weight, bias, config.quantize))


class TensorParallelRowLinear(SuperLayer):
    def __init__(self, linear, process_group):
        super().__init__(linear)
        self.process_group = process_group

    @classmethod
    def load(cls, config, prefix: str, weights, bias: bool):
        weight = weights.get_multi_weights_row(prefix, quantize=config.quantize)
        if bias and weights.process_group.rank() == 0:
            bias = weights.get_tensor(f"{prefix}.bias")  # bias only on the first rank
        else:
            bias = None
        return cls(get_linear(weight, bias, config.quantize), process_group=weights.process_group)

    def forward(self, x: torch.Tensor, **kw) -> torch.Tensor:
        if self.process_group.size() > 1:
            kw.pop("partial", None)  # the all-reduce needs the reduced f16 tensor
            out = self.linear.forward(x, **kw)
            pg = self.process_group
            collective(lambda t=out: torch.distributed.all_reduce(t, group=pg))
            return out
        return self.linear.forward(x, **kw)

    __call__ = forward


class TensorParallelEmbedding:
    def __init__(self, prefix: str, weights, reduce=True):
        self.weight = weights.get_partial_sharded(f"{prefix}.weight", dim=0).contiguous()
        num_embeddings = weights.get_shape(f"{prefix}.weight")[0]
        self.process_group = weights.process_group
        world_size = self.process_group.size()
        rank = self.process_group.rank()
        block_size = num_embeddings // world_size
        self.min_id = rank * block_size
        self.max_id = min(num_embeddings, (rank + 1) * block_size)
        self.reduce = reduce

    def forward(self, input_ids: torch.Tensor, positions=None, pos_table=None) -> torch.Tensor:
        # ids outside [min_id, max_id) read the null (zero) row, then shards are summed (layers.py:346-357)
        out = native.embedding(input_ids, self.weight, positions=positions, pos_table=pos_table,
                               id_offset=self.min_id)
        if self.reduce and self.process_group.size() > 1:
            pg = self.process_group
            collective(lambda t=out: torch.distributed.all_reduce(t, group=pg))
        return out

    __call__ = forward


class PositionRotaryEmbedding:
    """cos/sin caches: inv_freq and freqs in fp32, tables cast to the model dtype (layers.py:419-451)."""

    def __init__(self, inv_freq: torch.Tensor, scaling_factor: float = 1.0):
        self.inv_freq = inv_freq
        self.scaling_factor = scaling_factor
        self._seq_len_cached = 0
        self._cos_cached = None
        self._sin_cached = None
        # tables that were replaced by longer ones: captured decode graphs hold raw pointers into them, and every
        # position such a graph can see lies inside the table it captured, so they only have to stay allocated
        self._retired = []

    @classmethod
    def static(cls, dim, base, device, scaling_factor=1.0):
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, device=device, dtype=torch.float32) / dim))
        return cls(inv_freq, scaling_factor)

    def tables(self, dtype, device, seqlen: int):
        if seqlen > self._seq_len_cached or self._cos_cached.device != device or self._cos_cached.dtype != dtype:
            if self._cos_cached is not None:
                self._retired.append((self._cos_cached, self._sin_cached))
            self._seq_len_cached = seqlen
            t = torch.arange(seqlen, device=device, dtype=self.inv_freq.dtype)
            if self.scaling_factor != 1.0:
                t = t / self.scaling_factor
            freqs = torch.outer(t, self.inv_freq.to(device=t.device))
            self._cos_cached = torch.cos(freqs).to(dtype).contiguous()
            self._sin_cached = torch.sin(freqs).to(dtype).contiguous()
        return self._cos_cached, self._sin_cached
