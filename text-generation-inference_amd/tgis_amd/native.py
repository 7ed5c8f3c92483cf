"""ctypes binding of libtgis_hip.so (include/tgis_hip.h) for torch tensors on an MI355X.

This is the drop-in boundary #5 of SURVEY.md §8(b): where the reference imports CUDA extension modules
(`flash_attn_2_cuda`, `dropout_layer_norm`, `rotary_emb`, `exllamav2_kernels`; utils/flash_attn.py:23,
utils/layers.py:361,403-404, utils/gptq/exllamav2.py:7) this module loads one C-ABI shared library and
passes raw device pointers, sizes and the current HIP stream.  There is no CPU or eager-torch fallback:
if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.getenv("TGIS_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libtgis_hip.so")

F16, BF16 = 0, 1
KV_PAGE_TOKENS = 32

OP_GPTQ_GEMM, OP_ATTN, OP_DENSE_GEMM, OP_NORM, OP_ROPE_KV, OP_ACT, OP_SAMPLE = range(7)

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_f = ctypes.c_float
_vp = ctypes.c_void_p



# name -> (restype, argtypes); must list every symbol declared in include/tgis_hip.h
SIGNATURES = {
    "tgis_version": (ctypes.c_char_p, []),
    "tgis_arch": (ctypes.c_char_p, []),
    "tgis_clear_error": (None, []),
    "tgis_last_error": (ctypes.c_char_p, []),
    "tgis_device_info": (_c_int, [_c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64), ctypes.c_char_p, _c_int]),
    "tgis_timing_enable": (_c_int, [_c_int]),
    "tgis_timing_reset": (_c_int, []),
    "tgis_timing_read": (_c_int, [_c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(ctypes.c_double)]),
    "tgis_gptq_prepared_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "tgis_gptq_prepare": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _vp]),
    "tgis_gptq_gemm_fused_rows": (_c_i64, [_c_i64, _c_i64, _c_int, _c_int]),
    "tgis_gptq_gemm_workspace_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "tgis_gptq_gemm_f16": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                                    _c_int, _vp, _c_i64, _vp]),
    "tgis_gptq_gemm_partial_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "tgis_gptq_gemm_f16_partial": (_c_int, [_vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_i64,
                                            ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64), _vp]),
    "tgis_gptq_dequant_f16": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp]),
    "tgis_dense_gemm_rope": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_i64,
                                      _c_i64, _c_i64, _c_i64, _c_int, _vp]),
    "tgis_gptq_rope_ok": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_i64]),
    "tgis_gptq_fragments_ok": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int]),
    "tgis_dense_rope_ok": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgis_gptq_gemm_rope_f16": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _c_i64,
                                         _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp]),
    "tgis_dense_prepared_bytes": (_c_i64, [_c_i64, _c_i64]),
    "tgis_dense_prepare": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _c_int, _vp, _vp]),
    "tgis_dense_gemm_workspace_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "tgis_dense_gemm": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int,
                                 _c_int, _vp, _c_i64, _vp]),
    "tgis_dense_gemm_partial_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "tgis_dense_gemm_partial": (_c_int, [_vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _vp, _c_i64,
                                         ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64), _vp]),
    "tgis_rmsnorm_residual": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_f, _c_int, _vp]),
    "tgis_rmsnorm_residual_partial": (_c_int, [_vp, _c_int, _c_i64, _vp, _vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_f,
                                               _c_int, _vp]),
    "tgis_layernorm_residual": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_f, _c_int, _vp]),
    "tgis_layernorm_residual_partial": (_c_int, [_vp, _c_int, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64,
                                                 _c_f, _c_int, _vp]),
    "tgis_rope_kv_write": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _c_int,
                                    _c_int, _c_int, _vp]),
    "tgis_rope_kv_write_partial": (_c_int, [_vp, _c_int, _c_i64, _vp, _vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64,
                                            _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "tgis_rope_kv_write_prefill": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_i64,
                                            _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "tgis_attn_num_splits": (_c_int, [_c_i64, _c_int, _c_int, _c_i64, _c_i64]),
    "tgis_attn_workspace_bytes": (_c_i64, [_c_i64, _c_int, _c_int, _c_int, _c_int]),
    "tgis_attn_paged": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _c_i64, _vp, _vp, _vp, _c_i64, _c_i64, _c_int, _c_int,
                                 _c_int, _c_i64, _c_i64, _c_f, _c_int, _c_int, _vp, _c_i64, _vp]),
    "tgis_act_mul": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_int, _c_int, _vp]),
    "tgis_gelu": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _vp]),
    "tgis_embedding": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _vp]),
    "tgis_decode_slots": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _c_i64, _vp]),
    "tgis_decode_advance": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _vp, _vp, _c_i64, _vp]),
    "tgis_argmax_logprob": (_c_int, [_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _vp]),
    "tgis_argmax_scratch_bytes": (_c_i64, [_c_i64]),
    "tgis_warp_sample": (_c_int, [_vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64,
                                  _c_i64, _c_i64, _vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class TgisHipError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the C-ABI library and bind every declared symbol.  Needs no GPU (no compute is run)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise TgisHipError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')"
        )
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().tgis_last_error().decode()
        raise TgisHipError(f"{what} failed (code {rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return F16
    if dt == torch.bfloat16:
        return BF16
    raise TgisHipError(f"unsupported dtype {dt} (float16 / bfloat16 only)")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise TgisHipError("tensor is not on the GPU: the HIP path has no CPU fallback")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  The raw getter skips the Stream object that
    torch.cuda.current_stream() builds (a third of the host time of an eager decode step went there)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def version() -> str:
    return load_library().tgis_version().decode()


def timing_enable(on: bool):
    _check(load_library().tgis_timing_enable(int(on)), "tgis_timing_enable")


def timing_reset():
    _check(load_library().tgis_timing_reset(), "tgis_timing_reset")


def timing_read(op: int):
    c = _c_i64()
    ms = ctypes.c_double()
    _check(load_library().tgis_timing_read(op, ctypes.byref(c), ctypes.byref(ms)), "tgis_timing_read")
    return c.value, ms.value


# ---- workspaces ---------------------------------------------------------------------------------
class Workspace:
    """Zero-initialised scratch for split-K slabs / arrival counters and attention splits.

    One per stream of work.  The first 4096 bytes hold the arrival counters that the GEMM kernels leave
    at zero after every call, so the buffer is zeroed exactly once, at allocation."""

    def __init__(self, nbytes: int, device):
        self.buf = torch.zeros(max(int(nbytes), 4096), dtype=torch.uint8, device=device)
        # Buffers replaced by larger ones stay allocated: HIP graphs captured while they were current hold their raw
        # pointers (slabs, arrival counters, attention splits).  Each buffer's counters are only ever touched by the
        # launches that captured it, so old graphs and new launches never share state.
        self.retired = []

    def ensure(self, nbytes: int):
        if self.buf.numel() < nbytes:
            self.retired.append(self.buf)
            self.buf = torch.zeros(max(int(nbytes), 2 * self.buf.numel()), dtype=torch.uint8, device=self.buf.device)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def nbytes(self):
        return self.buf.numel()


# ---- activations in MFMA-fragment order (TGIS_LD_FRAGMENTS) ------------------------------------------------------------
LD_FRAGMENTS = -32


class FragAct:
    """A [M <= 64, K] f16 activation stored in fragment order (include/tgis_hip.h, TGIS_LD_FRAGMENTS): what the decode
    step's norm / attention / SiLU epilogue hand to the int4 GEMM behind them.  `buf` holds ceil(M / 32) * 32 * K elements."""

    def __init__(self, buf: torch.Tensor, M: int, K: int):
        assert 1 <= M <= 64 and K % 64 == 0 and buf.dtype == torch.float16 and buf.numel() == (M + 31) // 32 * 32 * K
        self.buf, self.M, self.K = buf, M, K
        self.dtype, self.device = buf.dtype, buf.device

    @property
    def shape(self):
        return (self.M, self.K)

    @staticmethod
    def empty(M: int, K: int, device) -> "FragAct":
        return FragAct(torch.empty((M + 31) // 32 * 32 * K, dtype=torch.float16, device=device), M, K)

    @staticmethod
    def from_rows(x: torch.Tensor) -> "FragAct":
        """Row-major [M, K] -> fragment order (torch ops; tests and cold paths only)."""
        M, K = x.shape
        rb = (M + 31) // 32
        full = torch.zeros((rb * 32, K), dtype=x.dtype, device=x.device)
        full[:M] = x
        # (row block, m, step, half, i, e) -> [row block][step][i][half][m][e]
        f = full.view(rb, 32, K // 64, 2, 4, 8).permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)
        return FragAct(f, M, K)

    def to_rows(self) -> torch.Tensor:
        K, rb = self.K, (self.M + 31) // 32
        return self.buf.view(rb, K // 64, 4, 2, 32, 8).permute(0, 4, 1, 3, 2, 5).reshape(rb * 32, K)[:self.M].contiguous()


def gptq_fragments_ok(M: int, w: "GptqWeight", act: int = 0) -> bool:
    """Should an M-row activation reach this int4 GEMM in fragment order (tgis_gptq_fragments_ok)?  act 3 = the rope image."""
    key = ("frag_ok", M, act)
    cache = w.__dict__.setdefault("_frag_ok", {})
    got = cache.get(key)
    if got is None:
        got = cache[key] = bool(load_library().tgis_gptq_fragments_ok(M, w.K, w.N, w.groups, int(w.perm is not None), act))
    return got


# ---- GPTQ ------------------------------------------------------------------------------------------
class GptqWeight:
    """Prepared (repacked) GPTQ matrix; owner of the device image. Mirrors the q_handle of
    Ex4bitLinearV2.post_init (utils/gptq/exllamav2.py:124-137)."""

    def __init__(self, qweight, qzeros, scales, g_idx, bits: int, groupsize: int, gate_up: bool = False,
                 rope: Optional[tuple] = None):
        """rope = (D, rotated heads): the image of a fused qkv projection for gptq_gemm_rope (TGIS_GPTQ_ROPE_IMAGE)."""
        self.flags = 1 if gate_up else 0
        if rope is not None:
            assert not gate_up
            self.flags = 2 | (int(rope[0]) << 8) | (int(rope[1]) << 20)
        self.partial_plan = {}  # pass count -> (slab bytes, S, ld) of the deferred-reduce form, filled on first use
        if bits != 4:
            raise TgisHipError("only 4-bit GPTQ is supported (exllamav2.py:105)")
        lib = load_library()
        self.K = qweight.shape[0] * 8
        self.N = qweight.shape[1]
        self.groups = qzeros.shape[0]
        if self.K % 32 or self.N % 32:
            raise TgisHipError("GPTQ height and width must be multiples of 32 (exllamav2.py:118-119)")
        dev = qweight.device
        nbytes = lib.tgis_gptq_prepared_bytes(self.K, self.N, self.groups)
        self.image = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        qweight = qweight.contiguous()
        qzeros = qzeros.contiguous()
        scales = scales.to(torch.float16).contiguous()
        self.perm = None
        self.in_features = self.K  # columns of the activation (differs from K only for padded act-order row shards)
        gi = None
        perm_buf = None
        explicit = None
        if isinstance(g_idx, tuple):  # ("perm", gather index [K] with -1 = zero, activation columns): rows are in image order
            _, explicit, self.in_features = g_idx
            assert explicit.numel() == self.K
            g_idx = None
        if g_idx is not None:
            gi = g_idx.to("cpu", torch.int32).contiguous()
            perm_buf = torch.empty(self.K, dtype=torch.int32, device=dev)
        _check(
            lib.tgis_gptq_prepare(
                _ptr(qweight), _ptr(qzeros), _ptr(scales), gi.data_ptr() if gi is not None else None,
                _ptr(perm_buf), self.K, self.N, self.groups, self.flags, _ptr(self.image), _stream()),
            "tgis_gptq_prepare")
        if gi is not None:
            gs = self.K // self.groups
            trivial = bool((gi == (torch.arange(self.K, dtype=torch.int32) // gs)).all())
            if not trivial:
                self.perm = perm_buf
        if explicit is not None:
            self.perm = explicit.to(dev, torch.int32).contiguous()
        torch.cuda.current_stream().synchronize()  # qweight/qzeros/scales may be freed by the caller

    def workspace_bytes(self, M: int) -> int:
        return load_library().tgis_gptq_gemm_workspace_bytes(M, self.K, self.N)

    def fused_rows(self, act: int = 0) -> int:
        """Largest M the fused kernels of tgis_gptq_gemm_f16 should be given (beyond: dequantise + library GEMM)."""
        return load_library().tgis_gptq_gemm_fused_rows(self.K, self.groups, int(self.perm is not None), act)


def gptq_gemm(x, w: GptqWeight, ws: Workspace, bias=None, act: int = 0, out=None, out_frag: bool = False):
    """act 0: out[M,N] = x @ dequant(W) (+bias); act 1: x is [M,2K], silu(x[:, :K]) * x[:, K:] is the operand;
    act 2 (weight prepared with gate_up=True): out[M,N/2] = silu(gate) * up.
    x may be a FragAct (decode, M <= 64); with act 2 the result may then leave as a FragAct too (out_frag)."""
    assert act != 2 or w.flags & 1, "act=2 needs a weight prepared with gate_up=True"
    if isinstance(x, FragAct):
        assert x.K == w.K and act in (0, 2) and w.perm is None
        M = x.M
        if out_frag:
            assert act == 2 and out is None
            res = FragAct.empty(M, w.N // 2, x.device)
            optr, ldo = _ptr(res.buf), LD_FRAGMENTS
        else:
            res = out if out is not None else torch.empty((M, w.N // 2 if act == 2 else w.N), dtype=torch.float16,
                                                          device=x.device)
            optr, ldo = _ptr(res), res.stride(0)
        ws.ensure(w.workspace_bytes(M))
        _check(
            load_library().tgis_gptq_gemm_f16(_ptr(x.buf), LD_FRAGMENTS, _ptr(w.image), _ptr(bias), None, optr, ldo, M, w.K,
                                              w.N, w.groups, act, ws.ptr, ws.nbytes, _stream()), "tgis_gptq_gemm_f16")
        return res
    assert not out_frag, "a fragment-order output needs a fragment-order activation"
    assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1
    M = x.shape[0]
    assert x.shape[1] == (2 * w.in_features if act == 1 else w.in_features), (x.shape, w.in_features, act)
    assert act != 1 or w.in_features == w.K
    if out is None:
        out = torch.empty((M, w.N // 2 if act == 2 else w.N), dtype=torch.float16, device=x.device)
    ws.ensure(w.workspace_bytes(M))
    _check(
        load_library().tgis_gptq_gemm_f16(_ptr(x), x.stride(0), _ptr(w.image), _ptr(bias), _ptr(w.perm), _ptr(out),
                                          out.stride(0), M, w.K, w.N, w.groups, act, ws.ptr, ws.nbytes, _stream()),
        "tgis_gptq_gemm_f16")
    return out


class Partial:
    """fp32 split-K partial sums [ceil(M/32), S, 32, ld] of a GEMM whose reduce is deferred to the consumer kernel
    (rmsnorm_residual / rope_kv_write accept it in place of the f16 activation)."""

    def __init__(self, slabs: torch.Tensor, S: int, ld: int, M: int, N: int, bias):
        self.slabs, self.S, self.ld, self.M, self.N, self.bias = slabs, S, ld, M, N, bias
        self.dtype = torch.float16
        self.device = slabs.device

    @property
    def shape(self):
        return (self.M, self.N)


PARTIAL_MAX_M = 256  # rows up to which a GEMM may leave its split-K sum to the consumer kernel


def gptq_gemm_partial(x: torch.Tensor, w: GptqWeight, bias=None, act: int = 0) -> Partial:
    """Launch the GEMM but leave the split-K reduce (and bias) to the consumer.  Slabs are stored in 32-row units:
    [ceil(M/32)][S][32][ld]."""
    lib = load_library()
    if isinstance(x, FragAct):  # decode, <= 64 rows: the fragment-order kernel and ITS split plan
        assert x.K == w.K and act == 0 and w.perm is None
        # the split plan of the fragment-order kernel depends on the row class (plan_wide: <= 32 rows / 33 - 64 rows)
        M, key, xp, ldx = x.M, ("frag", x.M > 32), _ptr(x.buf), LD_FRAGMENTS
    else:
        assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] <= PARTIAL_MAX_M
        M = x.shape[0]
        key, xp, ldx = ((M + 63) // 64 if M > 32 else 0), _ptr(x), x.stride(0)  # plan depends on M through its pass count
    plan = w.partial_plan.get(key)  # (slab bytes, S, ld): asked from the library once per weight and pass count
    if plan is None:
        nbytes = lib.tgis_gptq_gemm_partial_bytes(M, w.K, w.N)
        slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        S = _c_int()
        ld = _c_i64()
        _check(
            lib.tgis_gptq_gemm_f16_partial(xp, ldx, _ptr(w.image), _ptr(w.perm), M, w.K, w.N,
                                           w.groups, act, _ptr(slabs), nbytes, ctypes.byref(S), ctypes.byref(ld),
                                           _stream()), "tgis_gptq_gemm_f16_partial")
        w.partial_plan[key] = (nbytes, S.value, ld.value)
        return Partial(slabs, S.value, ld.value, M, w.N, bias)
    nbytes, S, ld = plan
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(
        lib.tgis_gptq_gemm_f16_partial(xp, ldx, _ptr(w.image), _ptr(w.perm), M, w.K, w.N,
                                       w.groups, act, _ptr(slabs), nbytes, None, None, _stream()),
        "tgis_gptq_gemm_f16_partial")
    return Partial(slabs, S, ld, M, w.N, bias)


def gptq_rope_ok(M: int, w: GptqWeight, D: int) -> bool:
    return bool(load_library().tgis_gptq_rope_ok(M, w.K, w.N, w.groups, int(w.perm is not None), D))


def rope_gemm_ok(M: int, w, D: int) -> bool:
    """Whether the fused qkv + rotary + cache-write launch should serve M rows of this weight (int4 or dense image)."""
    key = ("rope_ok", M)
    cache = w.__dict__.setdefault("_rope_ok", {})
    got = cache.get(key)
    if got is None:
        if isinstance(w, GptqWeight):
            got = gptq_rope_ok(M, w, D)
        else:
            got = bool(load_library().tgis_dense_rope_ok(M, w.K, w.N, D))
        cache[key] = got
    return got


def gptq_gemm_rope(x: torch.Tensor, w: GptqWeight, bias, cos, sin, positions, slots, k_pool, v_pool, H: int, Hkv: int,
                   D: int, out=None) -> torch.Tensor:
    """qkv projection + rotary embedding + cache write in one launch (decode, M <= 64; `w` is the rope image of the fused
    qkv weight).  Returns a [M, (H + 2 Hkv) D] tensor whose first H D columns hold the rotated q (the k / v columns are not
    written: they went straight into their cache pages)."""
    if isinstance(x, FragAct):
        assert x.K == w.K
        M, xp, ldx = x.M, _ptr(x.buf), LD_FRAGMENTS
    else:
        assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == w.K
        M, xp, ldx = x.shape[0], _ptr(x), x.stride(0)
    assert w.flags & 2 and w.N == (H + 2 * Hkv) * D
    assert positions.dtype == torch.int32 and slots.dtype == torch.int32 and cos.dtype == torch.float16
    assert cos.shape[1] * 2 == D, "the fused epilogue covers the full rotary span only"
    if out is None:
        out = torch.empty((M, w.N), dtype=torch.float16, device=x.device)
    _check(
        load_library().tgis_gptq_gemm_rope_f16(xp, ldx, _ptr(w.image), _ptr(bias), _ptr(positions), _ptr(slots),
                                               _ptr(cos), _ptr(sin), _ptr(out), out.stride(0), _ptr(k_pool), _ptr(v_pool),
                                               M, w.K, w.N, w.groups, H, Hkv, D, _stream()),
        "tgis_gptq_gemm_rope_f16")
    return out


def gptq_dequant(w: GptqWeight) -> torch.Tensor:
    """Dense f16 [K,N] (rows in the prepared order: permuted by w.perm for act-order matrices)."""
    out = torch.empty((w.K, w.N), dtype=torch.float16, device=w.image.device)
    _check(load_library().tgis_gptq_dequant_f16(_ptr(w.image), _ptr(out), w.K, w.N, w.groups, w.flags, _stream()),
           "tgis_gptq_dequant_f16")
    return out


# ---- dense ------------------------------------------------------------------------------------------
class DenseWeight:
    """torch-Linear weight [N,K] repacked into MFMA tile order.  gate_up=True: the weight is the Llama MLP's
    [gate | up] and the image interleaves the pairs for the SiLU*up epilogue (dense_gemm(act=2))."""

    def __init__(self, weight: torch.Tensor, gate_up: bool = False, rope: Optional[tuple] = None):
        lib = load_library()
        assert weight.dim() == 2
        self.N, self.K = weight.shape
        self.dtype = weight.dtype
        self.flags = 1 if gate_up else 0
        if rope is not None:  # (D, rotated heads): the image of a fused qkv projection for dense_gemm_rope
            assert not gate_up
            self.flags = 2 | (int(rope[0]) << 8) | (int(rope[1]) << 20)
        weight = weight.contiguous()
        self.image = torch.empty(lib.tgis_dense_prepared_bytes(self.N, self.K), dtype=torch.uint8,
                                 device=weight.device)
        _check(lib.tgis_dense_prepare(_ptr(weight), self.N, self.K, dtype_code(weight.dtype), self.flags,
                                      _ptr(self.image), _stream()), "tgis_dense_prepare")
        torch.cuda.current_stream().synchronize()

    def workspace_bytes(self, M: int) -> int:
        return load_library().tgis_dense_gemm_workspace_bytes(M, self.K, self.N)


def dense_gemm(x: torch.Tensor, w: DenseWeight, ws: Workspace, bias=None, out_f32: bool = False, act: int = 0,
               out=None) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == w.dtype
    M = x.shape[0]
    assert x.shape[1] == (2 * w.K if act == 1 else w.K)
    assert (act == 2) == bool(w.flags & 1), "act 2 runs on (and only on) a gate|up image"
    if out is None:
        out = torch.empty((M, w.N // 2 if act == 2 else w.N), dtype=torch.float32 if out_f32 else w.dtype,
                          device=x.device)
    ws.ensure(w.workspace_bytes(M))
    _check(
        load_library().tgis_dense_gemm(_ptr(x), x.stride(0), _ptr(w.image), _ptr(bias), _ptr(out), out.stride(0), M,
                                       w.K, w.N, dtype_code(w.dtype), int(out_f32), act, ws.ptr, ws.nbytes,
                                       _stream()), "tgis_dense_gemm")
    return out


def dense_gemm_partial(x: torch.Tensor, w: DenseWeight, bias=None, act: int = 0) -> Partial:
    """Launch the dense GEMM but leave the split-K reduce (and bias) to the consumer kernel."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == w.dtype and x.shape[0] <= PARTIAL_MAX_M
    lib = load_library()
    nbytes = lib.tgis_dense_gemm_partial_bytes(x.shape[0], w.K, w.N)
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    S = _c_int()
    ld = _c_i64()
    _check(
        lib.tgis_dense_gemm_partial(_ptr(x), x.stride(0), _ptr(w.image), x.shape[0], w.K, w.N, dtype_code(w.dtype), act,
                                    _ptr(slabs), nbytes, ctypes.byref(S), ctypes.byref(ld), _stream()),
        "tgis_dense_gemm_partial")
    p = Partial(slabs, S.value, ld.value, x.shape[0], w.N, bias)
    p.dtype = w.dtype
    return p


def dense_gemm_rope(x: torch.Tensor, w: DenseWeight, bias, cos, sin, positions, slots, k_pool, v_pool, H: int, Hkv: int,
                    D: int, out=None) -> torch.Tensor:
    """Dense qkv projection + rotary embedding + cache write in one launch (decode, M <= 64; `w` is the rope image of the
    fused qkv weight).  Returns [M, (H + 2 Hkv) D] whose first H D columns hold the rotated q."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == w.dtype and x.shape[1] == w.K
    assert w.flags & 2 and w.N == (H + 2 * Hkv) * D and cos.dtype == w.dtype and cos.shape[1] * 2 == D
    assert positions.dtype == torch.int32 and slots.dtype == torch.int32
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, w.N), dtype=w.dtype, device=x.device)
    _check(
        load_library().tgis_dense_gemm_rope(_ptr(x), x.stride(0), _ptr(w.image), _ptr(bias), _ptr(positions), _ptr(slots),
                                            _ptr(cos), _ptr(sin), _ptr(out), out.stride(0), _ptr(k_pool), _ptr(v_pool), M,
                                            w.K, w.N, H, Hkv, D, dtype_code(w.dtype), _stream()),
        "tgis_dense_gemm_rope")
    return out


def clear_error() -> None:
    """Drop a stale HIP error (aborted graph capture) so that it is not blamed on the next launch."""
    load_library().tgis_clear_error()


# ---- norms --------------------------------------------------------------------------------------------
def rmsnorm_residual(x, residual, weight, eps: float, y=None, res_out=None, frag: bool = False):
    """(y, res) = fused add + RMSNorm; mirrors LlamaRMSNorm.forward (flash_llama_modeling.py:113-152).
    frag (rows <= 32, f16): y is returned as a FragAct, the layout the int4 GEMM behind the norm reads."""
    rows, hidden = x.shape
    yf = None
    if frag:
        assert y is None and rows <= 64 and hidden % 64 == 0 and x.dtype == torch.float16
        yf = FragAct.empty(rows, hidden, x.device)
        y, ldy = yf.buf, LD_FRAGMENTS
    else:
        ldy = hidden
    if isinstance(x, Partial):
        if y is None:
            y = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
        if res_out is None:
            res_out = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)  # the reduced (+ residual) stream
        _check(
            load_library().tgis_rmsnorm_residual_partial(_ptr(x.slabs), x.S, x.ld, _ptr(x.bias), _ptr(residual),
                                                         _ptr(weight), _ptr(y), ldy, _ptr(res_out), rows, hidden,
                                                         float(eps), dtype_code(x.dtype), _stream()),
            "tgis_rmsnorm_residual_partial")
        return (yf if frag else y), res_out
    assert x.dim() == 2 and x.is_contiguous()
    if y is None:
        y = torch.empty_like(x)
    if res_out is None:
        res_out = torch.empty_like(x) if residual is not None else x
    _check(
        load_library().tgis_rmsnorm_residual(_ptr(x), _ptr(residual), _ptr(weight), _ptr(y), ldy,
                                             _ptr(res_out) if residual is not None else None, rows, hidden,
                                             float(eps), dtype_code(x.dtype), _stream()), "tgis_rmsnorm_residual")
    return (yf if frag else y), res_out


def layernorm_residual(x, residual, weight, bias, eps: float, y=None, res_out=None):
    """(y, res) = fused add + LayerNorm; mirrors FastLayerNorm.forward (utils/layers.py:363-396)."""
    if isinstance(x, Partial):
        rows, hidden = x.shape
        if y is None:
            y = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
        if res_out is None:
            res_out = torch.empty_like(y)  # always materialised: it is the reduced (+residual) stream
        _check(
            load_library().tgis_layernorm_residual_partial(_ptr(x.slabs), x.S, x.ld, _ptr(x.bias), _ptr(residual),
                                                           _ptr(weight), _ptr(bias), _ptr(y), _ptr(res_out), rows,
                                                           hidden, float(eps), dtype_code(x.dtype), _stream()),
            "tgis_layernorm_residual_partial")
        return y, res_out
    assert x.dim() == 2 and x.is_contiguous()
    rows, hidden = x.shape
    if y is None:
        y = torch.empty_like(x)
    if res_out is None:
        res_out = torch.empty_like(x) if residual is not None else x
    _check(
        load_library().tgis_layernorm_residual(_ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(y),
                                               _ptr(res_out) if residual is not None else None, rows, hidden,
                                               float(eps), dtype_code(x.dtype), _stream()),
        "tgis_layernorm_residual")
    return y, res_out


# ---- rope + kv write, attention --------------------------------------------------------------------------
def rope_kv_write(qkv, cos, sin, positions, slots, k_pool, v_pool, H: int, Hkv: int, D: int, rot_dim: int):
    """Rotates q,k in place and writes k,v to the cache.  `qkv` may be a Partial: the reduced, rotated activation is
    then materialised into a fresh [T, (H+2Hkv)D] tensor, which is returned (the plain form returns qkv itself)."""
    if isinstance(qkv, Partial):
        T = qkv.M
        out = torch.empty((T, qkv.N), dtype=qkv.dtype, device=qkv.device)
        _check(
            load_library().tgis_rope_kv_write_partial(_ptr(qkv.slabs), qkv.S, qkv.ld, _ptr(qkv.bias), _ptr(out),
                                                      out.stride(0), _ptr(cos), _ptr(sin), _ptr(positions),
                                                      _ptr(slots), _ptr(k_pool), _ptr(v_pool), T, H, Hkv, D, rot_dim,
                                                      dtype_code(out.dtype), _stream()), "tgis_rope_kv_write_partial")
        return out
    assert qkv.dim() == 2 and qkv.stride(1) == 1
    T = qkv.shape[0]
    _check(
        load_library().tgis_rope_kv_write(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(positions),
                                          _ptr(slots), _ptr(k_pool), _ptr(v_pool), T, H, Hkv, D, rot_dim,
                                          dtype_code(qkv.dtype), _stream()), "tgis_rope_kv_write")
    return qkv


def rope_kv_write_prefill(qkv, cos, sin, positions, cu_seqlens, block_tables, k_pool, v_pool, max_len: int, H: int,
                          Hkv: int, D: int, rot_dim: int):
    """rope_kv_write for a fresh prefill (token i of a sequence = cache position i): page-wise cache writes."""
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and block_tables.is_contiguous()
    B = block_tables.shape[0]
    _check(
        load_library().tgis_rope_kv_write_prefill(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(positions),
                                                  _ptr(cu_seqlens), _ptr(block_tables), block_tables.shape[1],
                                                  _ptr(k_pool), _ptr(v_pool), B, qkv.shape[0], max_len, H, Hkv, D,
                                                  rot_dim, dtype_code(qkv.dtype), _stream()),
        "tgis_rope_kv_write_prefill")
    return qkv


def attn_num_splits(B: int, Hkv: int, H: int, max_q_len: int, max_ctx: int) -> int:
    return load_library().tgis_attn_num_splits(B, Hkv, H, max_q_len, max_ctx)


def attn_workspace_bytes(total_q: int, H: int, Hkv: int, D: int, num_splits: int) -> int:
    return load_library().tgis_attn_workspace_bytes(total_q, H, Hkv, D, num_splits)


def attn_paged(q, ld_q: int, k_pool, v_pool, block_tables, ctx_lens, cu_seqlens_q, out, B: int, H: int, Hkv: int,
               D: int, max_q_len: int, max_ctx: int, scale: float, num_splits: int, ws: Optional[Workspace]):
    """q is a (view into a) [T, *] activation whose row stride is ld_q elements; out [T, H*D], or (decode, B <= 32) a
    FragAct of that shape for the o_proj GEMM."""
    assert block_tables.dtype == torch.int32 and ctx_lens.dtype == torch.int32 and cu_seqlens_q.dtype == torch.int32
    assert block_tables.is_contiguous()
    if isinstance(out, FragAct):
        assert max_q_len == 1 and out.K == H * D and out.M == B
        optr, ldo = _ptr(out.buf), LD_FRAGMENTS
    else:
        assert out.is_contiguous()
        optr, ldo = _ptr(out), H * D
    wptr, wbytes = (ws.ptr, ws.nbytes) if ws is not None else (None, 0)
    _check(
        load_library().tgis_attn_paged(_ptr(q), ld_q, _ptr(k_pool), _ptr(v_pool), _ptr(block_tables),
                                       block_tables.shape[1], _ptr(ctx_lens), _ptr(cu_seqlens_q), optr, ldo, B, H,
                                       Hkv, D, max_q_len, max_ctx, float(scale), dtype_code(q.dtype), num_splits,
                                       wptr, wbytes, _stream()), "tgis_attn_paged")
    return out


# ---- elementwise / sampling ---------------------------------------------------------------------------------
def act_mul(gate_up, I: int, out=None):
    T = gate_up.shape[0]
    assert gate_up.is_contiguous() and gate_up.shape[1] == 2 * I
    if out is None:
        out = torch.empty((T, I), dtype=gate_up.dtype, device=gate_up.device)
    _check(load_library().tgis_act_mul(_ptr(gate_up), _ptr(out), T, I, 1, dtype_code(gate_up.dtype), _stream()),
           "tgis_act_mul")
    return out


def gelu(x, tanh_approx: bool, out=None):
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    _check(load_library().tgis_gelu(_ptr(x), _ptr(out), x.numel(), int(tanh_approx), dtype_code(x.dtype), _stream()),
           "tgis_gelu")
    return out


def embedding(ids, table, positions=None, pos_table=None, id_offset: int = 0, out=None):
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.is_contiguous()
    T = ids.numel()
    E = table.shape[1]
    if out is None:
        out = torch.empty((T, E), dtype=table.dtype, device=table.device)
    _check(
        load_library().tgis_embedding(_ptr(ids), _ptr(table), _ptr(positions), _ptr(pos_table), _ptr(out), T, E,
                                      table.shape[0], id_offset, dtype_code(table.dtype), _stream()),
        "tgis_embedding")
    return out


def decode_slots(positions, block_tables, slots, ctx_lens):
    B = positions.numel()
    _check(
        load_library().tgis_decode_slots(_ptr(positions), _ptr(block_tables), block_tables.shape[1], _ptr(slots),
                                         _ptr(ctx_lens), B, _stream()), "tgis_decode_slots")


def decode_advance(ids, position_ids, all_input_ids=None, cu_seqlens=None, cu_seqlens_q=None, stage_ids=None,
                   stage_positions=None):
    """After a decode step (flash_causal_lm.py:457,499,533-535 in one launch): position_ids += 1 in place, the new ids
    scattered into all_input_ids at the new positions, cu_seqlens += cu_seqlens_q in place; returns a private copy of
    `ids`.  stage_ids / stage_positions (a decode graph's static inputs) receive the next step's inputs as well."""
    B = ids.numel()
    assert ids.dtype == torch.int64 and position_ids.dtype == torch.int64 and position_ids.numel() == B
    assert ids.is_contiguous() and position_ids.is_contiguous()
    out = torch.empty_like(ids)
    ld = 0
    if all_input_ids is not None:
        assert all_input_ids.dtype == torch.int64 and all_input_ids.stride(1) == 1 and all_input_ids.shape[0] >= B
        ld = all_input_ids.stride(0)
    if cu_seqlens is not None:
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens_q.dtype == torch.int32
        assert cu_seqlens.numel() == B + 1 and cu_seqlens_q.numel() == B + 1
    if stage_ids is not None:
        assert stage_ids.dtype == torch.int64 and stage_ids.numel() == B
    if stage_positions is not None:
        assert stage_positions.dtype == torch.int32 and stage_positions.numel() == B
    _check(
        load_library().tgis_decode_advance(_ptr(ids), _ptr(out), _ptr(position_ids), _ptr(all_input_ids), ld,
                                           _ptr(cu_seqlens), _ptr(cu_seqlens_q), _ptr(stage_ids), _ptr(stage_positions),
                                           B, _stream()), "tgis_decode_advance")
    return out


def argmax_scratch(B: int, device) -> torch.Tensor:
    """Scratch that lets argmax_logprob split the rows of a small batch over several workgroups each."""
    return torch.empty(max(16, load_library().tgis_argmax_scratch_bytes(B)), dtype=torch.uint8, device=device)


def argmax_logprob(logits, ids_out=None, logprob_out=None, scratch=None):
    assert logits.dim() == 2 and logits.stride(1) == 1
    B, V = logits.shape
    if ids_out is None:
        ids_out = torch.empty(B, dtype=torch.int64, device=logits.device)
    if logprob_out is None:
        logprob_out = torch.empty(B, dtype=torch.float32, device=logits.device)
    f32 = logits.dtype == torch.float32
    _check(
        load_library().tgis_argmax_logprob(_ptr(logits), logits.stride(0), B, V, int(f32),
                                           0 if f32 else dtype_code(logits.dtype), _ptr(ids_out),
                                           _ptr(logprob_out), _ptr(scratch), scratch.numel() if scratch is not None else 0,
                                           _stream()), "tgis_argmax_logprob")
    return ids_out, logprob_out


def warp_sample(logits, temperature=None, top_k=None, top_p_cut=None, typical_p=None, rep_penalty=None,
                input_ids=None, exclude_id: int = -1, eos_adjust=None, eos_id: int = -1, do_sample=None, rng=None):
    """The whole next-token chooser for a batch in one launch (tgis_warp_sample).  Returns (next_ids int64 [B],
    logprob f32 [B] of the chosen ids under the warped scores, lse f32 [B], warped scores f32 [B,V])."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B, V = logits.shape
    dev = logits.device
    for t, dt in ((temperature, torch.float32), (top_k, torch.int32), (top_p_cut, torch.float32),
                  (typical_p, torch.float32), (rep_penalty, torch.float32), (do_sample, torch.int32)):
        assert t is None or (t.dtype == dt and t.numel() == B and t.is_contiguous() and t.device == dev)
    assert rng is None or (rng.dtype == torch.int64 and rng.shape == (B, 2) and rng.is_contiguous())
    assert eos_adjust is None or (eos_adjust.dtype == torch.float32 and eos_adjust.shape == (B, 2)
                                  and eos_adjust.is_contiguous())
    L = ld_ids = 0
    if input_ids is not None:
        assert input_ids.dtype == torch.int64 and input_ids.dim() == 2 and input_ids.shape[0] == B
        assert input_ids.stride(1) == 1
        L, ld_ids = input_ids.shape[1], input_ids.stride(0)
    scores = torch.empty((B, V), dtype=torch.float32, device=dev)
    ids = torch.empty(B, dtype=torch.int64, device=dev)
    lps = torch.empty(B, dtype=torch.float32, device=dev)
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    _check(
        load_library().tgis_warp_sample(
            _ptr(logits), logits.stride(0), _ptr(scores), V, B, V, _ptr(temperature), _ptr(top_k), _ptr(top_p_cut),
            _ptr(typical_p), _ptr(rep_penalty), _ptr(input_ids), ld_ids, L, exclude_id, _ptr(eos_adjust), eos_id,
            _ptr(do_sample), _ptr(rng), _ptr(ids), _ptr(lps), _ptr(lse), _stream()), "tgis_warp_sample")
    return ids, lps, lse, scores
