"""ctypes binding of the experiments library (experiments/include/tgis_experiments.h): the measured-and-rejected kernels of
rounds 2 and 3.  `import native_experiments as nx` after `python experiments/build.py`; everything of tgis_amd.native is
re-exported, bound to experiments/lib/libtgis_experiments.so instead of libtgis_hip.so."""
import ctypes
import os
import sys
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "text-generation-inference_amd"))
os.environ.setdefault("TGIS_HIP_LIB", os.path.join(_HERE, "lib", "libtgis_experiments.so"))
from tgis_amd import native  # noqa: E402
from tgis_amd.native import *  # noqa: E402,F401,F403
from tgis_amd.native import (GptqWeight, DenseWeight, Partial, Workspace, _check, _ptr, _stream, _c_i64, _c_int, _c_f,  # noqa: E402
                             _vp, dtype_code)

OP_DECODE_TAIL = 7


class NormIn(ctypes.Structure):
    """tgis_norm_in of include/tgis_hip.h: the add + RMSNorm that runs as the first phase of a GEMM launch."""
    _fields_ = [("slabs", ctypes.c_void_p), ("num_slabs", ctypes.c_int), ("slab_ld", ctypes.c_int64),
                ("slab_bias", ctypes.c_void_p), ("x", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("weight", ctypes.c_void_p), ("eps", ctypes.c_float), ("y", ctypes.c_void_p), ("res_out", ctypes.c_void_p)]


class TailLinear(ctypes.Structure):
    """tgis_tail_linear of include/tgis_hip.h."""
    _fields_ = [("prepared", _vp), ("bias", _vp), ("K", _c_i64), ("N", _c_i64), ("groups", _c_i64)]


class TailArgs(ctypes.Structure):
    """tgis_tail_args of include/tgis_hip.h (same field order)."""
    _fields_ = [("M", _c_i64), ("hidden", _c_i64), ("eps", _c_f), ("attn_out", _vp), ("residual_in", _vp),
                ("o_proj", TailLinear), ("gate_up", TailLinear), ("down", TailLinear), ("qkv", TailLinear),
                ("norm1_weight", _vp), ("norm2_weight", _vp), ("y1", _vp), ("res1", _vp), ("act", _vp), ("y2", _vp),
                ("res2", _vp), ("slabs_o", _vp), ("slabs_down", _vp), ("slabs_qkv", _vp), ("qkv_out", _vp),
                ("cos", _vp), ("sin", _vp), ("positions", _vp), ("slots", _vp), ("k_pool", _vp), ("v_pool", _vp),
                ("H", _c_int), ("Hkv", _c_int), ("D", _c_int), ("rot_dim", _c_int), ("dtype", _c_int)]



EXPERIMENT_SIGNATURES = {
    "tgis_gptq_norm_gemm_ok": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int]),
    "tgis_gptq_norm_gemm_status": (_c_int, [_c_int]),
    "tgis_gptq_norm_gate_up_f16": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp]),
    "tgis_gptq_norm_qkv_rope_f16": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_i64,
                                             _c_i64, _c_i64, _c_i64, _c_i64, _vp]),
    "tgis_gptq_lean_ok": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int]),
    "tgis_gptq_lean_status": (_c_int, [_c_int]),
    "tgis_xsum_f16": (_c_int, [_vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _vp]),
    "tgis_gptq_gemm_f16_lean": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64,
                                         _c_i64, _c_int, _vp, _c_i64, _vp]),
    "tgis_gptq_gemm_f16_partial_lean": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp,
                                                 _c_i64, _vp, _vp, _vp]),
    "tgis_attn_decode_rope": (_c_int, [_vp, _c_i64, _vp, _c_int, _c_i64, _vp, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp,
                                       _c_i64, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f, _c_int,
                                       _c_int, _vp, _c_i64, _vp]),
    "tgis_llama_decode_tail_slab_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgis_llama_decode_tail_fits": (_c_int, [ctypes.POINTER(TailArgs)]),
    "tgis_llama_decode_tail": (_c_int, [ctypes.POINTER(TailArgs), _vp]),
    "tgis_llama_decode_tail_status": (_c_int, [_c_int]),
    "tgis_llama_decode_tail_trace": (_c_int, [_c_int, _vp, _c_int]),
}


def load_library():
    lib = native.load_library()
    if not getattr(lib, "_experiments_bound", False):
        for name, (res, args) in EXPERIMENT_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib._experiments_bound = True
    return lib


# ---- add + RMSNorm as the first phase of the int4 GEMM behind it ------------------------------------------------------
def gptq_norm_gemm_ok(M: int, w: GptqWeight, act: int) -> bool:
    """Whether the two-phase launch (norm rows -> grid barrier -> GEMM) serves M rows of this weight on this device; the
    first call allocates the library's grid barrier, so make it outside any graph capture."""
    cache = w.__dict__.setdefault("_norm_ok", {})
    got = cache.get((M, act))
    if got is None:
        got = cache[(M, act)] = bool(
            load_library().tgis_gptq_norm_gemm_ok(M, w.K, w.N, w.groups, int(w.perm is not None), act))
    return got


def gptq_norm_gemm_status(reset: bool = False) -> int:
    return load_library().tgis_gptq_norm_gemm_status(int(reset))


def _norm_in(x, residual, weight, eps: float):
    """(NormIn, y, res) for x = tensor or Partial; res is x itself when there is nothing to add (as rmsnorm_residual)."""
    n = NormIn()
    if isinstance(x, Partial):
        rows, hidden = x.shape
        n.slabs, n.num_slabs, n.slab_ld, n.slab_bias, n.x = _ptr(x.slabs), x.S, x.ld, _ptr(x.bias), None
        y = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
        res = torch.empty_like(y)
    else:
        assert x.dim() == 2 and x.is_contiguous()
        n.slabs, n.num_slabs, n.slab_ld, n.slab_bias, n.x = None, 0, 0, None, _ptr(x)
        y = torch.empty_like(x)
        res = torch.empty_like(x) if residual is not None else x
    n.residual, n.weight, n.eps = _ptr(residual), _ptr(weight), float(eps)
    n.y = _ptr(y)
    n.res_out = _ptr(res) if (isinstance(x, Partial) or residual is not None) else None
    return n, y, res


def gptq_norm_gate_up(x, residual, norm_weight, eps: float, w: GptqWeight, bias=None):
    """(silu(gate) * up [M, N/2], res) = post_attention_layernorm + gate_up_proj + activation in one launch."""
    n, y, res = _norm_in(x, residual, norm_weight, eps)
    M = y.shape[0]
    out = torch.empty((M, w.N // 2), dtype=torch.float16, device=y.device)
    _check(load_library().tgis_gptq_norm_gate_up_f16(ctypes.byref(n), _ptr(w.image), _ptr(bias), _ptr(out), out.stride(0), M,
                                                     w.K, w.N, w.groups, _stream()), "tgis_gptq_norm_gate_up_f16")
    return out, res


def gptq_norm_qkv_rope(x, residual, norm_weight, eps: float, w: GptqWeight, bias, cos, sin, positions, slots, k_pool, v_pool,
                       H: int, Hkv: int, D: int):
    """(qkv [M, (H + 2 Hkv) D] with the rotated q in its first H D columns, res) = input_layernorm + query_key_value +
    rotary embedding + cache write in one launch."""
    n, y, res = _norm_in(x, residual, norm_weight, eps)
    M = y.shape[0]
    out = torch.empty((M, w.N), dtype=torch.float16, device=y.device)
    _check(load_library().tgis_gptq_norm_qkv_rope_f16(ctypes.byref(n), _ptr(w.image), _ptr(bias), _ptr(positions), _ptr(slots),
                                                      _ptr(cos), _ptr(sin), _ptr(out), out.stride(0), _ptr(k_pool),
                                                      _ptr(v_pool), M, w.K, w.N, w.groups, H, Hkv, D, _stream()),
           "tgis_gptq_norm_qkv_rope_f16")
    return out, res


# ---- lean decode GEMM: x travels with the row sums its producer computed ---------------------------------------------
def xs_of(x: torch.Tensor) -> Optional[torch.Tensor]:
    """The row-sum side tensor `[M, K/16, 2]` fp32 a producer kernel attached to its f16 output (None if it did not)."""
    return getattr(x, "_tgis_xs", None)


def with_xs(x: torch.Tensor, xs: torch.Tensor) -> torch.Tensor:
    x._tgis_xs = xs
    return x


def xsum(x: torch.Tensor) -> torch.Tensor:
    """Row sums of an f16 matrix for the lean GEMM (stand-alone producer; the fused producers are rmsnorm_residual and
    the act=2 epilogue of gptq_gemm_lean)."""
    assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 16 == 0
    xs = torch.empty((x.shape[0], x.shape[1] // 16, 2), dtype=torch.float32, device=x.device)
    _check(load_library().tgis_xsum_f16(_ptr(x), x.stride(0), _ptr(xs), xs.stride(0) // 2, x.shape[0], x.shape[1],
                                        _stream()), "tgis_xsum_f16")
    return xs


def gptq_lean_status(reset: bool = False) -> int:
    """0, or the give-up code of a bounded wait inside the loader / consumer GEMM (synchronises)."""
    return load_library().tgis_gptq_lean_status(int(reset))


def gptq_lean_ok(M: int, w: GptqWeight, act: int = 0) -> bool:
    return bool(load_library().tgis_gptq_lean_ok(M, w.K, w.N, w.groups, int(w.perm is not None), act))


def gptq_gemm_lean(x: torch.Tensor, xs: torch.Tensor, w: GptqWeight, ws: Workspace, bias=None, act: int = 0, out=None,
                   want_xs: bool = False) -> torch.Tensor:
    """tgis_gptq_gemm_f16_lean; act=2 with want_xs attaches the row sums of the activated output to it."""
    assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == w.K
    assert xs.dtype == torch.float32 and xs.is_contiguous() and xs.shape[0] == x.shape[0]
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, w.N // 2 if act == 2 else w.N), dtype=torch.float16, device=x.device)
    xs_out = None
    if want_xs and act == 2:
        xs_out = torch.empty((M, w.N // 32, 2), dtype=torch.float32, device=x.device)
    ws.ensure(w.workspace_bytes(M))
    _check(
        load_library().tgis_gptq_gemm_f16_lean(_ptr(x), x.stride(0), _ptr(xs), xs.stride(0) // 2, _ptr(w.image),
                                               _ptr(bias), _ptr(out), out.stride(0), _ptr(xs_out), M, w.K, w.N,
                                               w.groups, act, ws.ptr, ws.nbytes, _stream()),
        "tgis_gptq_gemm_f16_lean")
    return with_xs(out, xs_out) if xs_out is not None else out


def gptq_gemm_partial_lean(x: torch.Tensor, xs: torch.Tensor, w: GptqWeight, bias=None) -> Partial:
    """tgis_gptq_gemm_f16_partial_lean: the slab geometry is the one of gptq_gemm_partial (same plan)."""
    assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] <= 32
    lib = load_library()
    M = x.shape[0]
    plan = w.partial_plan.get(("lean", 0))
    nbytes = plan[0] if plan else lib.tgis_gptq_gemm_partial_bytes(M, w.K, w.N)
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    S, ld = _c_int(), _c_i64()
    _check(
        lib.tgis_gptq_gemm_f16_partial_lean(_ptr(x), x.stride(0), _ptr(xs), xs.stride(0) // 2, _ptr(w.image), M, w.K,
                                            w.N, w.groups, _ptr(slabs), nbytes,
                                            None if plan else ctypes.byref(S), None if plan else ctypes.byref(ld),
                                            _stream()), "tgis_gptq_gemm_f16_partial_lean")
    if plan is None:
        plan = w.partial_plan[("lean", 0)] = (nbytes, S.value, ld.value)
    return Partial(slabs, plan[1], plan[2], M, w.N, bias)




def attn_decode_rope(qkv, cos, sin, positions, slots, k_pool, v_pool, block_tables, ctx_lens, cu_seqlens_q, out, B: int,
                     H: int, Hkv: int, D: int, rot_dim: int, max_ctx: int, scale: float, num_splits: int,
                     ws: Optional[Workspace]):
    """Decode attention straight from the qkv projection's output (a tensor or a Partial): rotary embedding of q and k
    and the cache write of the new token happen in the attention launch.  out [B, H*D]."""
    assert block_tables.dtype == torch.int32 and ctx_lens.dtype == torch.int32 and cu_seqlens_q.dtype == torch.int32
    assert block_tables.is_contiguous() and out.is_contiguous()
    wptr, wbytes = (ws.ptr, ws.nbytes) if ws is not None else (None, 0)
    if isinstance(qkv, Partial):
        qp, ld, slabs, S, sld, bias, dt = None, 0, _ptr(qkv.slabs), qkv.S, qkv.ld, _ptr(qkv.bias), qkv.dtype
    else:
        assert qkv.dim() == 2 and qkv.stride(1) == 1
        qp, ld, slabs, S, sld, bias, dt = _ptr(qkv), qkv.stride(0), None, 0, 0, None, qkv.dtype
    _check(
        load_library().tgis_attn_decode_rope(qp, ld, slabs, S, sld, bias, _ptr(cos), _ptr(sin), _ptr(positions),
                                             _ptr(slots), rot_dim, _ptr(k_pool), _ptr(v_pool), _ptr(block_tables),
                                             block_tables.shape[1], _ptr(ctx_lens), _ptr(cu_seqlens_q), _ptr(out), B, H,
                                             Hkv, D, max_ctx, float(scale), dtype_code(dt), num_splits, wptr, wbytes,
                                             _stream()), "tgis_attn_decode_rope")
    return out


# ---- persistent decode tail ---------------------------------------------------------------------------------------
def _groups(w) -> int:
    """tgis_tail_linear.groups: the GPTQ group count, 0 for a dense image."""
    return 0 if isinstance(w, DenseWeight) else w.groups


def _tail_linear(w, bias) -> TailLinear:
    return TailLinear(w.image.data_ptr(), _ptr(bias), w.K, w.N, _groups(w))


class DecodeTail:
    """Static part of one layer's tgis_llama_decode_tail call (weights of o_proj / gate_up / down, the post-attention
    norm, and the next layer's input norm + qkv — or the final norm when there is no next layer).  The four linears
    are either all GptqWeight (fp16; gate_up prepared with the fused SiLU*up epilogue) or all DenseWeight."""

    def __init__(self, o_proj, gate_up, down, norm1_weight, norm2_weight, eps: float, qkv=None, H=0, Hkv=0, D=0,
                 rot_dim=0):
        # each linear is (weight image, bias or None)
        self.dense = isinstance(o_proj[0], DenseWeight)
        kinds = {isinstance(l[0], DenseWeight) for l in (o_proj, gate_up, down, qkv) if l is not None}
        assert kinds == {self.dense}, "the decode tail runs layers whose linears are all int4 or all dense"
        self.dtype = o_proj[0].dtype if self.dense else torch.float16
        assert gate_up[0].flags & 1, "the decode tail needs the fused SiLU*up gate_up image"
        self.keep = (o_proj, gate_up, down, qkv, norm1_weight, norm2_weight)  # the struct only holds raw pointers
        self.o_proj, self.gate_up, self.down, self.qkv = o_proj, gate_up, down, qkv
        self.norm1_weight, self.norm2_weight, self.eps = norm1_weight, norm2_weight, float(eps)
        self.hidden = o_proj[0].N
        self.H, self.Hkv, self.D, self.rot_dim = H, Hkv, D, rot_dim
        self._slab_elems = {}

    def slab_elems(self, M: int):
        got = self._slab_elems.get(M)
        if got is None:
            lib = load_library()
            got = tuple(lib.tgis_llama_decode_tail_slab_bytes(M, w.K, w.N, _groups(w)) // 4 if w is not None else 0
                        for w in (self.o_proj[0], self.down[0], self.qkv[0] if self.qkv else None))
            self._slab_elems[M] = got
        return got

    def run(self, attn_out: torch.Tensor, residual: torch.Tensor, cos=None, sin=None, positions=None, slots=None,
            k_pool=None, v_pool=None):
        """Returns (y2, res2, qkv_out or None)."""
        assert attn_out.dtype == self.dtype and attn_out.is_contiguous() and residual.is_contiguous()
        M, dev = attn_out.shape[0], attn_out.device
        E, I = self.hidden, self.down[0].K
        h = lambda n: torch.empty((M, n), dtype=self.dtype, device=dev)  # noqa: E731
        y1, res1, act, y2, res2 = h(E), h(E), h(I), h(E), h(E)
        so, sd, sq = self.slab_elems(M)
        f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)  # noqa: E731
        slabs_o, slabs_d = f(so), f(sd)
        a = TailArgs()
        a.M, a.hidden, a.eps, a.dtype = M, E, self.eps, dtype_code(self.dtype)
        a.attn_out, a.residual_in = _ptr(attn_out), _ptr(residual)
        a.o_proj, a.gate_up, a.down = _tail_linear(*self.o_proj), _tail_linear(*self.gate_up), _tail_linear(*self.down)
        a.norm1_weight, a.norm2_weight = _ptr(self.norm1_weight), _ptr(self.norm2_weight)
        a.y1, a.res1, a.act, a.y2, a.res2 = _ptr(y1), _ptr(res1), _ptr(act), _ptr(y2), _ptr(res2)
        a.slabs_o, a.slabs_down = _ptr(slabs_o), _ptr(slabs_d)
        qkv_out = slabs_q = None
        if self.qkv is not None:
            a.qkv = _tail_linear(*self.qkv)
            qkv_out, slabs_q = h(self.qkv[0].N), f(sq)
            a.slabs_qkv, a.qkv_out = _ptr(slabs_q), _ptr(qkv_out)
            a.cos, a.sin, a.positions, a.slots = _ptr(cos), _ptr(sin), _ptr(positions), _ptr(slots)
            a.k_pool, a.v_pool = _ptr(k_pool), _ptr(v_pool)
            a.H, a.Hkv, a.D, a.rot_dim = self.H, self.Hkv, self.D, self.rot_dim
        _check(load_library().tgis_llama_decode_tail(ctypes.byref(a), _stream()), "tgis_llama_decode_tail")
        return y2, res2, qkv_out


def decode_tail_fits(M: int, o_proj, gate_up, down, qkv=None) -> bool:
    """Can tgis_llama_decode_tail run a layer with these linears (GptqWeight or DenseWeight; shapes and plans only)?"""
    a = TailArgs()
    a.M, a.hidden = M, o_proj.N
    a.dtype = dtype_code(o_proj.dtype) if isinstance(o_proj, DenseWeight) else F16
    for name, w in (("o_proj", o_proj), ("gate_up", gate_up), ("down", down), ("qkv", qkv)):
        if w is not None:
            setattr(a, name, TailLinear(None, None, w.K, w.N, _groups(w)))
    return bool(load_library().tgis_llama_decode_tail_fits(ctypes.byref(a)))


def decode_tail_status(reset: bool = False) -> int:
    return load_library().tgis_llama_decode_tail_status(int(reset))


