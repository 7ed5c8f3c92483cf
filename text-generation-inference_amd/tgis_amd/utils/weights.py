"""Checkpoint access + tensor-parallel slicing, mirroring utils/weights.py:14-229 of the reference.

`Weights` routes tensor names to safetensors files and hands out full / sharded tensors; the GPTQ
bundles follow the reference's rules exactly: column-parallel = shard qweight/qzeros/scales on dim 1 and
concatenate the fused prefixes, g_idx full and identical across prefixes (weights.py:115-142);
row-parallel = shard qweight on dim 0, qzeros/scales on dim 0 when groupsize >= 0, g_idx dropped for
tp > 1 (weights.py:144-201).  Act-order + row-parallel + tp > 1 is refused here (the reference falls back
to its Triton kernel, weights.py:150-156; this build has no second kernel family).

`DictWeights` serves the same interface from an in-memory dict (synthetic benchmark weights and tests)."""
import json
import logging
import math
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import torch

logger = logging.getLogger(__name__)

QUANTIZE_CONFIG_FILENAME = "quantize_config.json"


class _Base:
    device = None
    dtype = None
    process_group = None
    gptq_bits = None
    gptq_groupsize = None

    # -- to be provided -----------------------------------------------------------------
    def _full(self, name: str) -> torch.Tensor:
        raise NotImplementedError

    def _shape(self, name: str):
        raise NotImplementedError

    def _slice(self, name: str, dim: int, start: int, stop: int) -> torch.Tensor:
        raise NotImplementedError

    def has(self, name: str) -> bool:
        raise NotImplementedError

    # -- reference interface ---------------------------------------------------------------
    def _finish(self, tensor: torch.Tensor) -> torch.Tensor:
        # GPTQ u4 tensors are disguised as int32 and must not be converted (weights.py:72-75)
        if tensor.dtype not in (torch.int32, torch.int64):
            tensor = tensor.to(dtype=self.dtype)
        return tensor.to(device=self.device)

    def get_shape(self, name: str):
        return self._shape(name)

    def get_tensor(self, name: str) -> torch.Tensor:
        return self._finish(self._full(name))

    def get_partial_sharded(self, name: str, dim: int) -> torch.Tensor:
        world_size = self.process_group.size()
        rank = self.process_group.rank()
        size = self._shape(name)[dim]
        block_size = size // world_size
        return self._finish(self._slice(name, dim, rank * block_size, (rank + 1) * block_size))

    def get_sharded(self, name: str, dim: int) -> torch.Tensor:
        world_size = self.process_group.size()
        size = self._shape(name)[dim]
        assert size % world_size == 0, \
            f"The choosen size {size} is not compatible with sharding on {world_size} shards"
        return self.get_partial_sharded(name, dim)

    def get_multi_weights_col(self, prefixes: List[str], quantize: Optional[str], dim: int):
        if quantize == "gptq":
            if not all(self.has(f"{p}.qweight") for p in prefixes):
                raise RuntimeError("Cannot load `gptq` weight, make sure the model is already quantized")
            qweight = torch.cat([self.get_sharded(f"{p}.qweight", dim=1) for p in prefixes], dim=1)
            qzeros = torch.cat([self.get_sharded(f"{p}.qzeros", dim=1) for p in prefixes], dim=1)
            scales = torch.cat([self.get_sharded(f"{p}.scales", dim=1) for p in prefixes], dim=1)
            w = [self.get_tensor(f"{p}.g_idx") for p in prefixes]
            for w2 in w[1:]:
                torch.testing.assert_close(w2, w[0])
            g_idx = w[0]
            bits, groupsize = self._get_gptq_params()
            return (qweight, qzeros, scales, g_idx, bits, groupsize, bits == 4)
        w = [self.get_sharded(f"{p}.weight", dim=0) for p in prefixes]
        return torch.cat(w, dim=dim)

    def get_multi_weights_row(self, prefix: str, quantize: Optional[str]):
        if quantize == "gptq":
            bits, groupsize = self._get_gptq_params()
            tp = self.process_group.size()
            g_idx_full = self.get_tensor(f"{prefix}.g_idx") if self.has(f"{prefix}.g_idx") else None
            if tp > 1 and g_idx_full is not None:
                K = g_idx_full.shape[0]
                gs = groupsize if groupsize > 0 else K
                trivial = torch.equal(g_idx_full.cpu().to(torch.int32),
                                      (torch.arange(K, dtype=torch.int32) // gs))
                if not trivial and not bool((g_idx_full == 0).all()):
                    # The reference leaves exllama here and serves the shard with its g_idx (Triton) kernel: this rank's
                    # rows of qweight, the FULL scales / zero points, the rank's slice of g_idx
                    # (utils/weights.py:150-156,190-196; W[k] = (q[k] - z[g_idx[k]] - 1) * s[g_idx[k]],
                    # utils/gptq/quant_linear.py:159-192).  Same values here, in the streaming kernel's layout.
                    return self._act_order_row_shard(prefix, bits, groupsize, g_idx_full)
            qweight = self.get_sharded(f"{prefix}.qweight", dim=0)
            rows = qweight.shape[0] * (32 // bits)  # this rank's K
            if groupsize >= 0 and tp > 1 and rows % groupsize != 0:
                # The rank's rows do not start/end on group boundaries (llama-7B down_proj: 11008/4 = 2752 = 21.5
                # groups of 128).  The reference can only serve this through its g_idx (Triton) kernel; here the
                # groups are split into sub-groups of gcd(group, rows) rows that carry a copy of their parent's
                # scale / zero-point, which is numerically identical and keeps the streaming kernel's layout.
                sub = math.gcd(groupsize, rows)
                if sub % 8 != 0:
                    raise NotImplementedError(f"{prefix}: {rows} rows per rank cannot be regrouped from groups of "
                                              f"{groupsize} (sub-group {sub} is not a multiple of 8)")
                start = self.process_group.rank() * rows
                parent = (start + torch.arange(rows // sub) * sub) // groupsize
                qzeros = self.get_tensor(f"{prefix}.qzeros")[parent.to(self.device)].contiguous()
                scales = self.get_tensor(f"{prefix}.scales")[parent.to(self.device)].contiguous()
                groupsize = sub
            elif groupsize >= 0:
                qzeros = self.get_sharded(f"{prefix}.qzeros", dim=0)
                scales = self.get_sharded(f"{prefix}.scales", dim=0)
            else:
                qzeros = self.get_tensor(f"{prefix}.qzeros")
                scales = self.get_tensor(f"{prefix}.scales")
            g_idx = g_idx_full if tp == 1 else None
            return (qweight, qzeros, scales, g_idx, bits, groupsize, bits == 4)
        return self.get_sharded(f"{prefix}.weight", dim=1)

    def _act_order_row_shard(self, prefix: str, bits: int, groupsize: int, g_idx_full: torch.Tensor):
        """Row shard of an act-order GPTQ matrix.  The rank's rows [start, start + rows) belong to arbitrary groups, each
        group possibly present with any number of rows.  They are sorted by group (a permutation of the rank's OWN
        activation columns, applied while the GEMM stages its operand), every group's run is padded to a multiple of 32
        rows (pad rows: zero nibbles and the gather index -1, which reads as a zero activation), and each 32-row sub-group
        carries a copy of its parent group's scale / zero point from the FULL tables: numerically the reference's
        W[k] = (q[k] - z[g_idx[k]] - 1) * s[g_idx[k]] for k in the shard, in the layout the streaming kernel reads.
        Returns the usual bundle with ("perm", gather index int32 [K']) in the g_idx slot and group size 32."""
        SUB = 32
        qweight = self.get_sharded(f"{prefix}.qweight", dim=0)
        dev = qweight.device
        per = 32 // bits
        rows, N = qweight.shape[0] * per, qweight.shape[1]
        start = self.process_group.rank() * rows
        g = g_idx_full[start:start + rows].to("cpu", torch.int64)
        order = torch.argsort(g, stable=True)
        uniq, counts = torch.unique_consecutive(g[order], return_counts=True)
        padded = (counts + SUB - 1) // SUB * SUB
        Kp = int(padded.sum())
        # every group present in the shard costs up to SUB - 1 pad rows: at high TP degree K' approaches 2 K (image, streamed
        # bytes and the scale table grow alike; the numerics do not change) — say so once per weight
        logger.info("%s: act-order row shard of %d rows in %d groups -> %d padded rows (K'/K = %.2f)", prefix, rows,
                    int(uniq.numel()), Kp, Kp / max(rows, 1))
        perm = torch.full((Kp,), -1, dtype=torch.int32)
        parent = torch.empty(Kp // SUB, dtype=torch.int64)
        pos = src = 0
        for u, c, pc in zip(uniq.tolist(), counts.tolist(), padded.tolist()):
            perm[pos:pos + c] = order[src:src + c].to(torch.int32)
            parent[pos // SUB:(pos + pc) // SUB] = u
            pos += pc
            src += c
        shifts = (torch.arange(per, dtype=torch.int32, device=dev) * bits).view(1, per, 1)
        nib = ((qweight.unsqueeze(1) >> shifts) & ((1 << bits) - 1)).reshape(rows, N).to(torch.uint8)
        nib = torch.cat([nib, torch.zeros((1, N), dtype=torch.uint8, device=dev)])  # row `rows`: the pad row
        idx = torch.where(perm < 0, torch.full_like(perm, rows), perm).to(dev).long()
        packed = (nib[idx].view(Kp // per, per, N).to(torch.int32) << shifts).sum(dim=1, dtype=torch.int32)
        qzeros = self.get_tensor(f"{prefix}.qzeros")[parent.to(dev)].contiguous()
        scales = self.get_tensor(f"{prefix}.scales")[parent.to(dev)].contiguous()
        return (packed.contiguous(), qzeros, scales, ("perm", perm, rows), bits, SUB, bits == 4)

    def _get_gptq_params(self) -> Tuple[int, int]:
        if self.has("gptq_bits") and self.has("gptq_groupsize"):
            return int(self._full("gptq_bits").item()), int(self._full("gptq_groupsize").item())
        if self.gptq_bits is None or self.gptq_groupsize is None:
            raise RuntimeError("GPTQ parameters (bits, group_size) not found in checkpoint or config")
        return self.gptq_bits, self.gptq_groupsize

    def _set_gptq_params(self, model_config: Any, model_path: Optional[str]):
        config = model_config.to_dict() if hasattr(model_config, "to_dict") else dict(vars(model_config))
        quantize_config = config.get("quantization_config")
        if quantize_config is None and model_path is not None:
            filename = os.path.join(model_path, QUANTIZE_CONFIG_FILENAME)
            if not os.path.exists(filename):
                return
            with open(filename, "r") as f:
                quantize_config = json.load(f)
        if quantize_config is None:
            return
        self.gptq_bits = quantize_config["bits"]
        self.gptq_groupsize = quantize_config["group_size"]


class Weights(_Base):
    def __init__(self, filenames: List[Path], device, dtype, process_group,
                 aliases: Optional[Dict[str, List[str]]] = None):
        from safetensors import safe_open

        routing = {}
        for filename in filenames:
            with safe_open(filename, framework="pytorch") as f:
                for k in f.keys():
                    if k in routing:
                        raise RuntimeError(f"Key {k} was found in multiple files: {filename} and {routing[k]}")
                    routing[k] = filename
        self.aliases = aliases or {}
        self.routing = routing
        self.device = device
        self.dtype = dtype
        self.process_group = process_group
        self._handles = {}

    def _get_handle(self, filename):
        from safetensors import safe_open

        if filename not in self._handles:
            self._handles[filename] = safe_open(filename, framework="pytorch")
        return self._handles[filename]

    def get_filename(self, tensor_name: str):
        filename = self.routing.get(tensor_name, None)
        if filename is None:
            for alias in self.aliases.get(tensor_name, []):
                filename = self.routing.get(alias, None)
                if filename is not None:
                    return str(filename), alias
            raise RuntimeError(f"weight {tensor_name} does not exist")
        return str(filename), tensor_name

    def has(self, name: str) -> bool:
        try:
            self.get_filename(name)
            return True
        except RuntimeError:
            return False

    def _full(self, name):
        filename, name = self.get_filename(name)
        return self._get_handle(filename).get_tensor(name)

    def _shape(self, name):
        filename, name = self.get_filename(name)
        return self._get_handle(filename).get_slice(name).get_shape()

    def _slice(self, name, dim, start, stop):
        filename, name = self.get_filename(name)
        slice_ = self._get_handle(filename).get_slice(name)
        if dim == 0:
            return slice_[start:stop]
        if dim == 1:
            return slice_[:, start:stop]
        raise NotImplementedError("Let's make that generic when needed")


class DictWeights(_Base):
    """Same interface over an in-memory {name: tensor} dict."""

    def __init__(self, tensors: Dict[str, torch.Tensor], device, dtype, process_group):
        self.tensors = tensors
        self.device = device
        self.dtype = dtype
        self.process_group = process_group

    def has(self, name):
        return name in self.tensors

    def _full(self, name):
        if name not in self.tensors:
            raise RuntimeError(f"weight {name} does not exist")
        return self.tensors[name]

    def _shape(self, name):
        return list(self._full(name).shape)

    def _slice(self, name, dim, start, stop):
        t = self._full(name)
        if dim == 0:
            return t[start:stop]
        if dim == 1:
            return t[:, start:stop]
        raise NotImplementedError("Let's make that generic when needed")
