export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 AMD_LOG_LEVEL=0
CASES="sample t=0.8" timeout 300 python - <<'PY' 2>&1 | grep -v "amdgpu\|hostname" | tail -30
import os, sys, faulthandler
faulthandler.enable()
sys.path[:0] = ["/root/repo", "/root/repo/tools", "/root/repo/text-generation-inference_amd"]
import torch
from tgis_amd import native
orig = native.warp_sample
def traced(logits, **kw):
    torch.cuda.synchronize()
    print("warp_sample in:", tuple(logits.shape), logits.stride(), logits.dtype, "finite", bool(torch.isfinite(logits).all()),
          {k: (None if v is None else (tuple(v.shape), v.dtype) if hasattr(v, "shape") else v) for k, v in kw.items()}, flush=True)
    out = orig(logits, **kw)
    torch.cuda.synchronize()
    print("warp_sample ok", out[0][:4].tolist(), flush=True)
    return out
native.warp_sample = traced
import runpy
runpy.run_path("/root/repo/tools/sampling_bench.py", run_name="__main__")
PY
