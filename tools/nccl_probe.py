import os, torch
from datetime import timedelta
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
from torch.distributed import ProcessGroupNCCL
options = ProcessGroupNCCL.Options()
options.is_high_priority_stream = True
options._timeout = timedelta(seconds=60)
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", world_size=1, rank=0, timeout=timedelta(seconds=60), pg_options=options)
x = torch.ones(32, 4096, device="cuda", dtype=torch.float16)
torch.distributed.all_reduce(x)
out = torch.empty(2 * 1000, 32, device="cuda")[:1000]
torch.distributed.all_gather_into_tensor(torch.empty(1000, 32, device="cuda"), torch.ones(1000, 32, device="cuda"))
torch.cuda.synchronize()
print("nccl world=1 ok", float(x[0, 0]))
# graph capture of an all_reduce (single rank): does RCCL capture at all in this build?
g = torch.cuda.CUDAGraph()
y = torch.ones(32, 4096, device="cuda", dtype=torch.float16)
try:
    with torch.cuda.graph(g):
        torch.distributed.all_reduce(y)
    g.replay(); torch.cuda.synchronize()
    print("graph capture of all_reduce (1 rank): ok")
except Exception as e:
    print("graph capture failed:", str(e).splitlines()[0])
torch.distributed.destroy_process_group()
