"""The smallest RCCL job under the driver's launcher, time-stamped (tools/README.md): is a slow run the launcher's or ours?"""
import os, sys, time, atexit
t0 = time.time()
def stamp(m): print(f"[{time.time():.0f}] +{time.time()-t0:.1f}s {m}", file=sys.stderr, flush=True)
atexit.register(lambda: stamp("atexit"))
import torch, torch.distributed as dist
stamp("imported")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
stamp("init")
x = [torch.zeros(1, device="cuda", dtype=torch.int64)]
dist.all_gather(x, torch.ones(1, device="cuda", dtype=torch.int64))
torch.cuda.synchronize(); stamp("gathered")
dist.barrier(); dist.destroy_process_group(); stamp("destroyed")
