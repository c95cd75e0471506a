"""RCCL single-rank self-check (GPU box): the exact collective calls of interspeech_ser_amd/dist.py -- init with
device_id, broadcast_object_list, a flat fp32 bucket broadcast, all_reduce MAX / SUM on float64, barrier -- on the
"nccl" backend with world_size 1.  One GPU cannot host two RCCL ranks, so this is as far as the backend can be
exercised without a node; the N-rank control flow is rehearsed with gloo (tests/test_dist_gloo.py)."""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
obj = [[("a", (2, 3)), ("b", (5,))]]
dist.broadcast_object_list(obj, src=0)
flat = torch.arange(1 << 24, dtype=torch.float32, device=dev)
dist.barrier()
t0 = time.perf_counter()
dist.broadcast(flat, src=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
assert float(t.item()) == 3.5 and float(flat[12345].item()) == 12345.0
print(f"rccl self-check ok: backend={dist.get_backend()} broadcast 64 MiB in {dt * 1e3:.2f} ms", flush=True)
dist.destroy_process_group()
sys.exit(0)
