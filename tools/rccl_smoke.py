"""The torch.distributed calls bench.py makes for N > 1, on however many ranks it is started with (one is enough to
see that the RCCL group with a device id, the gloo side group, the barrier, the reductions and the padded gather
work on this image): python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_smoke.py"""
import os

import torch
import torch.distributed as dist

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
host_group = dist.new_group(backend="gloo")
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
s = torch.tensor([1.0], dtype=torch.float64, device="cuda")
dist.all_reduce(s, op=dist.ReduceOp.SUM)
mine = torch.arange(4, dtype=torch.float64) + dist.get_rank()
parts = [torch.empty(4, dtype=torch.float64) for _ in range(dist.get_world_size())] if dist.get_rank() == 0 else None
dist.gather(mine, parts, dst=0, group=host_group)
dist.barrier(group=host_group)
if dist.get_rank() == 0:
    print(f"rccl_smoke ok: world {dist.get_world_size()}, max {t.item():.0f}, sum {s.item():.0f}, gathered {len(parts)} parts")
dist.destroy_process_group()
