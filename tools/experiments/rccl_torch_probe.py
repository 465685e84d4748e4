"""Is RCCL itself healthy on this box?  torch.distributed (nccl backend = RCCL) with a world of one: init + all_gather."""
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29511", rank=0, world_size=1)
t = torch.ones(4, device="cuda:0")
out = [torch.zeros(4, device="cuda:0")]
dist.all_gather(out, t)
torch.cuda.synchronize()
print("torch RCCL all_gather ok:", out[0].tolist(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
