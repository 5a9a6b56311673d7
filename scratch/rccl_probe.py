"""One-rank RCCL check of the collectives bench.py / shard.py use (uint8 all_gather_into_tensor, barrier, fp64 MAX)."""
import os, torch, torch.distributed as dist
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
w = dist.get_world_size()
img = torch.full((1, 768, 768, 3), 7 + dist.get_rank(), dtype=torch.uint8, device=dev)
out = torch.empty((w,) + tuple(img.shape[1:]), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(out, img)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    out2 = torch.empty_like(out)
    dist.all_gather_into_tensor(out2, img)          # a collective issued under a side stream
dist.barrier(); torch.cuda.synchronize()
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
print("rank", dist.get_rank(), "world", w, out.shape, out.dtype, int(out.sum().item()), int(out2.sum().item()), float(tt.item()), flush=True)
dist.barrier()
dist.destroy_process_group()
