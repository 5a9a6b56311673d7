import os, torch, torch.distributed as dist, sys
sys.path.insert(0, ".")
import stable_diffusion_amd
from stable_diffusion_amd.shard import all_gather_images
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
img = torch.full((1, 768, 768, 3), 7 + dist.get_rank(), dtype=torch.uint8, device=dev)
out = all_gather_images(img, dist.get_world_size())
dist.barrier(); torch.cuda.synchronize()
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
print("rank", dist.get_rank(), "world", dist.get_world_size(), out.shape, out.dtype, int(out.sum().item()), float(tt.item()), flush=True)
dist.destroy_process_group()
