"""RCCL sanity on a 1-GPU box: process group of size 1 over the nccl (= RCCL) backend, the exact collective parallel.py issues."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from pdp_amd import parallel
loss = torch.arange(5, dtype=torch.float64, device="cuda"); grad = torch.randn(5, 9, dtype=torch.float64, device="cuda")
packed = torch.cat([grad, loss[:, None]], dim=1).contiguous(); out = torch.empty_like(packed)
dist.all_gather_into_tensor(out, packed); torch.cuda.synchronize()
assert torch.equal(out, packed)
L, G = parallel.gather_loss_grad(loss, grad, 5)
assert torch.equal(L, loss) and torch.equal(G, grad)
print("RCCL all_gather_into_tensor ok:", torch.cuda.get_device_name(0), "nccl version", torch.cuda.nccl.version())
dist.destroy_process_group()
