import torch, sys
sys.path.insert(0, "/root/repo")
from lion_amd.emd import earth_mover_distance_nograd
x1, x2 = torch.rand(32, 2048, 3, device="cuda"), torch.rand(32, 2048, 3, device="cuda")
for _ in range(6):
    earth_mover_distance_nograd(x1, x2, transpose=False)
torch.cuda.synchronize()
