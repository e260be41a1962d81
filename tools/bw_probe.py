import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.dev_bench import timeit
a = torch.empty(4096, 4096, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
ms = timeit(lambda: a.fill_(1.0), 50); print(f"fill 134MB: {ms*1e3:.1f} us  {a.numel()*8/ms/1e6:.0f} GB/s")
ms = timeit(lambda: b.copy_(a), 50); print(f"copy 134MB: {ms*1e3:.1f} us  {2*a.numel()*8/ms/1e6:.0f} GB/s (r+w)")
a = torch.empty(8*4096, 4096, dtype=torch.float64, device="cuda")
ms = timeit(lambda: a.fill_(1.0), 20); print(f"fill 1GB: {ms*1e3:.1f} us  {a.numel()*8/ms/1e6:.0f} GB/s")
