import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ucfp_amd import _lib, index
torch.cuda.set_device(0)
ctx = _lib.default_context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
import os
n, dim, nq, k = 1_000_000, 768, int(os.environ.get("NQ", "256")), 10
g = torch.Generator(device=dev); g.manual_seed(1)
rows = torch.randn((n, dim), dtype=torch.float32, device=dev, generator=g)
ids = torch.arange(n, dtype=torch.int64, device=dev)
ix = index.DeviceIndex(index.COSINE_F32, dim, index.APPEND_ONLY, ctx)
ix.append_dev(0, ids.data_ptr(), rows.data_ptr(), n, stream)
q = torch.randn((nq, dim), dtype=torch.float32, device=dev, generator=g)
o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev); o_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("REPS", "3"))):
    ix.search_dev(0, q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), 0, o_cnt.data_ptr(), stream)
torch.cuda.synchronize()
