import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
table = torch.arange(64, device=dev, dtype=torch.float32).reshape(64, 1).repeat(1, 1024).contiguous()
def run(tag):
    idx = torch.zeros(1, device=dev, dtype=torch.int32)
    out = torch.zeros(1024, device=dev)
    res = []
    big = torch.randn(4096, 4096, device=dev)
    for i in range(64):
        idx.fill_(i)
        ops.select_row(table, idx, out)
        res.append(out.clone())
        big = big @ big * 1e-3   # torch work in between
    torch.cuda.synchronize()
    bad = sum(int(not torch.all(r == i)) for i, r in enumerate(res))
    print(tag, "stream handle", torch.cuda.current_stream().cuda_stream, "mismatches", bad, "of 64")
run("default stream:")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run("explicit stream:")
