import torch, sys
sys.path.insert(0, "/root/repo")
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[2]
for mb in (1.3, 5.2, 21, 42, 84):
    n = int(mb * 1e6 / 2) // 8 * 8
    x = torch.randn(n, device=dev).half(); y = torch.empty_like(x)
    t_copy = timeit(lambda: y.copy_(x))
    t_mul = timeit(lambda: torch.mul(x, 2.0, out=y))
    t_add = timeit(lambda: ops.add(x, x, y))
    t_sum = timeit(lambda: x.sum())
    print(f"{mb:5.1f} MB: copy {t_copy:6.1f} us ({2*mb/t_copy:4.2f} TB/s)  mul {t_mul:6.1f} us ({2*mb/t_mul:4.2f})  lgd_add {t_add:6.1f} us ({3*mb/t_add:4.2f} TB/s 2r+1w)  sum {t_sum:6.1f} us ({mb/t_sum:4.2f} TB/s read)")
