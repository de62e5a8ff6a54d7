"""GPU-box tool: this library's GEMM against the vendor library GEMM torch dispatches to (hipBLASLt / rocBLAS) on the
benchmark's plain linear shapes — a yardstick for the operand-feed ceiling discussion in DESIGN.md, not a product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgd_amd  # noqa
from lgd_amd import ops
dev = torch.device("cuda:0")


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in [(65536, 320, 320), (65536, 960, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 1920, 640),
                  (16384, 5120, 640), (16384, 640, 2560), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120),
                  (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (65536, 320, 2880), (16384, 640, 5760), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    t_lib = timeit(lambda: torch.mm(x, w.t(), out=out))
    t_own = timeit(lambda: ops.linear(x, w, b, out=out))
    fl = 2.0 * M * N * K
    line = f"M{M:6d} N{N:6d} K{K:5d}: vendor {t_lib:7.1f} us {fl / t_lib / 1e6:7.1f} TF/s | own {t_own:7.1f} us {fl / t_own / 1e6:7.1f} TF/s"
    ref = x.float() @ w.float().t()
    for tile in [int(t) for t in os.environ.get("TILES", "").split(",") if t]:
        try:
            t = timeit(lambda: ops.linear(x, w, b, out=out, tile=tile, splits=1))
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            line += f" | tile {tile}: {t:7.1f} us {fl / t / 1e6:7.1f} TF/s err {err:.1e}"
        except RuntimeError as e:
            line += f" | tile {tile}: {e}"
    print(line, flush=True)
