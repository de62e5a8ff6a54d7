"""GPU-box tool: how much of a short-K GEMM is its epilogue?  Times M x N x K for K = 64..640 (same M, N): the intercept
of the line is prologue + epilogue + launch; also with / without the residual read."""
import os, sys, torch, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
M = 65536
for N in (320, 960, 2560):
    for tile in (33, 34):
        if N % 160 and tile == 33: continue
        for with_res in (False, True):
            line = f"N{N} tile{tile} res={int(with_res)}:"
            for K in (64, 128, 320, 640):
                g = torch.Generator().manual_seed(0)
                a = torch.randn(M, K, generator=g).to(dev).half()
                w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).half()
                bias = torch.randn(N, generator=g).to(dev)
                res = torch.randn(M, N, generator=g).to(dev).half() if with_res else None
                c = torch.empty(M, N, device=dev, dtype=torch.float16)
                d = ops.gemm_desc(a, w, c, M, N, K, c0=K, lda0=K, bias=bias, res=res, ldr=N, ldc=N, tile=tile, splits=1, taps=1)
                ops.gemm_launch(d); torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10): ops.gemm_launch(d)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 100)
                us = statistics.median(ts)
                mb = (M * K * 2 + M * N * 2 * (2 if with_res else 1)) / 1e6
                line += f"  K{K}: {us:6.1f}us ({mb / us:5.2f} TB/s... {mb:.0f}MB)"
            print(line, flush=True)
