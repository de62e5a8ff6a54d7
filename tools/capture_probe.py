"""Does a hipGraph capture on one host thread disturb plain work (sync H2D copies, allocations, launches) of another
thread?  Per capture_error_mode."""
import threading, time, sys
import torch
dev = torch.device("cuda:0")
x = torch.randn(1 << 20, device=dev)
for mode in ("global", "thread_local", "relaxed"):
    stop = False
    errs = {"capture": 0, "other": 0, "n_cap": 0, "n_other": 0, "first": None}
    def capturer():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y = torch.empty_like(x)
            for _ in range(60):
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=mode):
                        for _ in range(50):
                            torch.mul(x, 2.0, out=y)
                    g.replay()
                    errs["n_cap"] += 1
                except Exception as e:
                    errs["capture"] += 1
                    errs["first"] = errs["first"] or ("capture: " + str(e).split("\n")[0])
        torch.cuda.synchronize()
    def other():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop:
                try:
                    h = torch.zeros(4096)
                    d = h.to(dev)                      # pageable H2D: synchronous copy
                    e = torch.empty(1 << 16, device=dev).fill_(1.0)
                    float(d.sum() + e[0])
                    errs["n_other"] += 1
                except Exception as ex:
                    errs["other"] += 1
                    errs["first"] = errs["first"] or ("other: " + str(ex).split("\n")[0])
    tb = threading.Thread(target=other); tb.start()
    ta = threading.Thread(target=capturer); ta.start(); ta.join()
    stop = True; tb.join()
    print(mode, errs, flush=True)
