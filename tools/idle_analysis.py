"""GPU busy / idle analysis of a rocprofv3 kernel trace (csv): within the LAST contiguous burst of work (the timed
step of `bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline`), how much of the wall time has a kernel
running, and which kernels account for the busy time.  usage: idle_analysis.py <kernel_trace.csv> [step_ms]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
step_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
end = rows[-1][1]
t0 = end - step_ns if step_ns else rows[0][0]
sel = [r for r in rows if r[0] >= t0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = sel[-1][1] - sel[0][0]
gaps = sorted((sel[i + 1][0] - sel[i][1] for i in range(len(sel) - 1)), reverse=True)
print(f"dispatches {len(sel)}, span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {(span-busy)/1e6:.1f} ms")
pos = [g for g in gaps if g > 0]
print(f"gaps: n={len(pos)} mean {sum(pos)/max(len(pos),1)/1e3:.2f} us, >100us: {sum(1 for g in pos if g>1e5)} totalling {sum(g for g in pos if g>1e5)/1e6:.1f} ms, "
      f">1ms: {sum(1 for g in pos if g>1e6)} totalling {sum(g for g in pos if g>1e6)/1e6:.1f} ms; median {sorted(pos)[len(pos)//2]/1e3:.2f} us")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    k = n.split("(")[0][-60:] if "<" not in n else n[:n.index(">") + 1][-70:]
    by[k][0] += e - s; by[k][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{t/1e6:9.1f} ms {100*t/span:5.1f}% {c:7d} x {t/c/1e3:8.1f} us  {k}")
