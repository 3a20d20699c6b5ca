#!/usr/bin/env python3
"""GPU busy / idle time from a rocprofv3 kernel trace of the default (two-stream) bench:
    rocprofv3 --kernel-trace -d OUT --output-format csv -- python3 bench.py --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3
    python3 tools/trace_idle.py OUT
Prints, for the last 8 steps (delimited by the last kernel of the optimiser step: adam_segments_kernel, or adam_kernel with CE_ADAM_TILES=0): wall time per step, the time during which at least one
kernel was running, the time with two or more running, and the largest gaps with the kernels around them."""
import csv, glob, os, sys
import re
rows = []
queues = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        queues[(int(r["Start_Timestamp"]), r["Kernel_Name"])] = r.get("Stream_Id", r.get("Queue_Id", "?"))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"at::native::(\w+)<.*?(\w+Functor|\w+_add|\w+)<", n)
    if m:
        return f"torch:{m.group(1)[:24]}:{m.group(2)[:16]}"
    return n.split("(")[0][:44]
adam = [i for i, r in enumerate(rows) if "adam_segments_kernel" in r[2]] or [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
if len(adam) < 10:
    sys.exit("not enough steps in the trace")
lo, hi = adam[-9], adam[-1]
seg = rows[lo + 1: hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
steps = 8
events = []
for s, e, _ in seg:
    events.append((s, 1)); events.append((e, -1))
events.sort()
busy = over = 0
depth = 0
prev = t0
for t, d in events:
    if depth >= 1: busy += t - prev
    if depth >= 2: over += t - prev
    depth += d
    prev = t
wall = t1 - t0
print(f"per step: wall {wall / steps / 1e6:.3f} ms, >=1 kernel running {busy / steps / 1e6:.3f} ms, >=2 running {over / steps / 1e6:.3f} ms, "
      f"idle {(wall - busy) / steps / 1e6:.3f} ms, sum of kernel durations {sum(e - s for s, e, _ in seg) / steps / 1e6:.3f} ms")
# largest gaps
gaps = []
cur_end = t0
last = rows[lo][2]
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, short(last), short(n)))
    if e > cur_end:
        cur_end, last = e, n
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print(f"{len(gaps) / steps:.0f} gaps per step, total {tot / steps / 1e6:.3f} ms; the largest:")
for g in gaps[:12]:
    print(f"  {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}")

# timeline of the last complete step (start offset us, duration us, stream/queue, kernel) for offline reading
if len(sys.argv) > 2:
    lo2, hi2 = adam[-2], adam[-1]
    tz = rows[lo2][1]
    with open(sys.argv[2], "w") as f:
        for s_, e_, n_ in rows[lo2 + 1: hi2 + 1]:
            f.write(f"{(s_ - tz) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f} q{queues.get((s_, n_), '?'):>4s} {short(n_)}\n")
