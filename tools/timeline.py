"""Diagnostic: reconstruct the timeline of one steady-state step from a rocprofv3 --kernel-trace CSV.

    python tools/timeline.py <dir-or-csv> [step_index_from_end]

A step starts at a k_forward launch.  Prints every kernel of the chosen step with its start/end relative to the
step's start, the queue it ran on and the gap to the previous kernel on that queue, then the median step length
over the last 100 steps and the per-kernel medians."""
import csv, glob, os, re, sys, statistics as st
from collections import defaultdict

p = sys.argv[1]
cands = [p] if os.path.isfile(p) else glob.glob(p + "/**/*kernel_trace.csv", recursive=True) + glob.glob(p + "/**/*.db", recursive=True)
f = cands[0]
rows = []


def clean(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).replace("cafe::", "")


if f.endswith(".db"):       # rocprofv3's rocpd (sqlite) output
    import sqlite3
    for name, q, s, e, gx, wx in sqlite3.connect(f).execute("select name, queue_id, start, end, grid_x, workgroup_x from kernels"):
        rows.append((int(s), int(e), int(q), clean(name), int(gx) // max(1, int(wx))))
else:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), clean(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_forward")]
if len(starts) < 3:
    sys.exit("no steps found")
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a, b = starts[-back - 1], starts[-back]
t0 = rows[a][0]
last_end = {}
print(f"step of {b - a} kernels, {(rows[b][0] - t0) / 1e3:.1f} us fwd-to-fwd")
for s, e, q, name, wgs in rows[a:b]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print(f"  q{q} {name[:34]:34s} wgs={wgs:6d} start={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f} end={(e - t0) / 1e3:8.1f} gap={gap:6.1f}")
    last_end[q] = e
n = min(100, len(starts) - 1)
lens = [(rows[starts[-k]][0] - rows[starts[-k - 1]][0]) / 1e3 for k in range(1, n + 1)]
print(f"median step (fwd-to-fwd) over last {n}: {st.median(lens):.1f} us  min {min(lens):.1f}  max {max(lens):.1f}")
per = defaultdict(list)
cnt = defaultdict(int)
for k in range(1, n + 1):
    seen = defaultdict(float)
    for s, e, q, name, wgs in rows[starts[-k - 1]:starts[-k]]:
        seen[name] += (e - s) / 1e3
    for name, v in seen.items():
        per[name].append(v)
tot = 0.0
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    m = sum(v) / n
    tot += m
    print(f"  {name[:40]:40s} {m:7.1f} us/step (in {len(v)} of {n} steps)")
print(f"  sum of kernel time per step: {tot:.1f} us")
