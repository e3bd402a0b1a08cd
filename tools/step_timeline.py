#!/usr/bin/env python3
"""Timeline of one baroclinic step from a rocprofv3 kernel trace (csv): for every launch of the step its start relative to the
step's first kernel, its duration, its queue, and how much of it overlapped with launches on other queues.  A step starts at
a launch of k_init_fluxes* (the sequence's first kernel); the step printed is the last but one complete step of the trace."""
import csv, re, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)
starts = [i for i, r in enumerate(rows) if short(r[2]).startswith("k_init_fluxes")]
if len(starts) < 3:
    sys.exit("fewer than three steps in the trace")
# the bench's last steps run with stage timers (every stage on one queue): take the last step that used more than one queue
cands = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
multi = [ab for ab in cands if len({r[3] for r in rows[ab[0]:ab[1]]}) > 1]
a, b = (multi[-2] if len(multi) > 1 else multi[-1]) if multi else cands[-2]
step = rows[a:b]
print("# steps in the trace (launches, queues, us): " + " ".join(f"{y - x}/{len({r[3] for r in rows[x:y]})}/{(max(r[1] for r in rows[x:y]) - rows[x][0]) / 1e3:.0f}" for x, y in cands))
t0 = step[0][0]
queues = sorted({r[3] for r in step})
print(f"# one step: {len(step)} launches, {(max(r[1] for r in step) - t0) / 1e3:.1f} us from the first start to the last end; queues {queues}")
print(f"# {'start_us':>9} {'dur_us':>8} {'end_us':>9} q  kernel   [overlap with other queues, us]")
for s, e, n, q in step:
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2, q2 in step if q2 != q)
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f} {queues.index(q)}  {short(n)}" + (f"   [{ov / 1e3:.1f}]" if ov else ""))
busy = sorted((s, e) for s, e, _, _ in step)
tot = 0; cs, ce = busy[0]
for s, e in busy[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print(f"# some kernel running: {tot / 1e3:.1f} us; sum of durations {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")
