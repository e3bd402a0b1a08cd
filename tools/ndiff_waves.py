"""Per-wave durations of k_ndiff_flux from the debug timestamps bench.py --config hybrid writes with BLOM_NDIFF_PROF=<file.npy>
(one row per wave: start, end in 100 MHz ticks, the largest record count of its 64 faces): is the launch its mean wave or its slowest?
usage: ndiff_waves.py file.npy [ni]"""
import sys
import numpy as np
a = np.load(sys.argv[1])
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 216
d = (a[:, 1] - a[:, 0]) * 0.01          # us
on = d > 0
t0 = a[on, 0].min()
print(f"waves {len(a)}, with work {on.sum()}; launch span {(a[on,1].max() - t0) * 0.01:.1f} us; wave duration mean {d[on].mean():.1f} median {np.median(d[on]):.1f} "
      f"p90 {np.percentile(d[on], 90):.1f} p99 {np.percentile(d[on], 99):.1f} max {d[on].max():.1f} us")
print(f"start of the last wave to start: {(a[on,0].max() - t0) * 0.01:.1f} us after the first")
nb = len(a) // 2
idx = np.argsort(-d)[:12]
for w in idx:
    b = w % nb
    t = b * 64
    print(f"  wave {w} ({'v' if w >= nb else 'u'}-faces) j = {t // ni - 3} i = {t % ni - 3}..: {d[w]:.1f} us, records <= {a[w, 2]}, started at {(a[w,0]-t0)*0.01:.1f}")
r = a[on, 2]
s1 = (a[on, 3] - a[on, 0]) * 0.01; s2 = (a[on, 4] - a[on, 3]) * 0.01; s3 = (a[on, 1] - a[on, 4]) * 0.01
ok = (a[on, 3] > 0) & (a[on, 4] > 0)
print(f"phases (waves whose lane 0 has a face): first search {s1[ok].mean():.1f} us, surface alignment + snapping {s2[ok].mean():.1f} us, second search {s3[ok].mean():.1f} us; of the 200 slowest: {s1[ok][np.argsort(-d[on][ok])[:200]].mean():.1f} / {s2[ok][np.argsort(-d[on][ok])[:200]].mean():.1f} / {s3[ok][np.argsort(-d[on][ok])[:200]].mean():.1f}")
print("duration vs record count: corr", np.corrcoef(d[on], r)[0, 1])
for lo, hi in ((0, 50), (50, 100), (100, 150), (150, 200), (200, 400)):
    m = (r >= lo) & (r < hi)
    if m.any():
        print(f"  records {lo:3d}..{hi:3d}: {m.sum():5d} waves, mean {d[on][m].mean():8.1f} us")
