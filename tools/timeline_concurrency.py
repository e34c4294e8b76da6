#!/usr/bin/env python3
"""How many kernels of which kind are resident at once, and how long every stream sits idle between two iterations, from a
rocprofv3 --kernel-trace of the default bench command:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline -- python3 bench.py --steps 100 --warmup 10 --steady-steps 0 --no-cpu-baseline
    python tools/timeline_concurrency.py gpurun_out/timeline

The window looked at runs from the 30th to the 80th per cent of the photon-trace launches (the pipeline is full there)."""
import collections
import csv
import glob
import sys


def main(root):
    f = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)[0]
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').split('(')[0][:24], r['Stream_Id'])
                for r in csv.DictReader(open(f)))
    ph = [e for e in ev if e[2].startswith('k_trace_photon')]
    i0, i1 = int(len(ph) * 0.3), int(len(ph) * 0.8)
    lo, hi, n_iter = ph[i0][0], ph[i1][0], i1 - i0
    print(f'window {(hi - lo) / 1e6:.2f} ms, {n_iter} iterations, {(hi - lo) / 1e6 / n_iter:.3f} ms per iteration')

    def resident(pred):
        pts = []
        for s, e, n, _ in ev:
            if e > lo and s < hi and pred(n):
                pts += [(max(s, lo), 1), (min(e, hi), -1)]
        pts.sort()
        cur, last, h = 0, lo, collections.Counter()
        for t, d in pts:
            h[cur] += t - last
            last, cur = t, cur + d
        h[cur] += hi - last
        tot = sum(h.values())
        return {k: round(v / tot, 3) for k, v in sorted(h.items())}

    print('share of the time with N kernels resident')
    for label, pred in (('any kernel', lambda n: True), ('photon trace', lambda n: n.startswith('k_trace_photon')),
                        ('eye trace', lambda n: n.startswith('k_trace_eye')), ('any trace', lambda n: n.startswith('k_trace')),
                        ('gather (+ split)', lambda n: n.startswith('k_gather'))):
        print(f'  {label:18s}', resident(pred))
    dur = collections.defaultdict(list)
    for s, e, n, _ in ev:
        if s > lo and e < hi:
            dur[n].append((e - s) / 1e3)
    print('kernel: launches, average duration in the timeline (us), sum per iteration (ms)')
    for n, v in sorted(dur.items(), key=lambda x: -sum(x[1]))[:8]:
        print(f'  {n:26s} {len(v):4d} {sum(v) / len(v):8.1f} {sum(v) / 1e3 / n_iter:7.3f}')
    by = collections.defaultdict(list)
    for e in ev:
        if e[0] >= lo and e[1] <= hi:
            by[e[3]].append(e)
    print('stream: share of the window with a kernel of its own running; average wait between the resolve of one iteration and the first kernel of its next (us)')
    for sid, l in sorted(by.items()):
        busy = sum(e[1] - e[0] for e in l)
        waits = [(b[0] - a[1]) / 1e3 for a, b in zip(l, l[1:]) if a[2].startswith('k_resolve')]
        print(f'  stream {sid}: {busy / (hi - lo):.3f}   {sum(waits) / max(1, len(waits)):.0f}')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/timeline')
