"""Per-micro-step summary of a rocprofv3 kernel trace of tools/iter_timeline.py: for each of the last 8 micro-steps (cut at the
noise_target kernel that opens a micro-step) the span, kernel time per queue, and the kernels whose mean duration differs most
between a reference micro-step (the 3rd) and the 7th.  usage: python tools/exch_trace.py kernel_trace.csv [memory_copy_trace.csv]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows]
K.sort()
starts = [i for i, k in enumerate(K) if 'noise_target' in k[2]]
starts = starts[-17:]          # the two timed iterations + 1
print(f'{len(K)} kernels, {len(starts)} micro-step starts considered')
segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-16:-8]      # the first of the two timed iterations (the second has no successor to cut at)
per = []
for n, (a, b) in enumerate(segs):
    ks = K[a:b]
    t0, t1 = ks[0][0], K[b][0]
    byq = collections.defaultdict(float)
    names = collections.defaultdict(lambda: [0, 0.0])
    for s, e, nm, q in ks:
        byq[q] += (e - s) / 1e6
        import re
        key = re.sub(r'^void ', '', nm).replace('(anonymous namespace)::', '').split('(')[0][:60]
        names[key][0] += 1; names[key][1] += (e - s) / 1e3
    per.append(names)
    mainq = max(byq, key=lambda q: sum(1 for k in ks if k[3] == q))
    mk = [k for k in ks if k[3] == mainq]
    idle = sum(max(0, mk[i + 1][0] - mk[i][1]) for i in range(len(mk) - 1)) / 1e6
    print(f'micro-step {n + 1}: span {(t1 - t0) / 1e6:7.1f} ms, {len(ks)} kernels; kernel ms per queue: ' + ', '.join(f'{q}: {v:.1f}' for q, v in sorted(byq.items())) + f'; chain queue {mainq} idle between kernels {idle:.1f} ms')
ref, bad = per[2], per[6]
diff = sorted(((bad[k][1] - ref[k][1]) / 1e3, k, ref[k], bad[k]) for k in bad if k in ref)
print('largest differences micro-step 7 vs 3 (ms; calls, total us):')
for d, k, r, b in diff[-12:][::-1]:
    print(f'  {d:+7.2f} ms  {k:62s} {r[0]:5d} {r[1]:10.0f} -> {b[0]:5d} {b[1]:10.0f}')
only = [(bad[k][1] / 1e3, k, bad[k][0]) for k in bad if k not in ref]
for t, k, c in sorted(only)[::-1][:8]:
    print(f'  only in 7: {t:7.2f} ms {k} x{c}')
if len(sys.argv) > 2:
    C = list(csv.DictReader(open(sys.argv[2])))
    t00 = K[segs[0][0]][0]
    for r in C:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e - s > 5e6 and s >= t00:
            print(f"copy {r.get('Direction', '?')} {(s - t00) / 1e6:8.1f} -> {(e - t00) / 1e6:8.1f} ms")
