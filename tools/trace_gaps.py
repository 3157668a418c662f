"""Per-stream busy time and idle gaps of the last micro-steps in a rocprofv3 kernel trace (csv).
usage: python tools/trace_gaps.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections, re
def short(n):
    m = re.search(r'gemm_kernel<(\d+), (\d+), (\d+), (\d+), \d+, \d+, (\d+)', n)
    if m: return 'gemm<A%s,B%s,%sx%s,s%s>' % m.groups()
    m = re.search(r'gemm8_kernel<(\d+), (\d+)>', n)
    if m: return 'gemm8<256x%s%s>' % (m.group(1), ',geglu' if m.group(2) == '2' else '')
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0].split('<')[0][-34:]
path = sys.argv[1]; win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 380e6
rows = list(csv.DictReader(open(path)))
def col(r, *names):
    for n in names:
        if n in r: return r[n]
    raise KeyError(names)
ks = [(int(col(r, 'Start_Timestamp')), int(col(r, 'End_Timestamp')), col(r, 'Queue_Id', 'Stream_Id'), col(r, 'Kernel_Name')) for r in rows]
ks.sort()
tend = max(k[1] for k in ks); t0 = tend - win
ks = [k for k in ks if k[0] >= t0]
print(f'{len(ks)} kernels in the last {win/1e6:.0f} ms; queues:', collections.Counter(k[2] for k in ks))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
byq = collections.defaultdict(list)
for s, e, q, n in ks: byq[q].append((s, e, n))
print(f'union busy (any stream): {union([(s, e) for s, e, _, _ in ks])/1e6:.1f} ms of {win/1e6:.0f}')
for q, lst in byq.items():
    busy = union([(s, e) for s, e, _ in lst]); span = lst[-1][1] - lst[0][0]
    gaps = sorted(((lst[i + 1][0] - max(x[1] for x in lst[:i + 1][-8:]), lst[i][2], lst[i + 1][2]) for i in range(len(lst) - 1)), reverse=True)
    big = [g for g in gaps if g[0] > 5000]
    print(f'queue {q}: {len(lst)} kernels, busy {busy/1e6:.1f} ms of span {span/1e6:.1f} ms; gaps > 5 us: {len(big)} totalling {sum(g[0] for g in big)/1e6:.1f} ms')
    cls = collections.Counter()
    nxt = collections.Counter(); cnt = collections.Counter()
    for g in big: cls[(short(g[1]), short(g[2]))] += g[0]; nxt[short(g[2])] += g[0]; cnt[short(g[2])] += 1
    print('   idle time in front of (next kernel):')
    for k, t in nxt.most_common(14): print(f'     {t/1e6:6.2f} ms in {cnt[k]:5d} gaps (avg {t/cnt[k]/1e3:5.1f} us) before {k}')
    print('   by (previous -> next):')
    for (a, b), t in cls.most_common(14): print(f'     {t/1e6:6.2f} ms  {a}  ->  {b}')
    hist = collections.Counter(min(int(g[0] / 5000) * 5, 50) for g in big)
    print('   gap histogram (us: count):', sorted(hist.items()))

# the largest gaps of the busiest queue: what ran on the other queues meanwhile
mainq = max(byq, key=lambda q: len(byq[q]))
lst = byq[mainq]
gl = sorted(((lst[i + 1][0] - lst[i][1], lst[i][1], lst[i + 1][0], short(lst[i][2]), short(lst[i + 1][2])) for i in range(len(lst) - 1)), reverse=True)[:6]
for g, t_a, t_b, a, b in gl:
    print(f'gap {g/1e3:8.1f} us  {a} -> {b}')
    for q, l2 in byq.items():
        if q == mainq: continue
        ov = collections.Counter()
        for s, e, n in l2:
            if e > t_a and s < t_b: ov[short(n)] += 1
        print(f'     queue {q} meanwhile: {dict(ov)}')
    # the next kernel's own resources
for r in rows:
    n = short(r['Kernel_Name'])
    if n in ('attn_bwd_dkv_kernel', 'attn_bwd_dq_kernel', 'attn_fwd_kernel') or n.startswith('gemm<A1'):
        key = (n, r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size_X'])
        if key not in globals().setdefault('_seen', set()):
            _seen.add(key); print('resources', key)

# ---- phases of the last complete micro-step: forward = noise_target .. mse, backward = mse .. next noise_target --------------
starts = [s for s, e, q, n in ks if 'noise_target' in n]
losses = [s for s, e, q, n in ks if 'mse_kernel' in n]
if len(starts) >= 2:
    a, c = starts[-2], starts[-1]
    b = max(x for x in losses if a < x < c)
    for name, lo, hi in (('forward', a, b), ('backward', b, c)):
        sel = [(s, e, q, n) for s, e, q, n in ks if lo <= s < hi]
        print(f'{name}: span {(hi - lo)/1e6:.2f} ms, {len(sel)} kernels, busy (any queue) {union([(s, e) for s, e, _, _ in sel])/1e6:.2f} ms')
        for q in sorted(set(x[2] for x in sel)):
            l2 = sorted((s, e, n) for s, e, qq, n in sel if qq == q)
            busy = union([(s, e) for s, e, _ in l2])
            gaps = [(l2[i + 1][0] - max(x[1] for x in l2[:i + 1][-8:]), short(l2[i][2]), short(l2[i + 1][2])) for i in range(len(l2) - 1)]
            pos = [g for g in gaps if g[0] > 0]
            print(f'   queue {q}: {len(l2)} kernels, busy {busy/1e6:.2f} ms, idle between kernels {sum(g[0] for g in pos)/1e6:.2f} ms '
                  f'(median gap {sorted(g[0] for g in pos)[len(pos)//2]/1e3 if pos else 0:.1f} us)')
            tot = collections.Counter(); cnt = collections.Counter()
            for s, e, n in l2: tot[short(n)] += e - s; cnt[short(n)] += 1
            for k, t in tot.most_common(12): print(f'        {t/1e6:6.2f} ms {cnt[k]:5d} x {k}')
            cls = collections.Counter()
            for g in pos: cls[(g[1], g[2])] += g[0]
            for (x, y), t in cls.most_common(6): print(f'        gap {t/1e6:5.2f} ms  {x} -> {y}')

# ---- why the main queue idles in the backward pass: is the kernel after a gap released by a kernel that just ended on another queue? --
if len(starts) >= 2:
    a, c = starts[-2], starts[-1]
    b = max(x for x in losses if a < x < c)
    main = sorted((s, e, n) for s, e, q, n in ks if q == mainq and b <= s < c)
    other = sorted((e, s, n) for s, e, q, n in ks if q != mainq and b - 2e6 <= s < c)
    import bisect
    ends = [o[0] for o in other]
    dep = collections.Counter(); depn = collections.Counter(); free = collections.Counter(); freen = collections.Counter()
    for i in range(len(main) - 1):
        g0 = max(x[1] for x in main[max(0, i - 7):i + 1]); g1 = main[i + 1][0]
        if g1 - g0 <= 4000: continue
        j = bisect.bisect_right(ends, g1 + 500) - 1      # last kernel of another queue that ended before the next main kernel started
        if j >= 0 and ends[j] > g0 and g1 - ends[j] < 8000:
            k = (short(other[j][2]), short(main[i + 1][2])); dep[k] += g1 - g0; depn[k] += 1
        else:
            k = (short(main[i][2]), short(main[i + 1][2])); free[k] += g1 - g0; freen[k] += 1
    print(f'backward, main queue gaps > 4 us: released by a kernel ending on another queue {sum(dep.values())/1e6:.2f} ms in {sum(depn.values())} gaps; '
          f'no such kernel {sum(free.values())/1e6:.2f} ms in {sum(freen.values())} gaps')
    print('   released by (other queue kernel -> main kernel):')
    for k, t in dep.most_common(10): print(f'     {t/1e6:6.2f} ms in {depn[k]:4d} gaps (avg {t/depn[k]/1e3:5.1f} us)  {k[0]}  ->  {k[1]}')
    print('   not released by another queue (previous main kernel -> next):')
    for k, t in free.most_common(10): print(f'     {t/1e6:6.2f} ms in {freen[k]:4d} gaps (avg {t/freen[k]/1e3:5.1f} us)  {k[0]}  ->  {k[1]}')

# ---- GAP_DUMP="<prev short name>|<next short name>": the first gaps of that kind on the main queue with what the other queues did around them
import os
if os.environ.get('GAP_DUMP') and len(starts) >= 2:
    pa, pb = os.environ['GAP_DUMP'].split('|')
    shown = 0
    allk = sorted(ks)
    for i in range(len(main) - 1):
        if short(main[i][2]) != pa or short(main[i + 1][2]) != pb: continue
        g0, g1 = main[i][1], main[i + 1][0]
        if g1 - g0 < 15000: continue
        print(f'--- gap {(g1 - g0)/1e3:.1f} us: {pa} [{(main[i][0]-g0)/1e3:.1f} .. 0.0] -> {pb} [{(g1-g0)/1e3:.1f} .. {(main[i+1][1]-g0)/1e3:.1f}]   (us relative to the end of the previous main kernel)')
        for s, e, q, n in allk:
            if q != mainq and e > g0 - 60000 and s < g1 + 20000:
                print(f'       queue {q}: {short(n):40s} [{(s-g0)/1e3:8.1f} .. {(e-g0)/1e3:8.1f}]')
        shown += 1
        if shown >= 6: break
