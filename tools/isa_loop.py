"""Instruction mix of the MFMA-carrying loops of one kernel in a hipcc .s file: python tools/isa_loop.py file.s kernel-substring [--dump]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
dump = '--dump' in sys.argv
lines = open(path).read().split('\n')
# kernel body: from the label line "<mangled>:" containing key to the next ".end_amdhsa_kernel"/s_endpgm block end
start = next(i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_][\w$.]*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.section') or lines[i].strip().startswith('.rodata') or 's_endpgm' in lines[i] and False) if False else None
end = start
while end < len(lines) and not lines[end].strip().startswith('.Lfunc_end'): end += 1
body = lines[start:end]
# basic blocks
blocks, cur, name = [], [], 'entry'
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    else:
        s = l.strip()
        if s and not s.startswith(';') and not s.startswith('.'): cur.append(s)
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds:' + op
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'vmem:' + op.split()[0]
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_barrier'): return 's_barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_exp') or op.startswith('v_rcp') or op.startswith('v_log'): return 'trans:' + op
    if op.startswith('v_'): return 'valu:' + op
    return 'other:' + op
for name, ins in blocks:
    n_mfma = sum(1 for i in ins if i.startswith('v_mfma'))
    if n_mfma < 4: continue
    c = collections.Counter(cls(i.split()[0]) for i in ins)
    nv = sum(v for k, v in c.items() if k.startswith('valu') or k.startswith('trans'))
    print(f'== {name}: {len(ins)} instructions, {n_mfma} mfma, {nv} valu+trans')
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(f'   {v:4d} {k}')
    if dump:
        for i in ins: print('      ', i)
