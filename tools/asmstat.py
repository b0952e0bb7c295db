"""Instruction-class census of one kernel in a hipcc -S listing: python tools/asmstat.py file.s <name-substring>..."""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for want in sys.argv[2:]:
    for k, (i, name) in enumerate(starts):
        if want not in name: continue
        end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
        c = collections.Counter()
        for l in lines[i + 1:end]:
            l = l.strip()
            if not l or l[0] in '.;' or l.endswith(':'): continue
            c[l.split()[0]] += 1
        g = collections.Counter()
        for op, n in c.items():
            if op.startswith('v_mfma'): g['mfma'] += n
            elif op.startswith('v_accvgpr'): g['accvgpr_mov'] += n
            elif op.startswith('ds_'): g[op] += n
            elif op.startswith('scratch'): g['scratch'] += n
            elif op.startswith(('global_', 'buffer_')): g[op] += n
            elif op.startswith('v_'): g['valu'] += n
            elif op.startswith('s_waitcnt'): g['waitcnt'] += n
            elif op.startswith('s_nop'): g['s_nop'] += n
            else: g['salu/other'] += n
        print(name[:70], 'insts', sum(c.values()))
        print('   ', dict(g))
        print('    valu:', [(o, n) for o, n in c.most_common(60) if o.startswith('v_') and not o.startswith('v_mfma')][:16])
