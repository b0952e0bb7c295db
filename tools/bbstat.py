"""Per-basic-block census + schedule string for one kernel: python tools/bbstat.py file.s <name-substring> [min_mfma]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]; minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
i = next(k for k, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and want in l)
end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
def cls(o):
    if o.startswith('v_mfma'): return 'M'
    if o.startswith('v_accvgpr'): return 'a'
    if o.startswith('v_exp'): return 'E'
    if o.startswith('v_'): return 'v'
    if o.startswith('ds_read'): return 'r'
    if o.startswith('ds_write'): return 'W'
    if o.startswith(('global_load', 'buffer_load')): return 'g'
    if o.startswith(('global_store', 'buffer_store')): return 'G'
    if o.startswith('scratch'): return 'X'
    if o.startswith('s_waitcnt'): return '|'
    if o.startswith('s_barrier'): return 'B'
    if o.startswith('s_nop'): return 'n'
    return 's'
cur = 'entry'; bbs = collections.OrderedDict()
for l in lines[i + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: cur = m.group(1) + (' LOOP' if 'Loop' in l else ''); continue
    t = l.strip().split()
    if not t or t[0][0] in '.;': continue
    bbs.setdefault(cur, []).append(t[0])
for k, v in bbs.items():
    s = ''.join(cls(o) for o in v)
    if s.count('M') >= minm:
        print(k, 'n=%d mfma=%d acc=%d valu=%d wait=%d' % (len(v), s.count('M'), s.count('a'), s.count('v') + s.count('E'), s.count('|')))
        print('  ', s)
