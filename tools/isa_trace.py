"""Compact instruction-class trace of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only):
M mfma, r/w ds_read/ds_write, B buffer_load, G global_load, D global_load_lds, S global_store, X scratch, | s_barrier,
{..} s_waitcnt, v/s other vector / scalar instructions (with --all), j branches; run lengths as suffix.
usage: isa_trace.py FILE.s KERNEL_SUBSTRING [--all] [--around M] [--width N]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(pat) + r'\S*):', s, re.M)
start = m.end(); end = re.compile(r'^\.Lfunc_end\d+:', re.M).search(s, start).start()
body = s[start:end].split('\n')
allv = "--all" in sys.argv
def cls(l):
    l = l.strip()
    if l.startswith('v_mfma'): return 'M'
    if l.startswith('scratch_'): return 'X'
    if l.startswith('ds_read'): return 'r'
    if l.startswith('ds_write'): return 'w'
    if l.startswith('ds_'): return 'p'
    if l.startswith('buffer_load'): return 'B'
    if l.startswith('global_load_lds'): return 'D'
    if l.startswith('global_load'): return 'G'
    if l.startswith('global_store') or l.startswith('buffer_store'): return 'S'
    if l.startswith('global_atomic'): return 'A'
    if l.startswith('s_barrier'): return '|'
    if l.startswith('s_waitcnt'): return '{' + l.replace('s_waitcnt ', '').replace('vmcnt', 'vm').replace('lgkmcnt', 'lg') + '}'
    if l.startswith('s_cbranch') or l.startswith('s_branch'): return 'j'
    if allv and l.startswith('v_'): return 'v'
    if allv and l.startswith('s_'): return 's'
    if re.match(r'\.LBB', l): return '\n' + l.split(':')[0] + ' '
    return None
out = []; prev = None; cnt = 0
for l in body:
    c = cls(l)
    if c is None: continue
    if c == prev and not c.startswith('{'): cnt += 1
    else:
        if prev is not None: out.append(prev + (str(cnt) if cnt > 1 else ''))
        prev = c; cnt = 1
out.append(prev + (str(cnt) if cnt > 1 else ''))
txt = ' '.join(out)
w = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 4000
if "--around" in sys.argv:
    i = txt.index(sys.argv[sys.argv.index("--around") + 1])
    txt = txt[max(0, i - w // 2): i + w // 2]
print(len(body), "lines"); print(txt[:w])
