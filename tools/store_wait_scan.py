"""Static scan of `hipcc -S --cuda-device-only` listings for the pattern that serialized the epilogues of the 1x1 kernels in
round 2: a store, then a `s_waitcnt vmcnt(0|1)` (usually for a load issued between the stores: vmcnt also counts the earlier
STORES, so the wait is a round trip to memory), then the next store.  Prints kernels with >= 4 such sequences.
usage: store_wait_scan.py FILE.s [FILE.s ...]"""
import re, subprocess, sys
for f in sys.argv[1:]:
    s = open(f).read()
    for m in re.finditer(r'^(_Z\S+):', s, re.M):
        e = re.compile(r'^\.Lfunc_end\d+:', re.M).search(s, m.end())
        if not e: continue
        body = [l.strip().split(';')[0].strip() for l in s[m.end():e.start()].split('\n')]
        body = [l for l in body if l]
        seq = 0; state = 0
        for l in body:
            if re.match(r'(global|buffer|flat)_store', l):
                if state == 2: seq += 1
                state = 1
            elif l.startswith('s_waitcnt') and re.search(r'vmcnt\((0|1)\)', l) and state == 1:
                state = 2
            elif l.startswith('s_barrier') or l.startswith('s_endpgm'):
                state = 0
        if seq >= 4:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name); name = re.sub(r"\(.*", "", name)
            print(f"{f.split('/')[-1]:22s} {name[:70]:70s} store / vmcnt(0|1) / store sequences: {seq}")
