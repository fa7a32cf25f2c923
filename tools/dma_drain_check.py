"""Static check on a `hipcc -S --cuda-device-only` listing: for every LDS-DMA instruction (global_load_lds_*) of every kernel
whose name contains KERNEL, is there an `s_waitcnt vmcnt(N)` between it and the next MFMA / barrier that reaches it (N <= the VM operations issued
behind it)?  Memory operations
retire in order, so such a wait -- typically the compiler's wait for a scratch reload or an earlier load it issued BEFORE the
inline-asm DMA it does not know about -- also waits for the DMA: the prefetch is drained before the work it should overlap.
Round 2: 27 of 30 DMA instructions of conv3d_split_kernel are drained this way (every tap group starts with the round trip of
the NEXT group's weight slices).  usage: dma_drain_check.py FILE.s KERNEL"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):', s, re.M):
    start = m.end(); end = re.compile(r'^\.Lfunc_end\d+:', re.M).search(s, start).start()
    body = [l.strip().split(';')[0].strip() for l in s[start:end].split('\n')]
    body = [l for l in body if l]
    n = drained = 0
    for i, l in enumerate(body):
        if l.startswith('global_load_lds') or (l.startswith('buffer_load') and ' lds' in l):
            n += 1
            younger = 0  # VM operations issued behind the DMA: a wait vmcnt(N) reaches the DMA iff N <= younger
            for k in range(i + 1, min(i + 400, len(body))):
                b = body[k]
                if b.startswith('v_mfma') or b.startswith('s_barrier'): break
                if re.match(r'(buffer|global|scratch|flat)_(load|store|atomic)', b): younger += 1
                w = re.search(r'vmcnt\((\d+)\)', b) if b.startswith('s_waitcnt') else None
                if w and int(w.group(1)) <= younger:
                    drained += 1; break
    print(f"{m.group(1)[:90]:90s} LDS-DMA instructions {n:3d}, followed by a vmcnt wait before the next MFMA / barrier: {drained}")
