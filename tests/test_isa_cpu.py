"""Static checks on the compiled gfx950 code (no GPU: hipcc cross-compiles): properties of the hot kernels that no
numerical test can see and one compiler decision can silently undo."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lion_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _build_flags():
    """the compile flags of the product build, read from csrc/build.sh (one source of truth)"""
    txt = open(os.path.join(CSRC, "build.sh")).read()
    flags = re.search(r'^FLAGS="([^"]+)"', txt, re.M).group(1).split()
    return [f.replace("../../include", os.path.join(ROOT, "include")) for f in flags if f not in ("-fPIC",)]


def _listing(src, tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / (src + ".s"))
    subprocess.check_call([HIPCC] + _build_flags() + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(listing, name):
    for m in re.finditer(r'^(\S*' + re.escape(name) + r'\S*):', listing, re.M):
        end = re.compile(r'^\.Lfunc_end\d+:', re.M).search(listing, m.end()).start()
        body = [ln.strip().split(';')[0].strip() for ln in listing[m.end():end].split('\n')]
        yield m.group(1), [ln for ln in body if ln]


def _drained(body):
    """LDS-DMA instructions followed, before the next MFMA / barrier, by a vmcnt wait that reaches them (the rule of
    tools/dma_drain_check.py: wait vmcnt(N) reaches the DMA iff N <= VM operations issued behind it).  Returns
    (DMA instructions, drained ones, DMA instructions whose next MFMA comes before the next barrier, drained ones of those)"""
    n = drained = n_mfma = drained_mfma = 0
    for i, ln in enumerate(body):
        if ln.startswith('global_load_lds'):
            n += 1
            younger = 0
            hit = False
            before_mfma = False
            for b in body[i + 1:i + 400]:
                if b.startswith('v_mfma'):
                    before_mfma = True
                    break
                if b.startswith('s_barrier'):
                    break
                if re.match(r'(buffer|global|scratch|flat)_(load|store|atomic)', b):
                    younger += 1
                w = re.search(r'vmcnt\((\d+)\)', b) if b.startswith('s_waitcnt') else None
                if w and int(w.group(1)) <= younger and not hit:
                    hit = True
            drained += hit
            n_mfma += before_mfma
            drained_mfma += hit and before_mfma
    return n, drained, n_mfma, drained_mfma


@pytest.fixture(scope="module")
def conv_split_listing(tmp_path_factory):
    return _listing("conv3d_split.hip", tmp_path_factory)


def test_weight_dma_of_the_split_convolution_is_not_drained(conv_split_listing):
    """Round 2: with the DMA issued by inline asm the compiler's vmcnt(0) for its scratch reloads also waited for the
    DMA -- 27 of the 30 DMA instructions of conv3d_split_kernel were drained before the first MFMA of their tap group.
    Issued through the builtin the compiler counts them.  Round 3: the kernel carries two copies of the K walk (working
    waves / waves whose block holds no point), 27 DMA instructions each + 6 in the item prologue; in the working copy every
    group's DMA is issued behind barrier k, in front of the MFMAs of the group's last tap, and none may be waited for
    before those MFMAs (the idle copy has nothing between its DMA and the next barrier's wait: drained by construction)."""
    seen = 0
    for name, body in _kernels(conv_split_listing, "conv3d_split_kernelILi"):
        n, drained, n_mfma, drained_mfma = _drained(body)
        assert n == 60 and 27 <= n_mfma <= 33 and drained_mfma == 0, (name, n, drained, n_mfma, drained_mfma)
        seen += 1
    assert seen == 16
    for name, body in _kernels(conv_split_listing, "conv3d_split_pipe_kernel"):
        n, drained, _, _ = _drained(body)
        assert n > 0 and drained == 0, (name, n, drained)


def test_tap_loops_do_not_touch_scratch_between_mfmas(conv_split_listing):
    """spills are allowed around the staging, not inside a tap group: between the first and the last MFMA of a group there
    must be no scratch access (each one is a memory round trip in front of the matrix pipe)"""
    for name, body in _kernels(conv_split_listing, "conv3d_split_pipe_kernel"):
        assert not any(ln.startswith('scratch_') for ln in body), name          # the r = 8 kernel has no spills at all
    for name, body in _kernels(conv_split_listing, "conv3d_split_kernelILi"):
        # round 3: fragments are read a tap ahead and barrier k sits in front of the LAST tap of group k, so the rule is
        # simply: no scratch access between the first and the last MFMA of the kernel's (single, straight-line) tap walk
        mf = [i for i, ln in enumerate(body) if ln.startswith('v_mfma')]
        cb = int(re.search(r'kernelILi\d+ELi\d+ELi\d+ELi(\d+)E', name).group(1))
        assert len(mf) == 27 * 3 * cb * 2, (name, len(mf))
        bad = [ln for ln in body[mf[0]:mf[-1]] if ln.startswith('scratch_')]
        assert not bad, (name, len(bad))


def test_epilogue_stores_are_not_serialized_by_reload_waits(conv_split_listing):
    """Round 2: with the output computed and stored in one loop the compiler reloaded spilled values between the stores
    and waited for every reload with vmcnt(0|1) -- which also waits for the stores before it: 18-23 store / wait / store
    sequences (each a round trip to memory) per epilogue.  With a compute pass and a store pass there are none."""
    for name, body in list(_kernels(conv_split_listing, "conv3d_split_kernelILi")) + \
            list(_kernels(conv_split_listing, "conv3d_split_pipe_kernel")):
        seq = state = 0
        for ln in body:
            if re.match(r'(global|buffer|flat)_store', ln):
                if state == 2:
                    seq += 1
                state = 1
            elif ln.startswith('s_waitcnt') and re.search(r'vmcnt\((0|1)\)', ln) and state == 1:
                state = 2
            elif ln.startswith('s_barrier') or ln.startswith('s_endpgm'):
                state = 0
        assert seq <= 3, (name, seq)


def test_no_packed_fp32_arithmetic_in_the_library(tmp_path_factory):
    """Round 3: a wave that executes packed fp32 VALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
    while sharing its SIMD with the fp16-MFMA stream of conv3d_split_kernel computes wrong results -- one float2
    expression in fps_reg_kernel made 37-40 of 40 graph replays return wrong samples, the same kernel without it none of
    100+.  The SLP vectoriser introduced 116 of them in sampling.hip alone; the library is built with
    -fno-slp-vectorize and no kernel may contain the instruction class."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    assert "-fno-slp-vectorize" in _build_flags()
    tmp = tmp_path_factory.mktemp("isa_all")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    procs = [(f, subprocess.Popen([HIPCC] + _build_flags() + ["-S", "--cuda-device-only", os.path.join(CSRC, f), "-o",
                                                              str(tmp / (f + ".s"))], cwd=CSRC, stderr=subprocess.DEVNULL))
             for f in srcs]
    bad = {}
    for f, pr in procs:
        assert pr.wait() == 0, f
        n = len(re.findall(r'^\s*v_pk_(add|mul|fma)_f32', open(tmp / (f + ".s")).read(), re.M))
        if n:
            bad[f] = n
    assert not bad, bad


def test_1024_point_voxelize_kernels_keep_two_workgroups_per_cu(tmp_path_factory):
    """Round 5: clouds of <= 1024 points run one point per thread in 1024-thread workgroups, TWO per
    CU on half the LDS each (voxelize.hip::make_plan, two_per_cu) -- 32 waves per CU = 8 per SIMD, which the register file
    grants only up to 64 VGPRs (and 96 SGPRs) per wave.  The chunk-adoption block of the scatter kernel and a 16-deep
    prefetch of the ordered sums each pushed the NP = 1 instantiation past that ((128, 1024, 16): 31 -> 38 us,
    profiles/archive/r05b_scatter_adoption_ab.txt); both are compiled for NP >= 2 only.  The compiler's own occupancy figure
    of every NP = 1 instantiation must stay 8, and no voxelize kernel may spill."""
    lst = _listing("voxelize.hip", tmp_path_factory)
    seen = 0
    for m in re.finditer(r'^(\S*(?:vox_scatter_kernelILi1E|vox_fused_kernelILb[01]ELi1E)\S*):', lst, re.M):
        tail = lst[m.end():]
        occ = int(re.search(r'^; Occupancy: (\d+)', tail, re.M).group(1))
        vgpr = int(re.search(r'^; NumVgprs: (\d+)', tail, re.M).group(1))
        assert occ == 8 and vgpr <= 64, (m.group(1), occ, vgpr)
        seen += 1
    assert seen == 4, seen   # scatter <1, reader-aware / plain>, fused <with / without features, 1>
    for m in re.finditer(r'^; ScratchSize: (\d+)', lst, re.M):
        assert int(m.group(1)) == 0


def test_chunk_log_instrumentation_still_applies_to_the_voxelize_source(tmp_path):
    """tools/vox_chunk_log_build.py writes an instrumented copy of csrc/voxelize.hip by anchoring on source lines (the
    product file carries no timing code); a refactor that moves an anchor would silently rot the tool the scatter kernel's
    round-5 findings came from.  The copy must get all five stamps, the log buffer and the reader's entry point."""
    out = os.path.join(ROOT, "tools", "exp", "voxelize_chunk_log.hip")
    existed = os.path.exists(out)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "vox_chunk_log_build.py")], cwd=ROOT)
    try:
        s = open(out).read()
        for k in range(5):
            assert f"vlog[{k}] = wall_clock64();" in s, k
        assert "g_vlog[8192 * 8]" in s and "int lion_debug_vox_log(unsigned long long *host, int reset)" in s
        assert "lion_debug_vox_log" not in open(os.path.join(CSRC, "voxelize.hip")).read()   # the product stays clean
    finally:
        if not existed:
            os.remove(out)
