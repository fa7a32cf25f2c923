"""reader of the per-chunk wall-clock log of the voxelize scatter kernel (tools/vox_chunk_log_build.py): launch span, when the\nworkgroups end, a least-squares fit of every phase against (channels, channels x points), and the chunks of cloud 0 and of\nthe cloud that finishes last.  profiles/archive/r05b_scatter_adoption_ab.txt / r05b_scatter_units_wallclock_log.txt came from this."""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import _lib, fused_ops as fo
from lion_amd.functional.backend import _backend as bk
lib = _lib.load()
lib.lion_debug_vox_log.restype = ctypes.c_int
lib.lion_debug_vox_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
B = 32
chain = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "chain_clouds.npz"))
buf = np.zeros((8192, 8), dtype=np.uint64)
for C, n, r in ((64, 2048, 32), (128, 1024, 16), (32, 2048, 32)):
    for k in ("gauss", "step_0400"):
        if k == "gauss":
            co = torch.randn(B, 3, n, device="cuda")
        else:
            co = torch.from_numpy(np.ascontiguousarray(chain[k].transpose(0, 2, 1))).cuda().float()[:B]
            m = co.shape[2]
            while m > n:
                m //= 2
                co = bk.gather_features_forward(co, bk.furthest_point_sampling(co, m))
            co = co.contiguous()
        ft = torch.randn(B, C, n, device="cuda")
        plan = bk.voxel_index(co, r, True, 0.0)
        for _ in range(3): bk.voxel_scatter(ft, plan)
        torch.cuda.synchronize(); lib.lion_debug_vox_log(None, 1)
        bk.voxel_scatter(ft, plan); torch.cuda.synchronize()
        nrec = lib.lion_debug_vox_log(buf.ctypes.data, 1)
        rec = buf[:nrec].astype(np.int64)
        t = rec[:, :5] - rec[:, 0].min()
        pts, occ = rec[:, 5] >> 32, rec[:, 5] & 0xffffffff
        ch, slab, b, wg = rec[:, 6] >> 32, (rec[:, 6] >> 16) & 0xffff, rec[:, 6] & 0xffff, rec[:, 7]
        has = pts > 0
        t2 = np.where(has, t[:, 2], t[:, 1]); t3 = t[:, 3]
        d = np.stack([t[:, 1] - t[:, 0], t2 - t[:, 1], t3 - t2, t[:, 4] - t3], 1) / 100.0   # us
        print(f"== C={C} N={n} r={r} {k}: {nrec} units, launch span {t[:, 4].max() / 100.0:.1f} us; starts: {np.percentile(t[:,0],[0,50,100])/100.0}")
        nwg = len(set(wg.tolist()))
        wgend = {}
        for i in range(nrec): wgend[wg[i]] = max(wgend.get(wg[i], 0), t[i, 4])
        ends = np.array(sorted(wgend.values())) / 100.0
        print(f"   workgroup end times us: min {ends.min():.1f} p25 {np.percentile(ends,25):.1f} p50 {np.percentile(ends,50):.1f} p75 {np.percentile(ends,75):.1f} max {ends.max():.1f}  ({nwg} workgroups)")
        # least squares: unit time = a*ch + b*ch*pts + c*(pts>0)*ch + d
        A = np.stack([ch, ch * pts / 1000.0, has * ch, np.ones_like(ch)], 1).astype(np.float64)
        for nm, y in (("setup", d[:, 0]), ("B1", d[:, 1]), ("B2", d[:, 2]), ("C", d[:, 3]), ("total", d.sum(1))):
            co_, *_ = np.linalg.lstsq(A, y, rcond=None)
            print(f"   {nm:6s} mean {y.mean():6.2f} us  fit: {co_[0]:.3f}*ch + {co_[1]:.3f}*ch*kpts + {co_[2]:.3f}*ch*[pts>0] + {co_[3]:.2f}")
        last = b[np.argmax(t[:, 4])]
        for bb in (0, int(last)):
            print(f"   cloud {bb}:")
            for i in np.argsort(t[:, 0] + wg * 1e9):
                if b[i] == bb:
                    print(f"     wg {wg[i]:4d} slab {slab[i]:2d} pts {pts[i]:4d} occ {occ[i]:4d} ch {ch[i]:3d}  start {t[i,0]/100.0:6.1f}  setup {d[i,0]:5.1f} B1 {d[i,1]:5.1f} B2 {d[i,2]:5.1f} C {d[i,3]:5.1f}  end {t[i,4]/100.0:6.1f}")
