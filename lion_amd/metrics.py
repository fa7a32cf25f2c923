"""Generation metrics on top of the Chamfer / EMD kernels -- mirror of the callers in the reference's
``utils/evaluation_metrics_fast.py`` (distChamferCUDA :99, emd_approx :123, EMD_CD :184,
_pairwise_EMD_CD_ :321, knn :406, lgan_mmd_cov :448, compute_all_metrics :463).

Differences: a pair matrix is evaluated in large batches (``pair_batch`` pairs per kernel call instead
of one sample against N_ref/2 references), rows of the matrix can be sharded over ranks and
all-gathered (the reference computes everything on rank 0, base_trainer.py:491-495), and M_rs is
computed once (the reference computes it twice, :481-490 / :524-533).  JSD (CPU numpy/sklearn in the
reference) is out of scope."""
from __future__ import annotations

import torch

from .chamfer3d import chamfer_3DDist, chamfer_3DDist_nograd
from .emd import earth_mover_distance, earth_mover_distance_nograd


def distChamferCUDA(x, y):
    assert x.dim() == 3 and y.dim() == 3 and x.shape[-1] == 3 and y.shape[-1] == 3
    d1, d2, _, _ = chamfer_3DDist()(x.cuda(), y.cuda())
    return d1, d2


def distChamferCUDAnograd(x, y):
    assert x.shape[-1] == 3 and y.shape[-1] == 3, f'get {x.shape} and {y.shape}'
    d1, d2, _, _ = chamfer_3DDist_nograd()(x.cuda(), y.cuda())
    return d1, d2


def emd_approx(sample, ref, require_grad=True):
    fn = earth_mover_distance if require_grad else earth_mover_distance_nograd
    return fn(sample.cuda(), ref.cuda(), transpose=False)


def EMD_CD(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, reduced=True, require_grad=False):
    """paired CD / EMD between sample_pcs[i] and ref_pcs[i] (reference :184-226)."""
    assert sample_pcs.shape[0] == ref_pcs.shape[0], "REF:%d SMP:%d" % (ref_pcs.shape[0], sample_pcs.shape[0])
    cd_lst, emd_lst = [], []
    for s in range(0, sample_pcs.shape[0], batch_size):
        a, b = sample_pcs[s:s + batch_size].contiguous(), ref_pcs[s:s + batch_size].contiguous()
        dl, dr = (distChamferCUDA if require_grad else distChamferCUDAnograd)(a, b)
        cd_lst.append(dl.mean(dim=1) + dr.mean(dim=1))
        emd_lst.append(emd_approx(a, b, require_grad=require_grad))
    cd, emd = torch.cat(cd_lst), torch.cat(emd_lst)
    if reduced:
        cd, emd = cd.mean(), emd.mean()
    return {'MMD-CD': cd, 'MMD-EMD': emd}


@torch.no_grad()
def pairwise_distance(metric, sample_pcs, ref_pcs, pair_batch=256, rank=0, world=1):
    """[N_sample, N_ref] matrix of CD ('CD') or approximate EMD ('EMD') between every sample and every
    reference cloud (what _pairwise_EMD_CD_ returns, :321-355).  With world > 1 each rank computes a
    contiguous block of rows and the blocks are all-gathered."""
    ns, nr = sample_pcs.shape[0], ref_pcs.shape[0]
    sample_pcs, ref_pcs = sample_pcs.cuda().float().contiguous(), ref_pcs.cuda().float().contiguous()
    lo = ns * rank // world
    hi = ns * (rank + 1) // world
    out = torch.empty((hi - lo, nr), device=sample_pcs.device, dtype=torch.float32)
    ii, jj = torch.meshgrid(torch.arange(lo, hi, device=out.device), torch.arange(nr, device=out.device), indexing="ij")
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    flat = out.view(-1)
    for s in range(0, ii.numel(), pair_batch):
        a = sample_pcs.index_select(0, ii[s:s + pair_batch])
        b = ref_pcs.index_select(0, jj[s:s + pair_batch])
        if metric == 'CD':
            dl, dr = distChamferCUDAnograd(a, b)
            flat[s:s + pair_batch] = dl.mean(dim=1) + dr.mean(dim=1)
        elif metric == 'EMD':
            flat[s:s + pair_batch] = emd_approx(a, b, require_grad=False)
        else:
            raise NotImplementedError(metric)
    if world > 1:
        import torch.distributed as dist
        rows = [torch.empty((ns * (r + 1) // world - ns * r // world, nr), device=out.device) for r in range(world)]
        dist.all_gather(rows, out) if len({t.shape for t in rows}) == 1 else _ragged_gather(rows, out, rank)
        out = torch.cat(rows, dim=0)
    return out


def _ragged_gather(rows, mine, rank):
    import torch.distributed as dist
    for r, buf in enumerate(rows):
        if r == rank:
            buf.copy_(mine)
        dist.broadcast(buf, src=r)


def _pairwise_EMD_CD_(metric, sample_pcs, ref_pcs, batch_size, require_grad=False, accelerated_cd=True,
                      verbose=False):
    m = pairwise_distance(metric, sample_pcs, ref_pcs, pair_batch=max(int(batch_size), 1))
    return m, m


def knn(Mxx, Mxy, Myy, k, sqrt=False):
    """leave-one-out k-NN two-sample test (reference :406-445)."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat([torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.transpose(0, 1), Myy), 1)], 0)
    if sqrt:
        M = M.abs().sqrt()
    val, idx = (M + torch.diag(float('inf') * torch.ones(n0 + n1).to(Mxx))).topk(k, 0, False)
    count = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = torch.ge(count, (float(k) / 2) * torch.ones(n0 + n1).to(Mxx)).float()
    s = {'tp': (pred * label).sum(), 'fp': (pred * (1 - label)).sum(),
         'fn': ((1 - pred) * label).sum(), 'tn': ((1 - pred) * (1 - label)).sum()}
    s.update({'precision': s['tp'] / (s['tp'] + s['fp'] + 1e-10),
              'recall': s['tp'] / (s['tp'] + s['fn'] + 1e-10),
              'acc_t': s['tp'] / (s['tp'] + s['fn'] + 1e-10),
              'acc_f': s['tn'] / (s['tn'] + s['fp'] + 1e-10),
              'acc': torch.eq(label, pred).float().mean()})
    return s


def lgan_mmd_cov(all_dist):
    """all_dist [N_sample, N_ref] -> MMD / COV (reference :448-460)."""
    N_ref = all_dist.size(1)
    min_val_fromsmp, min_idx = torch.min(all_dist, dim=1)
    min_val, _ = torch.min(all_dist, dim=0)
    cov = torch.tensor(float(min_idx.unique().view(-1).size(0)) / float(N_ref)).to(all_dist)
    return {'lgan_mmd': min_val.mean(), 'lgan_cov': cov, 'lgan_mmd_smp': min_val_fromsmp.mean()}


@torch.no_grad()
def compute_all_metrics(sample_pcs, ref_pcs, batch_size=256, verbose=False, accelerated_cd=True,
                        metric1='CD', metric2='EMD', rank=0, world=1, **print_kwargs):
    """MMD / COV / 1-NNA under CD and EMD (reference :463-560); keys are the reference's
    ('lgan_mmd-CD', 'lgan_cov-CD', '1-NN-CD-acc', ... and the same with EMD)."""
    results = {}
    for metric in (metric1, metric2):
        if metric is None:
            continue
        M_rs = pairwise_distance(metric, ref_pcs, sample_pcs, batch_size, rank, world)
        res = lgan_mmd_cov(M_rs.t())
        results.update({'%s-%s' % (k, metric): v.item() for k, v in res.items()})
        M_rr = pairwise_distance(metric, ref_pcs, ref_pcs, batch_size, rank, world)
        M_ss = pairwise_distance(metric, sample_pcs, sample_pcs, batch_size, rank, world)
        one_nn = knn(M_rr, M_rs, M_ss, 1, sqrt=False)
        results.update({"1-NN-%s-%s" % (metric, k): v.item() for k, v in one_nn.items() if 'acc' in k})
    return results


# ---- JSD between the occupancy histograms of two point-cloud sets ------------------------------------------
# (utils/evaluation_metrics_fast.py:563-647, after latent_3d_points; the reference runs a CPU sklearn nearest-
# neighbour search per cloud -- here the whole set is assigned on the device in one batched distance + argmin)

def unit_cube_grid_point_cloud(resolution, clip_sphere=False, device=None):
    """centres of the resolution^3 cells of the unit cube [-0.5, 0.5]^3 (x slowest), optionally only those inside
    the unit-diameter sphere; returns (grid [G,3], spacing)."""
    spacing = 1.0 / float(resolution - 1)
    axis = torch.arange(resolution, dtype=torch.float32, device=device) * spacing - 0.5
    grid = torch.stack(torch.meshgrid(axis, axis, axis, indexing="ij"), dim=-1).reshape(-1, 3)
    if clip_sphere:
        grid = grid[grid.norm(dim=1) <= 0.5]
    return grid, spacing


def occupancy_histogram(pclouds, resolution=28, in_sphere=True, chunk=64):
    """(entropy of the per-cell Bernoulli occupancy, per-cell point counts [G]) of a set of clouds [S,N,3]:
    every point votes for its nearest grid cell (reference :601-640)."""
    pclouds = torch.as_tensor(pclouds, dtype=torch.float32)
    grid, _ = unit_cube_grid_point_cloud(resolution, in_sphere, device=pclouds.device)
    counts = torch.zeros(grid.shape[0], dtype=torch.float64, device=pclouds.device)
    hits = torch.zeros_like(counts)
    g2 = (grid * grid).sum(1)
    for s0 in range(0, pclouds.shape[0], chunk):
        pc = pclouds[s0:s0 + chunk]
        # argmin_g |p - g|^2 = argmin_g (|g|^2 - 2 p.g)
        idx = (g2[None, None, :] - 2.0 * pc @ grid.t()).argmin(dim=-1)          # [s, N]
        counts += torch.bincount(idx.reshape(-1), minlength=grid.shape[0]).double()
        onehot = torch.zeros(pc.shape[0], grid.shape[0], dtype=torch.bool, device=pc.device)
        onehot.scatter_(1, idx, True)
        hits += onehot.sum(0).double()
    p = hits / float(pclouds.shape[0])
    nz = p[(p > 0) & (p < 1)]
    ent = -(nz * nz.log() + (1 - nz) * (1 - nz).log()).sum() / grid.shape[0]
    return float(ent), counts


def jensen_shannon_divergence(P, Q):
    """base-2 JSD of two non-negative histograms (normalised here), reference :643-660."""
    P, Q = torch.as_tensor(P, dtype=torch.float64), torch.as_tensor(Q, dtype=torch.float64)
    if (P < 0).any() or (Q < 0).any():
        raise ValueError("Negative values.")
    if P.numel() != Q.numel():
        raise ValueError("Non equal size.")
    P, Q = P / P.sum(), Q / Q.sum()

    def h(x):
        x = x[x > 0]
        return -(x * x.log2()).sum()
    return float(h(0.5 * (P + Q)) - 0.5 * (h(P) + h(Q)))


def jsd_between_point_cloud_sets(sample_pcs, ref_pcs, resolution=28):
    """JSD between the voxel-occupancy histograms of two sets of clouds in the unit sphere (reference :587-598)."""
    return jensen_shannon_divergence(occupancy_histogram(sample_pcs, resolution, True)[1],
                                     occupancy_histogram(ref_pcs, resolution, True)[1])
