/*
 * lion_hip.h -- C ABI of liblion_hip.so: the MI355X (gfx950) implementation of the
 * LION / PVCNN hot path.
 *
 * Every entry point replaces one function of the reference's pybind/ATen operator modules
 * (cited per function as file:line under /root/reference).  Conventions:
 *   - plain device pointers + sizes, no torch types; all tensors C-contiguous, fp32 / int32;
 *   - no allocation, no synchronisation, no global state that results depend on: safe under hipGraph
 *     capture and callable from any host thread, for any device of the process (the device is the
 *     calling thread's current one; the only state kept is a per-device note of which kernels already
 *     had their dynamic-LDS limit raised -- the first call of a kernel on a device makes one
 *     hipFuncSetAttribute call, so run one call outside a capture first);
 *     scratch memory is passed in (`ws`, size from the matching lion_*_workspace_bytes());
 *   - every launch goes to the caller's `stream` (the reference mixes the legacy default stream
 *     and the current stream; here it is always the caller's stream);
 *   - outputs are fully overwritten (callers may pass torch.empty buffers) unless a parameter
 *     is documented "pre-zeroed";
 *   - return 0 on success, a positive hipError_t if a launch failed, or a negative LION_E* code
 *     for a rejected argument.  Nothing ever calls exit() (the reference does: cuda_utils.cuh:28-37).
 */
#ifndef LION_HIP_H_
#define LION_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *lionStream_t; /* hipStream_t */

#define LION_OK 0
#define LION_EINVAL (-1)      /* bad shape / null pointer */
#define LION_EUNSUPPORTED (-2) /* shape outside what the kernels implement */
#define LION_EWORKSPACE (-3)  /* workspace too small */

/* Library / device info. */
int lion_abi_version(void);

/* ---- K1+K2: avg_voxelize_forward, voxelization/vox.cpp:17-43 (vox.cu:18-72) -------------
 * feat f32[B,C,N], coords i32[B,3,N] -> out f32[B,C,r^3], ind i32[B,N], cnt i32[B,r^3].
 * Deterministic: each voxel sums its points in ascending point index (bit-exact vs oracle). */
size_t lion_avg_voxelize_workspace_bytes(int B, int C, int N, int r);
int lion_avg_voxelize_forward(const float *feat, const int32_t *coords, int B, int C, int N,
                              int r, float *out, int32_t *ind, int32_t *cnt, void *ws,
                              size_t ws_bytes, lionStream_t stream);

/* ---- P1+K1+K2 fused: Voxelization.forward, models/pvcnn2_ada.py:173-188 -----------------
 * coords f32[B,3,N] (raw point coords) -> norm_coords f32[B,3,N] (voxel units, clamped) and
 * everything lion_avg_voxelize_forward returns.  feat may be NULL (C ignored): only
 * norm_coords / ind / cnt are produced (pvcnn2_ada.py:185-186).
 * The coordinate mean uses the fixed summation tree documented in oracle/lion_oracle.c. */
int lion_voxelize_points_forward(const float *feat, const float *coords, int B, int C, int N,
                                 int r, int normalize, float eps, float *out,
                                 float *norm_coords, int32_t *ind, int32_t *cnt, void *ws,
                                 size_t ws_bytes, lionStream_t stream);

/* ---- P1+K1 once, K2 many times: the index plan of a (cloud, resolution) pair ----------------
 * One forward of the denoiser voxelises 4 distinct (coordinates, r) pairs 14 times (PVConv.forward,
 * models/pvcnn2_ada.py:235-243: every PVConv of a stage re-derives the same voxel ids from the same
 * coordinates).  lion_voxel_index computes what depends on the coordinates only -- norm_coords, ind, cnt
 * exactly as lion_voxelize_points_forward, plus the points of every voxel in ascending point index
 * (an opaque plan of lion_voxel_plan_bytes bytes, 0 = shape outside the fast path: use the fused entry
 * point) -- and lion_voxel_scatter mean-pools feat f32[B,C,N] into out f32[B,C,r^3] from that plan:
 * same sums in the same order, bit-identical to lion_voxelize_points_forward (vox.cu:48-72). */
size_t lion_voxel_plan_bytes(int B, int N, int r);
int lion_voxel_index(const float *coords, int B, int N, int r, int normalize, float eps,
                     float *norm_coords, int32_t *ind, int32_t *cnt, void *plan, size_t plan_bytes,
                     lionStream_t stream);
int lion_voxel_scatter(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N,
                       int r, float *out, lionStream_t stream);
/* Round 5: the same scatter for a grid whose ONLY reader is the sparse convolution that pops occ_m1
 * (lion_conv3d_tile_occupancy[_aware], margin 1; r in {16, 32}) -- conv1 of a PVConv in the fused inference path
 * (pvcnn2_ada.py:206-222).  That convolution stages the halo of the tiles with a point within one voxel and nothing
 * else: z-rows of the grid outside every such halo are NOT written (left as allocated; their content is all zeros in the
 * full scatter and has no reader). */
int lion_voxel_scatter_read(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int r,
                            const int32_t *occ_m1, float *out, lionStream_t stream);

/* ---- K3: avg_voxelize_backward, vox.cpp:54-79 (vox.cu:86-110) --------------------------
 * gy f32[B,C,r3], ind i32[B,N], cnt i32[B,r3] -> gx f32[B,C,N]. */
int lion_avg_voxelize_backward(const float *gy, const int32_t *ind, const int32_t *cnt, int B,
                               int C, int N, int r3, float *gx, lionStream_t stream);

/* ---- K4: trilinear_devoxelize_forward, interpolate/trilinear_devox.cpp:18-55 -------------
 * coords f32[B,3,N] (voxel units in [0,r-1]), feat f32[B,C,r^3] -> out f32[B,C,N];
 * training != 0 also writes inds i32[B,8,N], wgts f32[B,8,N] (else they may be NULL). */
int lion_trilinear_devoxelize_forward(const float *coords, const float *feat, int B, int C,
                                      int N, int r, int training, float *out, int32_t *inds,
                                      float *wgts, lionStream_t stream);

/* ---- K5: trilinear_devoxelize_backward, trilinear_devox.cpp:67-95 -------------------------
 * gy f32[B,C,N], inds i32[B,8,N], wgts f32[B,8,N] -> gx f32[B,C,r3] (fully written). */
int lion_trilinear_devoxelize_backward(const float *gy, const int32_t *inds, const float *wgts,
                                       int B, int C, int N, int r3, float *gx,
                                       lionStream_t stream);

/* ---- K6: ball_query, ball_query/ball_query.cpp:7-33 (ball_query.cu:19-50) ----------------
 * centers f32[B,3,M], points f32[B,3,N] -> idx i32[B,M,U]: the first U points (ascending
 * index) with d^2 < radius^2, padded with the first hit; all zero when there is none. */
int lion_ball_query(const float *centers, const float *points, int B, int M, int N,
                    float radius, int U, int32_t *idx, lionStream_t stream);

/* ---- K7/K8: grouping, grouping/grouping.cu:18-36 / :58-77 --------------------------------
 * feat f32[B,C,N], idx i32[B,M,U] -> out f32[B,C,M,U];  gy f32[B,C,M,U] -> gx f32[B,C,N]. */
int lion_grouping_forward(const float *feat, const int32_t *idx, int B, int C, int N, int M,
                          int U, float *out, lionStream_t stream);
/* BallQuery.forward (models/pvcnn2_ada.py:98-114) without the subtraction pass and the torch.cat of the grouped
 * activation: out f32[B, 3 + C, M, U], channels 0..2 = coords[b,:,idx[b,m,u]] - centers[b,:,m], 3.. = feat[b,:,idx]
 * (feat NULL with C = 0: coordinates only).  coords f32[B,3,N], centers f32[B,3,M], feat f32[B,C,N], idx i32[B,M,U]. */
int lion_group_points_forward(const float *coords, const float *centers, const float *feat, const int32_t *idx,
                              int B, int C, int N, int M, int U, float *out, lionStream_t stream);
int lion_grouping_backward(const float *gy, const int32_t *idx, int B, int C, int N, int M,
                           int U, float *gx, lionStream_t stream);

/* ---- K9: furthest_point_sampling, sampling/sampling.cpp:43-58 (sampling.cu:86-167) -------
 * coords f32[B,3,N] -> idx i32[B,M]; idx[.,0] = 0; ties resolved exactly as the reference's
 * 512-thread reduction does (lowest (k mod 512, k)). */
int lion_furthest_point_sampling(const float *coords, int B, int N, int M, int32_t *idx,
                                 lionStream_t stream);

/* ---- K10: gather_features, sampling/sampling.cu:17-31 / :52-66 --------------------------- */
int lion_gather_features_forward(const float *feat, const int32_t *idx, int B, int C, int N,
                                 int M, float *out, lionStream_t stream);
int lion_gather_features_backward(const float *gy, const int32_t *idx, int B, int C, int N,
                                  int M, float *gx, lionStream_t stream);

/* ---- K11+K12: three_nearest_neighbors_interpolate, interpolate/neighbor_interpolate.cu ----
 * points f32[B,3,N], centers f32[B,3,M], cfeat f32[B,C,M] ->
 * out f32[B,C,N], idx i32[B,3,N], wgt f32[B,3,N]   (:20-75 then :90-116);
 * backward gy f32[B,C,N] -> gx f32[B,C,M]           (:145-170). */
int lion_three_nn_interpolate_forward(const float *points, const float *centers,
                                      const float *cfeat, int B, int C, int N, int M,
                                      float *out, int32_t *idx, float *wgt,
                                      lionStream_t stream);
/* PointNetFPModule.forward (pvcnn2_ada.py:403-411) with its two torch.cat folded in: out f32[B][C1+C2+C3][N] = [interp(cfeat
 * f32[B][C1][M]) ; interp(temb[b*ld_t + c] taken as a [C2][M] constant row) ; skip f32[B][C3][N]] (C2 / C3 may be 0).  Same
 * three products and two sums per value as lion_three_nn_interpolate_forward on the materialised concatenation. */
int lion_three_nn_interpolate_cat_forward(const float *points, const float *centers, const float *cfeat, const float *temb,
                                          int ld_t, const float *skip, int B, int C1, int C2, int C3, int N, int M,
                                          float *out, int32_t *idx, float *wgt, lionStream_t stream);
int lion_three_nn_interpolate_backward(const float *gy, const int32_t *idx, const float *wgt,
                                       int B, int C, int N, int M, float *gx,
                                       lionStream_t stream);

/* ---- E1: chamfer_3D.forward / .backward, chamfer3D/chamfer_cuda.cpp:17-33 ----------------
 * xyz1 f32[B,N,3], xyz2 f32[B,M,3] (point-major) -> dist1 f32[B,N], idx1 i32[B,N],
 * dist2 f32[B,M], idx2 i32[B,M]; lowest index wins ties (chamfer3D.cu:36,126).
 * backward: gxyz1 f32[B,N,3], gxyz2 f32[B,M,3] fully written (chamfer3D.cu:155-185). */
int lion_chamfer_forward(const float *xyz1, const float *xyz2, int B, int N, int M,
                         float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                         lionStream_t stream);
int lion_chamfer_backward(const float *xyz1, const float *xyz2, const float *gdist1,
                          const float *gdist2, const int32_t *idx1, const int32_t *idx2, int B,
                          int N, int M, float *gxyz1, float *gxyz2, lionStream_t stream);

/* ---- E2: emd_ext.approxmatch_forward / matchcost_forward / matchcost_backward ------------
 * PyTorchEMD/cuda/emd.cpp:7-27 (emd_kernel.cu:24-156, :199-241, :285-353).
 * xyz1 f32[B,N,3], xyz2 f32[B,M,3] -> match f32[B,M,N]; cost f32[B]. */
size_t lion_emd_workspace_bytes(int B, int N, int M);
int lion_emd_approxmatch(const float *xyz1, const float *xyz2, int B, int N, int M,
                         float *match, void *ws, size_t ws_bytes, lionStream_t stream);
/* evaluation path (emd_nograd.py:19-44): approxmatch + matchcost without the [B,N,M] match matrix -> cost f32[B] */
int lion_emd_cost(const float *xyz1, const float *xyz2, int B, int N, int M, float *cost, void *ws,
                  size_t ws_bytes, lionStream_t stream);
int lion_emd_matchcost(const float *xyz1, const float *xyz2, const float *match, int B, int N,
                       int M, float *cost, void *ws, size_t ws_bytes, lionStream_t stream);
int lion_emd_matchcost_backward(const float *grad_cost, const float *xyz1, const float *xyz2,
                                const float *match, int B, int N, int M, float *grad1,
                                float *grad2, lionStream_t stream);

/* ---- D1: one denoiser update, utils/diffusion_pvd.py -------------------------------------
 * DDIM (:451-467): out = x*s + (c*eps + sigma*z).            z may be NULL when sigma == 0.
 * DDPM (:283-296 + get_q_posterior_mean :475-486):
 *   t>0 : out = k_outer*(x - k_a*eps/k_b) + scale*z*temp
 *   t==0: out = k_outer*(x - k_a*eps)
 * out may alias x. */
int lion_ddim_update(const float *x, const float *eps, const float *z, size_t numel, float s,
                     float c, float sigma, float *out, lionStream_t stream);
int lion_ddpm_update(const float *x, const float *eps, const float *z, size_t numel,
                     int t_is_zero, float k_outer, float k_a, float k_b, float scale,
                     float temp, float *out, lionStream_t stream);
/* The same updates for a chain that is replayed from ONE captured graph (whole-step capture, noise
 * on the device): the host supplies nothing per step.
 *   table f32[n_steps][8] = {t_model, a0..a5, -} per step, counter i32[1] (device), cur f32[8] (device).
 * lion_chain_begin_step: i = clamp(*counter); t_out[0..B) = table[i][0]; cur[0..7) = table[i][0..7),
 *   cur[7] = bits(i); *counter = i + 1.
 * lion_chain_update_noise: z ~ N(0,1) drawn in the kernel (Philox4x32-10, key = the 64-bit seed in device memory, counter = (element/4,
 *   cur step, stream_id); Box-Muller) replacing the reference's per-step CPU randn + H2D copy
 *   (diffusion_pvd.py:465-466);  mode 0 (DDIM): out = x*a0 + (a1*eps + a2*z);
 *   mode 1 (DDPM): a5 != 0: out = a0*(x - a1*eps), else out = a0*(x - a1*eps/a2) + (a3*z)*a4.
 *   z_out (may be NULL) receives the noise that was used (tests / trajectories).  out may alias x. */
int lion_chain_begin_step(const float *table, int n_steps, int32_t *counter, float *t_out, int B, float *cur,
                          lionStream_t stream);
int lion_chain_update_noise(int mode, const float *x, const float *eps, size_t numel, const float *cur,
                            const uint32_t *seed /* device u32[2]: lo, hi */, uint32_t stream_id, float *out,
                            float *z_out, lionStream_t stream);
/* Round 6 -- no ATen kernel inside a captured step.  lion_chain_begin_step_temb additionally copies row i of the chain's
 * time-embedding table temb_table f32[n_steps][temb_width] to temb_out (was an index_select per step).
 * lion_chain_update_noise_cm: x / out f32[B][N][4] (the point-major latent of the local prior, latent_dim 1), eps_cm the
 * denoiser's CHANNEL-major output f32[B][4][N] -- the permute(0, 2, 1).contiguous() of
 * models/latent_points_ada_localprior.py:84 folded into the update's read; same arithmetic, same Philox counters. */
int lion_chain_begin_step_temb(const float *table, int n_steps, int32_t *counter, float *t_out, int B, float *cur,
                               const float *temb_table, int temb_width, float *temb_out, lionStream_t stream);
int lion_chain_update_noise_cm(int mode, const float *x, const float *eps_cm, int B, int N, const float *cur,
                               const uint32_t *seed, uint32_t stream_id, float *out, float *z_out, lionStream_t stream);
/* x f32[B][N][D] (point-major latent; 3 <= D <= 8) -> channel-major all f32[B][D][N], coords f32[B][3][N] (rows 0-2),
 * rest f32[B][D-3][N]; each output may be NULL: the view / permute / slice / contiguous head of
 * models/latent_points_ada_localprior.py:72-84 + models/latent_points_ada.py:118-121 in one launch. */
int lion_latent_unpack(const float *x, int B, int N, int D, float *all, float *coords, float *rest, lionStream_t stream);
/* out f32[B][Ca+Ct][N] = cat(a f32[B][Ca][N], t[b*ld_t + c] broadcast along N): torch.cat([features, temb], dim=1) of
 * models/latent_points_ada.py:139,147 (temb = the time embedding expanded over the points).  N % 4 == 0. */
int lion_concat_broadcast(const float *a, const float *t, int B, int Ca, int Ct, int N, int ld_t, float *out,
                          lionStream_t stream);

/* ---- C3: nn.Conv3d(kernel 3, stride 1, padding 1) of PVConv, models/pvcnn2_ada.py:211-222 --------
 * fp32-input MFMA implicit GEMM (exact fp32).  Weights are re-packed once per weight tensor:
 * w f32[Cout,Cin,3,3,3] -> wp f32[ceil4(Cin),27,Cout] (lion_conv3d_packed_floats floats).
 * x f32[B,Cin,r,r,r] with Cin % 4 == 0, r in {8,16,32}, Cout % 32 == 0 -> y f32[B,Cout,r,r,r];
 * bias f32[Cout] or NULL.  Other shapes: LION_EUNSUPPORTED (callers keep the library convolution). */
size_t lion_conv3d_packed_floats(int Cout, int Cin);
int lion_conv3d_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream);
int lion_conv3d_k3_forward(const float *x, const float *wp, const float *bias, int B, int Cin,
                           int Cout, int r, float *y, lionStream_t stream);
/* training: the data gradient is lion_conv3d_k3_forward(gy, pack(mirrored, channel-swapped w)); the weight gradient
 * gw f32[Cout,Cin,3,3,3] = sum_{b,v} gy * shifted x runs on the same MFMA instruction with the voxels on the k axis
 * (ws: lion_conv3d_wgrad_workspace_floats floats of partial sums, reduced in a fixed order). */
size_t lion_conv3d_wgrad_workspace_floats(int B, int Cin, int Cout, int r);
int lion_conv3d_k3_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int r, float *gw, float *ws,
                         size_t ws_floats, lionStream_t stream);
/* the same gradient on the 16-bit matrix pipe at fp32 accuracy (round 4): both operands cut into fp16 pairs in registers
 * (one power-of-two scale per tensor), v_mfma_f32_32x32x16_f16 with 16 voxels on K, fp32 accumulation; same arguments and
 * workspace as lion_conv3d_k3_wgrad; Cin % 8 == 0, x / gy 16-byte aligned (LION_EUNSUPPORTED otherwise). */
int lion_conv3d_k3_wgrad_split(const float *x, const float *gy, int B, int Cin, int Cout, int r, float *gw, float *ws,
                               size_t ws_floats, lionStream_t stream);
/* TRAINING, round 6: the same gradient when x is a freshly voxelised grid (the first Conv3d of a PVConv, models/pvcnn2_ada.py:211):
 * cnt i32[B, r^3] = the voxelisation's per-voxel point counts (16-byte aligned).  Tiles without a point within one voxel add nothing
 * and are neither loaded nor multiplied (a one-workgroup-per-sample kernel derives the tile map first).  Without counts the split
 * kernel still skips the arithmetic of a tile whose staged window turns out to be all zero.  Workspace:
 * lion_conv3d_wgrad_sparse_workspace_floats (the dense size + the tile map). */
size_t lion_conv3d_wgrad_sparse_workspace_floats(int B, int Cin, int Cout, int r);
int lion_conv3d_k3_wgrad_split_sparse(const float *x, const float *gy, const int32_t *cnt, int B, int Cin, int Cout, int r,
                                      float *gw, float *ws, size_t ws_floats, lionStream_t stream);

/* ---- P2+P3+P4 folded into C3 / K4 (inference): PVConv.forward voxel branch, pvcnn2_ada.py:211-226 ----
 * conv -> AdaGN -> Swish -> conv -> AdaGN -> SE3d -> devoxelize without a single stand-alone pass over
 * the grid: the conv epilogue emits per-tile channel sums (stats f32[B,Cout,T,2], T =
 * lion_conv3d_stat_tiles(r, Cout, B, sparse): the spatial tiling depends on the shape), lion_groupnorm_fold turns them into per-(batch, channel) scalars
 * A, Bs (GroupNorm(G) x adaptive affine fac/gbias, models/adagn.py:61-64; fac/gbias rows are ld_fg floats
 * apart, so the two halves of the [B,2C] style projection are consumed in place) and the channel mean,
 * the next conv applies swish(x*A+Bs) while staging its input (pro_a/pro_b f32[B,Cin]), and
 * lion_trilinear_devoxelize_affine_forward interpolates scale*feat+shift (second AdaGN x SE gate). */
int lion_conv3d_stat_tiles(int r, int Cout, int B, int sparse); /* sparse = 1: the call will be given an occ */
int lion_conv3d_k3_fused_forward(const float *x, const float *wp, const float *bias, int B, int Cin,
                                 int Cout, int r, const float *pro_a, const float *pro_b, const float *pro_bias,
                                 const float *tconst, float *y, float *stats, int32_t *occ, lionStream_t stream);
/* The first convolution of a PVConv reads the voxelised grid (>= 94 % zeros); spatial tiles whose whole halo is
 * empty produce exactly bias.  occ i32[lion_conv3d_occupancy_ints(r,Cout,B)] (per-tile flags: 1 = some voxel of
 * the tile's halo holds a point; a work list, occupied tiles first; a queue counter) is derived from the
 * voxelisation's cnt i32[B,r^3]; passing it to a fused forward (only without pro_a/pro_b) skips the K loop of
 * empty tiles and balances the occupied ones over all CUs through the queue -- bit-identical output.  The queue
 * re-arms itself when the launch's last workgroup leaves (round 5): one buffer serves any number of launches in
 * stream order (never two concurrently).
 * Layout: [B*tiles flags][B*tiles work list][queue counter, exit counter, mode, pad].  Flag bits 0-3 = wave w's
 * 64-voxel block of the tile has a point within the margin (0 = empty tile), bit 8 = the tile's output has a reader
 * (consumer-aware buffers), bit 9 (margin-2 words) = occupied at margin 1.  These are the only words the shipped
 * kernels read (round 5's active-voxel bit maps belonged to the rejected compaction experiment and are gone). */
size_t lion_conv3d_occupancy_ints(int r, int Cout, int B);
/* The second convolution sees swish(AdaGN(conv1)): a per-channel constant c = swish(A*bias1+Bs) wherever conv1 saw no
 * point, plus a sparse delta.  lion_conv3d_const_response turns c into tconst f32[B][27][Cout], the exact response of
 * the convolution to the constant field for each of the 27 border configurations (wsum f32[27][Cin][Cout] = weights
 * summed over the in-grid taps); with pro_bias (= bias1) and tconst the fused forward convolves only the delta, adds
 * tconst in the epilogue, and may take an occ built with margin 2 (within fp32 rounding of the dense evaluation). */
int lion_conv3d_const_response(const float *wsum, const float *bias2, const float *bias1, const float *pro_a,
                               const float *pro_b, int B, int Cin, int Cout, float *tconst, lionStream_t stream);
int lion_conv3d_tile_occupancy(const int32_t *cnt, int B, int r, int Cout, int32_t *occ_m1, int32_t *occ_m2,
                               lionStream_t stream); /* margin 1 / margin 2 lists; either may be NULL; r in {16, 32} */
/* Round 5: the same with consumer_aware != 0 for the pair of a PVConv (pvcnn2_ada.py:206-233: conv on the voxelised grid
 * -> delta conv -> trilinear devoxelisation).  Nobody reads the first conv's output outside the tiles the second one
 * stages (a point within 2 voxels, and their neighbours), nor the second one's outside the tiles around the points: a
 * split-kernel launch that pops such a buffer writes NO output for an empty tile without a reader (its GroupNorm sums
 * are still produced, in closed form).  The rest of y is left as allocated.  With consumer_aware == 0 (and through
 * lion_conv3d_tile_occupancy) every voxel of y is written, as before. */
int lion_conv3d_tile_occupancy_aware(const int32_t *cnt, int B, int r, int Cout, int32_t *occ_m1, int32_t *occ_m2,
                                     int consumer_aware, lionStream_t stream);
int lion_groupnorm_fold(const float *stats, int B, int C, int T, int G, int voxels, const float *gamma,
                        const float *beta, const float *fac, const float *gbias, int ld_fg, float eps,
                        float *A, float *Bs, float *chmean, lionStream_t stream);
/* ---- C3 on the 16-bit matrix pipe at fp32 accuracy (csrc/conv3d_split.hip) -----------------------------------
 * The same convolution and the same modes as lion_conv3d_k3_fused_forward with every fp32 operand cut into two fp16
 * pieces (a = a_h + a_l/2048; main += W_h X_h, corr += W_h X_l + W_l X_h in fp32 accumulators; 3 MFMAs of
 * v_mfma_f32_32x32x16_f16 per K = 16 instead of 8 fp32 MFMAs) and block-scaled by exact powers of two (one scale per
 * weight tensor, one monotone scale per workgroup tile for the activations): error vs float64 of the fp32 kernel's
 * class (2.6e-7 rms), no clamp, any fp32 range.  Cin % 16 == 0, Cout % 32 == 0, r in {8,16,32}; otherwise
 * LION_EUNSUPPORTED (callers keep the fp32 kernel).
 *   wp: lion_conv3d_split_packed_halfs(Cout, Cin) uint16 (16-byte aligned), filled by lion_conv3d_split_pack_weights
 *       (3 small launches: max |w|, scale, cut);
 *   stats f32[B,Cout,lion_conv3d_split_stat_tiles(r,Cout),2]; occ / tconst exactly as for the fp32 kernel's sparse
 *   plan (lion_conv3d_tile_occupancy, lion_conv3d_const_response). */
size_t lion_conv3d_split_packed_halfs(int Cout, int Cin);
int lion_conv3d_split_pack_weights(const float *w, int Cout, int Cin, uint16_t *wp, lionStream_t stream);
int lion_conv3d_split_stat_tiles(int r, int Cout);
int lion_conv3d_k3_split_forward(const float *x, const uint16_t *wp, const float *bias, int B, int Cin, int Cout,
                                 int r, const float *pro_a, const float *pro_b, const float *pro_bias,
                                 const float *tconst, float *y, float *stats, int32_t *occ, lionStream_t stream);
/* ---- D2: global denoiser, models/score_sde/resnet.py:60-90, :195-218 -----------------------------------
 * The 1x1 convs of the [B, C, 1, 1] style-latent network as 32-row GEMMs on channel-major activations
 * f32[nb][C][32] (batch padded to 32 per slab), split over K into lion_skinny_splits workgroup rows that write raw
 * partial tiles; the consumer's operand load sums the partials and applies the producer's bias / ReLU (+ an added
 * tensor: ResBlockSEDrop's x + t), lion_skinny_finish applies the squeeze-excite tail or the last layer's bias.
 * wp = lion_skinny_pack_weights(w f32[Cout,Cin]) (lion_skinny_packed_floats floats, tile-major, 4 k-steps per
 * 16-byte lane load).  Cout % 32 == 0. */
size_t lion_skinny_packed_floats(int Cout, int Cin);
int lion_skinny_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream);
int lion_skinny_splits(int Cin, int Cout);
int lion_skinny_gemm(const float *pin, int ks_in, const float *bias_in, int act_in, const float *addT,
                     const float *wp, int nb, int Cin, int Cout, float *pout, lionStream_t stream);
int lion_skinny_finish(const float *A, int ks_a, const float *bias_a, const float *Bp, int ks_b, const float *resid,
                       int nb, int C, int mode, float *y, lionStream_t stream);
/* Round 6: the second squeeze-excite GEMM of a residual block WITH the block's tail (resnet.py:77-86) in its epilogue:
 * y = resid + relu(sum_q A[q] + bias_a) * sigmoid(W act_in(sum pin + bias_in)) == lion_skinny_gemm + lion_skinny_finish(mode 1)
 * bit for bit, one launch.  Needs lion_skinny_splits(Cin, Cout) == 1. */
int lion_skinny_gemm_se_finish(const float *pin, int ks_in, const float *bias_in, int act_in, const float *wp, int nb, int Cin,
                               int Cout, const float *A, int ks_a, const float *bias_a, const float *resid, float *y,
                               lionStream_t stream);
/* activations of the global denoiser [B, C, 1, 1] <-> the channel-major [ceil(B/32)][C][32] form its layers work on
 * (models/score_sde/resnet.py:195-218 keeps [B, C, 1, 1]); a / b: two tensors per launch (either may be NULL), row strides
 * lda / ldb in floats (0 = one row broadcast over the batch: the time embedding of a chain step). */
int lion_to_channel_major(const float *a, int lda, int Ca, float *oa, const float *b, int ldb, int Cb, float *ob, int B,
                          lionStream_t stream);
int lion_from_channel_major(const float *x, int B, int C, float *y, lionStream_t stream);
/* SE3d (pvcnn2_ada.py:27-41) on the folded scalars: A, Bs f32[B,C] are multiplied in place by
 * sigmoid(W2 relu(W1 (A*chmean + Bs))), w1 f32[H,C], w2 f32[C,H] (C <= 1024, H <= 128). */
/* lion_groupnorm_fold + lion_se_gate in one launch (C <= 256, H <= 128): the second AdaGN of a PVConv with its SE3d gate
 * (models/pvcnn2_ada.py:27-41, :219-226). */
int lion_groupnorm_fold_se(const float *stats, int B, int C, int T, int G, int voxels, const float *gamma,
                           const float *beta, const float *fac, const float *gbias, int ld_fg, float eps, const float *w1,
                           const float *w2, int H, float *A, float *Bs, lionStream_t stream);
int lion_se_gate(const float *chmean, const float *w1, const float *w2, int B, int C, int H, float *A,
                 float *Bs, lionStream_t stream);
int lion_trilinear_devoxelize_affine_forward(const float *coords, const float *feat, const float *scale,
                                             const float *shift, int B, int C, int N, int r, float *out,
                                             lionStream_t stream);
/* K4 in two steps (r = 32, N <= 2048; inference): everything the coordinates alone decide -- which 32-byte pieces of the
 * z-rows some point interpolates from (trilinear_devox.cu:36-66), their slots in the kernel's LDS ring, every point's 8
 * corner offsets -- is computed once per cloud (lion_trilinear_devoxelize_plan), and each feature tensor devoxelised at
 * those coordinates starts its first DMA one round trip after the kernel begins (PVConv: four devoxelisations per
 * (cloud, r = 32) and forward, pvcnn2_ada.py:235-243).  plan: lion_devoxelize_plan_bytes(B, N, r) bytes (0 = shape not
 * covered), 16-byte aligned; coords must be the tensor the plan was made from; scale / shift both or neither (the affine
 * form of lion_trilinear_devoxelize_affine_forward).  Bit-identical to the one-step entry points. */
size_t lion_devoxelize_plan_bytes(int B, int N, int r);
int lion_trilinear_devoxelize_plan(const float *coords, int B, int N, int r, void *plan, size_t plan_bytes,
                                   lionStream_t stream);
int lion_trilinear_devoxelize_planned_forward(const void *plan, size_t plan_bytes, const float *coords, const float *feat,
                                              const float *scale, const float *shift, int B, int C, int N, int r,
                                              float *out, lionStream_t stream);

/* ---- P2+P3(+P6) for 1-D / 2-D SharedMLP layers (inference), pvcnn2_ada.py:120-164, :375-377 --------
 * after the 1x1 convolution: row sums -> lion_groupnorm_fold (T = 1) -> y = swish(x*A+Bs), optionally
 * reduced with max over the U neighbours of each centre.  x is [rows = B*C, L] (L = N or M*U). */
int lion_row_stats(const float *x, int rows, int L, float *stats, lionStream_t stream);
/* G1 itself: the 1x1 Conv1d / Conv2d of SharedMLP (pvcnn2_ada.py:120) as an fp32-MFMA GEMM over
 * x f32[B,Cin,L] (L = N or M*U) with the previous layer's AdaGN+Swish applied to the operand in flight
 * (pro_a/pro_b f32[B,Cin] or both NULL) and this layer's GroupNorm sums in the epilogue
 * (stats f32[B,Cout,lion_pwconv_stat_tiles(Cout,Cin,L),2] or NULL; fold with lion_groupnorm_fold).
 * w f32[Cout,Cin] -> wp f32[ceil2(Cin),ceil64(Cout)] once per weight (lion_pwconv_packed_floats floats).
 * Any Cout (channel tiles of 32..256 output rows; rows beyond Cout carry zero weights and are never stored); the
 * weight slice of a tile must fit LDS (150 KiB), otherwise LION_EUNSUPPORTED. */
size_t lion_pwconv_packed_floats(int Cout, int Cin);
int lion_pwconv_pack_weights(const float *w, int Cout, int Cin, float *wp, lionStream_t stream);
int lion_pwconv_stat_tiles(int Cout, int Cin, int L);
int lion_pwconv_forward(const float *x, const float *wp, const float *bias, int B, int Cin, int Cout, int L,
                        const float *pro_a, const float *pro_b, float *y, float *stats, lionStream_t stream);
/* The LAST layer of a set-abstraction MLP without its output (pvcnn2_ada.py:120-164 + :375-377: conv -> AdaGN -> Swish
 * -> max over the U = 32 neighbours of a centre; x f32[B,Cin,M*32]): call once with out_a == NULL -- only the GroupNorm
 * sums are written (stats as lion_pwconv_forward) --, fold them (lion_groupnorm_fold), call again with out_a / out_b
 * f32[B,Cout] = this layer's AdaGN scalars and ymax f32[B,Cout,M]: max_u swish(y*a+b), bit-identical to
 * lion_pwconv_forward + lion_affine_swish_max, while y (268 MB in the first set-abstraction module) is never stored.
 * LION_EUNSUPPORTED outside the large-activation tiling (L % 32 != 0, L <= 4096, small grids). */
int lion_pwconv_forward_max(const float *x, const float *wp, const float *bias, int B, int Cin, int Cout, int L,
                            const float *pro_a, const float *pro_b, const float *out_a, const float *out_b,
                            float *stats, float *ymax, lionStream_t stream);
/* G1 on the 16-bit matrix pipe at fp32 accuracy (csrc/pwconv_split.hip): same contract as lion_pwconv_forward with both
 * operands cut into fp16 hi / lo pieces (power-of-two block scaling per tensor for w, per column and 16-channel chunk for
 * the activation, fp32 accumulation; error within that of the fp32 MFMA chain).  w f32[Cout,Cin] -> wp once per weight
 * (lion_pwconv_split_packed_halfs uint16, 16-byte aligned); stats f32[B,Cout,lion_pwconv_split_stat_tiles(Cout,Cin,L),2].
 * Any Cout, Cin, L with Cin * L < 2^29. */
size_t lion_pwconv_split_packed_halfs(int Cout, int Cin);
int lion_pwconv_split_pack_weights(const float *w, int Cout, int Cin, uint16_t *wp, lionStream_t stream);
/* TRAINING: the packed form of W^T (the matrix of the data gradient, autograd's mm(W^T, gy)) from W f32[Cout,Cin] as stored and
 * wp = lion_pwconv_split_pack_weights(W) (its tail carries the tensor's scale, which the transposition does not change):
 * wpt = lion_pwconv_split_packed_halfs(Cin, Cout) uint16.  One launch; no transposed copy, no second pass for max |w|. */
int lion_pwconv_split_pack_weights_t(const float *w, int Cout, int Cin, const uint16_t *wp, uint16_t *wpt, lionStream_t stream);
int lion_pwconv_split_stat_tiles(int Cout, int Cin, int L);
int lion_pwconv_split_forward(const float *x, const uint16_t *wp, const float *bias, int B, int Cin, int Cout, int L,
                              const float *pro_a, const float *pro_b, float *y, float *stats, lionStream_t stream);
/* ---- P5: LinearAttention (models/pvcnn2_ada.py:43-71) between its two 1x1 convolutions ---------------------
 * qkv f32[B, 3*H*D, N] (to_qkv's output, channel order (qkv, head, d)) -> out f32[B, H*D, N] (to_out's input):
 * softmax over the N points of every k row, ctx = softmax(k) v^T (D x D), out = ctx^T q.  D = 32 (every
 * LinearAttention of the models); one workgroup per (batch, head), fp32 MFMA. */
int lion_linear_attention_core(const float *qkv, int B, int H, int D, int N, float *out, lionStream_t stream);
/* its gradient (training): gout f32[B, H*D, N] -> gqkv f32[B, 3*H*D, N] (softmax backward included; the row dot product
 * sum_n p gp is the 32 x 32 contraction sum_e gctx[d][e] ctx[d][e]: no second sweep over the points). */
int lion_linear_attention_core_backward(const float *qkv, const float *gout, int B, int H, int D, int N, float *gqkv,
                                        lionStream_t stream);
/* nn.Linear on [B,K] (time-embedding MLP, batched AdaGN style projections): y[b][o] = act(bias[o] + sum_k x[b][k] W[o][k]),
 * wp = lion_pwconv_pack_weights(W f32[O,K]); act 0 none / 1 relu / 2 leaky-relu(slope).  One launch, fixed summation
 * order (4 K-quarters combined in order). */
int lion_linear_forward(const float *x, const float *wp, const float *bias, int B, int K, int O, int act, float slope,
                        float *y, lionStream_t stream);
int lion_affine_swish(const float *x, const float *A, const float *Bs, int rows, int L, float *y,
                      lionStream_t stream);
int lion_affine_swish_max(const float *x, const float *A, const float *Bs, int rows, int M, int U,
                          float *y, lionStream_t stream);
/* y = swish(x * A + Bs) + addend (rows x L, same layout): the point branch's last activation with PVConv's residual sum
 * (models/pvcnn2_ada.py:276-278) behind it. */
/* emb f32[B, D]: [sin, cos] of (t_b * scale) * row_i, i < half (D = 2 half, or 2 half + 1 with a zero column):
 * models/latent_points_ada.py get_timestep_embedding (reference latent_points_ada.py:101-115) in one launch. */
int lion_timestep_embedding(const float *t, const float *row, float scale, int B, int half, int D, float *emb,
                            lionStream_t stream);
int lion_affine_swish_add(const float *x, const float *A, const float *Bs, const float *addend, int rows, int L, float *y,
                          lionStream_t stream);

/* ---- backward scatters without float atomics (training; K5, K8, K12-grad) -------------------------------------------
 * gx f32[B,C,bins] = sum over the entries e with idx[b,e] = bin of (w ? w[b,e] : 1) * gy[b,c, e mod S]:
 *   grouping backward (grouping.cu:58-80):            idx [B, M*U], w NULL,       gy [B,C,M*U], S = E = M*U, bins = N
 *   3-NN interpolation backward (neighbor_interpolate.cu:145-170): idx / w [B, 3*N], gy [B,C,N], S = N, E = 3N, bins = M
 *   devoxelize backward (trilinear_devox.cu:119-162):  idx / w [B, 8*N], gy [B,C,N], S = N, E = 8N, bins = r^3
 * The inverse index is built once per sample, every channel is a gather-sum in ascending entry order: deterministic
 * (the reference's atomicAdd order is not), and not bound by the LDS float-atomic rate (csrc/scatter_csr.hip). */
size_t lion_scatter_csr_workspace_bytes(int B, int E, int bins);
int lion_scatter_csr(const float *gy, const int32_t *idx, const float *w, int B, int C, int S, int E, int bins, void *ws,
                     size_t ws_bytes, float *gx, lionStream_t stream);

/* ---- weight gradient of the 1x1 convolutions (training): gw[o][i] = sum_b sum_l gy[b][o][l] x[b][i][l] --------------
 * x f32[B,Cin,L], gy f32[B,Cout,L] (16-byte aligned), gw f32[Cout,Cin]; ws from lion_pwconv_wgrad_workspace_bytes.  What
 * autograd's mm([O, B L] x [B L, I]) of models/pvcnn2_ada.py's SharedMLP layers computes, without the transposing copies;
 * exact fp32 products on the MFMA pipe, slices summed in fixed order (csrc/pwconv_wgrad.hip).  gb f32[Cout] or NULL: the bias
 * gradient sum_b sum_l gy[b][o][l], taken from the same staged rows of gy (no second pass over it). */
size_t lion_pwconv_wgrad_workspace_bytes(int B, int Cin, int Cout, int L);
int lion_pwconv_wgrad(const float *x, const float *gy, int B, int Cin, int Cout, int L, void *ws, size_t ws_bytes,
                      float *gw, float *gb, lionStream_t stream);

/* ---- training forms of GroupNorm / AdaGN (+ Swish): models/adagn.py:45-65, models/pvcnn2_ada.py:78-84 --------------
 * Per row (b, c) of length L a GroupNorm followed by per-(batch, channel) scalars and an activation is
 * y = act(A x + Bs) (act: 0 identity, 1 swish).  Forward: lion_row_stats + lion_affine_act.  Backward:
 * lion_affine_act_bwd_stats -> S f32[rows, 2] = {sum da, sum da x} with da = gy act'(A x + Bs); the caller combines them
 * into Q, R f32[rows] (group means of the normalisation's gradient) and the parameter gradients;
 * lion_affine_act_bwd_apply writes dx = A da + Q + R x.  (csrc/norm_train.hip, lion_amd/train_ops.py) */
/* the [B, C] scalar algebra of both directions (double inside).  fac / bias f32[B, *] with row strides (views of the
 * [B, 2C] AdaGN projection) or NULL (plain GroupNorm); dfac / dbias f32[B, *] with row stride d_stride (the halves of one [B, 2C] buffer, or C) may be NULL; pw f32[B,C,3] = per-sample terms of the
 * GroupNorm weight / bias gradients (the caller sums them over the batch). */
int lion_gn_train_fold(const float *stats, const float *gw, const float *gb, const float *fac, int fac_stride,
                       const float *bias, int bias_stride, int B, int C, int G, int L, float eps, float *A, float *Bs,
                       float *mean, float *rstd, lionStream_t stream);
/* the same with the row sums taken on shifted values and carried in double (|mean| >> std rows keep their variance):
 * what lion_amd/train_ops.py uses */
int lion_row_stats64(const float *x, int rows, int L, double *stats, lionStream_t stream);
int lion_gn_train_fold64(const double *stats, const float *gw, const float *gb, const float *fac, int fac_stride,
                         const float *bias, int bias_stride, int B, int C, int G, int L, float eps, float *A,
                         float *Bs, float *mean, float *rstd, lionStream_t stream);
int lion_gn_train_bwd_fold(const float *S, const float *mean, const float *rstd, const float *gw, const float *gb,
                           const float *fac, int fac_stride, int B, int C, int G, int L, float *Q, float *R, float *dfac,
                           float *dbias, int d_stride, float *pw, const float *A, const double *xstats, const float *gate,
                           const float *qse, float *Aout, lionStream_t stream);
/* pw f32[B,C,3]: the per-sample terms of {d GroupNorm weight, d GroupNorm bias, sum over the row of dx}.  The third needs A (of the
 * forward fold) and xstats = the forward's lion_row_stats64 result (column 0 = row sums of x); both NULL: it is written as 0.
 * Summed over the batch it is the BIAS GRADIENT of the convolution that produced x (sum_l dx = A S1 + L Q + R sum_l x): Conv3d layers
 * in front of an AdaGN take it from here instead of a pass over dx (lion_amd/conv_ops.py).
 * gate / qse f32[B,C] (both or neither; with them A and Aout are required): an SE3d gate sits directly behind this AdaGN and the two
 * run as one op (lion_gn_se_gate_fwd / _bwd below) -- the upstream gradient was du = g gy + qse, S holds ITS row sums, and the apply
 * pass gets Aout = A g and Q = A qse + Q_gn: dx = Aout gy + Q + R x.
 * lion_gn_train_param_grads: dgw / dgb / dxs (or NULL) f32[C] = pw summed over the batch (ascending b). */
int lion_gn_train_param_grads(const float *pw, int B, int C, float *dgw, float *dgb, float *dxs, lionStream_t stream);
/* TRAINING, SE3d gate (reference models/pvcnn2_ada.py:27-41: x * sigmoid(W2 relu(W1 mean_voxels(x))), both Linear layers bias-free):
 * the [B, C] algebra between the row-sum pass (lion_row_stats) and the scaling pass (lion_affine_act with A = g, Bs = zero).
 * fwd: stats f32[B*C,2] (column 0 = row sums over the L voxels), w1 f32[Cr,C], w2 f32[C,Cr] -> mean f32[B,C], h f32[B,Cr] (after the
 * ReLU), g f32[B,C] (after the sigmoid), zero f32[B,C] (zeros).  bwd: S f32[B*C,2] (column 1 = sum_voxels gy*x, from
 * lion_affine_act_bwd_stats) -> dpre2 f32[B,C], dpre1 f32[B,Cr] (scratch outputs), Q f32[B,C] (per-voxel gradient through the mean,
 * for lion_affine_act_bwd_apply), dw1 f32[Cr,C], dw2 f32[C,Cr] (summed over the batch in ascending order).  C <= 1024, Cr <= 128. */
int lion_se_gate_fwd(const float *stats, const float *w1, const float *w2, int B, int C, int Cr, int L, float *mean, float *h,
                     float *g, float *zero, lionStream_t stream);
int lion_se_gate_bwd(const float *S, const float *g, const float *h, const float *mean, const float *w1, const float *w2, int B,
                     int C, int Cr, int L, float *dpre2, float *dpre1, float *Q, float *dw1, float *dw2, lionStream_t stream);
/* The tail of every PVConv's voxel branch is Conv3d -> AdaGN -> SE3d with NO activation between the last two (reference
 * pvcnn2_ada.py:211-226): both are affine per (sample, channel), so they run as ONE op -- u = A x + Bs is never written.
 * fwd: xstats f64[B*C,2] (lion_row_stats64 of x), A / Bs (lion_gn_train_fold64) -> mean f32[B,C] (channel means of u), h, g, and
 * A2 = g A, B2 = g Bs for one lion_affine_act pass (act 0) over x.  bwd: S f32[B*C,2] = {sum gy, sum gy x} (one
 * lion_affine_act_bwd_stats pass, act 0) -> Qse f32[B,C], Sp f32[B*C,2] (the row sums of du = g gy + Qse that
 * lion_gn_train_bwd_fold takes, with gate = g, qse = Qse), dpre2 / dpre1 (scratch), dw1, dw2; then one lion_affine_act_bwd_apply
 * pass with (Aout, Q, R).  3 passes over the grid forward and 5 backward instead of 6 and 10. */
/* ... and with the devoxelisation behind them (AdaGN -> SE3d -> trilinear_devoxelize, pvcnn2_ada.py:211-233: all linear): the gated
 * grid is never written either.  Forward: devoxelise x itself, apply (A2, B2) to the [B,C,N] result.  Backward, from the gradient g
 * f32[B,C,N] of that result: lion_rows_dot2 -> S f32[B*C,2] = {sum_p g ws, sum_p g v} (ws f32[B,N] = each point's 8 corner weights
 * summed, v = the devoxelised x saved by the forward) = the row sums {sum gy, sum gy x} of the scatter without a pass over the grid;
 * lion_gn_se_gate_bwd + lion_gn_train_bwd_fold as above; lion_trilinear_devoxelize_backward_affine writes
 * dx = A' scatter(g) + Q + R x in one pass (slab = Q + R x, corner contributions added in LDS; x 16-byte aligned, r3 * 4 <= 128 KiB,
 * r3 % 8 == 0).  Dense passes over the grid per PVConv tail: 1 forward (the row sums) and 2 backward -- were 6 + 10 as separate ops. */
int lion_rows_dot2(const float *g, const float *v, const float *ws, int B, int C, int N, float *S, lionStream_t stream);
int lion_trilinear_devoxelize_backward_affine(const float *gy, const int32_t *inds, const float *wgts, const float *x,
                                              const float *Ap, const float *Q, const float *R, int B, int C, int N, int r3,
                                              float *dx, lionStream_t stream);
int lion_gn_se_gate_fwd(const double *xstats, const float *A, const float *Bs, const float *w1, const float *w2, int B, int C, int Cr,
                        int L, float *mean, float *h, float *g, float *A2, float *B2, lionStream_t stream);
int lion_gn_se_gate_bwd(const float *S, const double *xstats, const float *A, const float *Bs, const float *g, const float *h,
                        const float *mean, const float *w1, const float *w2, int B, int C, int Cr, int L, float *Sp, float *dpre2,
                        float *dpre1, float *Qse, float *dw1, float *dw2, lionStream_t stream);
int lion_affine_act(const float *x, const float *A, const float *Bs, int rows, int L, int act, float *y,
                    lionStream_t stream);
int lion_affine_act_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int L, int act,
                              float *S, lionStream_t stream);
int lion_affine_act_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                              const float *R, int rows, int L, int act, float *dx, lionStream_t stream);
/* TRAINING, round 6: the same three passes with nn.Dropout behind the activation (PVConv's voxel branch: Conv3d -> AdaGN -> Swish ->
 * Dropout, reference pvcnn2_ada.py:211-222; pvcnn2.py:170-186): y = act(A x + Bs) * m / keep, m ~ Bernoulli(keep) per element.
 * Element e is kept when word (e & 3) of Philox4x32-10(counter = {e >> 2, 'DROP', 0}, key = *seed) < keep * 2^32; seed: a 64-bit
 * word in DEVICE memory (drawn per call by the framework's generator, so that a replayed hipGraph draws a fresh mask); the two
 * backward passes regenerate the mask from the same seed (gy is masked and scaled on the fly) -- no mask tensor exists.
 * keep in (0, 1]. */
int lion_affine_act_dropout(const float *x, const float *A, const float *Bs, int rows, int L, int act, const uint64_t *seed,
                            float keep, float *y, lionStream_t stream);
int lion_affine_act_dropout_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int L, int act,
                                      const uint64_t *seed, float keep, float *S, lionStream_t stream);
int lion_affine_act_dropout_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                                      const float *R, int rows, int L, int act, const uint64_t *seed, float keep, float *dx,
                                      lionStream_t stream);
/* Round 6 -- the same op with the set-abstraction pooling behind it (pvcnn2_ada.py:375-377): y f32[rows, M] = max over the U
 * neighbours of act(A x + Bs), x f32[rows, M, U] (U in {8, 16, 32, 64}).  The activated tensor is never written and its gradient
 * never materialised: backward = lion_affine_act_max_bwd_stats (S = {sum da, sum da x}, da = gy act'(.) at each group's FIRST
 * arg-max, 0 elsewhere) -> lion_gn_train_bwd_fold -> lion_affine_act_max_bwd_apply (dx = [arg-max] A da + Q + R x). */
int lion_affine_act_max(const float *x, const float *A, const float *Bs, int rows, int M, int U, int act, float *y,
                        lionStream_t stream);
int lion_affine_act_max_bwd_stats(const float *x, const float *gy, const float *A, const float *Bs, int rows, int M, int U,
                                  int act, float *S, lionStream_t stream);
int lion_affine_act_max_bwd_apply(const float *x, const float *gy, const float *A, const float *Bs, const float *Q,
                                  const float *R, int rows, int M, int U, int act, float *dx, lionStream_t stream);

/* ---- optimizer update (training): torch.optim.Adam with L2 weight decay (reference utils/utils.py:115-121; optimizer.step() of
 * trainers/hvae_trainer.py:150-154, train_2prior.py:405-410) over EVERY parameter tensor in one launch (csrc/optim.hip).
 * table u64[T,lion_adam_row()]: device addresses {param, grad, exp_avg, exp_avg_sq, step, ema} of T float32 tensors (step: that
 * parameter's f32 step count, incremented by the call, then used for the bias corrections; ema: 0, or the moving average of the
 * weights the reference's EMA wrapper keeps (utils/ema.py:60-75: ema = ema * ema_decay + (1 - ema_decay) * param after the update,
 * started from the updated parameter at the parameter's first step)); numel i32[T]; blockmap i32[blocks,2]: {tensor, chunk} per
 * workgroup, chunks of lion_adam_chunk() elements (all three in DEVICE memory, written by the caller once per gradient layout);
 * lr f32[1] (device).
 * Same arithmetic as torch.optim.Adam's single-tensor path (fp32 op for op, bias corrections in double). */
int lion_adam_chunk(void);
int lion_adam_row(void);
int lion_adam_step(const uint64_t *table, const int32_t *numel, const int32_t *blockmap, int blocks, int tensors, const float *lr,
                   float beta1, float beta2, float eps, float weight_decay, float ema_decay, lionStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LION_HIP_H_ */
