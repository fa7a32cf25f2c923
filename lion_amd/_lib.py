"""ctypes binding of ``lion_amd/csrc/liblion_hip.so`` (C ABI declared in ``include/lion_hip.h``).

There is NO fallback: if the shared library is missing or a call returns non-zero, a
``RuntimeError`` is raised (reference convention: precondition -> exception,
third_party/pvcnn/functional/src/utils.hpp:7-18; the reference's ``exit(-1)`` on launch
failure, cuda_utils.cuh:28-37, is deliberately not reproduced).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LION_HIP_SO") or os.path.join(_HERE, "csrc", "liblion_hip.so")  # env: A/B of kernel builds

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/lion_hip.h one to one
SIGNATURES = {
    "lion_abi_version": (_i, []),
    "lion_avg_voxelize_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "lion_avg_voxelize_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lion_voxelize_points_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp,
                                          _vp, _sz, _vp]),
    "lion_voxel_plan_bytes": (_sz, [_i, _i, _i]),
    "lion_voxel_index": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lion_voxel_scatter": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    "lion_voxel_scatter_read": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp, _vp, _vp]),
    "lion_avg_voxelize_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_trilinear_devoxelize_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lion_trilinear_devoxelize_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_ball_query": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "lion_grouping_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lion_group_points_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lion_grouping_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lion_furthest_point_sampling": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "lion_gather_features_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_gather_features_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_three_nn_interpolate_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lion_three_nn_interpolate_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_chamfer_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lion_chamfer_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "lion_emd_workspace_bytes": (_sz, [_i, _i, _i]),
    "lion_emd_approxmatch": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_emd_cost": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_emd_matchcost": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_emd_matchcost_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "lion_conv3d_packed_floats": (_sz, [_i, _i]),
    "lion_conv3d_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_conv3d_k3_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_conv3d_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "lion_conv3d_k3_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_conv3d_k3_wgrad_split": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_conv3d_wgrad_sparse_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "lion_conv3d_k3_wgrad_split_sparse": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lion_conv3d_stat_tiles": (_i, [_i, _i, _i, _i]),
    "lion_conv3d_k3_fused_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_conv3d_const_response": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_conv3d_occupancy_ints": (_sz, [_i, _i, _i]),
    "lion_conv3d_tile_occupancy": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "lion_conv3d_tile_occupancy_aware": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "lion_conv3d_split_packed_halfs": (_sz, [_i, _i]),
    "lion_conv3d_split_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_conv3d_split_stat_tiles": (_i, [_i, _i]),
    "lion_conv3d_k3_split_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_groupnorm_fold": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "lion_groupnorm_fold_se": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "lion_skinny_packed_floats": (_sz, [_i, _i]),
    "lion_skinny_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_skinny_splits": (_i, [_i, _i]),
    "lion_skinny_gemm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_skinny_gemm_se_finish": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "lion_skinny_finish": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "lion_to_channel_major": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "lion_from_channel_major": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_se_gate": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "lion_trilinear_devoxelize_affine_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_devoxelize_plan_bytes": (_sz, [_i, _i, _i]),
    "lion_trilinear_devoxelize_plan": (_i, [_vp, _i, _i, _i, _vp, _sz, _vp]),
    "lion_trilinear_devoxelize_planned_forward": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_row_stats": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_pwconv_packed_floats": (_sz, [_i, _i]),
    "lion_pwconv_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_pwconv_stat_tiles": (_i, [_i, _i, _i]),
    "lion_pwconv_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lion_pwconv_forward_max": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_pwconv_split_packed_halfs": (_sz, [_i, _i]),
    "lion_pwconv_split_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_pwconv_split_pack_weights_t": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "lion_pwconv_split_stat_tiles": (_i, [_i, _i, _i]),
    "lion_pwconv_split_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lion_linear_attention_core": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_linear_attention_core_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_linear_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "lion_affine_swish": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "lion_affine_swish_max": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_timestep_embedding": (_i, [_vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    "lion_affine_swish_add": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "lion_scatter_csr_workspace_bytes": (_sz, [_i, _i, _i]),
    "lion_scatter_csr": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "lion_pwconv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "lion_pwconv_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "lion_gn_train_fold": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "lion_row_stats64": (_i, [_vp, _i, _i, _vp, _vp]),
    "lion_gn_train_fold64": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "lion_gn_train_bwd_fold": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp]),
    "lion_gn_train_param_grads": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "lion_adam_chunk": (_i, []),
    "lion_adam_row": (_i, []),
    "lion_adam_step": (_i, [_vp, _vp, _vp, _i, _i, _vp, _f, _f, _f, _f, _f, _vp]),
    "lion_se_gate_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lion_se_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_rows_dot2": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_trilinear_devoxelize_backward_affine": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_gn_se_gate_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_gn_se_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lion_affine_act": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_affine_act_bwd_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_affine_act_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lion_affine_act_dropout": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp]),
    "lion_affine_act_dropout_bwd_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp]),
    "lion_affine_act_dropout_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp]),
    "lion_affine_act_max": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_affine_act_max_bwd_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_affine_act_max_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lion_ddim_update": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _vp, _vp]),
    "lion_ddpm_update": (_i, [_vp, _vp, _vp, _sz, _i, _f, _f, _f, _f, _f, _vp, _vp]),
    "lion_chain_begin_step": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp]),
    "lion_chain_update_noise": (_i, [_i, _vp, _vp, _sz, _vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "lion_chain_begin_step_temb": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "lion_chain_update_noise_cm": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "lion_latent_unpack": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lion_concat_broadcast": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lion_three_nn_interpolate_cat_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
}



_ERR = {-1: "LION_EINVAL (bad shape / null pointer)",
        -2: "LION_EUNSUPPORTED (shape outside what the kernels implement)",
        -3: "LION_EWORKSPACE (workspace too small)"}

_lib = None


def load() -> C.CDLL:
    """Load liblion_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"lion_amd: HIP extension not built: {SO_PATH} is missing. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or lion_amd/csrc/build.sh. "
            f"There is no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise RuntimeError(f"lion_amd: {SO_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = _ERR.get(code, f"hipError {code}" if code > 0 else f"error {code}")
        raise RuntimeError(f"lion_amd: {what} failed: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    """Same preconditions the reference enforces with TORCH_CHECK (src/utils.hpp:7-18)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("lion_amd: expected a CUDA(HIP) tensor; there is no CPU path")
        if not t.is_contiguous():
            raise RuntimeError("lion_amd: expected a contiguous tensor")
