"""ctypes binding of libn2m_hip.so (the C ABI declared in include/n2m_hip.h).

The HIP library IS the product: if it is missing or a call fails, this module raises -- there is no CPU or
PyTorch fallback anywhere in the package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("N2M_HIP_LIB") or os.path.join(_HERE, "lib", "libn2m_hip.so")   # override: A/B builds of the same ABI

_u32, _i32, _f32, _vp, _int, _u64 = ctypes.c_uint32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64

# name -> argtypes (restype is always int); mirrors include/n2m_hip.h one to one
SIGNATURES = {
    "n2m_near_far_from_aabb": [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    "n2m_sph_from_ray": [_vp, _vp, _f32, _u32, _vp, _vp],
    "n2m_morton3D": [_vp, _u32, _vp, _vp],
    "n2m_morton3D_invert": [_vp, _u32, _vp, _vp],
    "n2m_packbits": [_vp, _u32, _f32, _vp, _vp],
    "n2m_packbits_dev": [_vp, _u32, _vp, _vp, _vp],
    "n2m_flatten_rays": [_vp, _u32, _u32, _vp, _vp],
    "n2m_march_rays_train": [_vp, _vp, _vp, _f32, _int, _f32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_march_rays_train_write": [_vp, _vp, _vp, _f32, _int, _f32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "n2m_march_rays_train_fused": [_vp, _vp, _vp, _f32, _int, _f32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _u64, _vp],
    "n2m_march_fused_workspace_bytes": [_u32],
    "n2m_march_fallback_count": [_vp],
    "n2m_debug_fill_times": [_int, _vp],
    "n2m_stream_copy": [_vp, _vp, _u64, _u32, _vp],
    "n2m_composite_live_counts": [_vp, _vp],
    "n2m_sample_order_live_first": [_vp, _vp, _vp, _u32, _u32, _vp, _vp],
    "n2m_grid_backward_sample_order": [_vp],
    "n2m_grid_backward_config": [_int, _f32],
    "n2m_grid_backward_merge_levels": [_u32],
    "n2m_grid_backward_mid_event": [_vp],
    "n2m_composite_rays_train_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _int, _vp, _vp, _vp, _vp, _vp],
    "n2m_composite_rays_train_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _int, _vp, _vp, _vp],
    "n2m_composite_loss_train": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_composite_loss_train_ent": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp],
    "n2m_composite_loss_train_ex": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _int, _vp],
    "n2m_march_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _int, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_composite_rays": [_u32, _u32, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_compact_alive": [_vp, _u32, _vp, _vp, _vp],
    "n2m_select_positive": [_vp, _u32, _u32, _vp, _vp, _vp],
    "n2m_march_rays_dev": [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _f32, _int, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_composite_rays_dev": [_vp, _u32, _u32, _u32, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_compact_alive_dev": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    "n2m_grid_encode_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _int, _u32, _int, _vp],
    "n2m_grid_encode_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _int, _u32, _int, _vp],
    "n2m_grad_total_variation": [_vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _int, _vp],
    "n2m_grid_encode_forward_bm": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _int, _vp],
    "n2m_grid_encode_backward_bm": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _int, _vp],
    "n2m_grid_binned_workspace_bytes": [_u32, _u32, _u32, _u32, _vp, _int, _int],          # returns uint64 (RESTYPES)
    "n2m_grid_encode_backward_binned": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _int,
                                        _vp, _f32, _f32, _f32, _vp, _vp, _vp, _u64, _vp],
    "n2m_grid_encode_forward_pair": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _vp],
    "n2m_grid_encode_forward_packed": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _vp],
    "n2m_grid_encode_forward_packed_tv": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _vp, _vp],
    "n2m_grid_encode_forward_packed_tvterms": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp],
    "n2m_grid_backward_tv_corners": [_vp],
    "n2m_grid_encode_forward_packed_levels": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _u32, _u32, _vp],
    "n2m_grid_binned_pair_workspace_bytes": [_u32, _u32, _vp],                                  # returns uint64 (RESTYPES)
    "n2m_grid_encode_backward_binned_pair": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                             _vp, _f32, _f32, _f32, _vp, _vp, _f32, _f32, _int, _vp, _u64, _vp],
    "n2m_grid_encode_backward_binned_pair_half": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                                  _vp, _f32, _f32, _f32, _vp, _vp, _f32, _f32, _int, _vp, _u64, _vp, _int],
    "n2m_sdf_fold_plan": [_vp, _u32, _f32, _f32, _u32, _u32, _f32, _u32, _int, _vp, _vp, _vp, _u32, _vp, _u32, _vp],
    "n2m_sdf_fold_gather": [_vp, _u32, _u32, _vp, _vp, _u32, _vp, _u32, _vp, _vp],
    "n2m_grid_encode_backward_binned_lists": [_vp, _vp, _u32, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _vp, _int, _vp, _u64, _vp],
    "n2m_grid_encode_backward_binned_pair_fold": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                                  _vp, _f32, _f32, _f32, _vp, _vp, _f32, _f32, _int, _vp, _u64, _vp, _vp, _f32, _f32, _vp],
    "n2m_occupancy_update_partials": [_u32],                                                     # returns uint32 (RESTYPES)
    "n2m_occupancy_points": [_vp, _vp, _vp, _f32, _f32, _vp, _u32, _vp],
    "n2m_occupancy_update": [_vp, _vp, _f32, _u32, _f32, _vp, _vp, _vp, _vp, _vp],
    "n2m_grid_pair_fuse_plan": [_u32, _u32, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)],
    "n2m_grid_encode_backward_binned_pair_adam": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                                  _vp, _f32, _f32, _f32, _vp, _vp, _f32, _f32, _vp, _u64, _vp, _vp],
    "n2m_adam_fuse_restore": [_vp, _vp, _u32, _vp, _vp],
    "n2m_grid_tv_terms": [_vp, _vp, _vp, _u32, _u32, _f32, _u32, _u32, _int, _u32, _f32, _f32, _f32, _vp, _f32, _f32, _vp, _vp],
    "n2m_grid_encode_backward_binned_pair_tvt": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                                 _vp, _vp, _f32, _f32, _int, _vp, _u64, _vp, _int],
    "n2m_grad_total_variation_binned": [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _vp, _u64, _vp],
    "n2m_marching_cubes_workspace_bytes": [_u32, _u32, _u32],                                  # returns uint64 (RESTYPES)
    "n2m_marching_cubes_count": [_vp, _u32, _u32, _u32, ctypes.c_double, _vp, _u64, _vp, _vp],
    "n2m_marching_cubes_emit": [_vp, _u32, _u32, _u32, ctypes.c_double, _vp, _u64, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _int,
                                _u32, _vp, _u32, _vp],
    "n2m_texture_pad_nearest": [_vp, _vp, _u32, _u32, _u32, _u32, _vp],
    "n2m_freq_encode_forward": [_vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_freq_encode_backward": [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_get_rays": [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "n2m_batch_rays": [_vp, _vp, _u32, _u32, _u32, _u32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_batch_rays_cnf": [_vp, _vp, _u32, _u32, _u32, _u32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_adam_step": [_vp, ctypes.c_double, ctypes.c_double, _f32, _vp, _vp, _vp, _vp],
    "n2m_adam_step_scaler": [_vp, ctypes.c_double, ctypes.c_double, _f32, _vp, _vp, _vp, _vp, _vp],
    "n2m_ema_update": [_vp, _f32, _vp],
    "n2m_scaler_update": [_vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _f32, _f32, _f32, _vp],
    "n2m_scaler_update_slots": [_vp, _vp, _vp, _vp, _vp, _u32, ctypes.c_double, ctypes.c_double, _f32, _f32, _f32, _vp],
    "n2m_scaler_update_slots_loss": [_vp, _vp, _vp, _vp, _vp, _u32, ctypes.c_double, ctypes.c_double, _f32, _f32, _f32, _vp, _u32, _u32, _vp, _vp, _vp],
    "n2m_scaler_update_slots_loss2": [_vp, _vp, _vp, _vp, _vp, _u32, ctypes.c_double, ctypes.c_double, _f32, _f32, _f32, _vp, _u32, _u32, _vp, _vp,
                                      _vp, _u32, _f32, _vp],
    "n2m_scaler_update_slots_loss3": [_vp, _vp, _vp, _vp, _vp, _u32, ctypes.c_double, ctypes.c_double, _f32, _f32, _f32, _vp, _u32, _u32, _vp, _vp,
                                      _vp, _u32, _f32, _vp, _u32, _f32, _vp],
    "n2m_sdf_offsets": [_vp, _u32, _f32, _f32, _vp, _vp, _vp],
    "n2m_sdf_alpha_forward": [_vp, _vp, _vp, _vp, _u32, _vp, _f32, _f32, _vp, _vp, _vp, _vp],
    "n2m_sdf_alpha_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _f32, _f32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    "n2m_photo_loss_forward": [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _u32, _vp, _vp, _vp, _vp],
    "n2m_photo_loss_backward": [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _u32, _vp, _vp, _vp, _vp],
    "n2m_sh_encode_forward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp],
    "n2m_sh_encode_backward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    # include/n2m_mlp.h
    "n2m_field_forward": [_vp] * 11 + [_u32, _int, _int] + [_vp] * 4,
    "n2m_field_backward": [_vp] * 11 + [_u32, _int, _int] + [_vp] * 14,
    "n2m_field_forward_train": [_vp] * 11 + [_u32, _int, _int] + [_vp] * 5,
    "n2m_field_backward_train": [_vp] * 11 + [_u32, _int, _int] + [_vp] * 13 + [_f32, _vp, _vp],
    "n2m_field_spec_partials": [],
    # include/n2m_raster.h
    "n2m_rasterize_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "n2m_rasterize_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_interpolate_forward": [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_interpolate_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "n2m_interpolate_backward_strided": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "n2m_antialias_build_topology": [_vp, _u32, _vp, _u32, _vp],
    "n2m_antialias_forward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_antialias_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _vp, _vp, _vp],
    "n2m_antialias_backward_seeded": [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _vp, _vp, _vp],
    "n2m_to_clip": [_vp, _vp, _u32, _vp, _vp],
    "n2m_to_clip_backward": [_vp, _vp, _u32, _vp, _vp],
    "n2m_laplacian_forward": [_vp, _vp, _vp, _u32, _vp, _f32, _f32, _f32, _u32, _vp, _vp, _vp, _vp],
    "n2m_laplacian_backward": [_vp, _vp, _vp, _vp, _u32, _vp, _f32, _vp, _f32, _f32, _u32, _vp, _vp, _vp],
    "n2m_laplacian_backward_acc": [_vp, _vp, _vp, _vp, _u32, _vp, _f32, _vp, _f32, _f32, _u32, _vp, _vp, _vp],
    "n2m_gather_rows": [_vp, _vp, _u32, _u32, _vp, _vp],
    "n2m_gather_rows_strided": [_vp, _vp, _u32, _u32, _u32, _vp, _u32, _vp],
    "n2m_scatter_rows_strided": [_vp, _vp, _u32, _u32, _u32, _vp, _u32, _vp],
    "n2m_scatter_rows": [_vp, _vp, _u32, _u32, _vp, _vp],
    "n2m_stage1_head": [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp],
    # include/n2m_peer.h
    "n2m_peer_alloc": [ctypes.c_size_t, _int, ctypes.POINTER(_vp)],
    "n2m_peer_free": [_vp],
    "n2m_peer_export": [_vp, _vp],
    "n2m_peer_import": [_vp, ctypes.POINTER(_vp)],
    "n2m_peer_unmap": [_vp],
    "n2m_peer_signal": [_vp, _u32, _vp],
    "n2m_peer_wait": [_vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "n2m_peer_copy": [_vp, _vp, ctypes.c_size_t, _vp],
    "n2m_peer_reduce_slices": [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "n2m_adam_step_peer": [_vp, ctypes.c_double, ctypes.c_double, _f32, _vp, _vp, _vp, _vp, _vp],
    "n2m_grid_backward_peer_route": [_vp],
    "n2m_prof_enable": [_int],
    "n2m_prof_reset": [],
    "n2m_prof_read": [_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    "n2m_prof_seen": [_int, ctypes.POINTER(ctypes.c_uint64)],
}

RESTYPES = {"n2m_occupancy_update_partials": _u32, "n2m_grid_binned_workspace_bytes": _u64, "n2m_grid_binned_pair_workspace_bytes": _u64, "n2m_march_fused_workspace_bytes": _u64,
            "n2m_marching_cubes_workspace_bytes": _u64, "n2m_field_spec_partials": _u32}   # everything else returns an int status

F32, F16 = 0, 1
ADAM_MAX = 16


class AdamDesc(ctypes.Structure):
    """N2mAdamDesc of include/n2m_hip.h."""
    _fields_ = [("param", _vp * ADAM_MAX), ("grad", _vp * ADAM_MAX), ("exp_avg", _vp * ADAM_MAX), ("exp_avg_sq", _vp * ADAM_MAX),
                ("half_shadow", _vp * ADAM_MAX), ("numel", _u32 * ADAM_MAX), ("lr", _f32 * ADAM_MAX), ("grad_is_half", _i32 * ADAM_MAX),
                ("shadow_mode", _i32 * ADAM_MAX), ("clear_grad", _i32 * ADAM_MAX), ("slot", _i32 * ADAM_MAX), ("count", _u32)]


class ScalerTail(ctypes.Structure):
    """N2mScalerTail of include/n2m_hip.h."""
    _fields_ = [("growth_tracker", _vp), ("steps", _vp), ("participants", _u32), ("growth_factor", _f32), ("backoff_factor", _f32),
                ("growth_interval", _f32), ("loss_partial", _vp), ("n_partial", _u32), ("n_rays", _u32), ("loss", _vp), ("loss_sum", _vp),
                ("extra_partial", _vp), ("n_extra", _u32), ("extra_scale", _f32), ("extra2_partial", _vp), ("n_extra2", _u32),
                ("extra2_scale", _f32), ("ticket", _vp)]


EMA_MAX = 16


class EmaDesc(ctypes.Structure):
    """N2mEmaDesc of include/n2m_hip.h."""
    _fields_ = [("shadow", _vp * EMA_MAX), ("param", _vp * EMA_MAX), ("numel", _u32 * EMA_MAX), ("count", _u32)]


PEER_MAX = 8


class PeerPtrs(ctypes.Structure):
    """N2mPeerPtrs of include/n2m_peer.h."""
    _fields_ = [("ptr", _vp * PEER_MAX), ("count", _u32)]


class AdamPeer(ctypes.Structure):
    """N2mAdamPeer of include/n2m_peer.h."""
    _fields_ = [("world", _u32), ("slots", (_vp * PEER_MAX) * 16), ("packed_local", _vp), ("packed_remote", _vp * PEER_MAX), ("n_remote", _u32)]


class PeerRoute(ctypes.Structure):
    """N2mPeerRoute of include/n2m_peer.h."""
    _fields_ = [("world", _u32), ("split_row", _u32), ("rows_c", _u32), ("rows_f", _u32), ("g1", (_vp * PEER_MAX) * 2), ("g2", (_vp * PEER_MAX) * 2)]


class AdamFuse(ctypes.Structure):
    """N2mAdamFuse of include/n2m_hip.h."""
    _fields_ = [("p_in", _vp * 2), ("m_in", _vp * 2), ("v_in", _vp * 2), ("p_out", _vp * 2), ("m_out", _vp * 2), ("v_out", _vp * 2),
                ("packed", _vp), ("first_level", _u32), ("lr", _f32 * 2), ("slot", _i32 * 2), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", _f32), ("scale", _vp), ("bias", _vp)]


KERNEL_IDS = {"grid_encode_forward": 0, "grid_encode_backward": 1, "grad_total_variation": 2, "march_rays_train_count": 3,
              "march_rays_train_write": 4, "composite_rays_train_forward": 5, "composite_rays_train_backward": 6,
              "near_far_from_aabb": 7, "packbits": 8, "mlp_forward": 9, "mlp_backward": 10, "rasterize": 11,
              "grid_encode_forward_packed": 12, "adam_step": 13,
              "interpolate_forward": 14, "interpolate_backward": 15, "antialias_forward": 16, "antialias_backward": 17, "rasterize_backward": 18}

_lib = None


def lib():
    """Load the library once. Raises if the HIP build is not present (no fallback by design)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch ships, not load a second copy before torch does
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"nerf2mesh_amd: {LIB_PATH} is missing -- build it with `python -m nerf2mesh_amd.build` "
                "(hipcc, gfx950). There is no CPU / PyTorch fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError here = header/library mismatch: fail loudly
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, ctypes.c_int)
        L.n2m_last_error.restype = ctypes.c_char_p
        L.n2m_prof_name.restype = ctypes.c_char_p
        L.n2m_prof_name.argtypes = [_int]
        L.n2m_abi_version.restype = ctypes.c_int
        if L.n2m_abi_version() != 1:
            raise RuntimeError("nerf2mesh_amd: libn2m_hip.so ABI version mismatch")
        _lib = L
    return _lib


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.n2m_last_error().decode()}")


_BWD_CFG = [None]


def grid_backward_config(tv_stride=1, overflow_div=1.0):
    """n2m_grid_backward_config is PROCESS-wide state of the binned backward (row stride of the TV table, fp16 overflow margin).  Every
    Python caller of a binned backward / binned TV entry point states what it needs right before its call -- a sharded engine (stride 2,
    margin W) and a plain trainer or single-GPU engine (1, 1) can then live in one process without inheriting each other's setting.  The
    last value is cached: the common case costs one tuple compare."""
    want = (int(tv_stride), float(overflow_div))
    if _BWD_CFG[0] != want:
        call("n2m_grid_backward_config", want[0], want[1])
        _BWD_CFG[0] = want


_WORKSPACE = {}


def workspace(device, nbytes, slot=0):
    """Grow-only device scratch of the binned grid kernels: one buffer per (device, slot); uses on one stream are ordered, so a
    slot is only needed per concurrently running stream."""
    import torch
    key = (device, slot)
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _WORKSPACE.pop(key, None)
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WORKSPACE[key] = buf
    return buf


_SIDE_STREAMS = {}


def side_stream(device, slot=1, priority=0):
    """A second stream per device for work that may overlap the main stream (created once).  priority > 0 asks for the lowest
    priority the device offers: its workgroups are dispatched only into slots the main stream leaves free."""
    import torch
    key = (device, slot)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=int(priority))
        _SIDE_STREAMS[key] = st
    return st


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device (the C call behind torch.cuda.current_stream())."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def check_cuda(**tensors):
    """The reference's CHECK_CUDA / CHECK_CONTIGUOUS (gridencoder.cu:15-18): RuntimeError like TORCH_CHECK."""
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous tensor")


def prof_enable(on=True):
    """True/1: time every launch; n > 1: every n-th launch per kernel; False/0: off."""
    call("n2m_prof_enable", int(on))


def prof_reset():
    call("n2m_prof_reset")


def prof_seen(kernel):
    """Launches of `kernel` seen since the last reset, timed or not (sampled timing: per-step cost = mean timed duration x seen / steps)."""
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    n = ctypes.c_uint64(0)
    call("n2m_prof_seen", kid, ctypes.byref(n))
    return int(n.value)


def prof_read(kernel):
    """(launches, total_ms, algorithmic_bytes) of the timed launches of `kernel` since the last reset."""
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    n, ms, by = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_double(0)
    call("n2m_prof_read", kid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
    return int(n.value), float(ms.value), float(by.value)
