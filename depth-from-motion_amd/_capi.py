"""ctypes binding of lib/libdfm_hip.so (C ABI: include/dfm_hip.h).

Loud by design: a missing library is an ImportError with the build command,
never a silent fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DFM_HIP_LIB', os.path.join(_HERE, 'lib', 'libdfm_hip.so'))

DFM_F32, DFM_BF16 = 0, 1

# every symbol include/dfm_hip.h declares (tests/test_capi_symbols.py checks
# the header against this list and against the built library)
EXPORTS = (
    'dfm_version',
    'dfm_last_error',
    'dfm_profile_begin',
    'dfm_profile_end',
    'dfm_camera_prepare',
    'dfm_plane_sweep_workspace_bytes',
    'dfm_plane_sweep_fwd',
    'dfm_plane_sweep_cl_workspace_bytes',
    'dfm_plane_sweep_fwd_channels_last',
    'dfm_plane_sweep_fwd_nhwc',
    'dfm_plane_sweep_fwd_from_nhwc',
    'dfm_plane_sweep_bwd',
    'dfm_plane_sweep_grid',
    'dfm_plane_sweep_last_kernel',
    'dfm_plane_sweep_bwd_last_kernel',
    'dfm_plane_sweep_fwd_opts',
    'dfm_plane_sweep_bwd_opts',
    'dfm_plane_sweep_bwd_channels_last',
    'dfm_plane_sweep_bwd_cur_nhwc',
    'dfm_plane_sweep_bwd_prev_gather_workspace_bytes',
    'dfm_plane_sweep_bwd_prev_gather',
    'dfm_plane_sweep_bwd_gather',
    'dfm_plane_sweep_autotune',
    'dfm_plane_sweep_tuning',
    'dfm_plane_sweep_reset_tuning',
    'dfm_store_probe',
    'dfm_clock_probe',
    'dfm_point_sample_mv_workspace_bytes',
    'dfm_point_sample_mv_fwd',
    'dfm_frustum_to_voxel_workspace_bytes',
    'dfm_frustum_to_voxel_fwd',
    'dfm_depth_head_fwd',
    'dfm_depth_head_stats_fwd',
    'dfm_frustum_to_voxel_fused_fwd',
    'dfm_frustum_to_voxel_bwd_workspace_bytes',
    'dfm_frustum_to_voxel_bwd',
    'dfm_frustum_to_voxel_fused_bwd',
    'dfm_frustum_to_voxel_bwd_gather_workspace_bytes',
    'dfm_frustum_to_voxel_bwd_gather',
    'dfm_frustum_to_voxel_bwd_gather_cl',
    'dfm_point_sample_mv_fwd_batched',
    'dfm_point_sample_mv_bwd_workspace_bytes',
    'dfm_point_sample_mv_bwd',
    'dfm_depth_head_bwd',
    'dfm_sweep_conv_weight_bytes',
    'dfm_sweep_conv_pack_weights',
    'dfm_sweep_conv_stats_splits',
    'dfm_sweep_conv_fwd',
    'dfm_cost_gate_weight_bytes',
    'dfm_cost_gate_pack_weights',
    'dfm_cost_gate_fwd',
    'dfm_conv3d_k3_c32_weight_bytes',
    'dfm_conv3d_k3_c32_pack_weights',
    'dfm_conv3d_k3_c32_stats_splits',
    'dfm_conv3d_k3_c32_fwd',
    'dfm_conv3d_k3_c32_fwd_strided',
    'dfm_conv3d_k3_c32_fwd_slices',
    'dfm_conv3d_k3_c32_to1_fwd',
    'dfm_group_norm_coefficients',
    'dfm_conv3d_to1_norm_fwd',
    'dfm_conv3d_to1_bwd_data',
    'dfm_conv3d_to1_wgrad_workspace_bytes',
    'dfm_conv3d_to1_wgrad',
    'dfm_bilinear_resize_bwd_nhwc',
    'dfm_depth_pool_fwd',
    'dfm_depth_pool_bwd',
    'dfm_cost_gate_mfma_weight_bytes',
    'dfm_cost_gate_mfma_pack_weights',
    'dfm_cost_gate_mfma_fwd',
    'dfm_conv3d_g_weight_bytes',
    'dfm_conv3d_g_pack_weights',
    'dfm_conv3d_g_pack_weights_2d',
    'dfm_conv3d_g_fwd',
    'dfm_conv3d_g_fwd_f32',
    'dfm_conv3d_g_plan',
    'dfm_conv3d_wgrad_workspace_bytes',
    'dfm_conv3d_wgrad',
    'dfm_conv3d_wgrad_to',
    'dfm_depth_loss_fwd',
    'dfm_depth_loss_bwd',
    'dfm_depth_loss_fused_fwd',
    'dfm_depth_loss_fused_bwd',
    'dfm_voxel_sample_fwd',
    'dfm_voxel_sample_bwd',
    'dfm_spp_tail_workspace_bytes',
    'dfm_spp_tail_fwd',
    'dfm_group_norm_workspace_bytes',
    'dfm_group_norm_fwd',
    'dfm_group_norm_fwd_channels_last',
    'dfm_group_norm_apply_channels_last',
    'dfm_group_norm_fwd_channels_last_res',
    'dfm_group_norm_apply_channels_last_res',
    'dfm_group_norm_bwd',
    'dfm_group_norm_bwd_channels_last',
    'dfm_group_norm_bwd_channels_last_xmask',
)


class SweepDesc(ctypes.Structure):
    """struct dfm_sweep_desc"""
    _fields_ = [
        ('batch', ctypes.c_int32),
        ('channels', ctypes.c_int32),
        ('h_in', ctypes.c_int32),
        ('w_in', ctypes.c_int32),
        ('num_depths', ctypes.c_int32),
        ('h_out', ctypes.c_int32),
        ('w_out', ctypes.c_int32),
        ('feat_sample_factor', ctypes.c_float),
        ('cost_sample_factor', ctypes.c_float),
        ('img_scale_factor', ctypes.c_float),
        ('crop_x', ctypes.c_float),
        ('crop_y', ctypes.c_float),
        ('org_w', ctypes.c_float),
        ('flip', ctypes.c_int32),
        ('dtype', ctypes.c_int32),
    ]


class SweepOpts(ctypes.Structure):
    """struct dfm_sweep_opts: launch options of ONE plane-sweep call (0 = library default)"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        'kernel', 'lanes_per_workgroup', 'lds_kib', 'blocks_per_group', 'planes_per_workgroup',
        'bands_per_chunk', 'points_per_lane', 'pipeline', 'store_align_points', 'pair_stores',
        'unpack')] + [('reserved', ctypes.c_int32 * 1)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != 'reserved'}


class MvDesc(ctypes.Structure):
    """struct dfm_mv_desc"""
    _fields_ = [
        ('num_views', ctypes.c_int32),
        ('num_frames', ctypes.c_int32),
        ('channels', ctypes.c_int32),
        ('feat_h', ctypes.c_int32),
        ('feat_w', ctypes.c_int32),
        ('nx', ctypes.c_int32),
        ('ny', ctypes.c_int32),
        ('nz', ctypes.c_int32),
        ('num_points', ctypes.c_int64),
        ('scale_x', ctypes.c_float),
        ('scale_y', ctypes.c_float),
        ('crop_x', ctypes.c_float),
        ('crop_y', ctypes.c_float),
        ('flip', ctypes.c_int32),
        ('pad_h', ctypes.c_float),
        ('pad_w', ctypes.c_float),
        ('mode', ctypes.c_int32),
        ('aggregate', ctypes.c_int32),
        ('valid_sample', ctypes.c_int32),
        ('dtype', ctypes.c_int32),
        ('out_channels_last', ctypes.c_int32),
        ('feats_channels_last', ctypes.c_int32),
    ]


class F2vDesc(ctypes.Structure):
    """struct dfm_f2v_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        'batch', 'channels', 'd', 'h', 'w', 'ds', 'hs', 'ws', 'sem_channels', 'hsem', 'wsem', 'nz',
        'ny', 'nx')] + [(n, ctypes.c_float) for n in ('pad_h', 'pad_w', 'depth_min', 'depth_span')
                        ] + [('dtype', ctypes.c_int32), ('stereo_channels_last', ctypes.c_int32),
                           ('out_channels_last', ctypes.c_int32), ('stereo_atten', ctypes.c_int32),
                           ('no_sem_atten', ctypes.c_int32), ('sem_channels_last', ctypes.c_int32)]


class SppDesc(ctypes.Structure):
    """struct dfm_spp_desc"""
    _fields_ = [('batch', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('num_sources', ctypes.c_int32), ('source_channels', ctypes.c_int32 * 4),
                ('num_branches', ctypes.c_int32), ('in_channels', ctypes.c_int32),
                ('spp_channels', ctypes.c_int32), ('pooled_h', ctypes.c_int32 * 4),
                ('pooled_w', ctypes.c_int32 * 4), ('eps', ctypes.c_float)]


class VsDesc(ctypes.Structure):
    """struct dfm_vs_desc"""
    _fields_ = [('channels', ctypes.c_int32), ('nx', ctypes.c_int32), ('ny', ctypes.c_int32),
                ('nz', ctypes.c_int32), ('num_depths', ctypes.c_int32), ('h_out', ctypes.c_int32),
                ('w_out', ctypes.c_int32), ('downsample_factor', ctypes.c_float),
                ('scale_x', ctypes.c_float), ('scale_y', ctypes.c_float), ('crop_x', ctypes.c_float),
                ('crop_y', ctypes.c_float), ('flip', ctypes.c_int32), ('ori_w', ctypes.c_float),
                ('voxel_range', ctypes.c_float * 6), ('voxel_size', ctypes.c_float * 3),
                ('proj_inv', ctypes.c_float * 16), ('mode', ctypes.c_int32),
                ('dtype', ctypes.c_int32)]


class DepthLossDesc(ctypes.Structure):
    """struct dfm_depth_loss_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ('batch', 'num_depths', 'h', 'w', 'target', 'focal')] + \
        [(n, ctypes.c_float) for n in ('min_depth', 'max_depth', 'interval', 'sigma', 'alpha', 'gamma')] + \
        [('dtype', ctypes.c_int32)]


class Conv3dDesc(ctypes.Structure):
    """struct dfm_conv3d_desc"""
    _fields_ = [('n', ctypes.c_int32), ('cin', ctypes.c_int32), ('cout', ctypes.c_int32),
                ('in_size', ctypes.c_int32 * 3), ('out_size', ctypes.c_int32 * 3),
                ('stride', ctypes.c_int32 * 3), ('padding', ctypes.c_int32 * 3),
                ('transposed', ctypes.c_int32 * 3), ('relu', ctypes.c_int32),
                ('in_channel_stride', ctypes.c_int32), ('kernel1', ctypes.c_int32 * 3)]


class Conv3dWgradDesc(ctypes.Structure):
    """struct dfm_conv3d_wgrad_desc"""
    _fields_ = [('n', ctypes.c_int32), ('a', ctypes.c_int32), ('b', ctypes.c_int32),
                ('g_size', ctypes.c_int32 * 3), ('x_size', ctypes.c_int32 * 3),
                ('stride', ctypes.c_int32 * 3), ('padding', ctypes.c_int32 * 3),
                ('g_stride', ctypes.c_int64 * 4), ('x_stride', ctypes.c_int64 * 4)]


DL_LINEAR, DL_HARD, DL_GAUSSIAN, DL_LAPLACIAN = 0, 1, 2, 3


DFM_ERR_UNSUPPORTED = -2  # include/dfm_hip.h


class DfmHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    h = ctypes.CDLL(LIB_PATH)
    vp, fp, i32, sz = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
    dp = ctypes.POINTER(SweepDesc)
    h.dfm_version.restype = ctypes.c_int
    h.dfm_last_error.restype = ctypes.c_char_p
    h.dfm_profile_begin.restype = ctypes.c_int
    h.dfm_profile_begin.argtypes = [ctypes.c_int]
    h.dfm_profile_end.restype = ctypes.c_int
    h.dfm_profile_end.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    h.dfm_camera_prepare.restype = ctypes.c_int
    h.dfm_camera_prepare.argtypes = [fp, i32, i32, i32, fp, fp, vp]
    h.dfm_plane_sweep_workspace_bytes.restype = sz
    h.dfm_plane_sweep_workspace_bytes.argtypes = [dp]
    h.dfm_plane_sweep_fwd.restype = ctypes.c_int
    h.dfm_plane_sweep_fwd.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp]
    h.dfm_plane_sweep_cl_workspace_bytes.restype = sz
    h.dfm_plane_sweep_cl_workspace_bytes.argtypes = [dp]
    h.dfm_plane_sweep_fwd_channels_last.restype = ctypes.c_int
    h.dfm_plane_sweep_fwd_channels_last.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp]
    h.dfm_plane_sweep_fwd_nhwc.restype = ctypes.c_int
    h.dfm_plane_sweep_fwd_nhwc.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp]
    h.dfm_plane_sweep_fwd_from_nhwc.restype = ctypes.c_int
    h.dfm_plane_sweep_fwd_from_nhwc.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp]
    h.dfm_plane_sweep_bwd.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd.argtypes = [dp, vp, fp, fp, fp, fp, fp, fp, vp]
    h.dfm_plane_sweep_grid.restype = ctypes.c_int
    h.dfm_plane_sweep_grid.argtypes = [dp, i32, fp, fp, fp, fp, fp, fp, vp]
    h.dfm_plane_sweep_last_kernel.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_last_kernel.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_last_kernel.argtypes = []
    op = ctypes.POINTER(SweepOpts)
    h.dfm_plane_sweep_fwd_opts.restype = ctypes.c_int
    h.dfm_plane_sweep_fwd_opts.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp, op]
    h.dfm_plane_sweep_bwd_opts.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_opts.argtypes = [dp, vp, fp, fp, fp, fp, fp, fp, vp, op]
    h.dfm_store_probe.restype = ctypes.c_int
    h.dfm_store_probe.argtypes = [vp, i32, i32, ctypes.c_int64, i32, i32, vp]
    h.dfm_clock_probe.restype = ctypes.c_int
    h.dfm_clock_probe.argtypes = [vp, i32, vp]
    h.dfm_plane_sweep_bwd_channels_last.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_channels_last.argtypes = [dp, vp, fp, fp, fp, fp, fp, fp, vp, sz, vp]
    h.dfm_plane_sweep_bwd_cur_nhwc.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_cur_nhwc.argtypes = [dp, vp, fp, fp, fp, fp, fp, vp]
    h.dfm_plane_sweep_bwd_prev_gather_workspace_bytes.restype = ctypes.c_size_t
    h.dfm_plane_sweep_bwd_prev_gather_workspace_bytes.argtypes = [dp]
    h.dfm_plane_sweep_bwd_prev_gather.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_prev_gather.argtypes = [dp, vp, fp, fp, fp, fp, fp, vp, sz, vp]
    h.dfm_plane_sweep_bwd_gather.restype = ctypes.c_int
    h.dfm_plane_sweep_bwd_gather.argtypes = [dp, i32, vp, i32, fp, fp, fp, fp, fp, i32, vp, sz, vp]
    h.dfm_plane_sweep_autotune.restype = ctypes.c_int
    h.dfm_plane_sweep_autotune.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, sz, vp, op]
    h.dfm_plane_sweep_tuning.restype = ctypes.c_int
    h.dfm_plane_sweep_tuning.argtypes = [dp, op]
    h.dfm_plane_sweep_reset_tuning.restype = None
    mp = ctypes.POINTER(MvDesc)
    h.dfm_point_sample_mv_workspace_bytes.restype = sz
    h.dfm_point_sample_mv_workspace_bytes.argtypes = [mp]
    h.dfm_point_sample_mv_fwd.restype = ctypes.c_int
    h.dfm_point_sample_mv_fwd.argtypes = [mp, vp, fp, fp, fp, vp, vp, vp, sz, vp]
    h.dfm_point_sample_mv_fwd_batched.restype = ctypes.c_int
    h.dfm_point_sample_mv_fwd_batched.argtypes = [vp, i32, vp, fp, i32, fp, fp, vp, vp, vp]
    h.dfm_frustum_to_voxel_fwd.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_fwd.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, vp, fp, fp, vp, vp,
                                           ctypes.c_size_t, vp]
    h.dfm_frustum_to_voxel_workspace_bytes.restype = ctypes.c_size_t
    h.dfm_frustum_to_voxel_workspace_bytes.argtypes = [ctypes.POINTER(F2vDesc)]
    h.dfm_depth_head_fwd.restype = ctypes.c_int
    h.dfm_depth_head_fwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, fp, vp, vp, vp, vp]
    h.dfm_depth_head_stats_fwd.restype = ctypes.c_int
    h.dfm_depth_head_stats_fwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, fp, fp, fp, vp, vp]
    h.dfm_frustum_to_voxel_fused_fwd.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_fused_fwd.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, fp, fp, i32, vp, fp, fp, vp, vp,
                                                 ctypes.c_size_t, vp]
    h.dfm_frustum_to_voxel_bwd.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_bwd.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, fp, fp, fp, fp, vp,
                                           ctypes.c_size_t, vp]
    h.dfm_frustum_to_voxel_fused_bwd.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_fused_bwd.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, fp, fp, i32, fp, fp, fp, fp, vp,
                                                 ctypes.c_size_t, vp]
    h.dfm_frustum_to_voxel_bwd_gather_workspace_bytes.restype = ctypes.c_size_t
    h.dfm_frustum_to_voxel_bwd_gather_workspace_bytes.argtypes = [ctypes.POINTER(F2vDesc)]
    h.dfm_frustum_to_voxel_bwd_gather.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_bwd_gather.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, vp, fp, fp, i32, fp,
                                                  ctypes.POINTER(ctypes.c_float), fp, fp, fp, vp, sz, vp]
    h.dfm_frustum_to_voxel_bwd_gather_cl.restype = ctypes.c_int
    h.dfm_frustum_to_voxel_bwd_gather_cl.argtypes = [ctypes.POINTER(F2vDesc), vp, vp, vp, fp, fp, i32, fp,
                                                     ctypes.POINTER(ctypes.c_float), fp, vp, fp, vp, sz, vp]
    h.dfm_frustum_to_voxel_bwd_workspace_bytes.restype = ctypes.c_size_t
    h.dfm_frustum_to_voxel_bwd_workspace_bytes.argtypes = [ctypes.POINTER(F2vDesc)]
    h.dfm_point_sample_mv_bwd.restype = ctypes.c_int
    h.dfm_point_sample_mv_bwd.argtypes = [mp, vp, fp, fp, fp, fp, vp, sz, vp]
    h.dfm_point_sample_mv_bwd_workspace_bytes.restype = sz
    h.dfm_point_sample_mv_bwd_workspace_bytes.argtypes = [mp]
    h.dfm_depth_head_bwd.restype = ctypes.c_int
    h.dfm_depth_head_bwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, fp, vp, vp, vp, fp, vp]
    h.dfm_sweep_conv_weight_bytes.restype = sz
    h.dfm_sweep_conv_pack_weights.restype = ctypes.c_int
    h.dfm_sweep_conv_pack_weights.argtypes = [vp, vp, i32, vp, vp]
    h.dfm_sweep_conv_stats_splits.restype = ctypes.c_int
    h.dfm_sweep_conv_stats_splits.argtypes = [dp, i32]
    h.dfm_sweep_conv_fwd.restype = ctypes.c_int
    h.dfm_sweep_conv_fwd.argtypes = [dp, vp, vp, fp, fp, fp, fp, vp, vp, vp, fp, fp, i32, vp]
    h.dfm_cost_gate_fwd.restype = ctypes.c_int
    h.dfm_cost_gate_fwd.argtypes = [i32, i32, ctypes.c_int64, i32, vp, vp, vp, vp, vp]
    h.dfm_cost_gate_weight_bytes.restype = sz
    h.dfm_cost_gate_weight_bytes.argtypes = [i32]
    h.dfm_cost_gate_pack_weights.restype = ctypes.c_int
    h.dfm_cost_gate_pack_weights.argtypes = [vp, i32, i32, vp, vp]
    h.dfm_conv3d_k3_c32_weight_bytes.restype = sz
    h.dfm_conv3d_k3_c32_pack_weights.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_pack_weights.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    h.dfm_conv3d_k3_c32_fwd.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_fwd.argtypes = [i32, i32, i32, i32, vp, vp, fp, vp, i32, i32, i32, fp, vp]
    h.dfm_conv3d_k3_c32_fwd_slices.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_fwd_slices.argtypes = [i32, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp]
    h.dfm_conv3d_k3_c32_fwd_strided.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_fwd_strided.argtypes = [i32, i32, i32, i32, vp, i32, vp, fp, vp, i32, i32, i32, fp, vp]
    h.dfm_conv3d_k3_c32_to1_fwd.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_to1_fwd.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, i32, vp]
    h.dfm_group_norm_coefficients.restype = ctypes.c_int
    h.dfm_group_norm_coefficients.argtypes = [i32, i32, i32, ctypes.c_float, vp, i32, vp, vp, vp, vp]
    h.dfm_conv3d_to1_bwd_data.restype = ctypes.c_int
    h.dfm_conv3d_to1_bwd_data.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, vp]
    h.dfm_conv3d_to1_wgrad_workspace_bytes.restype = sz
    h.dfm_conv3d_to1_wgrad_workspace_bytes.argtypes = []
    h.dfm_conv3d_to1_wgrad.restype = ctypes.c_int
    h.dfm_conv3d_to1_wgrad.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, sz, vp]
    h.dfm_conv3d_to1_norm_fwd.restype = ctypes.c_int
    h.dfm_conv3d_to1_norm_fwd.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, vp, i32, vp]
    h.dfm_cost_gate_mfma_weight_bytes.restype = sz
    h.dfm_cost_gate_mfma_weight_bytes.argtypes = [i32]
    h.dfm_cost_gate_mfma_pack_weights.restype = ctypes.c_int
    h.dfm_cost_gate_mfma_pack_weights.argtypes = [vp, i32, i32, vp, vp]
    h.dfm_cost_gate_mfma_fwd.restype = ctypes.c_int
    h.dfm_cost_gate_mfma_fwd.argtypes = [i32, i32, ctypes.c_int64, vp, vp, vp, vp, vp]
    h.dfm_bilinear_resize_bwd_nhwc.restype = ctypes.c_int
    h.dfm_bilinear_resize_bwd_nhwc.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, vp, fp, i32, vp, fp, i32, vp, vp]
    for fn in (h.dfm_depth_pool_fwd, h.dfm_depth_pool_bwd):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int64, i32, ctypes.c_int64, i32, vp, vp, vp]
    h.dfm_conv3d_k3_c32_stats_splits.restype = ctypes.c_int
    h.dfm_conv3d_k3_c32_stats_splits.argtypes = [i32, i32, i32, i32, i32]
    cp = ctypes.POINTER(Conv3dDesc)
    h.dfm_conv3d_g_weight_bytes.restype = sz
    h.dfm_conv3d_g_weight_bytes.argtypes = [i32, i32]
    h.dfm_conv3d_g_pack_weights.restype = ctypes.c_int
    h.dfm_conv3d_g_pack_weights.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    h.dfm_conv3d_g_pack_weights_2d.restype = ctypes.c_int
    h.dfm_conv3d_g_pack_weights_2d.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    h.dfm_conv3d_g_fwd.restype = ctypes.c_int
    h.dfm_conv3d_g_fwd.argtypes = [cp, vp, vp, fp, fp, vp, vp, vp]
    h.dfm_conv3d_g_fwd_f32.restype = ctypes.c_int
    h.dfm_conv3d_g_fwd_f32.argtypes = [cp, vp, vp, vp, vp, vp]
    h.dfm_conv3d_g_plan.restype = ctypes.c_int
    h.dfm_conv3d_g_plan.argtypes = [cp, ctypes.POINTER(ctypes.c_int64)]
    wp = ctypes.POINTER(Conv3dWgradDesc)
    h.dfm_conv3d_wgrad_workspace_bytes.restype = sz
    h.dfm_conv3d_wgrad_workspace_bytes.argtypes = [wp]
    h.dfm_conv3d_wgrad_to.restype = ctypes.c_int
    h.dfm_conv3d_wgrad_to.argtypes = [wp, vp, vp, vp, i32, vp, sz, vp]
    h.dfm_conv3d_wgrad.restype = ctypes.c_int
    h.dfm_conv3d_wgrad.argtypes = [wp, vp, vp, fp, vp, sz, vp]
    lp = ctypes.POINTER(DepthLossDesc)
    h.dfm_depth_loss_fwd.restype = ctypes.c_int
    h.dfm_depth_loss_fwd.argtypes = [lp, vp, fp, fp, fp, vp, vp]
    h.dfm_depth_loss_bwd.restype = ctypes.c_int
    h.dfm_depth_loss_bwd.argtypes = [lp, vp, fp, fp, fp, vp, vp]
    h.dfm_depth_loss_fused_fwd.restype = ctypes.c_int
    h.dfm_depth_loss_fused_fwd.argtypes = [lp, vp, i32, fp, fp, fp, vp, vp]
    h.dfm_depth_loss_fused_bwd.restype = ctypes.c_int
    h.dfm_depth_loss_fused_bwd.argtypes = [lp, vp, i32, fp, fp, fp, fp, vp]
    h.dfm_voxel_sample_fwd.restype = ctypes.c_int
    h.dfm_voxel_sample_fwd.argtypes = [ctypes.POINTER(VsDesc), vp, fp, vp, vp]
    h.dfm_voxel_sample_bwd.restype = ctypes.c_int
    h.dfm_voxel_sample_bwd.argtypes = [ctypes.POINTER(VsDesc), vp, fp, fp, vp]
    h.dfm_spp_tail_workspace_bytes.restype = ctypes.c_size_t
    h.dfm_spp_tail_workspace_bytes.argtypes = [ctypes.POINTER(SppDesc)]
    h.dfm_spp_tail_fwd.restype = ctypes.c_int
    pp = ctypes.POINTER(ctypes.c_void_p)
    h.dfm_spp_tail_fwd.argtypes = [ctypes.POINTER(SppDesc), pp, pp, pp, pp, pp, vp, vp, sz, vp]
    i64, f32 = ctypes.c_int64, ctypes.c_float
    h.dfm_group_norm_workspace_bytes.restype = sz
    h.dfm_group_norm_workspace_bytes.argtypes = [i32, i32, i64, i32]
    h.dfm_group_norm_fwd.restype = ctypes.c_int
    h.dfm_group_norm_fwd.argtypes = [i32, i32, i64, i32, f32, i32, i32, vp, fp, fp, vp, fp, fp, vp, sz, vp]
    h.dfm_group_norm_fwd_channels_last.restype = ctypes.c_int
    h.dfm_group_norm_fwd_channels_last.argtypes = h.dfm_group_norm_fwd.argtypes
    h.dfm_group_norm_apply_channels_last.restype = ctypes.c_int
    h.dfm_group_norm_apply_channels_last.argtypes = [i32, i32, i64, i32, f32, i32, i32, vp, fp, fp, vp, fp, fp, fp,
                                                     i32, vp, sz, vp]
    h.dfm_group_norm_fwd_channels_last_res.restype = ctypes.c_int
    h.dfm_group_norm_fwd_channels_last_res.argtypes = [i32, i32, i64, i32, f32, i32, i32, vp, fp, fp, vp, vp, fp, fp,
                                                       vp, sz, vp]
    h.dfm_group_norm_apply_channels_last_res.restype = ctypes.c_int
    h.dfm_group_norm_apply_channels_last_res.argtypes = [i32, i32, i64, i32, f32, i32, i32, vp, fp, fp, vp, vp, fp,
                                                         fp, fp, i32, vp, sz, vp]
    h.dfm_group_norm_bwd.restype = ctypes.c_int
    h.dfm_group_norm_bwd.argtypes = [i32, i32, i64, i32, i32, i32, vp, vp, vp, fp, fp, fp, vp, fp, fp, vp, sz,
                                     vp]
    h.dfm_group_norm_bwd_channels_last_xmask.restype = ctypes.c_int
    h.dfm_group_norm_bwd_channels_last_xmask.argtypes = [i32, i32, i64, i32, i32, vp, vp, fp, fp, fp, fp, vp, fp, fp, vp, sz, vp]
    h.dfm_group_norm_bwd_channels_last.restype = ctypes.c_int
    h.dfm_group_norm_bwd_channels_last.argtypes = [i32, i32, i64, i32, i32, i32, vp, vp, vp, fp, fp, fp, vp, vp, fp,
                                                   fp, vp, sz, vp]
    _lib = h
    return h


def check(rc):
    if rc != 0:
        msg = lib().dfm_last_error().decode('utf-8', 'replace')
        raise DfmHipError(f'libdfm_hip error {rc}: {msg}')
