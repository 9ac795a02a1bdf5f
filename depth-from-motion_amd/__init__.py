"""depth-from-motion_amd -- MI355X (gfx950) native plane-sweep cost-volume path
of Depth-from-Motion.

The directory name carries a hyphen (it is the project name); import it as
``import dfm_amd`` (alias module at the repo root) or
``importlib.import_module('depth-from-motion_amd')``.

Everything numerical runs in hand-written HIP kernels behind the C ABI of
``include/dfm_hip.h`` (``lib/libdfm_hip.so``).  There is NO CPU or eager
PyTorch fallback: if the shared library is missing or no GPU is visible the
ops raise.
"""
from . import _capi  # noqa: F401
from . import modules, registry  # noqa: F401  (registers the module classes)
from .plane_sweep import build_dfm_cost, plane_sweep_grid  # noqa: F401
from .depth_head import LazyDepthDistribution, depth_head_forward, depth_head_statistics  # noqa: F401
from .group_norm import HipGroupNorm, group_norm  # noqa: F401
from .geometry import prepare_coordinates_3d, prepare_depth  # noqa: F401
from .frustum_to_voxel import frustum_to_voxel_sample  # noqa: F401
from .integration import (DfMStereoPath, MultiViewDfMMixin, MultiViewVoxelPath,  # noqa: F401
                          enable_fast_path, inject_detector_attributes, patch_reference)
from .conv3d import MfmaPathError, fallback_policy, set_fallback_policy, set_fp32_mode  # noqa: F401
from .depth_head import depth_distribution_loss  # noqa: F401
from .data_geometry import (fold_ref_frame_matrices, select_ref_frames, stage_geometry,  # noqa: F401
                            video_cur2prevs)
from .point_sample import (mv_feature_transformation, point_sample, voxel_centers,  # noqa: F401
                           voxel_sample)

__all__ = ['build_dfm_cost', 'plane_sweep_grid', 'point_sample', 'mv_feature_transformation',
           'voxel_centers', 'voxel_sample', 'frustum_to_voxel_sample', 'depth_head_forward', 'prepare_depth',
           'prepare_coordinates_3d', 'group_norm', 'HipGroupNorm', 'DfMStereoPath', 'MultiViewDfMMixin',
           'MultiViewVoxelPath', 'inject_detector_attributes', 'patch_reference', 'enable_fast_path', 'set_fallback_policy', 'set_fp32_mode',
           'fallback_policy', 'MfmaPathError', 'select_ref_frames', 'fold_ref_frame_matrices', 'video_cur2prevs',
           'stage_geometry', 'depth_distribution_loss',
           'depth_head_statistics', 'LazyDepthDistribution']
