"""dhd_amd -- MI355X (gfx950) implementation of DHD's height-decoupled LSS view transform.

Host-side mirror of the reference's plugin surface for this path
(projects/mmdet3d_plugin: ops/bev_pool_v2, models/necks/lss_heightmap.py, models/necks/mix.py),
over the C ABI of csrc/libdhd_amd.so (include/dhd_amd.h).  Importing the package does not need
a GPU; calling any operator does, and fails loudly without the HIP library.
"""
from .registry import NECKS, BACKBONES, HEADS, DETECTORS, HOOKS, build_neck, build_backbone, build_head, build_detector, build_hook  # noqa: F401
from .bev_pool_v2 import bev_pool_v2, QuickCumsumCuda  # noqa: F401
from .lss_heightmap import MGHS, MGHS_Depth, MGHS_Stereo  # noqa: F401
from .mix import SFA, channel_spatial_stage  # noqa: F401
from .depthnet import HeightNet, DepthNet  # noqa: F401
from .detector import DHD  # noqa: F401  (also registers ResNet, CustomFPN, CustomResNet, FPN_LSS, UNet, Identity, predictor)
from .swin import SwinTransformer  # noqa: F401
from .ema import ModelEMA, MEGVIIEMAHook, SyncbnControlHook, SequentialControlHook  # noqa: F401
from .config import Config  # noqa: F401
from .checkpoint import load_checkpoint, load_state_dict, save_checkpoint  # noqa: F401
from .amp_weights import HalfWeightCache  # noqa: F401

__version__ = '0.1.0'
