"""microaligner_amd -- MI355X-native implementation of microaligner's optical-flow registration
hot path (OptFlowRegistrator.register() + Warper.warp()) behind the reference's Python API.

    from microaligner_amd import OptFlowRegistrator, Warper

mirrors `from microaligner import OptFlowRegistrator, Warper` (microaligner/__init__.py:18-20).
Compute runs in hand-written HIP kernels for gfx950 behind a C-ABI (include/microaligner_hip.h);
there is no CPU fallback.
"""
from .feature_reg import FeatureRegistrator
from .optflow_reg import OptFlowRegistrator, TileFlowCalc, Warper, farneback, merge_two_flows
from .shared_modules.utils import max_project_and_normalize, pad_to_shape, transform_img_with_tmat

__all__ = ["FeatureRegistrator", "OptFlowRegistrator", "Warper", "TileFlowCalc", "farneback", "merge_two_flows", "pad_to_shape",
           "transform_img_with_tmat", "max_project_and_normalize"]
__version__ = "0.1.0"
