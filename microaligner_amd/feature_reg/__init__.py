"""Feature-based affine registration (counterpart of microaligner/feature_reg)."""
from .feature_registrator import FeatureRegistrator

__all__ = ["FeatureRegistrator"]
