"""Drop-in for the reference's ``diff_surfel_rasterization`` extension module.

Put ``<repo>/vings_mono_amd/dropin`` on PYTHONPATH and the imports at
/root/reference/scripts/gaussian/gaussian_base.py:5-6 resolve here:

    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from diff_surfel_rasterization import SparseGaussianAdam
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from vings_mono_amd.surfel import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: E402,F401
                                   SparseGaussianAdam)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "SparseGaussianAdam"]
