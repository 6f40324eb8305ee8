"""Drop-in for the reference's ``droid_backends`` extension module.

Put ``<repo>/vings_mono_amd/dropin`` on PYTHONPATH and the imports at
/root/reference/scripts/frontend/depth_video.py:4 and
/root/reference/scripts/frontend/modules/corr.py:4 resolve here.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from vings_mono_amd.dba import (BACore, altcorr_backward, altcorr_forward, ba,  # noqa: E402,F401
                                corr_index_backward, corr_index_forward, frame_distance)

__all__ = ["ba", "BACore", "frame_distance", "corr_index_forward", "corr_index_backward",
           "altcorr_forward", "altcorr_backward"]
