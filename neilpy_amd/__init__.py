"""neilpy_amd: MI355X-native implementation of neilpy's SMRF bare-earth path.

Drop-in for ``neilpy.smrf`` / ``progressive_filter`` / ``create_dem`` /
``inpaint_nans_by_springs`` (and the ``disk`` / ``opening`` seam they use).  All compute runs
in hand-written gfx950 HIP kernels in ``libsmrf_hip.so`` (C ABI: ``include/smrf_hip.h``);
there is no CPU fallback.
"""
from ._lib import SmrfHipError, load as load_library, LIB_PATH          # noqa: F401
from .affine import Affine, edges_from_IT, from_origin, write_worldfile                 # noqa: F401
from .api import (create_dem, dilation, disk, erosion, inpaint_nans_by_fda, inpaint_nans_by_springs,   # noqa: F401
                  last_stats,
                  opening, progressive_filter, pssm, smrf)
from .las import read_las, read_las_xyz, write_las                         # noqa: F401
from .synth import synth_dem, synth_points                               # noqa: F401

__version__ = "0.1.0"
