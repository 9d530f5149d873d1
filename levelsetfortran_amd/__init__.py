"""levelsetfortran_amd -- MI355X (gfx950) implementation of the hot path of musheen/LevelSetFortran:
WENO5 Hamilton-Jacobi signed-distance reinitialisation and min/max-flow smoothing on a uniform 3-D
grid, behind the reference's own procedure interface (see levelset.py, include/lsf.h).
"""
from .levelset import (LsfError, LsfNaNError, SweepReport, advectNodes, minmaxFlow, mode_word, narrowBand, peer_selftest,  # noqa: F401
                       phi0Init, reinit, reinit_multi)
from . import fields  # noqa: F401

__version__ = "0.1.6"
