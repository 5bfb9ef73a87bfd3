"""cipkkt -- MI355X-native KKT-solve path for ConicIP-style interior-point solvers.

Host-side mirror (Python over ctypes) of the reference's plugin interface for its
Newton-step hot path, bound to the hand-written HIP library ``libcipkkt.so``
(C ABI: include/cipkkt.h).  GPU-only: importing works anywhere, but any compute
call raises when the library is not built or no HIP device is visible.
"""
from . import _lib
from ._lib import (CONE_Q, CONE_R, CONE_S, MAT_A, MAT_G, MAT_Q, OP_F, OP_FINV, OP_FINVT, OP_FT, ROUTE_FULL3X3,
                   ROUTE_SCHUR, CipError)
from .kkt import KKTSystem, kktsolver_2x2_hip, kktsolver_hip, kktsolver_hip_full3x3, pivot
from . import blocks
from .driver import Solution, conicIP
from .preprocess import imcols, preprocess_conicIP

__all__ = ["KKTSystem", "kktsolver_hip", "kktsolver_hip_full3x3", "kktsolver_2x2_hip", "pivot", "blocks", "conicIP", "Solution", "CipError", "imcols",
           "preprocess_conicIP"]
