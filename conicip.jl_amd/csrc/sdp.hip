// S-cone (semidefinite) kernels -- device counterparts of nestod_sdc, VecCongurance,
// xsdc!/dsdc!, maxstep_sdc (src/ConicIP.jl:35-40, :69, :196-210, :272-303, :347-360).
// Round-1 state: not yet implemented on the device; every entry point reports
// CIP_E_UNSUPPORTED (cip_create refuses S cones), nothing falls back to the CPU.
#include "cip_internal.h"
#include "../../include/cipkkt.h"

static int unsupported(const char *what) {
    cip_set_error("S cones: %s not implemented on the device yet", what);
    return CIP_E_UNSUPPORTED;
}
int cip_sdp_nt_scaling(hipStream_t, const ConeSet &, const double *, const double *, double *) { return unsupported("nt_scaling"); }
int cip_sdp_apply(hipStream_t, const ConeSet &, int, const double *, double *) { return unsupported("apply"); }
int cip_sdp_prod(hipStream_t, const ConeSet &, const double *, const double *, double *) { return unsupported("cone_prod"); }
int cip_sdp_div(hipStream_t, const ConeSet &, const double *, const double *, double *) { return unsupported("cone_div"); }
int cip_sdp_maxstep(hipStream_t, const ConeSet &, const double *, const double *, double, double *) { return unsupported("maxstep"); }
int cip_sdp_scale_At(hipStream_t, const ConeSet &, int, const double *, long, double *, long) { return unsupported("scale_At"); }
