// placeholder until the phase-shift kernels land (replaced in the next commit)
#include "common.h"
extern "C" int impdar_phaseshift(impdar_ctx *, const void *, int, int, int, int, const double *, const double *, double, const double *, double, const double *, int, double, double, void *)
{ impdar_set_error("phaseshift not built yet"); return IMPDAR_ERR_UNSUPPORTED; }
extern "C" int impdar_taper(impdar_ctx *, void *, int, int, int, double, double)
{ impdar_set_error("taper not built yet"); return IMPDAR_ERR_UNSUPPORTED; }
