// placeholder until the rocFFT path lands (replaced in the next commit)
#include "common.h"
extern "C" int impdar_stolt(impdar_ctx *, const void *, int, int, int, const double *, const double *, double, double, double, void *)
{ impdar_set_error("stolt not built yet"); return IMPDAR_ERR_UNSUPPORTED; }
extern "C" int impdar_stolt_dev(impdar_ctx *, const void *, int, int, int, const double *, const double *, double, double, double, void *)
{ impdar_set_error("stolt not built yet"); return IMPDAR_ERR_UNSUPPORTED; }
