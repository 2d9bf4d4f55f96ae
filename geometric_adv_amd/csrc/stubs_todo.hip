// Entry points declared in include/geoadv.h whose kernels land later in the round.
#include "common.h"
#define TODO(name) set_error(name ": not implemented in this build yet"); return GEOADV_EINVAL
using namespace geoadv;
extern "C" size_t geoadv_approx_match_temp_floats(int b, int n, int m) { return (size_t)b * (n + m) * 2; }
extern "C" int geoadv_approx_match(int, int, int, const float *, const float *, float *, float *, void *) { TODO("approx_match"); }
extern "C" int geoadv_match_cost(int, int, int, const float *, const float *, const float *, float *, void *) { TODO("match_cost"); }
extern "C" int geoadv_match_cost_grad(int, int, int, const float *, const float *, const float *, float *, float *, void *) { TODO("match_cost_grad"); }
