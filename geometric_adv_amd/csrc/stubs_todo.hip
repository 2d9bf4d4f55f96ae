// Entry points declared in include/geoadv.h whose kernels land later in the round.
#include "common.h"
#define TODO(name) set_error(name ": not implemented in this build yet"); return GEOADV_EINVAL
using namespace geoadv;
extern "C" size_t geoadv_approx_match_temp_floats(int b, int n, int m) { return (size_t)b * (n + m) * 2; }
extern "C" int geoadv_approx_match(int, int, int, const float *, const float *, float *, float *, void *) { TODO("approx_match"); }
extern "C" int geoadv_match_cost(int, int, int, const float *, const float *, const float *, float *, void *) { TODO("match_cost"); }
extern "C" int geoadv_match_cost_grad(int, int, int, const float *, const float *, const float *, float *, float *, void *) { TODO("match_cost_grad"); }
extern "C" int geoadv_query_ball_point(int, int, int, float, int, const float *, const float *, int *, int *, void *) { TODO("query_ball_point"); }
extern "C" int geoadv_selection_sort(int, int, int, int, const float *, int *, float *, void *) { TODO("selection_sort"); }
extern "C" int geoadv_group_point(int, int, int, int, int, const float *, const int *, float *, void *) { TODO("group_point"); }
extern "C" int geoadv_group_point_grad(int, int, int, int, int, const float *, const int *, float *, void *) { TODO("group_point_grad"); }
extern "C" int geoadv_knn_point(int, int, int, int, const float *, const float *, float *, int *, void *) { TODO("knn_point"); }
extern "C" int geoadv_knn_dists(int, int, int, const float *, float *, void *) { TODO("knn_dists"); }
