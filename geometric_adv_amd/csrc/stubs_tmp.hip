#include "common.h"
extern "C" size_t geoadv_approx_match_temp_floats(int, int, int) { return 0; }
extern "C" size_t geoadv_ae_workspace_bytes(const geoadv_ae *, int) { return 0; }
