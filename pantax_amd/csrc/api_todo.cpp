// api_todo.cpp -- entry points whose stages are not built yet in this commit; they fail loudly.
#include "common.hpp"
using namespace ptx;
extern "C" {
int pantax_hip_profile(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *) { return fail(ctx, PANTAX_HIP_E_STATE, "profile: not built yet"); }
}
