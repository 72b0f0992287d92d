// api_todo.cpp -- entry points whose stages are not built yet in this commit; they fail loudly.
#include "common.hpp"
using namespace ptx;
extern "C" {
int pantax_hip_strain_profile(pantax_hip_ctx *ctx, pantax_hip_db *, const pantax_hip_strain_config *, const uint8_t *, const double *, pantax_hip_hap_metrics *, pantax_hip_solve_info *) { return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: not built yet"); }
int pantax_hip_pao_solve(pantax_hip_ctx *ctx, uint32_t, const int64_t *, const double *, const uint64_t *, uint32_t, const uint64_t *, const uint32_t *, uint32_t, const uint32_t *, const uint8_t *, double *, float *, double *, int32_t *) { return fail(ctx, PANTAX_HIP_E_STATE, "pao_solve: not built yet"); }
int pantax_hip_profile(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *) { return fail(ctx, PANTAX_HIP_E_STATE, "profile: not built yet"); }
}
