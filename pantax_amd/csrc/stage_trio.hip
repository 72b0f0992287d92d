// placeholder until the radix-sort based trio index lands (next commit)
#include "common.hpp"
namespace ptx { int trio_index_build(Ctx *ctx, Db *) { return fail(ctx, PANTAX_HIP_E_STATE, "trio_index: not built yet"); } }
extern "C" {
int pantax_hip_trio_index(pantax_hip_ctx *ctx, pantax_hip_db *db, uint64_t *) { return ptx::trio_index_build(ctx, db); }
int pantax_hip_trio_get(pantax_hip_ctx *ctx, const pantax_hip_db *, uint32_t *, uint32_t *, int64_t *, uint64_t *) { return ptx::fail(ctx, PANTAX_HIP_E_STATE, "trio_get: not built yet"); }
}
