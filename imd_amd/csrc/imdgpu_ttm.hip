// imdgpu_ttm.hip -- two-temperature-model kernels (placeholder until the TTM
// milestone; the force path does not depend on it).
#include "imdgpu_internal.h"

struct TtmState { int dummy; };

int imdgpu_ttm_step_internal(imdgpu_ctx *ctx, int steps) { (void) steps; return imdgpu_fail(ctx, "TTM not initialised"); }
void imdgpu_ttm_free(imdgpu_ctx *ctx) { delete ctx->ttm; ctx->ttm = nullptr; }
