// pnp.hip -- placeholder until the batched DLS-PnP/RANSAC kernels land (next milestone).
#include "chip_internal.h"
namespace chip {
int pnp_create(Ctx *) { return CHIP_OK; }
void pnp_destroy(Ctx *) {}
}
extern "C" int chip_pnp_ransac(chip_ctx *, const double *, const double *, int32_t, const chip_ransac_params *,
                               double *, float *, uint8_t *, chip_ransac_summary *)
{
    return CHIP_ERR_UNSUPPORTED;
}
