// vit.hip -- DINOv2 ViT-S/14 forward (row A1). Placeholder translation unit: replaced by the MFMA
// implementation in a later commit of this round; until then the entry points report VFM_EINVAL.
#include "common.h"

VFM_EXPORT size_t vfm_vit_weights_bytes(const vfm_vit_config*) { return 0; }
VFM_EXPORT size_t vfm_vit_workspace_bytes(const vfm_vit_config*, int) { return 0; }
VFM_EXPORT int vfm_vit_forward(const vfm_vit_config*, const void*, const uint8_t*, int, int, int, float*, void*, size_t,
                               vfm_stream_t) {
    return vfm_fail(VFM_EINVAL, "vfm_vit_forward: not built yet");
}
