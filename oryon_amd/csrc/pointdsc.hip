// PointDSC registration (K3-K10) -- placeholder entry points while the kernels are being brought up.
#include "common.h"
using namespace oryon;
struct oryon_pointdsc { oryon_pointdsc_config_t cfg; };
#define NOT_YET() do { set_error("%s: not implemented yet", __func__); return ORYON_ERR_STATE; } while (0)
extern "C" int oryon_pointdsc_create(oryon_pointdsc_t **h, const oryon_pointdsc_config_t *cfg) { NOT_YET(); }
extern "C" void oryon_pointdsc_destroy(oryon_pointdsc_t *h) {}
extern "C" int oryon_pointdsc_load_param(oryon_pointdsc_t *h, const char *, const float *, int64_t) { NOT_YET(); }
extern "C" int oryon_pointdsc_finalize(oryon_pointdsc_t *h, void *) { NOT_YET(); }
extern "C" size_t oryon_pointdsc_workspace_bytes(const oryon_pointdsc_t *h, int, int) { return 0; }
extern "C" int oryon_pointdsc_register(oryon_pointdsc_t *, const float *, const float *, const int32_t *, int, int, const int32_t *, void *, size_t, float *, uint8_t *, int32_t *, void *) { NOT_YET(); }
extern "C" int oryon_pointdsc_encode(oryon_pointdsc_t *, const float *, const float *, const int32_t *, int, int, void *, size_t, float *, float *, void *) { NOT_YET(); }
extern "C" int oryon_pointdsc_seeds(oryon_pointdsc_t *, const float *, const float *, const int32_t *, int, int, int, int32_t *, int32_t *, void *) { NOT_YET(); }
extern "C" int oryon_pointdsc_hypotheses(oryon_pointdsc_t *, const float *, const float *, const float *, const int32_t *, const int32_t *, const int32_t *, int, int, int, void *, size_t, float *, float *, int32_t *, void *) { NOT_YET(); }
extern "C" int oryon_pointdsc_refine(oryon_pointdsc_t *, const float *, const float *, const int32_t *, int, int, const float *, float *, uint8_t *, void *) { NOT_YET(); }
