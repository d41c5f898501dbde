// ht_camshift.hip — camshift.Tracker on the device (placeholder until the kernels land).
#include "ht_internal.h"

extern "C" ht_status ht_camshift_reserve(ht_ctx *c, int32_t) { return ht_fail(c, HT_ERR_STATE, "camshift: not built yet"); }
extern "C" ht_status ht_camshift_init_batch(ht_ctx *c, int32_t, int32_t, const ht_cs_rect *) { return ht_fail(c, HT_ERR_STATE, "camshift: not built yet"); }
extern "C" ht_status ht_camshift_track_batch(ht_ctx *c, int32_t, int32_t, int32_t, ht_cs_trackobj *) { return ht_fail(c, HT_ERR_STATE, "camshift: not built yet"); }
extern "C" ht_status ht_allgather_records(ht_ctx *const *, int32_t, void *const *, size_t) { return HT_ERR_STATE; }
