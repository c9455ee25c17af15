// Tracking (placeholder while the acquisition path is brought up).
#include <hip/hip_runtime.h>
#include "bds_internal.h"
namespace bds {
struct TrackState {};
void track_state_free(TrackState *t) { delete t; }
}
extern "C" int bds_track(bds_ctx *ctx, const bds_settings *, const char *, int, const bds_channel *, bds_track_out *) {
    return bds::fail(ctx, BDS_ERR_UNSUPPORTED, "bds_track: not built yet");
}
extern "C" int bds_track_mem(bds_ctx *ctx, const bds_settings *, const int8_t *, size_t, int, const bds_channel *, bds_track_out *) {
    return bds::fail(ctx, BDS_ERR_UNSUPPORTED, "bds_track_mem: not built yet");
}
extern "C" int bds_track_correlate(bds_ctx *ctx, const bds_settings *, const int8_t *, size_t, int, const int32_t *, const double *, double *) {
    return bds::fail(ctx, BDS_ERR_UNSUPPORTED, "bds_track_correlate: not built yet");
}
