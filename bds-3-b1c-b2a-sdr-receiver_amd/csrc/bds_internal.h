// Internal declarations shared by the translation units of libbds_mi355x.so.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bds_mi355x.h"

namespace bds {

// host code generation (bds_codes.cpp)
int gen_primary(int signal, bool pilot, int prn, int8_t *out /*10230*/);
int gen_secondary(int prn, int8_t *out /*1800*/);

// MATLAB round(): half away from zero
inline double m_round(double x) { return x >= 0 ? (double)(long long)(x + 0.5) : -(double)(long long)(-x + 0.5); }

// samplesPerCode = round(fs / (codeFreqBasis / codeLength))   (B2a/acquisition.m:130-131)
inline long samples_per_code(const bds_settings &s) {
    return (long)m_round(s.samplingFreq / (s.codeFreqBasis / s.codeLength));
}

struct AcqState;    // bds_acq.hip
struct TrackState;  // bds_track.hip
void acq_state_free(AcqState *);
void track_state_free(TrackState *);

}  // namespace bds

struct bds_ctx {
    int device = 0;
    void *stream = nullptr;   // hipStream_t: everything is ordered on this stream ...
    void *stream2 = nullptr;  // ... except the search's column pass, which overlaps the next row pass
    std::string err;
    std::string devname;
    bds::AcqState *acq = nullptr;
    bds::TrackState *trk = nullptr;
    bds_timing timing{};
};

namespace bds {
int fail(bds_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
}

#define BDS_HIP(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return bds::fail((ctx), BDS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);                       \
    } while (0)
