// Bilinear-sampling geometry shared by the DCNv2 kernels.
#pragma once
#include "common.h"

struct Tap {
    float w00, w01, w10, w11;   // bilinear corner weights (0 where the corner is outside the image)
    float lh, lw;
    int h0, w0;
    bool ok00, ok01, ok10, ok11;
};

__device__ static inline Tap make_tap(float py, float px, int H, int W) {
    Tap t;
    const float fh = floorf(py), fw = floorf(px);
    t.h0 = (int)fh;
    t.w0 = (int)fw;
    t.lh = py - fh;
    t.lw = px - fw;
    const bool h0ok = t.h0 >= 0 && t.h0 <= H - 1, h1ok = t.h0 + 1 >= 0 && t.h0 + 1 <= H - 1;
    const bool w0ok = t.w0 >= 0 && t.w0 <= W - 1, w1ok = t.w0 + 1 >= 0 && t.w0 + 1 <= W - 1;
    t.ok00 = h0ok && w0ok; t.ok01 = h0ok && w1ok; t.ok10 = h1ok && w0ok; t.ok11 = h1ok && w1ok;
    t.w00 = t.ok00 ? (1.f - t.lh) * (1.f - t.lw) : 0.f;
    t.w01 = t.ok01 ? (1.f - t.lh) * t.lw : 0.f;
    t.w10 = t.ok10 ? t.lh * (1.f - t.lw) : 0.f;
    t.w11 = t.ok11 ? t.lh * t.lw : 0.f;
    return t;
}

__device__ static inline float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

