// Shared between vae_conv.hip (entry points, the row-tile kernels) and vae_conv3.hip (the 2-D-tile LDS-DMA kernel).
#pragma once
#include "td_common.h"

struct VaeConvP {
  const uint16_t* x;      // [B, Ti, Hi, Wi, Ci] bf16 (batch stride xs_b elements)
  const uint16_t* w;      // [Co, taps * Ci] bf16, K ordered (dt, dh, dw, c)
  const uint16_t* bias;   // [Co] or null
  const uint16_t* res;    // same layout as y, or null
  uint16_t* y;            // [B, To', Ho, Wo, Co'] bf16 (batch stride ys_b elements)
  int64_t xs_b, ys_b;
  int B, Ti, Hi, Wi, Ci;
  int To, Ho, Wo, Co;     // the GEMM's output grid (To = Ti; Ho = Hi or 2 Hi)
  int kt, kh, kw;
  int up2;                // 1: nearest x2 up-sampling of H, W before the convolution
  int interleave;         // 1: time up-sampler output mapping (y has 2 To frames of Co / 2 channels)
  int64_t M;              // B * To * Ho * Wo
  int halves;             // taps * Ci / 32
  int st, ss;             // output strides in time / space (the encoder's down-samplers: 2); 1 = plain
  int t_fast;             // row-tile kernel: tile order (w tiles, FRAMES, rows) instead of (w tiles, rows, frames)
  int pt, ph, pw;         // zero frames / rows / columns on the LEFT (kt - 1, kh / 2, kw / 2 = causal in time, centred in space;
                          // the encoder's ZeroPad2d((0, 1, 0, 1)) and its unpadded stride-2 time convolution pass 0); whatever the
                          // output grid reaches beyond the right edge is zero too
};

// vae_conv3.hip: eligible = plain geometry (stride 1, centred in space, causal in time), 3x3 in space, C_in % 32 == 0,
// C_out % 96 == 0 or <= 32, no time-up-sampler mapping.  order bit 0: 0 = (n, w, h) tiles of a frame then the next frame, 1 = frames
// first; bit 1: 256-position tiles (16 x 16) with two workgroups per CU instead of 512-position tiles with one.
bool vae_conv3_eligible(const VaeConvP& p, bool plain);
int vae_conv3_launch(const VaeConvP& p, int order, hipStream_t st);
