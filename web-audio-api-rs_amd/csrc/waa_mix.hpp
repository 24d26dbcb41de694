// waa_mix.hpp — AudioRenderQuantum::mix (src/render/quantum.rs:285-505) on register tiles: v[C][K] holds K values per lane of
// up to C channels; `from` channels are mixed to `to` (speakers: the W3C up- / down-mix table for 1, 2, 4, 6 channels;
// discrete and every other pair: pad with silence / truncate).  Shared by the static chain kernels (waa_kernels.hip) and the
// dynamic-count kernel (waa_dyn.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace waa {
namespace {

template <int C, int K>
__device__ __forceinline__ void mix_regs(float (&v)[C][K], int from, int to, int interp) {
  constexpr int TILE_K = K;
  if (from == to) return;
  if (interp == 1 || from > 6 || to > 6) {  // discrete: pad with silence / truncate
#pragma unroll
    for (int c = 0; c < C; c++)
      if (c >= from && c < to) {
#pragma unroll
        for (int i = 0; i < TILE_K; i++) v[c][i] = 0.f;
      }
    return;
  }
  if constexpr (C >= 2) {
    if (from == 1 && to == 2) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) v[1][i] = v[0][i];
      return;
    }
    if (from == 2 && to == 1) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) v[0][i] = 0.5f * (v[0][i] + v[1][i]);
      return;
    }
  }
  if constexpr (C >= 4) {
    if (from == 1 && to == 4) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[1][i] = v[0][i];
        v[2][i] = 0.f;
        v[3][i] = 0.f;
      }
      return;
    }
    if (from == 2 && to == 4) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[2][i] = 0.f;
        v[3][i] = 0.f;
      }
      return;
    }
    if (from == 4 && to == 1) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) v[0][i] = 0.25f * (v[0][i] + v[1][i] + v[2][i] + v[3][i]);
      return;
    }
    if (from == 4 && to == 2) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[0][i] = 0.5f * (v[0][i] + v[2][i]);
        v[1][i] = 0.5f * (v[1][i] + v[3][i]);
      }
      return;
    }
  }
  if constexpr (C >= 6) {
    const float sqrt05 = 0.70710678118654752440f;
    if (from == 1 && to == 6) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[2][i] = v[0][i];
        v[0][i] = v[1][i] = v[3][i] = v[4][i] = v[5][i] = 0.f;
      }
      return;
    }
    if (from == 2 && to == 6) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) v[2][i] = v[3][i] = v[4][i] = v[5][i] = 0.f;
      return;
    }
    if (from == 4 && to == 5) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[4][i] = v[3][i];
        v[3][i] = v[2][i];
        v[2][i] = 0.f;
      }
      return;
    }
    if (from == 4 && to == 6) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[4][i] = v[2][i];
        v[5][i] = v[3][i];
        v[2][i] = v[3][i] = 0.f;
      }
      return;
    }
    if (from == 6 && to == 1) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++)
        v[0][i] = __builtin_fmaf(sqrt05, v[0][i] + v[1][i], __builtin_fmaf(0.5f, v[4][i] + v[5][i], v[2][i]));
      return;
    }
    if (from == 6 && to == 2) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        v[0][i] += sqrt05 * (v[2][i] + v[4][i]);
        v[1][i] += sqrt05 * (v[2][i] + v[5][i]);
      }
      return;
    }
    if (from == 6 && to == 4) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        float c = v[2][i];
        v[0][i] += sqrt05 * c;
        v[1][i] += sqrt05 * c;
        v[2][i] = v[4][i];
        v[3][i] = v[5][i];
      }
      return;
    }
  }
  // all other speaker layouts: pad with silence / truncate
#pragma unroll
  for (int c = 0; c < C; c++)
    if (c >= from && c < to) {
#pragma unroll
      for (int i = 0; i < TILE_K; i++) v[c][i] = 0.f;
    }
}


// the pairs of mix_regs that COMPUTE their result (a down-mix through make_mut: the quantum is no longer silent by pointer
// identity afterwards, quantum.rs:96-104) — everything else copies channels or pads with the silent block
__host__ __device__ inline bool mix_is_computed(int from, int to, int interp) {
  if (interp == 1 || from > 6 || to > 6) return false;
  return (from == 2 && to == 1) || (from == 4 && to == 1) || (from == 6 && to == 1) || (from == 4 && to == 2) ||
         (from == 6 && to == 2) || (from == 6 && to == 4);
}

}  // namespace
}  // namespace waa
