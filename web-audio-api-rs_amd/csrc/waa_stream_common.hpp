// waa_stream_common.hpp — device helpers shared by the streaming recurrence kernels
// (waa_biquad_stream.hip, waa_iir_stream.hip): tile geometry, DPP / readlane helpers and the rare-case
// source loader.
#pragma once
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {
namespace {
constexpr int NV4 = TILE_K / 4;
constexpr int LDS_ROW = TILE_K + 4;

// DPP row shift right by N lanes within each row of 16; lanes without a source read 0.
template <int N>
__device__ __forceinline__ double row_shr(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}
// same shift, but a lane without a source keeps `fallback` (used to compose with the identity map)
template <int N>
__device__ __forceinline__ double row_shr_keep(double v, double fallback) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(__double2loint(fallback), lo, 0x110 + N, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(__double2hiint(fallback), hi, 0x110 + N, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// generic (rare) input path for one channel: end-of-buffer / loop wrap / slow track / silent quanta
__device__ __noinline__ void load_channel_generic(const InputRef& in, const SrcInst& si, const SrcSchedule& sc, int ch,
                                                  uint32_t tile, int lane, uint32_t n_quanta, float* out32) {
  const float* chp = si.base + (uint64_t)ch * si.ch_stride;
  if (si.linear_all) {
    // the render's partial last tile of a source that is one linear run (SrcInst::linear_all; every tile in front of it is
    // in the fast prefix): vector loads bounded by the render's end instead of eight dependent record / sample round trips
    // at the end of EVERY wave
    const float* p = chp + si.linear_start + (int64_t)tile * TILE;
    const uint64_t f_end = (uint64_t)n_quanta * RQ;
    for (int j = 0; j < NV4; j++) {
      const uint64_t f = (uint64_t)tile * TILE + j * 256 + lane * 4;
      const bool ok = f + 3 < f_end;
      f4v t = load_global_f4(ok ? p + j * 256 + lane * 4 : chp + si.linear_start);
      if (!ok) t = f4v{0.f, 0.f, 0.f, 0.f};
      out32[j * 4 + 0] = t.x;
      out32[j * 4 + 1] = t.y;
      out32[j * 4 + 2] = t.z;
      out32[j * 4 + 3] = t.w;
    }
    return;
  }
  for (int j = 0; j < NV4; j++) {
    const uint32_t fq = j * 256 + lane * 4;
    const uint32_t q = tile * QUANTA_PER_TILE + fq / RQ;
    const bool valid_q = q < n_quanta;
    const QRec r = load_global(sc.qrec + (valid_q ? q : 0));
    const uint32_t mode = valid_q ? r.mode : (uint32_t)Q_SILENT;
    for (int e = 0; e < 4; e++) {
      const uint32_t i = (fq % RQ) + e;
      float o = 0.f;
      if (mode == Q_FAST || mode == Q_FAST_LOOP) {
        uint64_t bi = (uint64_t)r.start + i;
        bool ok = true;
        if (bi >= si.frames) {
          if (mode == Q_FAST_LOOP)
            bi = bi % si.frames;
          else
            ok = false;
        }
        o = ok ? load_global(chp + bi) : 0.f;
      } else if (mode == Q_SLOW) {
        const SlowRec s = load_global(sc.slow + (uint64_t)q * RQ + i);
        if (s.prev >= 0) {
          const double prev_sample = (double)load_global(chp + s.prev);
          double next_sample;
          if (s.next >= 0)
            next_sample = (double)load_global(chp + s.next);
          else if (s.next == -1)
            next_sample = 0.;
          else
            next_sample = 2. * prev_sample - (double)load_global(chp + s.prev - 1);
          o = (float)__builtin_fma(1. - s.k, prev_sample, s.k * next_sample);
        }
      }
      out32[j * 4 + e] = o;
    }
  }
}


}  // namespace
}  // namespace waa
