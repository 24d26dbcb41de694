// waa_hrtf_fft_tables.hpp — host side of waa_hrtf_fft.hip: the partition spectra H_p of one HRIR pair (both ears in one complex
// table, 1 / 256 folded in) in the lane-major layout the kernel reads (waa_hrtf_fft.hpp).  Host-only; shared by
// waa_frozen_host.cpp and tools/hrtf_fft_emulate.cpp.
#pragma once
#include <cmath>
#include <vector>

#include "waa_hrtf_fft.hpp"
#include "waa_osfft_tables.hpp"

namespace waa {
namespace hrtffft {

// pair: [>= taps][2] interleaved (left, right), as plan_hrtf builds it for hrtf8_kernel; taps <= 128 * PARTS
inline std::vector<float> make_tables(const float* pair, int taps) {
  const double two_pi = 6.283185307179586476925286766559;
  std::vector<float> out((size_t)PARTS * TAB_SLOTS * 2, 0.f);
  // (one table of cos / sin: the exponent only enters modulo 256)
  double cs[256], sn[256];
  for (int j = 0; j < 256; j++) {
    cs[j] = std::cos(-two_pi * (double)j / 256.);
    sn[j] = std::sin(-two_pi * (double)j / 256.);
  }
  for (int p = 0; p < PARTS; p++)
    for (int kp = 0; kp < 256; kp++) {
      double re = 0., im = 0.;
      for (int n = 0; n < 128; n++) {
        const int tap = 128 * p + n;
        if (tap >= taps) break;
        const double hl = (double)pair[(size_t)tap * 2], hr = (double)pair[(size_t)tap * 2 + 1];
        const int e = (kp * n) & 255;
        // (hl + i hr) (cs + i sn)
        re += hl * cs[e] - hr * sn[e];
        im += hl * sn[e] + hr * cs[e];
      }
      const int t = kp & 15, j = kp >> 4;  // kp = t + 16 j, slot s with K16(s) = j
      const size_t slot = (size_t)t * ROW + (size_t)K16(j);
      out[((size_t)p * TAB_SLOTS + slot) * 2 + 0] = (float)(re / 256.);
      out[((size_t)p * TAB_SLOTS + slot) * 2 + 1] = (float)(im / 256.);
    }
  return out;
}

}  // namespace hrtffft
}  // namespace waa
