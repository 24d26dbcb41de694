// waa_osfft_tables.hpp — host side of waa_osfft.hip: the filter of a rubato FftFixedInOut stage (restated)
// and the per-branch spectral tables U_r, V_r of the polyphase form (waa_osfft.hpp), in the lane-major layout the kernel
// reads.  Host-only; shared by waa_frozen_host.cpp and tools/osfft_emulate.cpp.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "waa_osfft.hpp"

namespace waa {
namespace osfft {

// rubato's windowed sinc of fi taps (sinc.rs / windows.rs, BlackmanHarris2, cutoff 0.4^(16 / fi), times fo / fi when
// down-sampling), normalised to unit sum and divided by the transform length 2 fi — evaluated in f32 like the crate
inline std::vector<float> rubato_filter_taps(int fi, int fo) {
  const float cutoff = fi > fo ? std::pow(0.4f, 16.0f / (float)fi) * (float)fo / (float)fi : std::pow(0.4f, 16.0f / (float)fi);
  const float pi = 3.14159265358979323846f;
  const float pi2 = 2.f * pi, pi4 = 4.f * pi, pi6 = 6.f * pi, np = (float)fi;
  std::vector<float> y((size_t)fi);
  float sum = 0.f;
  for (int x = 0; x < fi; x++) {
    const float xf = (float)x;
    const float bh = 0.35875f - 0.48829f * std::cos(pi2 * xf / np) + 0.14128f * std::cos(pi4 * xf / np) - 0.01168f * std::cos(pi6 * xf / np);
    const float arg = (xf - (float)(fi / 2)) * cutoff / 1.f;
    const float sinc = arg == 0.f ? 1.f : std::sin(arg * pi) / (arg * pi);
    const float val = bh * bh * sinc;
    sum += val;
    y[(size_t)x] = val;
  }
  for (int n = 0; n < fi; n++) y[(size_t)n] = (y[(size_t)n] / sum) / (float)(2 * fi);
  return y;
}

// bins 0 .. n_bins-1 of the (2 fi)-point DFT of the taps, f64
inline void filter_spectrum(const std::vector<float>& g, int fi, int n_bins, std::vector<double>* re, std::vector<double>* im) {
  const double two_pi = 6.283185307179586476925286766559;
  re->assign((size_t)n_bins, 0.);
  im->assign((size_t)n_bins, 0.);
  for (int k = 0; k < n_bins; k++) {
    double r = 0., i = 0.;
    for (int n = 0; n < fi; n++) {
      const double a = -two_pi * (double)((int64_t)k * n % (2 * fi)) / (double)(2 * fi);
      r += (double)g[(size_t)n] * std::cos(a);
      i += (double)g[(size_t)n] * std::sin(a);
    }
    (*re)[(size_t)k] = r;
    (*im)[(size_t)k] = i;
  }
}

// 2 R tables of TAB_SLOTS complex f32: U_0 .. U_{R-1}, V_0 .. V_{R-1}; T[t * ROW + s] = table[t + 16 K16(s)]
inline std::vector<float> make_tables(int R) {
  const int fo = 128 * R;
  const double two_pi = 6.283185307179586476925286766559, pi = 3.14159265358979323846264338327950288;
  std::vector<double> ur, ui, dr, di;
  filter_spectrum(rubato_filter_taps(128, fo), 128, 129, &ur, &ui);   // up: bins 0..128 of 256
  filter_spectrum(rubato_filter_taps(fo, 128), fo, 128, &dr, &di);    // down: bins 0..127 of 256 R
  std::vector<float> out((size_t)2 * R * TAB_SLOTS * 2, 0.f);
  for (int r = 0; r < R; r++)
    for (int kp = 0; kp < 256; kp++) {
      const int k = kp <= 128 ? kp : kp - 256;           // the signed bin
      const int ka = k < 0 ? -k : k;
      // U_r: F_up[k] (hermitian) times the branch's shift; bin 128 collects +128 and -128
      double fr = ur[(size_t)ka], fi_ = k < 0 ? -ui[(size_t)ka] : ui[(size_t)ka];
      double sr, si;
      if (kp == 128) {
        sr = 2. * std::cos(pi * (double)r / (double)R);
        si = 0.;
      } else {
        const double a = two_pi * (double)k * (double)r / (double)(256 * R);
        sr = std::cos(a);
        si = std::sin(a);
      }
      const double Ure = fr * sr - fi_ * si, Uim = fr * si + fi_ * sr;
      // V_r: F_dn[k] (hermitian) times e^{-2 pi i r k / 256 R}; nothing at |k| = 128
      double Vre = 0., Vim = 0.;
      if (kp != 128) {
        const double gr = dr[(size_t)ka], gi = k < 0 ? -di[(size_t)ka] : di[(size_t)ka];
        const double a = -two_pi * (double)k * (double)r / (double)(256 * R);
        Vre = gr * std::cos(a) - gi * std::sin(a);
        Vim = gr * std::sin(a) + gi * std::cos(a);
      }
      const int t = kp & 15, j = kp >> 4;                 // kp = t + 16 j, slot s with K16(s) = j  (K16 is an involution)
      const size_t slot = (size_t)t * ROW + (size_t)K16(j);
      out[((size_t)r * TAB_SLOTS + slot) * 2 + 0] = (float)Ure;
      out[((size_t)r * TAB_SLOTS + slot) * 2 + 1] = (float)Uim;
      out[((size_t)(R + r) * TAB_SLOTS + slot) * 2 + 0] = (float)Vre;
      out[((size_t)(R + r) * TAB_SLOTS + slot) * 2 + 1] = (float)Vim;
    }
  return out;
}

// (the tables depend on R only: 5 ms of filter-tap and spectrum arithmetic per plan -> once per process)
inline const std::vector<float>& tables(int R) {
  static std::mutex lock;
  static std::map<int, std::vector<float>> cache;
  std::lock_guard<std::mutex> l(lock);
  auto it = cache.find(R);
  if (it == cache.end()) it = cache.emplace(R, make_tables(R)).first;
  return it->second;
}

// exp(-2 pi i j / 256), j < 256
inline std::vector<float> tw256() {
  std::vector<float> tw(512);
  for (int j = 0; j < 256; j++) {
    const double a = -6.283185307179586476925286766559 * (double)j / 256.;
    tw[(size_t)2 * j] = (float)std::cos(a);
    tw[(size_t)2 * j + 1] = (float)std::sin(a);
  }
  return tw;
}

}  // namespace osfft
}  // namespace waa
