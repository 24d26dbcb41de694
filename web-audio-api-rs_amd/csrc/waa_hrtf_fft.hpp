// waa_hrtf_fft.hpp — the HRTF panner's FIR as 256-point transforms (waa_hrtf_fft.hip), for a direction that does not change
// during the render (one HRIR pair for the whole batch: PannerNode and AudioListener at rest — the usual scene).
//
// Definition (panner.rs:781-829 + crate hrtf, restated): out_q[i] = gain * sum_j h[j] x[q * 128 + i - j] per ear,
// x continued into the previously PROCESSED quanta.  With h the same for every quantum this is one linear convolution, and the
// direct form spends taps * 128 * 2 fused multiply-adds per quantum on it (415 taps at 48 kHz: 106 k; hrtf8_kernel runs at 0.6
// of the packed-f32 peak and is still the slowest BASELINE-sized kernel of the library).  Uniform partitioned overlap-add on the
// machinery of waa_osfft.hpp (a 256-point complex transform on 16 lanes x 16 values, one LDS exchange):
//   h = h_0 | h_1 | h_2 | h_3          partitions of 128 taps (taps <= 512), both ears in one complex table:
//   H_p   = DFT256(hL_p + i hR_p, zero-padded) / 256                        (host, f64 -> f32, lane-major rows like U_r / V_r)
//   Z_q   = DFT256(x_q, zero-padded)                                         x_q real (the mono mix of the quantum)
//   Y_q   = Z_q H_0 + Z_{q-1} H_1 + Z_{q-2} H_2 + Z_{q-3} H_3                q - p: the p-th processed quantum in front of q
//   o     = IDFT256(Y_q);  out_q[n] = o[n] + carry[n],  carry[n] <- o[n + 128]     re = left ear, im = right ear
// (x real and H_p = DFT(hL) + i DFT(hR) give IDFT(Z H) = x * hL + i x * hR: both ears ride in one transform.)  Two transforms and
// four spectral products per quantum: ~28 k flops instead of 212 k.  The three previous spectra and the carry are the node's whole
// state; a group of 16 lanes walks a run of quanta with them in registers, and a run starts by rendering the FOUR processed quanta
// in front of it without storing them (three give the spectra, the fourth the carry).  Frozen history (panner.rs:697-711: a skipped
// quantum leaves the state alone; the tail) is the `prev` table of link_kernel, exactly as for the oversampled WaveShaper.
//
// Host-compilable like waa_osfft.hpp: tools/hrtf_fft_emulate.cpp replays the choreography against the f64 direct form
// (tests/test_hrtf_fft_emulation.py).
#pragma once
#include "waa_osfft.hpp"

namespace waa {
namespace hrtffft {

using namespace osfft;
constexpr int PARTS = 4;       // partitions of 128 taps: HRIRs up to 512 taps (44.1 kHz: 512, 48 kHz: 415)
constexpr int HEADS = PARTS;   // processed quanta rendered in front of a run: PARTS - 1 spectra + the carry

struct HLane {
  c2v a[16];             // the forward transform in flight / the quantum's spectrum, slot order
  c2v Y[16];             // the output spectrum, then the inverse transform in flight
  c2v Zh[PARTS - 1][16]; // spectra of the last three processed quanta, slot order (Zh[0] the newest)
  c2v ocar[8];           // carry: o[t + 16 j + 128]
  Tw tws;
};
F3_FN void lane_reset_if(HLane& L, bool fresh) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    L.ocar[j].x = fresh ? 0.f : L.ocar[j].x;
    L.ocar[j].y = fresh ? 0.f : L.ocar[j].y;
  }
#pragma unroll
  for (int p = 0; p < PARTS - 1; p++)
#pragma unroll
    for (int s = 0; s < 16; s++) {
      L.Zh[p][s].x = fresh ? 0.f : L.Zh[p][s].x;
      L.Zh[p][s].y = fresh ? 0.f : L.Zh[p][s].y;
    }
}
// the quantum's mono frames t + 16 j, j < 8 -> pass 1 of Z
F3_FN void ph_in(HLane& L, const float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) L.a[j] = c2v{x[j], 0.f};
  pass1_half<false>(L.a, L.tws);
}
// (exchange of a) -> Z;  Y = Z H_0 + Zh[0] H_1 + Zh[1] H_2 + Zh[2] H_3 in natural order;  the history moves on when the node
// processes this quantum;  pass 1 of o
F3_FN void ph_spec(HLane& L, cldsp tab, int t, bool proc) {
  pass2<false>(L.a);
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int p = 0; p < PARTS; p++) {
      c2v w[8];
      tab_read8(w, tab + p * TAB_SLOTS, t, h);
#pragma unroll
      for (int s = 8 * h; s < 8 * h + 8; s += 4) {
        const int o = s - 8 * h;
        if (p == 0)
          cmul4<false>(L.Y[K16(s)], L.Y[K16(s + 1)], L.Y[K16(s + 2)], L.Y[K16(s + 3)], L.a[s], L.a[s + 1], L.a[s + 2], L.a[s + 3], w[o], w[o + 1],
                       w[o + 2], w[o + 3]);
        else
          cmac4(L.Y[K16(s)], L.Y[K16(s + 1)], L.Y[K16(s + 2)], L.Y[K16(s + 3)], L.Zh[p - 1][s], L.Zh[p - 1][s + 1], L.Zh[p - 1][s + 2],
                L.Zh[p - 1][s + 3], w[o], w[o + 1], w[o + 2], w[o + 3]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 16; s++) {
#pragma unroll
    for (int p = PARTS - 2; p > 0; p--) {
      L.Zh[p][s].x = proc ? L.Zh[p - 1][s].x : L.Zh[p][s].x;
      L.Zh[p][s].y = proc ? L.Zh[p - 1][s].y : L.Zh[p][s].y;
    }
    L.Zh[0][s].x = proc ? L.a[s].x : L.Zh[0][s].x;
    L.Zh[0][s].y = proc ? L.a[s].y : L.Zh[0][s].y;
  }
  pass1<true>(L.Y, L.tws);
}
// (exchange of Y) -> o; the quantum's output frames t + 16 j, j < 8: re = left, im = right
F3_FN void ph_out(HLane& L, bool proc, c2v (&o)[8]) {
  pass2<true>(L.Y);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    o[j] = L.Y[K16(j)] + L.ocar[j];
    const c2v nxt = L.Y[K16(j + 8)];
    L.ocar[j].x = proc ? nxt.x : L.ocar[j].x;
    L.ocar[j].y = proc ? nxt.y : L.ocar[j].y;
  }
}

}  // namespace hrtffft
}  // namespace waa
