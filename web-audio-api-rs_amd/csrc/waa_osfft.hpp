// waa_osfft.hpp — the WaveShaperNode's 2x / 4x oversampling as in-register / in-LDS transforms (waa_osfft.hip), the form
// rubato's FftFixedInOut stages have in the reference (waveshaper.rs:290-347, 409-481), instead of dense matrix products.
//
// One rubato stage (fi -> fo): rfft(2 fi) of the zero-padded block, bins [0, new_len) times the filter
// spectrum, irfft(2 fo), first half + kept overlap = output, second half = new overlap.  Up (128 -> 128 R), curve, down
// (128 R -> 128).  Written out in polyphase form, with n = R m + r for the samples at the high rate, EVERY transform of
// both stages is a 256-point complex one and both channels of a stereo quantum ride in one of them as z = L + i R (all
// steps between the transforms are real-linear or act on re / im separately):
//   Z      = DFT256(x, zero-padded)                                     x = the quantum, 128 frames
//   y_r    = IDFT256(Z .* U_r)            r = 0 .. R-1                  U_r[k'] = F_up[k'] s_r(k'),  s_r = e^{2 pi i k r / 256 R}
//                                                                       with k = k' (k' < 128), k' - 256 (k' > 128),
//                                                                       s_r(128) = 2 cos(pi r / R)   (bins +128 and -128)
//   u[R m + r] = y_r[m] + carry_r[m],  carry_r[m] <- y_r[m + 128]       m < 128: the up-sampled quantum, then the curve
//   D_r    = DFT256(curve(u[R m + r]), zero-padded)
//   Z3     = sum_r V_r .* D_r                                           V_r[k'] = F_dn[k] e^{-2 pi i r k / 256 R},  V_r[128] = 0
//   o      = IDFT256(Z3);  out[n] = o[n] + ocarry[n],  ocarry[n] <- o[n + 128]
// (tools/osfft_proto.py checks this against the stage-by-stage definition to 1e-15 in f64.)  2 + 2 R transforms per
// stereo quantum.  The tables U_r, V_r are computed on the host in f64 from the crate's f32 filter taps.
//
// A 256-point transform runs on SIXTEEN lanes that hold 16 complex values each, as 16 x 16 with one exchange through LDS:
//   n = 16 n1 + n2,  k = k1 + 16 k2:   X[k1 + 16 k2] = sum_n2 W_16^(n2 k2) [ W_256^(n2 k1) sum_n1 x[16 n1 + n2] W_16^(n1 k1) ]
//   pass 1  lane n2:  dft16 over n1, times W_256^(n2 k1)      -> E[k1][n2]
//   pass 2  lane k1:  dft16 over n2                           -> X[k1 + 16 k2] in slot s, k2 = K16(s)
// so that input AND output are distributed as "lane t holds the indices congruent t mod 16" — the overlap of a block
// (index m + 128 = m + 8 * 16) stays in the lane that needs it, and consecutive transforms chain without re-distribution.
// A wavefront carries four such groups; a group walks a run of render quanta of one instance with the carries in registers.
//
// This header is also compiled for the HOST (tools/osfft_emulate.cpp, tests/test_osfft_emulation.py): the same operations in
// the same order, the choreography of one group replayed lane by lane, checked against the f64 definition without a GPU.
#pragma once
#include "waa_fft3.hpp"

namespace waa {
namespace osfft {

using fft3::c2v;
using fft3::cldsp;
using fft3::ldsp;
using fft3::K16;

constexpr int ROW = 18;              // pitch (8-byte slots) of a 16-slot row: 144 B lane stride, conflict-free 16-byte accesses
constexpr int XSLOTS = 16 * ROW + 16;  // one group's exchange buffer E[k1][n2] (288 slots) + 128 B: the buffers of the four groups of a
                                      // wavefront start in alternating halves of the banks (2304 B apart they collide pairwise)
constexpr int TAB_SLOTS = 16 * ROW;  // one table (U_r or V_r) in lane-major rows: T[t][s] = table[t + 16 K16(s)]

// The lane's pass-1 twiddles W_256^(t k1), k1 = 1..15, kept as six values instead of fifteen (the kernel is short of
// registers: 18 fewer): W^(t k1) = W^(t (k1 & 3)) * W^(t (k1 & 12)), lo[k] = W_256^(t k), hi[k] = W_256^(4 t k), k = 1..3; the
// nine composite ones cost one more complex multiply per element (~1 ulp on those twiddles, like fwd_pass1_compute_half of
// waa_fft3.hpp).  tw256[j] = exp(-2 pi i j / 256).
struct Tw {
  c2v lo[4], hi[4];  // [0] unused
};
F3_FN void load_tw(const c2v* tw256, int t, Tw& w) {
#pragma unroll
  for (int k = 1; k < 4; k++) {
    w.lo[k] = tw256[(t * k) & 255];
    w.hi[k] = tw256[(t * 4 * k) & 255];
  }
  w.lo[0] = w.hi[0] = c2v{1.f, 0.f};
}
// acc + a * w as two packed FMAs: (acc.re + a.re w.re, acc.im + a.re w.im), then (.. - a.im w.im, .. + a.im w.re)
F3_FN c2v cmac(c2v acc, c2v a, c2v w) {
#if F3_DEV
  c2v r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(r));
  return r;
#else
  const c2v r = c2v{__builtin_fmaf(a.x, w.x, acc.x), __builtin_fmaf(a.x, w.y, acc.y)};
  return c2v{__builtin_fmaf(a.y, -w.y, r.x), __builtin_fmaf(a.y, w.x, r.y)};
#endif
}

// (cmul4: four independent complex multiplies per asm block, fft3::cmul4 in waa_fft3.hpp)
using fft3::cmul4;
// four accumulating ones (cmac), the two fused steps of an element four instructions apart
F3_FN void cmac4(c2v& z0, c2v& z1, c2v& z2, c2v& z3, c2v a0, c2v a1, c2v a2, c2v a3, c2v w0, c2v w1, c2v w2, c2v w3) {
#if F3_DEV
  asm("v_pk_fma_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+&v"(z0), "+&v"(z1), "+&v"(z2), "+&v"(z3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
#else
  z0 = cmac(z0, a0, w0);
  z1 = cmac(z1, a1, w1);
  z2 = cmac(z2, a2, w2);
  z3 = cmac(z3, a3, w3);
#endif
}

// x[s] *= W_256^(t K16(s)), s = 1..15 (INV: conjugated): the nine composite twiddles first, then the products, four at a time
template <bool INV>
F3_FN void apply_tw(c2v (&x)[16], const Tw& w) {
  // tw[k1], k1 = 1..15: lo[a] (b == 0), hi[b] (a == 0), lo[a] * hi[b]
  c2v tw[16];
  tw[1] = w.lo[1]; tw[2] = w.lo[2]; tw[3] = w.lo[3];
  tw[4] = w.hi[1]; tw[8] = w.hi[2]; tw[12] = w.hi[3];
  cmul4<false>(tw[5], tw[6], tw[7], tw[9], w.lo[1], w.lo[2], w.lo[3], w.lo[1], w.hi[1], w.hi[1], w.hi[1], w.hi[2]);
  cmul4<false>(tw[10], tw[11], tw[13], tw[14], w.lo[2], w.lo[3], w.lo[1], w.lo[2], w.hi[2], w.hi[2], w.hi[3], w.hi[3]);
  tw[15] = fft3::cmul(w.lo[3], w.hi[3]);
  // slot s holds k1 = K16(s)
  cmul4<INV>(x[1], x[2], x[3], x[4], x[1], x[2], x[3], x[4], tw[K16(1)], tw[K16(2)], tw[K16(3)], tw[K16(4)]);
  cmul4<INV>(x[5], x[6], x[7], x[8], x[5], x[6], x[7], x[8], tw[K16(5)], tw[K16(6)], tw[K16(7)], tw[K16(8)]);
  cmul4<INV>(x[9], x[10], x[11], x[12], x[9], x[10], x[11], x[12], tw[K16(9)], tw[K16(10)], tw[K16(11)], tw[K16(12)]);
  cmul4<INV>(x[13], x[14], x[15], tw[0], x[13], x[14], x[15], x[15], tw[K16(13)], tw[K16(14)], tw[K16(15)], tw[K16(15)]);
}

// pass 1 of a 256-point transform: x[n1] natural -> slot s holds (sum_n1 x[n1] W_16^(n1 K16(s))) W_256^(t K16(s))
template <bool INV>
F3_FN void pass1(c2v (&x)[16], const Tw& tws) {
  fft3::dft16<INV>(x);
  apply_tw<INV>(x, tws);
}
// ... with the inputs 8..15 zero (a zero-padded block): the first radix-4 stage degenerates
template <bool INV>
F3_FN void pass1_half(c2v (&x)[16], const Tw& tws) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const c2v a0 = x[j], a1 = x[j + 4];
    x[j] = a0 + a1;
    x[j + 8] = a0 - a1;
    x[j + 4] = INV ? fft3::add_pib(a0, a1) : fft3::add_nib(a0, a1);
    x[j + 12] = INV ? fft3::add_nib(a0, a1) : fft3::add_pib(a0, a1);
  }
#define OSF_TW2(J)                                        \
  x[J + 4] = fft3::mulw32<2 * J * 1, INV>(x[J + 4]);      \
  x[J + 8] = fft3::mulw32<2 * J * 2, INV>(x[J + 8]);      \
  x[J + 12] = fft3::mulw32<2 * J * 3, INV>(x[J + 12]);
  OSF_TW2(1) OSF_TW2(2) OSF_TW2(3)
#undef OSF_TW2
#pragma unroll
  for (int q = 0; q < 16; q += 4) fft3::bfly4<INV>(x[q], x[q + 1], x[q + 2], x[q + 3]);
  apply_tw<INV>(x, tws);
}
// exchange: lane t (= n2) leaves slot s in row K16(s), column t ...
F3_FN void xwrite(const c2v (&x)[16], ldsp ex, int t) {
#pragma unroll
  for (int s = 0; s < 16; s++) ex[K16(s) * ROW + t] = x[s];
}
// ... and lane t (= k1) picks up its row: x[n2], natural
F3_FN void xread(c2v (&x)[16], cldsp ex, int t) {
  const F3_LDS fft3::f4v_* row = (const F3_LDS fft3::f4v_*)(ex + t * ROW);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const fft3::f4v_ v = row[j];
    x[2 * j] = c2v{v.x, v.y};
    x[2 * j + 1] = c2v{v.z, v.w};
  }
}
// pass 2: slot s <- element t + 16 K16(s) of the transform
template <bool INV>
F3_FN void pass2(c2v (&x)[16]) {
  fft3::dft16<INV>(x);
}
// half h (slots 8 h .. 8 h + 7) of a lane's row of a table (lane-major, slot order): tab[t * ROW + s]
F3_FN void tab_read8(c2v (&w)[8], cldsp tab, int t, int h) {
  const F3_LDS fft3::f4v_* row = (const F3_LDS fft3::f4v_*)(tab + t * ROW + 8 * h);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const fft3::f4v_ v = row[j];
    w[2 * j] = c2v{v.x, v.y};
    w[2 * j + 1] = c2v{v.z, v.w};
  }
}

// waveshaper.rs:555-573, branch-free: the interpolation is always evaluated (on indices clamped into the table, which changes
// nothing for 0 < v < n - 1) and the two saturated cases are selected afterwards — the same values as the reference's three
// branches (and as shape_curve_lds, waa_frozen.hip) for every input, NaN included (index 0, NaN result), without 64 divergent
// branches per quantum.  c_first / c_last = curve[0] / curve[n - 1].
template <typename P>
F3_FN float shape(P curve, int nn, float c_first, float c_last, float input) {
  const float n = (float)nn;
  const float v = (n - 1.f) / 2.0f * (input + 1.f);
  const float k = __builtin_floorf(v);
  const float f = v - k;
  int ki = (int)k;
  const int hi = nn >= 2 ? nn - 2 : 0;
  ki = ki < 0 ? 0 : (ki > hi ? hi : ki);
  const float r = (1.f - f) * curve[ki] + f * curve[ki + (nn >= 2 ? 1 : 0)];
  return nn == 0 ? 0.f : (v <= 0.f ? c_first : (v >= n - 1.f ? c_last : r));
}

// both channels of one frame: the arithmetic as packed operations (IEEE, element-wise: the same values as shape() per channel)
// PADDED: the table holds one more element (a copy of the last one), so curve[k + 1] is always readable and the two reads
// of a lookup are one ds_read2_b32
template <bool PADDED, typename P>
F3_FN c2v shape2(P curve, int nn, float c_first, float c_last, c2v in) {
  const float n = (float)nn;
  const float hn = (n - 1.f) / 2.0f;
  const c2v v = c2v{hn, hn} * (in + c2v{1.f, 1.f});
  const c2v k = c2v{__builtin_floorf(v.x), __builtin_floorf(v.y)};
  const c2v f = v - k;
  const int hi = nn >= 2 ? nn - 2 : 0, st = PADDED ? 1 : (nn >= 2 ? 1 : 0);
  int k0 = (int)k.x, k1 = (int)k.y;
  k0 = k0 < 0 ? 0 : (k0 > hi ? hi : k0);
  k1 = k1 < 0 ? 0 : (k1 > hi ? hi : k1);
  const c2v ca = c2v{curve[k0], curve[k1]}, cb = c2v{curve[k0 + st], curve[k1 + st]};
  const c2v r = (c2v{1.f, 1.f} - f) * ca + f * cb;
  // (sequential selects, innermost first: nested conditionals with the uniform nn == 0 test became 32 branches per quantum;
  // an empty curve never gets here — a WaveShaperNode without a curve is an identity the planner aliases)
  c2v o;
  o.x = v.x >= n - 1.f ? c_last : r.x;
  o.y = v.y >= n - 1.f ? c_last : r.y;
  o.x = v.x <= 0.f ? c_first : o.x;
  o.y = v.y <= 0.f ? c_first : o.y;
  return o;
}

// ---- one lane's part of a quantum, phase by phase (an exchange through LDS sits between two phases: xwrite, wave
// barrier, xread — the caller's; the emulator runs every phase for the 16 lanes of a group in turn) ------------------------
template <int R>
struct Lane {
  c2v a[16];        // the transform in flight
  c2v Z[16];        // spectrum of the input quantum, slot order
  c2v Z3[16];       // spectrum of the output quantum being accumulated over r, slot order
  c2v ycar[R][8];   // overlap of the up-sampling stage, per polyphase branch: y_r[m + 128], m = t + 16 j
  c2v ocar[8];      // overlap of the down-sampling stage
  Tw tws;           // the lane's pass-1 twiddles
};
template <int R>
F3_FN void lane_reset(Lane<R>& L) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    L.ocar[j] = c2v{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; r++) L.ycar[r][j] = c2v{0.f, 0.f};
  }
}
// the same under a per-lane predicate, as selects: a divergent branch around the reset makes the compiler keep a second copy
// of the whole carried state across the kernel's loop (33 registers over the budget of two wavefronts per SIMD, measured)
template <int R>
F3_FN void lane_reset_if(Lane<R>& L, bool fresh) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    L.ocar[j].x = fresh ? 0.f : L.ocar[j].x;
    L.ocar[j].y = fresh ? 0.f : L.ocar[j].y;
#pragma unroll
    for (int r = 0; r < R; r++) {
      L.ycar[r][j].x = fresh ? 0.f : L.ycar[r][j].x;
      L.ycar[r][j].y = fresh ? 0.f : L.ycar[r][j].y;
    }
  }
}
// the quantum's frames 16 j + t, j < 8 (re = channel 0, im = channel 1 or 0) -> pass 1 of Z
template <int R>
F3_FN void ph_in(Lane<R>& L, const c2v (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) L.a[j] = x[j];
  pass1_half<false>(L.a, L.tws);
}
// (exchange) -> Z
template <int R>
F3_FN void ph_spec(Lane<R>& L) {
  pass2<false>(L.a);
#pragma unroll
  for (int s = 0; s < 16; s++) L.Z[s] = L.a[s];
}
// Z .* U_r (the lane's row of U_r, slot order, read in two halves) -> pass 1 of y_r
template <int R>
F3_FN void ph_up(Lane<R>& L, cldsp tab_r, int t) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    c2v w[8];
    tab_read8(w, tab_r, t, h);
#pragma unroll
    for (int s = 8 * h; s < 8 * h + 8; s += 4)  // natural order: bin t + 16 K16(s)
      cmul4<false>(L.a[K16(s)], L.a[K16(s + 1)], L.a[K16(s + 2)], L.a[K16(s + 3)], L.Z[s], L.Z[s + 1], L.Z[s + 2], L.Z[s + 3],
                   w[s - 8 * h], w[s + 1 - 8 * h], w[s + 2 - 8 * h], w[s + 3 - 8 * h]);
  }
  pass1<true>(L.a, L.tws);
}
// (exchange) -> y_r; the up-sampled frames R (t + 16 j) + r, j < 8, of the quantum; the overlap moves on when the node
// really processes this quantum (`proc`: a group whose quantum is skipped runs the same instructions, masked)
template <int R>
F3_FN void ph_up_out(Lane<R>& L, int r, bool proc, c2v (&u)[8]) {
  pass2<true>(L.a);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u[j] = L.a[K16(j)] + L.ycar[r][j];
    const c2v nxt = L.a[K16(j + 8)];
    L.ycar[r][j] = proc ? nxt : L.ycar[r][j];
  }
}
// the curve's output at those frames -> pass 1 of D_r
template <int R>
F3_FN void ph_dn(Lane<R>& L, const c2v (&d)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) L.a[j] = d[j];
  pass1_half<false>(L.a, L.tws);
}
// (exchange) -> D_r;  Z3 (+)= V_r .* D_r
template <int R>
F3_FN void ph_dn_acc(Lane<R>& L, int r, cldsp tab_r, int t) {
  pass2<false>(L.a);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    c2v w[8];
    tab_read8(w, tab_r, t, h);
#pragma unroll
    for (int s = 8 * h; s < 8 * h + 8; s += 4) {
      const int o = s - 8 * h;
      if (r == 0)
        cmul4<false>(L.Z3[s], L.Z3[s + 1], L.Z3[s + 2], L.Z3[s + 3], L.a[s], L.a[s + 1], L.a[s + 2], L.a[s + 3], w[o], w[o + 1], w[o + 2],
                     w[o + 3]);
      else
        cmac4(L.Z3[s], L.Z3[s + 1], L.Z3[s + 2], L.Z3[s + 3], L.a[s], L.a[s + 1], L.a[s + 2], L.a[s + 3], w[o], w[o + 1], w[o + 2], w[o + 3]);
    }
  }
}
// Z3 -> pass 1 of o
template <int R>
F3_FN void ph_out(Lane<R>& L) {
#pragma unroll
  for (int j = 0; j < 16; j++) L.a[j] = L.Z3[K16(j)];
  pass1<true>(L.a, L.tws);
}
// (exchange) -> o; the quantum's output frames t + 16 j, j < 8
template <int R>
F3_FN void ph_out_end(Lane<R>& L, bool proc, c2v (&o)[8]) {
  pass2<true>(L.a);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    o[j] = L.a[K16(j)] + L.ocar[j];
    const c2v nxt = L.a[K16(j + 8)];
    L.ocar[j] = proc ? nxt : L.ocar[j];
  }
}

}  // namespace osfft
}  // namespace waa
