// waa_fft3.hpp — the 16384-point complex FFT of the ConvolverNode path as THREE register passes (32 x 32 x 16) with two
// exchanges through LDS, for 512 threads that hold 32 complex values each (waa_conv3.hip).
//
// The round-1/2 kernel (conv_fft_pipe_kernel, waa_conv.hip) ran the same transform as radix-4 stages IN LDS: five write +
// five read sweeps over the 128 KB buffer per transform and ~2100 vector instructions per thread of which 40 % were LDS
// address arithmetic and moves; measured, the transform (not HBM) bounded both FFT kernels.  Here the data
// lives in registers; LDS is only the transposition between passes (2 writes + 2 reads of the buffer), every LDS address is
// one base register plus an immediate, and the complex arithmetic is written for the packed-f32 instructions with their
// operand selects (a complex multiply is 2 instructions, a +-i rotation inside a butterfly is free).
//
// Index algebra (forward, W_n = exp(-2 pi i / n); the inverse is the exact mirror with conjugated twiddles):
//   n = n1 * 512 + m,  m = m1 * 16 + m2,      k = k1 + 32 * k2 + 1024 * k3      (n1, m1, k1, k2 < 32; m2, k3 < 16)
//   W_N^(n k) = W_32^(n1 k1) * W_N^(m k1) * W_32^(m1 k2) * W_512^(m2 k2) * W_16^(m2 k3)
//   pass 1  thread m:        y_k1[m]     = (sum_n1 z[n1 * 512 + m] W_32^(n1 k1)) * W_N^(m k1)            -> E1[k1][m]
//   pass 2  thread (k1, m2): u_k1k2[m2]  = sum_m1 y_k1[m1 * 16 + m2] W_32^(m1 k2)                        -> E2[k1 * 32 + k2][m2]
//   pass 3  thread row r = k1 * 32 + k2: X[k1 + 32 k2 + 1024 k3] = sum_m2 (u_r[m2] W_512^(m2 k2)) W_16^(m2 k3)  -> position k3 * 1024 + r
// (the W_512 twiddles ride on pass 3's inputs, where they depend on k2 = t mod 32 only — the same 15 values for both rows
// of a thread, kept in registers like pass 1's 31; on pass 2's outputs they would differ per lane AND per output.)
// The spectrum is stored in THAT position order (the element-wise product with the impulse response's spectra does not
// care, as long as H, X and Y share it): for a fixed k3 consecutive threads hold consecutive positions (coalesced 8-byte
// accesses), and the time-domain side of pass 1 is coalesced over m.
//
// This header is also compiled for the HOST (tests/test_fft3_emulation.py, clang++): the primitives then fall back to plain
// C++ with the same operations in the same order, and the test replays the kernel's thread / LDS choreography against a
// float64 DFT — the index maps are checked without a GPU.
#pragma once
#include <cstdint>

#if defined(__HIP_DEVICE_COMPILE__)
#define F3_DEV 1
#define F3_FN __device__ __forceinline__
#else
#define F3_DEV 0
#if defined(__HIPCC__)
#define F3_FN __device__ __forceinline__
#else
#define F3_FN inline
#endif
#endif

namespace waa {
namespace fft3 {

typedef float c2v __attribute__((ext_vector_type(2)));  // (re, im) in one 64-bit register pair
// pointers into the workgroup's LDS carry their address space explicitly on the device (a volatile access through a
// generic pointer would become a flat load)
#if F3_DEV
#define F3_LDS __attribute__((address_space(3)))
#else
#define F3_LDS
#endif
typedef F3_LDS c2v* ldsp;
typedef const F3_LDS c2v* cldsp;

constexpr int N = 16384, NT = 512, B = N / 2;
// LDS map, in 8-byte slots.  E1[k1][m]: rows of 512 + 16 slots — the 128-byte skew puts the two half-waves of a pass-2 read
// (k1 and k1 + 1) on disjoint bank halves.  E2[r][m2]: rows of 16 + 2 slots (a thread reads / writes its row with eight
// conflict-free 16-byte accesses, lane stride 144 B), 32 rows per k1 plus 16 slots of skew (same reason).  E1 and E2 share
// the space.
constexpr int E1_ROW = 528, E2_ROW = 18, E2_K1 = 32 * E2_ROW + 16;
constexpr int E_SLOTS = 32 * E2_K1;               // 18944 slots >= 32 * E1_ROW = 16896
constexpr int LDS_SLOTS = E_SLOTS;
constexpr int LDS_BYTES = LDS_SLOTS * 8;          // 151552 <= 163840
constexpr int e1(int k1, int m) { return k1 * E1_ROW + m; }
constexpr int e2(int r, int m2) { return (r >> 5) * E2_K1 + (r & 31) * E2_ROW + m2; }

// slot <-> index maps of the in-register transforms (decimation in frequency: natural order in, permuted out)
constexpr int K32(int s) { return (s >> 3) + 4 * ((s >> 1) & 3) + 16 * (s & 1); }  // slot s of dft32 holds output K32(s)
constexpr int S32(int k) { return 8 * (k & 3) + 2 * ((k >> 2) & 3) + (k >> 4); }   // inverse map
constexpr int K16(int s) { return (s >> 2) + 4 * (s & 3); }                        // an involution

// ---- complex primitives -----------------------------------------------------------------------------------------------
// a + (-i) b = (a.re + b.im, a.im - b.re)
F3_FN c2v add_nib(c2v a, c2v b) {
#if F3_DEV
  c2v r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return c2v{a.x + b.y, a.y - b.x};
#endif
}
// a + (+i) b = (a.re - b.im, a.im + b.re)
F3_FN c2v add_pib(c2v a, c2v b) {
#if F3_DEV
  c2v r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return c2v{a.x - b.y, a.y + b.x};
#endif
}
// a * w:  re = fma(a.re, w.re, -(a.im * w.im)),  im = fma(a.re, w.im, a.im * w.re)  — the operations of waa_conv.hip's cmul
F3_FN c2v cmul(c2v a, c2v w) {
#if F3_DEV
  c2v t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
#else
  return c2v{__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x)};
#endif
}
// a * conj(w):  re = fma(a.re, w.re, a.im * w.im),  im = fma(a.re, -w.im, a.im * w.re)
F3_FN c2v cmulc(c2v a, c2v w) {
#if F3_DEV
  c2v t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
#else
  return c2v{__builtin_fmaf(a.x, w.x, a.y * w.y), __builtin_fmaf(a.x, -w.y, a.y * w.x)};
#endif
}
// Four independent complex multiplies in one block, the four products first and the four fused steps after them: a packed-f32
// result cannot be forwarded to the very next instruction (the compiler puts an s_nop between a dependent pair — 250 of them
// per quantum with one multiply at a time), so dependent halves are kept four instructions apart.  Same operations as
// cmul / cmulc, element by element.
template <bool CONJ>
F3_FN void cmul4(c2v& r0, c2v& r1, c2v& r2, c2v& r3, c2v a0, c2v a1, c2v a2, c2v a3, c2v w0, c2v w1, c2v w2, c2v w3) {
#if F3_DEV
  c2v t0, t1, t2, t3;
  if (!CONJ)
    asm("v_pk_mul_f32 %0, %4, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %5, %9 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %6, %10 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %7, %11 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  else
    asm("v_pk_mul_f32 %0, %4, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %5, %9 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %6, %10 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %7, %11 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  r0 = t0;
  r1 = t1;
  r2 = t2;
  r3 = t3;
#else
  const c2v t0 = CONJ ? cmulc(a0, w0) : cmul(a0, w0), t1 = CONJ ? cmulc(a1, w1) : cmul(a1, w1);
  const c2v t2 = CONJ ? cmulc(a2, w2) : cmul(a2, w2), t3 = CONJ ? cmulc(a3, w3) : cmul(a3, w3);
  r0 = t0;
  r1 = t1;
  r2 = t2;
  r3 = t3;
#endif
}
// the same with a compile-time twiddle: a scalar register pair, no vector registers and no moves
template <bool CONJ>
F3_FN c2v cmul_k(c2v a, const c2v w) {
#if F3_DEV
  c2v t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  if (!CONJ)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
#else
  return CONJ ? cmulc(a, w) : cmul(a, w);
#endif
}
// a * (-i) = (a.im, -a.re);  a * (+i) = (-a.im, a.re)
F3_FN c2v rot_ni(c2v a) {
#if F3_DEV
  c2v r;
  const c2v k = {1.f, -1.f};
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(k));
  return r;
#else
  return c2v{a.y, -a.x};
#endif
}
F3_FN c2v rot_pi(c2v a) {
#if F3_DEV
  c2v r;
  const c2v k = {-1.f, 1.f};
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(k));
  return r;
#else
  return c2v{-a.y, a.x};
#endif
}

// exp(-2 pi i e / 32), (float) of the f64 value; the quarter turns and the odd eighth turns are special-cased below
constexpr float W32_RE[32] = {0x1.000000p+0f,  0x1.f6297cp-1f,  0x1.d906bcp-1f,  0x1.a9b662p-1f,  0x1.6a09e6p-1f,  0x1.1c73b4p-1f,
                              0x1.87de2ap-2f,  0x1.8f8b84p-3f,  0.f,             -0x1.8f8b84p-3f, -0x1.87de2ap-2f, -0x1.1c73b4p-1f,
                              -0x1.6a09e6p-1f, -0x1.a9b662p-1f, -0x1.d906bcp-1f, -0x1.f6297cp-1f, -0x1.000000p+0f, -0x1.f6297cp-1f,
                              -0x1.d906bcp-1f, -0x1.a9b662p-1f, -0x1.6a09e6p-1f, -0x1.1c73b4p-1f, -0x1.87de2ap-2f, -0x1.8f8b84p-3f,
                              0.f,             0x1.8f8b84p-3f,  0x1.87de2ap-2f,  0x1.1c73b4p-1f,  0x1.6a09e6p-1f,  0x1.a9b662p-1f,
                              0x1.d906bcp-1f,  0x1.f6297cp-1f};
constexpr float W32_IM[32] = {0.f,             -0x1.8f8b84p-3f, -0x1.87de2ap-2f, -0x1.1c73b4p-1f, -0x1.6a09e6p-1f, -0x1.a9b662p-1f,
                              -0x1.d906bcp-1f, -0x1.f6297cp-1f, -0x1.000000p+0f, -0x1.f6297cp-1f, -0x1.d906bcp-1f, -0x1.a9b662p-1f,
                              -0x1.6a09e6p-1f, -0x1.1c73b4p-1f, -0x1.87de2ap-2f, -0x1.8f8b84p-3f, 0.f,             0x1.8f8b84p-3f,
                              0x1.87de2ap-2f,  0x1.1c73b4p-1f,  0x1.6a09e6p-1f,  0x1.a9b662p-1f,  0x1.d906bcp-1f,  0x1.f6297cp-1f,
                              0x1.000000p+0f,  0x1.f6297cp-1f,  0x1.d906bcp-1f,  0x1.a9b662p-1f,  0x1.6a09e6p-1f,  0x1.1c73b4p-1f,
                              0x1.87de2ap-2f,  0x1.8f8b84p-3f};
constexpr float C4 = 0x1.6a09e6p-1f;  // sqrt(1/2)

// a * W_32^E (INV: a * conj(W_32^E)), E a compile-time constant
template <int E, bool INV>
F3_FN c2v mulw32(c2v a) {
  constexpr int e = ((INV ? -E : E) % 32 + 32) % 32;
  if constexpr (e == 0) return a;
  else if constexpr (e == 8) return rot_ni(a);
  else if constexpr (e == 16) return -a;
  else if constexpr (e == 24) return rot_pi(a);
  else if constexpr (e == 4) return add_nib(a, a) * c2v{C4, C4};     // (1 - i) / sqrt 2
  else if constexpr (e == 12) return add_pib(a, a) * c2v{-C4, -C4};  // (-1 - i) / sqrt 2
  else if constexpr (e == 20) return add_nib(a, a) * c2v{-C4, -C4};  // (-1 + i) / sqrt 2
  else if constexpr (e == 28) return add_pib(a, a) * c2v{C4, C4};    // (1 + i) / sqrt 2
  else return cmul_k<false>(a, c2v{W32_RE[e], W32_IM[e]});
}

// radix-4 butterfly, outputs in natural order: a_q <- sum_i a_i W_4^(i q)   (INV: conjugated)
template <bool INV>
F3_FN void bfly4(c2v& a0, c2v& a1, c2v& a2, c2v& a3) {
  const c2v s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  a0 = s02 + s13;
  a2 = s02 - s13;
  a1 = INV ? add_pib(d02, d13) : add_nib(d02, d13);
  a3 = INV ? add_nib(d02, d13) : add_pib(d02, d13);
}

// 32-point DFT in registers, natural order in, slot s <- output K32(s).  ODD_ONLY: only the odd slots (outputs 16..31) are
// needed (overlap-save keeps the second half of the inverse transform): the sums of the last radix-2 stage are skipped.
template <bool INV, bool ODD_ONLY = false>
F3_FN void dft32(c2v (&x)[32]) {
#pragma unroll
  for (int j = 0; j < 8; j++) bfly4<INV>(x[j], x[j + 8], x[j + 16], x[j + 24]);
  // twiddles W_32^(j q) on slot j + 8 q
#define F3_TW1(J)                                   \
  x[J + 8] = mulw32<J * 1, INV>(x[J + 8]);          \
  x[J + 16] = mulw32<J * 2, INV>(x[J + 16]);        \
  x[J + 24] = mulw32<J * 3, INV>(x[J + 24]);
  F3_TW1(1) F3_TW1(2) F3_TW1(3) F3_TW1(4) F3_TW1(5) F3_TW1(6) F3_TW1(7)
#undef F3_TW1
#pragma unroll
  for (int b = 0; b < 32; b += 8) {
    bfly4<INV>(x[b], x[b + 2], x[b + 4], x[b + 6]);
    bfly4<INV>(x[b + 1], x[b + 3], x[b + 5], x[b + 7]);
    x[b + 3] = mulw32<4, INV>(x[b + 3]);   // W_8^(1 q2) on slot 1 + 2 q2
    x[b + 5] = mulw32<8, INV>(x[b + 5]);
    x[b + 7] = mulw32<12, INV>(x[b + 7]);
  }
#pragma unroll
  for (int i = 0; i < 32; i += 2) {
    const c2v u = x[i], v = x[i + 1];
    if (!ODD_ONLY) x[i] = u + v;
    x[i + 1] = u - v;
  }
}
// 16-point DFT in registers, natural order in, slot s <- output K16(s)
template <bool INV>
F3_FN void dft16(c2v (&x)[16]) {
#pragma unroll
  for (int j = 0; j < 4; j++) bfly4<INV>(x[j], x[j + 4], x[j + 8], x[j + 12]);
#define F3_TW2(J)                                   \
  x[J + 4] = mulw32<2 * J * 1, INV>(x[J + 4]);      \
  x[J + 8] = mulw32<2 * J * 2, INV>(x[J + 8]);      \
  x[J + 12] = mulw32<2 * J * 3, INV>(x[J + 12]);
  F3_TW2(1) F3_TW2(2) F3_TW2(3)
#undef F3_TW2
#pragma unroll
  for (int q = 0; q < 16; q += 4) bfly4<INV>(x[q], x[q + 1], x[q + 2], x[q + 3]);
}

// One 8-byte LDS read that stays ONE ds_read_b64: the compiler pairs neighbouring reads into ds_read2_b64, which the LDS
// serves at half the rate of two single reads (MI355X_MICROARCH.md, LDS table: 8 cycles against 2 + 2).
F3_FN c2v lds_rd(cldsp p) {
#if F3_DEV
  return *(const volatile F3_LDS c2v*)p;
#else
  return *p;
#endif
}

// ---- the passes (one thread's part; barriers are the caller's) ---------------------------------------------------------
// Twiddle sets a thread keeps for the whole kernel: W_N^(t * k1), k1 = 1..31 — by output slot for the forward pass 1
// (tws[s] = W_N^(t * K32(s))), by k1 for the inverse pass 3.  `tw` = exp(-2 pi i j / N), j < N.
F3_FN void load_tw1_slots(const c2v* tw, int t, c2v (&tws)[32]) {
#pragma unroll
  for (int s = 1; s < 32; s++) tws[s] = tw[(t * K32(s)) & (N - 1)];
  tws[0] = c2v{1.f, 0.f};
}
F3_FN void load_tw1_natural(const c2v* tw, int t, c2v (&twn)[32]) {
#pragma unroll
  for (int k = 1; k < 32; k++) twn[k] = tw[(t * k) & (N - 1)];
  twn[0] = c2v{1.f, 0.f};
}
// W_512^(m2 k2), m2 = 1..15, for the rows r = t and t + 512 of pass 3 (k2 = r mod 32 = t mod 32 for both)
F3_FN void load_tw3(const c2v* tw, int t, c2v (&tw3)[16]) {
#pragma unroll
  for (int m2 = 1; m2 < 16; m2++) tw3[m2] = tw[(m2 * (t & 31) * 32) & (N - 1)];
  tw3[0] = c2v{1.f, 0.f};
}

// forward pass 1: x[n1] = z[n1 * 512 + t] (the caller's barrier sits between the two halves: E1 overwrites the rows the
// previous block's pass 3 reads)
F3_FN void fwd_pass1_compute(c2v (&x)[32], const c2v (&tws)[32]) {
  dft32<false>(x);
  // (four at a time: no hazard nop between a product and its fused step; slot 0's twiddle is 1)
  x[1] = cmul(x[1], tws[1]);
  x[2] = cmul(x[2], tws[2]);
  x[3] = cmul(x[3], tws[3]);
#pragma unroll
  for (int s = 4; s < 32; s += 4) cmul4<false>(x[s], x[s + 1], x[s + 2], x[s + 3], x[s], x[s + 1], x[s + 2], x[s + 3], tws[s], tws[s + 1], tws[s + 2], tws[s + 3]);
}
// The same with HALF the twiddle registers (the filter-stage kernel is short of them): W_N^(t k1) for k1 < 16 and
// W_N^(16 t); the upper half is their product (one more complex multiply per element, ~1 ulp on those twiddles).
F3_FN void load_tw1_half(const c2v* tw, int t, c2v (&twh)[16], c2v& w16) {
#pragma unroll
  for (int k = 1; k < 16; k++) twh[k] = tw[(t * k) & (N - 1)];
  twh[0] = c2v{1.f, 0.f};
  w16 = tw[(t * 16) & (N - 1)];
}
F3_FN void fwd_pass1_compute_half(c2v (&x)[32], const c2v (&twh)[16], const c2v w16) {
  dft32<false>(x);
#pragma unroll
  for (int s = 1; s < 32; s++) {
    const int k1 = K32(s);
    if (k1 < 16)
      x[s] = cmul(x[s], twh[k1]);
    else if (k1 == 16)
      x[s] = cmul(x[s], w16);
    else
      x[s] = cmul(x[s], cmul(twh[k1 - 16], w16));
  }
}
F3_FN void fwd_pass1_write(const c2v (&x)[32], ldsp lds, int t) {
#pragma unroll
  for (int s = 0; s < 32; s++) lds[e1(K32(s), 0) + t] = x[s];
}
// forward pass 2, thread (k1, m2) = (t >> 4, t & 15): read, transform — the caller puts a barrier between this
// and fwd_pass2_write (E2 overwrites E1)
F3_FN void fwd_pass2_compute(c2v (&x)[32], cldsp lds, int t) {
  const int base = e1(t >> 4, t & 15);
#pragma unroll
  for (int m1 = 0; m1 < 32; m1++) x[m1] = lds_rd(lds + base + m1 * 16);
  dft32<false>(x);
}
F3_FN void fwd_pass2_write(const c2v (&x)[32], ldsp lds, int t) {
  const int base = e2((t >> 4) * 32, t & 15);
#pragma unroll
  for (int s = 0; s < 32; s++) lds[base + K32(s) * E2_ROW] = x[s];
}
// forward pass 3 for row r: y[s] <- X[(r & 31) ... ] with k3 = K16(s); the caller stores y[s] at position K16(s) * 1024 + r
typedef float f4v_ __attribute__((ext_vector_type(4)));
F3_FN void fwd_pass3(c2v (&y)[16], const c2v (&tw3)[16], cldsp lds, int r) {
  const F3_LDS f4v_* row = (const F3_LDS f4v_*)(lds + e2(r, 0));
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const f4v_ v = row[j];
    y[2 * j] = c2v{v.x, v.y};
    y[2 * j + 1] = c2v{v.z, v.w};
  }
  y[1] = cmul(y[1], tw3[1]);
  y[2] = cmul(y[2], tw3[2]);
  y[3] = cmul(y[3], tw3[3]);
#pragma unroll
  for (int m2 = 4; m2 < 16; m2 += 4)
    cmul4<false>(y[m2], y[m2 + 1], y[m2 + 2], y[m2 + 3], y[m2], y[m2 + 1], y[m2 + 2], y[m2 + 3], tw3[m2], tw3[m2 + 1], tw3[m2 + 2], tw3[m2 + 3]);
  dft16<false>(y);
}

// inverse pass 1 for row r: y[k3] = Y[k3 * 1024 + r] -> E2[r][m2]
F3_FN void inv_pass1_compute(c2v (&y)[16], const c2v (&tw3)[16]) {
  dft16<true>(y);
  y[1] = cmulc(y[1], tw3[K16(1)]);
  y[2] = cmulc(y[2], tw3[K16(2)]);
  y[3] = cmulc(y[3], tw3[K16(3)]);
#pragma unroll
  for (int s = 4; s < 16; s += 4)
    cmul4<true>(y[s], y[s + 1], y[s + 2], y[s + 3], y[s], y[s + 1], y[s + 2], y[s + 3], tw3[K16(s)], tw3[K16(s + 1)], tw3[K16(s + 2)], tw3[K16(s + 3)]);
}
F3_FN void inv_pass1_write(const c2v (&y)[16], ldsp lds, int r) {
  F3_LDS f4v_* row = (F3_LDS f4v_*)(lds + e2(r, 0));
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const c2v a = y[K16(2 * j)], b = y[K16(2 * j + 1)];  // slot of output index m2 (K16 is an involution)
    row[j] = f4v_{a.x, a.y, b.x, b.y};
  }
}
// inverse pass 2, thread (k1, m2): transform over k2 -> m1 (barrier, then the write: E1 overwrites E2)
F3_FN void inv_pass2_compute(c2v (&x)[32], cldsp lds, int t) {
  const int base = e2((t >> 4) * 32, t & 15);
#pragma unroll
  for (int k2 = 0; k2 < 32; k2++) x[k2] = lds_rd(lds + base + k2 * E2_ROW);
  dft32<true>(x);
}
F3_FN void inv_pass2_write(const c2v (&x)[32], ldsp lds, int t) {
  const int base = e1(t >> 4, t & 15);
#pragma unroll
  for (int s = 0; s < 32; s++) lds[base + K32(s) * 16] = x[s];
}
// inverse pass 3, thread m = t: x[s] (odd s) <- 16384 * z[K32(s) * 512 + t], the second half of the window
F3_FN void inv_pass3(c2v (&x)[32], const c2v (&twn)[32], cldsp lds, int t) {
#pragma unroll
  for (int k1 = 0; k1 < 32; k1++) x[k1] = lds_rd(lds + e1(k1, 0) + t);
  x[1] = cmulc(x[1], twn[1]);
  x[2] = cmulc(x[2], twn[2]);
  x[3] = cmulc(x[3], twn[3]);
#pragma unroll
  for (int k1 = 4; k1 < 32; k1 += 4)
    cmul4<true>(x[k1], x[k1 + 1], x[k1 + 2], x[k1 + 3], x[k1], x[k1 + 1], x[k1 + 2], x[k1 + 3], twn[k1], twn[k1 + 1], twn[k1 + 2], twn[k1 + 3]);
  dft32<true, true>(x);
}

}  // namespace fft3
}  // namespace waa
