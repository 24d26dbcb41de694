// waa_conv3.hip — the N = 16384 transforms of the ConvolverNode path (B = 8192: impulse responses of 2^16 .. 2^17.6 frames,
// the parking-garage response of BASELINE configs 3 / 4 and of the north-star graph) as three register passes with two
// LDS exchanges (waa_fft3.hpp has the algebra, the LDS map and the per-thread pass bodies).
//
// Kernel shape, as before: ONE workgroup of 512 threads per CU (the exchange buffer takes 148 KB of the 160 KB), persistent
// over the blocks of one (instance pair, channel), software-pipelined: the next block's input is requested into registers
// before the current transform starts.  A thread owns frames t, t + 512, t + 1024, ... of the window (forward) resp. the
// positions r, r + 1024, ... of the spectrum (inverse), so every global access of a wavefront is one contiguous run; the
// shared half of consecutive overlap-save windows stays in registers (input read once).
//
// Per block and thread: ~760 packed-f32 instructions, 64 + 64 LDS reads and 64 LDS writes, four barriers —
// against ~1230 f32 + ~880 integer / move instructions, 128 + 136 LDS accesses and five barriers of conv_fft_pipe_kernel.
// WAA_CONV_FFT_R4=1 (read when the batch is planned) keeps the round-2 kernels for same-box A/Bs.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_fft3.hpp"
#include "waa_internal.hpp"
#include "waa_stream_common.hpp"

namespace waa {

namespace {

using namespace fft3;

// workgroup barrier for LDS hand-offs: this wave's LDS traffic has completed, then the barrier — NOT __syncthreads(),
// whose release fence also waits for every outstanding global store (waa_conv.hip, pipe_barrier)
__device__ __forceinline__ void f3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// forces the wait for prefetched registers to sit HERE (in front of a block's stores: loads and stores share one counter)
template <int CNT>
__device__ __forceinline__ void f3_settle(c2v (&v)[CNT]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < CNT; r++) asm volatile("" : "+v"(v[r])::"memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int f3_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// half a window: frames f0 + j * 512 + t (j < 16) of both instances.  Loads are unconditional (clamped address); frames
// outside [0, valid) are zeroed when the half is staged (f3_mask) — a load behind `if` makes the compiler wait for it on
// the spot (waa_conv.hip, pipe_load_half).
__device__ __forceinline__ void f3_load_half(const float* pa, const float* pb, int64_t f0, uint64_t valid, int t, c2v (&v)[16]) {
  const bool inside = f0 >= 0 && (uint64_t)f0 + B <= valid;  // uniform
  if (inside) {
    const float* qa = pa + f0 + t;
    const float* qb = pb + f0 + t;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      v[j].x = qa[j * 512];
      v[j].y = qb[j * 512];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int64_t f = f0 + j * 512 + t;
      const int64_t fc = f >= 0 && (uint64_t)f < valid ? f : 0;
      v[j].x = pa[fc];
      v[j].y = pb[fc];
    }
  }
}
__device__ __forceinline__ c2v f3_mask(c2v v, int64_t f, uint64_t valid, bool has_b) {
  const bool ok = f >= 0 && (uint64_t)f < valid;
  return c2v{ok ? v.x : 0.f, ok && has_b ? v.y : 0.f};
}

enum { F3_FWD = 0, F3_IR = 2 };

template <int MODE>
__global__ __launch_bounds__(NT) void conv_fft3_fwd_kernel(const ConvDesc d, int blocks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  ldsp lds = (ldsp)lds_raw;
  const int t = threadIdx.x;
  const c2v* twg = reinterpret_cast<const c2v*>(d.tw);
  c2v tws[32], tw3[16];
  load_tw1_slots(twg, t, tws);
  load_tw3(twg, t, tw3);
  if (MODE == F3_IR) {
    // spectrum of IR partition k of IR channel c: h[kB .. (k+1)B) zero-padded to 2B, imaginary part 0
    const int k = blockIdx.x, c = blockIdx.y;
    const float* h = d.ir + (uint64_t)c * d.ir_len;
    c2v x[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t idx = (uint64_t)k * B + j * 512 + t;
      x[j] = c2v{idx < d.ir_len ? h[idx] : 0.f, 0.f};
      x[j + 16] = c2v{0.f, 0.f};
    }
    fwd_pass1_compute(x, tws);
    fwd_pass1_write(x, lds, t);
    f3_barrier();
    fwd_pass2_compute(x, lds, t);
    f3_barrier();
    fwd_pass2_write(x, lds, t);
    f3_barrier();
    c2v* dst = reinterpret_cast<c2v*>(const_cast<Cplx*>(d.H)) + ((uint64_t)c * d.parts + k) * N;
#pragma unroll
    for (int set = 0; set < 2; set++) {
      const int r = t + set * NT;
      c2v y[16];
      fwd_pass3(y, tw3, lds, r);
#pragma unroll
      for (int s = 0; s < 16; s++) dst[K16(s) * 1024 + r] = y[s];
    }
    return;
  }
  const int c = blockIdx.y;
  const uint32_t pair = blockIdx.z;
  const int k0 = d.kb0 + blockIdx.x * blocks_per_wg;
  const int k1 = k0 + blocks_per_wg < d.kb1 ? k0 + blocks_per_wg : d.kb1;
  if (k0 >= k1) return;
  const uint32_t ia = pair * 2, ib = pair * 2 + 1;
  const bool has_b = ib < d.n_inst;
  const float* pa = d.in.base + (uint64_t)ia * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
  const float* pb = d.in.base + (uint64_t)(has_b ? ib : ia) * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
  c2v* xbase = reinterpret_cast<c2v*>(d.X) + ((uint64_t)pair * d.cin + c) * d.nb * N;
  c2v hold[16], pre[16];  // the window's first half (masked already) and the prefetched second half (raw)
  f3_load_half(pa, pb, ((int64_t)k0 - 1) * B, d.in_valid, t, hold);
  f3_load_half(pa, pb, (int64_t)k0 * B, d.in_valid, t, pre);
#pragma unroll
  for (int j = 0; j < 16; j++) hold[j] = f3_mask(hold[j], ((int64_t)k0 - 1) * B + j * 512 + t, d.in_valid, has_b);
  // (no load may be pending on loop entry: the wait counts at the loop header merge this path with the back edge, where
  // the previous block's stores are in flight)
  f3_settle(hold);
  f3_settle(pre);
  for (int k = k0; k < k1; k++) {
    const int tk = f3_opaque(t);  // (keeps the loop-invariant LDS / global address arithmetic from being hoisted into registers)
    c2v x[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      x[j] = hold[j];
      x[j + 16] = f3_mask(pre[j], (int64_t)k * B + j * 512 + tk, d.in_valid, has_b);
      hold[j] = x[j + 16];
    }
    // the next block's new half: in flight during the transform (past the last block: the clamped, zeroed form)
    f3_load_half(pa, pb, k + 1 < k1 ? ((int64_t)k + 1) * B : (int64_t)d.in_valid, d.in_valid, tk, pre);
    fwd_pass1_compute(x, tws);
    f3_barrier();  // the previous block's pass 3 has read its rows
    fwd_pass1_write(x, lds, tk);
    f3_barrier();
    fwd_pass2_compute(x, lds, tk);
    f3_barrier();
    fwd_pass2_write(x, lds, tk);
    f3_barrier();
    f3_settle(pre);
    c2v* dst = xbase + (uint64_t)k * N;
#pragma unroll
    for (int set = 0; set < 2; set++) {
      const int r = tk + set * NT;
      c2v y[16];
      fwd_pass3(y, tw3, lds, r);
#pragma unroll
      for (int s = 0; s < 16; s++) dst[K16(s) * 1024 + r] = y[s];
    }
  }
}

// ---- forward transform with the BiquadFilterNode in front of the convolver folded into its input stage ----------------
// T1 / C4 (the north-star graph): source -> Biquad -> Convolver.  As two launches the filtered signal is written and read
// back once (7.9 GB of the 49 GB a T1 render moved, and a 1.3-1.6 ms launch of its own); here the workgroup that transforms
// block k of (pair, channel) first FILTERS the block's 8192 new frames of both instances:
//   * the raw frames arrive coalesced (16 B per lane, prefetched one block ahead) and are transposed through LDS so that
//     thread (half h, j) owns the 32 consecutive frames [32 j, 32 j + 32) of instance a (h = 0) or b (h = 1): four
//     wavefronts per stream, in time order;
//   * the recurrence is the scheme of waa_biquad_stream.hip, one level deeper: zero-state response per lane, DPP scan of
//     the affine maps inside each wavefront (A = M^32, powers A .. A^8 within a row, A^16 across rows), the wavefronts' end
//     states combined through LDS with A^64 — and then the reference's evaluation order, unfused, from every lane's true
//     incoming state (biquad_filter.rs:877-883); the FIR part is recomputed in the second pass instead of kept (registers);
//   * the results go back through LDS into pass 1's layout (thread m holds frames m, m + 512, ... of both instances).
// The streams are serial in time, so a workgroup owns a whole (pair, channel): blocks_per_wg = nb.
namespace bq {
// LDS map of the filter stage.  Rows of 32 frames + 4 (conflict-free 16-byte row accesses), 256 rows per instance, in the
// exchange buffer's space (free until pass 1 writes E1); the small f64 tables live behind the exchange buffer.
constexpr int ROW = 36, HALF = 256 * ROW;            // floats
constexpr int TAB = 160;                              // doubles per instance half
constexpr int T_A1 = 0, T_A2 = 4, T_A4 = 8, T_A8 = 12, T_A16 = 16, T_A64 = 20, T_CO = 24, T_AJ = 32, T_W = 96, T_CY = 104, T_CX = 106;
constexpr int TW3_OFF = LDS_BYTES + 2 * TAB * 8;    // bytes: pass 3's W_512 twiddles, [m2][k2] (they depend on k2 = t mod 32 only)
constexpr int LDS_BYTES_BQ = TW3_OFF + 16 * 32 * 8;
struct M2 {
  double a, b, c, d;
};
__device__ __forceinline__ M2 mm(const M2& x, const M2& y) {
  M2 r;
  r.a = __builtin_fma(x.a, y.a, x.b * y.c);
  r.b = __builtin_fma(x.a, y.b, x.b * y.d);
  r.c = __builtin_fma(x.c, y.a, x.d * y.c);
  r.d = __builtin_fma(x.c, y.b, x.d * y.d);
  return r;
}
typedef __attribute__((address_space(3))) double* ldsd;
__device__ __forceinline__ M2 ld_m2(const __attribute__((address_space(3))) double* p) { return M2{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void mv(const M2& m, double x1, double x2, double e1, double e2, double& o1, double& o2) {
  o1 = __builtin_fma(m.a, x1, __builtin_fma(m.b, x2, e1));
  o2 = __builtin_fma(m.c, x1, __builtin_fma(m.d, x2, e2));
}
}  // namespace bq

__global__ __launch_bounds__(NT) void conv_fft3_fwd_bq_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  ldsp lds = (ldsp)lds_raw;
  typedef __attribute__((address_space(3))) float* ldsf;
  const ldsf ldf = (ldsf)lds_raw;
  const int t = threadIdx.x;
  const c2v* twg = reinterpret_cast<const c2v*>(d.tw);
  c2v twh[16], w16;
  load_tw1_half(twg, t, twh, w16);
  // (pass 3's twiddles live in LDS here, not in registers: the filter stage needs them)
  const ldsp tw3s = (ldsp)(lds_raw + bq::TW3_OFF / 4);
  tw3s[t] = twg[((t >> 5) * (t & 31) * 32) & (N - 1)];  // entry [m2 = t >> 5][k2 = t & 31]
  const int c = blockIdx.y;
  const uint32_t pair = blockIdx.z;
  const uint32_t ia = pair * 2, ib = pair * 2 + 1;
  const bool has_b = ib < d.n_inst;
  // f64 denormals: flush inputs and outputs, like the reference's FTZ/DAZ render scope (waa_biquad_stream.hip)
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  // filter-stage roles: half h filters instance a (0) or b (1); j = the thread's row, wv = its wavefront in time order
  const int h = __builtin_amdgcn_readfirstlane(t >> 8);
  const int j = t & 255, lane = t & 63, row = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((t >> 6) & 3);
  const uint32_t inst_h = h && has_b ? ib : ia;
  const float* ph = d.in.base + (uint64_t)inst_h * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
  c2v* xbase = reinterpret_cast<c2v*>(d.X) + ((uint64_t)pair * d.cin + c) * d.nb * N;
  const bq::ldsd tab = (bq::ldsd)(lds_raw + LDS_BYTES / 4) + h * bq::TAB;
  double* stp = d.pre_state + (uint64_t)inst_h * STATE_STRIDE + c * 4;
  {
    // coefficients and the powers of the 32-step transition A = M^32, M = [[-a1, -a2], [1, 0]]: once per workgroup, in LDS
    const double* cp = d.pre_coefs + (uint64_t)inst_h * d.pre_coef_stride;
    const double a1 = cp[3], a2 = cp[4];
    bq::M2 m = {-a1, -a2, 1., 0.};
#pragma unroll
    for (int sq = 0; sq < 5; sq++) m = bq::mm(m, m);
    const bq::M2 A1 = m, A2 = bq::mm(A1, A1), A4 = bq::mm(A2, A2), A8 = bq::mm(A4, A4), A16 = bq::mm(A8, A8);
    const bq::M2 A32 = bq::mm(A16, A16), A64 = bq::mm(A32, A32);
    if (j < 16) {  // A^(lane % 16)
      bq::M2 Aj = {1., 0., 0., 1.};
      if (j & 1) Aj = bq::mm(Aj, A1);
      if (j & 2) Aj = bq::mm(Aj, A2);
      if (j & 4) Aj = bq::mm(Aj, A4);
      if (j & 8) Aj = bq::mm(Aj, A8);
      tab[bq::T_AJ + j * 4 + 0] = Aj.a;
      tab[bq::T_AJ + j * 4 + 1] = Aj.b;
      tab[bq::T_AJ + j * 4 + 2] = Aj.c;
      tab[bq::T_AJ + j * 4 + 3] = Aj.d;
    }
    if (j == 0) {
      const bq::M2 ms[6] = {A1, A2, A4, A8, A16, A64};
#pragma unroll
      for (int q = 0; q < 6; q++) {
        tab[q * 4 + 0] = ms[q].a;
        tab[q * 4 + 1] = ms[q].b;
        tab[q * 4 + 2] = ms[q].c;
        tab[q * 4 + 3] = ms[q].d;
      }
#pragma unroll
      for (int q = 0; q < 5; q++) tab[bq::T_CO + q] = cp[q];
      // carried state: x[n-1], x[n-2], y[n-1], y[n-2] (biquad_filter.rs:761)
      tab[bq::T_CX + 0] = stp[0];
      tab[bq::T_CX + 1] = stp[1];
      tab[bq::T_CY + 0] = stp[2];
      tab[bq::T_CY + 1] = stp[3];
    }
  }
  // raw input of a block, coalesced: thread (h, j) asks for frames 4 (j + 256 r) .. + 3, r < 8, of its instance
  f4v_ raw[8];
  auto load_raw = [&](int64_t f0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int64_t f = f0 + 4 * (int64_t)(j + 256 * r);
      const int64_t fc = (uint64_t)f + 3 < d.in_valid ? f : 0;  // (clamped; zeroed when staged)
      raw[r] = *reinterpret_cast<const f4v_*>(ph + fc);
    }
  };
  load_raw(0);
  c2v hold[16];
#pragma unroll
  for (int q = 0; q < 16; q++) hold[q] = c2v{0.f, 0.f};  // block 0's first half: frames before the stream
#pragma unroll
  for (int r = 0; r < 8; r++) asm volatile("" : "+v"(raw[r])::"memory");
  for (int k = 0; k < d.nb; k++) {
    const int tk = f3_opaque(t);
    const int jk = tk & 255;
    f3_barrier();  // the previous block's pass 3 has read its rows; the filter tables are written
    // ---- stage the raw block: row (frame >> 5), column (frame & 31) of the half's buffer
    {
      const ldsf rb = ldf + h * bq::HALF;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int64_t f = (int64_t)k * B + 4 * (int64_t)(jk + 256 * r);
        const bool ok = (uint64_t)f + 3 < d.in_valid;
        const f4v_ z = {0.f, 0.f, 0.f, 0.f};
        const int fl = 4 * (jk + 256 * r);
        *(__attribute__((address_space(3))) f4v_*)(rb + (fl >> 5) * bq::ROW + (fl & 31)) = ok ? raw[r] : z;
      }
    }
    load_raw(k + 1 < d.nb ? ((int64_t)k + 1) * B : (int64_t)d.in_valid);  // the next block: in flight during this one
    f3_barrier();
    // ---- this thread's 32 frames, the two before them, and the block's last two (next block's x history)
    const ldsf myrow = ldf + h * bq::HALF + jk * bq::ROW;
    float x[32];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const f4v_ v = *(const __attribute__((address_space(3))) f4v_*)(myrow + q * 4);
      x[q * 4 + 0] = v.x;
      x[q * 4 + 1] = v.y;
      x[q * 4 + 2] = v.z;
      x[q * 4 + 3] = v.w;
    }
    double xs1, xs2;  // x[n-1], x[n-2] in front of this thread's chunk
    {
      const ldsf prow = ldf + h * bq::HALF + (jk > 0 ? jk - 1 : 0) * bq::ROW;
      const float p1 = prow[31], p2 = prow[30];
      const double c1 = tab[bq::T_CX + 0], c2 = tab[bq::T_CX + 1];
      xs1 = jk > 0 ? (double)p1 : c1;
      xs2 = jk > 0 ? (double)p2 : c2;
    }
    const ldsf lrow = ldf + h * bq::HALF + 255 * bq::ROW;
    const float nx1 = lrow[31], nx2 = lrow[30];
    const double b0 = tab[bq::T_CO + 0], b1 = tab[bq::T_CO + 1], b2 = tab[bq::T_CO + 2], a1 = tab[bq::T_CO + 3], a2 = tab[bq::T_CO + 4];
    // ---- sweep 1: zero-state response of the chunk (fused multiply-adds: this pass only feeds the scan, whose result is
    // an incoming STATE accurate to f64 rounding; the samples come out of sweep 2 in the reference's own order)
    double z1 = 0., z2 = 0.;
    {
      double x1 = xs1, x2 = xs2;
#pragma unroll
      for (int i = 0; i < 32; i++) {
        double xd = (double)x[i];
        // (ties frame i's conversion to frame i - 2's result: without it the compiler hoists all 32 conversions and
        // products — they do not depend on the recurrence — and spills 46 registers; same trick as waa_iir_stream.hip)
        asm volatile("" : "+v"(xd) : "v"(z2));
        const double w = __builtin_fma(b2, x2, __builtin_fma(b1, x1, b0 * xd));
        x2 = x1;
        x1 = xd;
        const double u = __builtin_fma(-a2, z2, w);
        const double y = __builtin_fma(-a1, z1, u);
        z2 = z1;
        z1 = y;
      }
    }
    // in-row inclusive scan (rows of 16 lanes): R_l = sum_{i in row, i <= l} A^(l - i) z_i
    double r1 = z1, r2 = z2;
    {
      bq::M2 P = bq::ld_m2(tab + bq::T_A1);
      double q1 = row_shr<1>(r1), q2 = row_shr<1>(r2);
      bq::mv(P, q1, q2, r1, r2, r1, r2);
      P = bq::ld_m2(tab + bq::T_A2);
      q1 = row_shr<2>(r1);
      q2 = row_shr<2>(r2);
      bq::mv(P, q1, q2, r1, r2, r1, r2);
      P = bq::ld_m2(tab + bq::T_A4);
      q1 = row_shr<4>(r1);
      q2 = row_shr<4>(r2);
      bq::mv(P, q1, q2, r1, r2, r1, r2);
      P = bq::ld_m2(tab + bq::T_A8);
      q1 = row_shr<8>(r1);
      q2 = row_shr<8>(r2);
      bq::mv(P, q1, q2, r1, r2, r1, r2);
    }
    const bq::M2 A16 = bq::ld_m2(tab + bq::T_A16);
    const double e01 = read_lane(r1, 15), e02 = read_lane(r2, 15), e11 = read_lane(r1, 31), e12 = read_lane(r2, 31);
    const double e21 = read_lane(r1, 47), e22 = read_lane(r2, 47), e31 = read_lane(r1, 63), e32 = read_lane(r2, 63);
    {
      // zero-state end state of this wavefront's 2048 frames
      double u1 = e01, u2 = e02;
      bq::mv(A16, u1, u2, e11, e12, u1, u2);
      bq::mv(A16, u1, u2, e21, e22, u1, u2);
      bq::mv(A16, u1, u2, e31, e32, u1, u2);
      if (lane == 0) {
        tab[bq::T_W + wv * 2 + 0] = u1;
        tab[bq::T_W + wv * 2 + 1] = u2;
      }
    }
    f3_barrier();
    // ---- the state entering this wavefront, its rows, this lane
    double t1 = tab[bq::T_CY + 0], t2 = tab[bq::T_CY + 1];
    {
      const bq::M2 A64 = bq::ld_m2(tab + bq::T_A64);
      for (int w = 0; w < wv; w++) bq::mv(A64, t1, t2, tab[bq::T_W + w * 2], tab[bq::T_W + w * 2 + 1], t1, t2);
    }
    double T1 = t1, T2 = t2;
    {
      double v1 = t1, v2 = t2;
      bq::mv(A16, v1, v2, e01, e02, v1, v2);
      if (row == 1) {
        T1 = v1;
        T2 = v2;
      }
      bq::mv(A16, v1, v2, e11, e12, v1, v2);
      if (row == 2) {
        T1 = v1;
        T2 = v2;
      }
      bq::mv(A16, v1, v2, e21, e22, v1, v2);
      if (row == 3) {
        T1 = v1;
        T2 = v2;
      }
    }
    double s1, s2;
    {
      const bq::M2 Aj = bq::ld_m2(tab + bq::T_AJ + (lane & 15) * 4);
      const double ex1 = row_shr<1>(r1), ex2 = row_shr<1>(r2);
      bq::mv(Aj, T1, T2, ex1, ex2, s1, s2);
    }
    // ---- sweep 2: the reference's evaluation order from the true incoming state (biquad_filter.rs:877-883); the results go
    // into the thread's own row four at a time (the row's input values are in registers)
    double y1 = s1, y2 = s2;
    {
      float badacc = 0.f;  // becomes NaN as soon as one output is inf / NaN, off the critical path
      double x1 = xs1, x2 = xs2;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        float yo[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double xd = (double)x[q * 4 + e];
          const double w = (b0 * xd + b1 * x1) + b2 * x2;
          x2 = x1;
          x1 = xd;
          const double y = (w - a1 * y1) - a2 * y2;
          y2 = y1;
          y1 = y;
          yo[e] = (float)y;
          badacc = __builtin_fmaf(yo[e], 0.f, badacc);
        }
        *(__attribute__((address_space(3))) f4v_*)(myrow + q * 4) = f4v_{yo[0], yo[1], yo[2], yo[3]};
      }
      if (__any(badacc != badacc)) {  // inf / NaN somewhere: redo with the explicit flush
        y1 = __builtin_isfinite(s1) ? s1 : 0.;
        y2 = __builtin_isfinite(s2) ? s2 : 0.;
        x1 = xs1;
        x2 = xs2;
#pragma unroll  // (fully: a rolled loop indexes x[] dynamically, which moves the whole array to scratch memory — 64 KB of
                // extra stores per block, measured as +3.96 GB of WRITE_SIZE on T1)
        for (int q = 0; q < 8; q++) {
          float yo[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const double xd = (double)x[q * 4 + e];
            const double w = (b0 * xd + b1 * x1) + b2 * x2;
            x2 = x1;
            x1 = xd;
            double y = (w - a1 * y1) - a2 * y2;
            if (!__builtin_isnormal(y)) y = 0.;
            y2 = y1;
            y1 = y;
            yo[e] = (float)y;
          }
          *(__attribute__((address_space(3))) f4v_*)(myrow + q * 4) = f4v_{yo[0], yo[1], yo[2], yo[3]};
        }
      }
    }
    // the last thread of the stream publishes the carried state
    if (jk == 255) {
      tab[bq::T_CY + 0] = y1;
      tab[bq::T_CY + 1] = y2;
      tab[bq::T_CX + 0] = (double)nx1;
      tab[bq::T_CX + 1] = (double)nx2;
    }
    f3_barrier();
    // ---- pass 1's layout: thread m holds frames m, m + 512, ... of both instances
    c2v xx[32];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int f = q * 512 + tk;
      const int o = (f >> 5) * bq::ROW + (f & 31);
      const float va = ldf[o], vb = ldf[bq::HALF + o];
      xx[q] = hold[q];
      xx[q + 16] = c2v{va, has_b ? vb : 0.f};
      hold[q] = xx[q + 16];
    }
    fwd_pass1_compute_half(xx, twh, w16);
    f3_barrier();  // every thread has read its frames
    fwd_pass1_write(xx, lds, tk);
    f3_barrier();
    fwd_pass2_compute(xx, lds, tk);
    f3_barrier();
    fwd_pass2_write(xx, lds, tk);
    f3_barrier();
#pragma unroll
    for (int r = 0; r < 8; r++) asm volatile("" : "+v"(raw[r])::"memory");  // (the prefetch settles in front of the stores)
    c2v* dst = xbase + (uint64_t)k * N;
    c2v tw3[16];
#pragma unroll
    for (int m2 = 1; m2 < 16; m2++) tw3[m2] = tw3s[m2 * 32 + (tk & 31)];
    tw3[0] = c2v{1.f, 0.f};
#pragma unroll
    for (int set = 0; set < 2; set++) {
      const int r = tk + set * NT;
      c2v y[16];
      fwd_pass3(y, tw3, lds, r);
#pragma unroll
      for (int sl = 0; sl < 16; sl++) dst[K16(sl) * 1024 + r] = y[sl];
    }
  }
  // final filter state (the batch's state buffer; a later launch of a block-scheduled plan would continue from it)
  f3_barrier();
  if (j == 0 && (h == 0 || has_b)) {
    stp[0] = tab[bq::T_CX + 0];
    stp[1] = tab[bq::T_CX + 1];
    stp[2] = tab[bq::T_CY + 0];
    stp[3] = tab[bq::T_CY + 1];
  }
}

__global__ __launch_bounds__(NT) void conv_fft3_inv_kernel(const ConvDesc d, int blocks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  ldsp lds = (ldsp)lds_raw;
  const int t = threadIdx.x;
  const c2v* twg = reinterpret_cast<const c2v*>(d.tw);
  c2v twn[32], tw3[16];
  load_tw1_natural(twg, t, twn);
  load_tw3(twg, t, tw3);
  const int c = blockIdx.y;
  const uint32_t pair = blockIdx.z;
  const int k0 = d.kb0 + blockIdx.x * blocks_per_wg;
  const int k1 = k0 + blocks_per_wg < d.kb1 ? k0 + blocks_per_wg : d.kb1;
  if (k0 >= k1) return;
  const uint32_t ia = pair * 2, ib = pair * 2 + 1;
  const bool has_b = ib < d.n_inst;
  float* pa = d.out.base + (uint64_t)ia * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
  float* pb = d.out.base + (uint64_t)(has_b ? ib : ia) * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
  const c2v* ybase = reinterpret_cast<const c2v*>(d.Y) + ((uint64_t)pair * d.cout + c) * d.nb * N;
  const float scale = 1.f / (float)N;
  c2v pre[32];  // [set * 16 + k3] = Y_k[k3 * 1024 + t + set * 512]
  {
    const c2v* src = ybase + (uint64_t)k0 * N + t;
#pragma unroll
    for (int i = 0; i < 32; i++) pre[i] = src[(i & 15) * 1024 + (i >> 4) * NT];
  }
  f3_settle(pre);
  for (int k = k0; k < k1; k++) {
    const int tk = f3_opaque(t);
    c2v y0[16], y1[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      y0[i] = pre[i];
      y1[i] = pre[16 + i];
    }
    {
      // the next spectrum: in flight during the transform (the last block re-reads itself: harmless)
      const int kn = k + 1 < k1 ? k + 1 : k;
      const c2v* src = ybase + (uint64_t)kn * N + tk;
#pragma unroll
      for (int i = 0; i < 32; i++) pre[i] = src[(i & 15) * 1024 + (i >> 4) * NT];
    }
    inv_pass1_compute(y0, tw3);
    inv_pass1_compute(y1, tw3);
    f3_barrier();  // the previous block's pass 3 has read its columns
    inv_pass1_write(y0, lds, tk);
    inv_pass1_write(y1, lds, tk + NT);
    f3_barrier();
    c2v x[32];
    inv_pass2_compute(x, lds, tk);
    f3_barrier();
    inv_pass2_write(x, lds, tk);
    f3_barrier();
    inv_pass3(x, twn, lds, tk);
    f3_settle(pre);
    // overlap-save: the second half of the window is the linear convolution; re -> instance a, im -> instance b
    const uint64_t fb = (uint64_t)k * B + tk;
    if ((uint64_t)(k + 1) * B <= d.frames) {  // (uniform; only the last block of a stream can be partial)
      float* qa = pa + fb;
      float* qb = pb + fb;
#pragma unroll
      for (int s = 1; s < 32; s += 2) qa[(K32(s) - 16) * 512] = x[s].x * scale;
      if (has_b) {
#pragma unroll
        for (int s = 1; s < 32; s += 2) qb[(K32(s) - 16) * 512] = x[s].y * scale;
      }
    } else {
#pragma unroll
      for (int s = 1; s < 32; s += 2) {
        const uint64_t f = fb + (uint64_t)(K32(s) - 16) * 512;
        if (f < d.frames) {
          pa[f] = x[s].x * scale;
          if (has_b) pb[f] = x[s].y * scale;
        }
      }
    }
  }
}

}  // namespace

static void f3_allow_lds() {
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft3_fwd_kernel<F3_FWD>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft3_fwd_kernel<F3_IR>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft3_inv_kernel));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft3_fwd_bq_kernel));
}
// blocks per persistent workgroup: whole (pair, channel) streams when there are enough of them to fill the chip, shorter
// runs otherwise
static int f3_blocks_per_wg(const ConvDesc& d, int channels) {
  const int streams = (int)d.n_pairs * channels, nbl = d.kb1 - d.kb0;
  int segs = streams >= 512 ? 1 : (512 + streams - 1) / streams;
  if (segs > nbl) segs = nbl;
  return (nbl + segs - 1) / segs;
}

void launch_conv3_ir_spectra(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  hipLaunchKernelGGL(conv_fft3_fwd_kernel<F3_IR>, dim3(d.parts, d.ir_nch, 1), dim3(NT), (size_t)LDS_BYTES, (hipStream_t)stream, d, 1);
}
void launch_conv3_forward(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  if (d.pre_coefs) {  // the filter in front is serial in time: one workgroup per (pair, channel) stream
    hipLaunchKernelGGL(conv_fft3_fwd_bq_kernel, dim3(1, d.cin, d.n_pairs), dim3(NT), (size_t)bq::LDS_BYTES_BQ, (hipStream_t)stream, d);
    return;
  }
  const int bpw = f3_blocks_per_wg(d, d.cin);
  hipLaunchKernelGGL(conv_fft3_fwd_kernel<F3_FWD>, dim3((d.kb1 - d.kb0 + bpw - 1) / bpw, d.cin, d.n_pairs), dim3(NT), (size_t)LDS_BYTES,
                     (hipStream_t)stream, d, bpw);
}
void launch_conv3_inverse(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  const int bpw = f3_blocks_per_wg(d, d.cout);
  hipLaunchKernelGGL(conv_fft3_inv_kernel, dim3((d.kb1 - d.kb0 + bpw - 1) / bpw, d.cout, d.n_pairs), dim3(NT), (size_t)LDS_BYTES,
                     (hipStream_t)stream, d, bpw);
}

}  // namespace waa
