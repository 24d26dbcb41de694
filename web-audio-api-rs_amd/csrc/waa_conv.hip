// waa_conv.hip — ConvolverNode on gfx950: node-major partitioned overlap-save convolution.
//
// The reference (convolver.rs:343-490 -> crate fft-convolver) runs, per 128-frame quantum and per
// convolver, a 2048-point real FFT, a complex MAC against P = ceil(len/1024) IR partitions held in
// a frequency-domain delay line, and an inverse FFT (overlap-add).  That arithmetic is the linear
// convolution of the input with the (normalised, scaled) impulse response.  Because a whole offline
// render is available node-major, the same linear convolution is evaluated here time-tiled:
//
//   conv_fft_kernel<FWD>  one workgroup per (block k, input channel, instance pair): the two
//                         instances' real streams are packed as z = a + i b (the IR spectra are shared
//                         by every instance, so (a + i b) * h = a*h + i b*h), window [(k-1)B, (k+1)B),
//                         in-place radix-4 DIF FFT in LDS, spectrum stored in bit-reversed order.
//   conv_mac_kernel       Y_k[pos] = sum_p H_p[pos] * X_{k-p}[pos], register-tiled over 16 output
//                         blocks so every input spectrum is read ~2.4x instead of P times.
//   conv_fft_kernel<INV>  radix-4 DIT inverse straight from the bit-reversed spectrum, last B samples
//                         (overlap-save), real part -> instance a, imaginary part -> instance b.
//
// f32 throughout (the reference convolver is f32).  No MFMA: the FFTs are LDS/HBM bound and the MAC is
// a streaming f32 FMA kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"

namespace waa {

namespace {

__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) {
  Cplx r;
  r.re = __builtin_fmaf(a.re, b.re, -(a.im * b.im));
  r.im = __builtin_fmaf(a.re, b.im, a.im * b.re);
  return r;
}
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return Cplx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return Cplx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cplx mul_negi(Cplx a) { return Cplx{a.im, -a.re}; }  // a * (-i)
__device__ __forceinline__ Cplx mul_posi(Cplx a) { return Cplx{-a.im, a.re}; }  // a * (+i)
__device__ __forceinline__ Cplx conj(Cplx a) { return Cplx{a.re, -a.im}; }

// in-place radix-4 decimation-in-frequency FFT: natural order in, bit-reversed order out. n = 4^m.
// n = 4^m or 2 * 4^m (a trailing radix-2 stage on adjacent pairs)
__device__ __forceinline__ void fft_dif(Cplx* a, const Cplx* tw, int n, int tid, int nthreads) {
  int L = n;
  for (; L >= 4; L >>= 2) {
    const int q = L >> 2;
    const int tstep = n / (4 * q);
    for (int b = tid; b < (n >> 2); b += nthreads) {
      const int j = b % q, base = (b / q) * 4 * q + j;
      const Cplx x0 = a[base], x1 = a[base + q], x2 = a[base + 2 * q], x3 = a[base + 3 * q];
      const Cplx s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = mul_negi(csub(x1, x3));
      const Cplx w1 = tw[j * tstep];
      const Cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1);
      a[base] = cadd(s02, s13);
      a[base + q] = cmul(csub(s02, s13), w2);
      a[base + 2 * q] = cmul(cadd(d02, d13), w1);
      a[base + 3 * q] = cmul(csub(d02, d13), w3);
    }
    __syncthreads();
  }
  if (L == 2) {
    for (int b = tid; b < (n >> 1); b += nthreads) {
      const Cplx u = a[2 * b], v = a[2 * b + 1];
      a[2 * b] = cadd(u, v);
      a[2 * b + 1] = csub(u, v);
    }
    __syncthreads();
  }
}
// in-place radix-4 decimation-in-time inverse FFT: bit-reversed order in, natural order out (unscaled).
__device__ __forceinline__ void fft_dit_inv(Cplx* a, const Cplx* tw, int n, int tid, int nthreads) {
  int q0 = 1;
  if (__builtin_ctz(n) & 1) {  // n = 2 * 4^m: leading radix-2 stage on adjacent pairs
    for (int b = tid; b < (n >> 1); b += nthreads) {
      const Cplx u = a[2 * b], v = a[2 * b + 1];
      a[2 * b] = cadd(u, v);
      a[2 * b + 1] = csub(u, v);
    }
    __syncthreads();
    q0 = 2;
  }
  for (int q = q0; q <= (n >> 2); q <<= 2) {
    const int tstep = n / (4 * q);
    for (int b = tid; b < (n >> 2); b += nthreads) {
      const int j = b % q, base = (b / q) * 4 * q + j;
      const Cplx w1 = conj(tw[j * tstep]);
      const Cplx w2 = cmul(w1, w1);
      const Cplx x0 = a[base], x1 = cmul(a[base + q], w2), x2 = a[base + 2 * q], x3 = cmul(a[base + 3 * q], w2);
      const Cplx a0 = cadd(x0, x1), a1 = csub(x0, x1);
      const Cplx a2 = cmul(cadd(x2, x3), w1), a3 = mul_posi(cmul(csub(x2, x3), w1));
      a[base] = cadd(a0, a2);
      a[base + 2 * q] = csub(a0, a2);
      a[base + q] = cadd(a1, a3);
      a[base + 3 * q] = csub(a1, a3);
    }
    __syncthreads();
  }
}


// ---- convolver FFTs (n = 4^m >= 256): padded LDS layout + register radix-16 tail ---------------------------
// Element i lives at pad(i) = i + 2*(i/16) complex slots (16 B of padding per 128 B), so a thread can read or
// write 16 consecutive elements with eight conflict-free ds_*_b128 (lane stride 144 B).  The two innermost
// radix-4 stages (strides 4 and 1) then run on registers instead of taking 4- and 8-way bank conflicts in LDS.
__device__ __forceinline__ int pad(int i) { return i + ((i >> 4) << 1); }

__device__ __forceinline__ void radix4_dif(Cplx& x0, Cplx& x1, Cplx& x2, Cplx& x3, Cplx w1, bool use_tw) {
  const Cplx s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = mul_negi(csub(x1, x3));
  const Cplx y0 = cadd(s02, s13), y1 = csub(s02, s13), y2 = cadd(d02, d13), y3 = csub(d02, d13);
  if (use_tw) {
    const Cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1);
    x0 = y0;
    x1 = cmul(y1, w2);
    x2 = cmul(y2, w1);
    x3 = cmul(y3, w3);
  } else {
    x0 = y0;
    x1 = y1;
    x2 = y2;
    x3 = y3;
  }
}
__device__ __forceinline__ void radix4_dit(Cplx& x0, Cplx& x1, Cplx& x2, Cplx& x3, Cplx w1c, bool use_tw) {
  Cplx t1 = x1, t2 = x2, t3 = x3;
  if (use_tw) {
    const Cplx w2 = cmul(w1c, w1c);
    t1 = cmul(x1, w2);
    t3 = cmul(x3, w2);
  }
  const Cplx a0 = cadd(x0, t1), a1 = csub(x0, t1);
  Cplx a2 = cadd(t2, t3), a3 = csub(t2, t3);
  if (use_tw) {
    a2 = cmul(a2, w1c);
    a3 = cmul(a3, w1c);
  }
  a3 = mul_posi(a3);
  x0 = cadd(a0, a2);
  x2 = csub(a0, a2);
  x1 = cadd(a1, a3);
  x3 = csub(a1, a3);
}

__device__ __forceinline__ void fft_dif_padded(Cplx* a, const Cplx* tw, int n, int tid, int nthreads) {
  for (int L = n; L >= 64; L >>= 2) {  // LDS stages with stride q >= 16
    const int q = L >> 2;
    const int tstep = n / (4 * q);
    for (int b = tid; b < (n >> 2); b += nthreads) {
      const int j = b % q, base = (b / q) * 4 * q + j;
      const int i0 = pad(base), i1 = pad(base + q), i2 = pad(base + 2 * q), i3 = pad(base + 3 * q);
      Cplx x0 = a[i0], x1 = a[i1], x2 = a[i2], x3 = a[i3];
      radix4_dif(x0, x1, x2, x3, tw[j * tstep], true);
      a[i0] = x0;
      a[i1] = x1;
      a[i2] = x2;
      a[i3] = x3;
    }
    __syncthreads();
  }
  // strides 4 and 1 on registers: thread t owns elements [16t, 16t+16)
  for (int t = tid; t < (n >> 4); t += nthreads) {
    float4* row = reinterpret_cast<float4*>(a + 18 * t);
    Cplx x[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 v = row[k];
      x[2 * k] = Cplx{v.x, v.y};
      x[2 * k + 1] = Cplx{v.z, v.w};
    }
    const int tstep = n >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++) radix4_dif(x[j], x[j + 4], x[j + 8], x[j + 12], tw[j * tstep], true);
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dif(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], Cplx{1.f, 0.f}, false);
#pragma unroll
    for (int k = 0; k < 8; k++) row[k] = make_float4(x[2 * k].re, x[2 * k].im, x[2 * k + 1].re, x[2 * k + 1].im);
  }
  __syncthreads();
}
__device__ __forceinline__ void fft_dit_inv_padded(Cplx* a, const Cplx* tw, int n, int tid, int nthreads) {
  for (int t = tid; t < (n >> 4); t += nthreads) {
    float4* row = reinterpret_cast<float4*>(a + 18 * t);
    Cplx x[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 v = row[k];
      x[2 * k] = Cplx{v.x, v.y};
      x[2 * k + 1] = Cplx{v.z, v.w};
    }
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dit(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], Cplx{1.f, 0.f}, false);
    const int tstep = n >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++) radix4_dit(x[j], x[j + 4], x[j + 8], x[j + 12], conj(tw[j * tstep]), true);
#pragma unroll
    for (int k = 0; k < 8; k++) row[k] = make_float4(x[2 * k].re, x[2 * k].im, x[2 * k + 1].re, x[2 * k + 1].im);
  }
  __syncthreads();
  for (int q = 16; q <= (n >> 2); q <<= 2) {
    const int tstep = n / (4 * q);
    for (int b = tid; b < (n >> 2); b += nthreads) {
      const int j = b % q, base = (b / q) * 4 * q + j;
      const int i0 = pad(base), i1 = pad(base + q), i2 = pad(base + 2 * q), i3 = pad(base + 3 * q);
      Cplx x0 = a[i0], x1 = a[i1], x2 = a[i2], x3 = a[i3];
      radix4_dit(x0, x1, x2, x3, conj(tw[j * tstep]), true);
      a[i0] = x0;
      a[i1] = x1;
      a[i2] = x2;
      a[i3] = x3;
    }
    __syncthreads();
  }
}

enum { MODE_FWD = 0, MODE_INV = 1, MODE_IR = 2 };

template <int MODE>
__global__ void conv_fft_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  Cplx* a = reinterpret_cast<Cplx*>(lds_raw);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = d.n, B = d.block;
  const int k = blockIdx.x + (MODE == MODE_IR ? 0 : d.kb0);  // output block (FWD/INV) or partition (IR)
  const int c = blockIdx.y;  // channel
  const uint32_t pair = blockIdx.z;
  if (MODE == MODE_IR) {
    // spectrum of IR partition k of IR channel c: h[kB .. (k+1)B) zero-padded to 2B
    const float* h = d.ir + (uint64_t)c * d.ir_len;
    for (int i = tid; i < n; i += nt) {
      const uint64_t idx = (uint64_t)k * B + i;
      a[pad(i)] = Cplx{(i < B && idx < d.ir_len) ? h[idx] : 0.f, 0.f};
    }
    __syncthreads();
    fft_dif_padded(a, d.tw, n, tid, nt);
    Cplx* dst = const_cast<Cplx*>(d.H) + ((uint64_t)c * d.parts + k) * n;
    for (int i = tid; i < n; i += nt) dst[i] = a[pad(i)];
    return;
  }
  const uint32_t ia = pair * 2, ib = pair * 2 + 1;
  const bool has_b = ib < d.n_inst;
  if (MODE == MODE_FWD) {
    const float* pa = d.in.base + (uint64_t)ia * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
    const float* pb = d.in.base + (uint64_t)(has_b ? ib : ia) * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
    const int64_t f0 = ((int64_t)k - 1) * B;
    // 16 B per lane per stream: 4 consecutive frames of instance a and of instance b -> 4 complex samples
    for (int i4 = tid; i4 < (n >> 2); i4 += nt) {
      const int64_t f = f0 + 4 * (int64_t)i4;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (f >= 0 && (uint64_t)f + 3 < d.in_valid) {
        va = *reinterpret_cast<const float4*>(pa + f);
        if (has_b) vb = *reinterpret_cast<const float4*>(pb + f);
      }
      float4* dst4 = reinterpret_cast<float4*>(a + pad(4 * i4));
      dst4[0] = make_float4(va.x, vb.x, va.y, vb.y);
      dst4[1] = make_float4(va.z, vb.z, va.w, vb.w);
    }
    __syncthreads();
    fft_dif_padded(a, d.tw, n, tid, nt);
    float4* dst = reinterpret_cast<float4*>(d.X + (((uint64_t)pair * d.cin + c) * d.nb + k) * n);
    for (int i = tid; i < (n >> 1); i += nt) dst[i] = *reinterpret_cast<const float4*>(a + pad(2 * i));
  } else {
    const float4* src = reinterpret_cast<const float4*>(d.Y + (((uint64_t)pair * d.cout + c) * d.nb + k) * n);
    for (int i = tid; i < (n >> 1); i += nt) *reinterpret_cast<float4*>(a + pad(2 * i)) = src[i];
    __syncthreads();
    fft_dit_inv_padded(a, d.tw, n, tid, nt);
    float* pa = d.out.base + (uint64_t)ia * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
    float* pb = d.out.base + (uint64_t)(has_b ? ib : ia) * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
    const float scale = 1.f / (float)n;
    for (int i4 = tid; i4 < (B >> 2); i4 += nt) {
      const uint64_t f = (uint64_t)k * B + 4 * (uint64_t)i4;
      if (f + 3 < d.frames) {
        // overlap-save: the last B samples are the linear convolution; re -> instance a, im -> instance b
        const float4 p0 = reinterpret_cast<const float4*>(a + pad(B + 4 * i4))[0];
        const float4 p1 = reinterpret_cast<const float4*>(a + pad(B + 4 * i4))[1];
        *reinterpret_cast<float4*>(pa + f) = make_float4(p0.x * scale, p0.z * scale, p1.x * scale, p1.z * scale);
        if (has_b) *reinterpret_cast<float4*>(pb + f) = make_float4(p0.y * scale, p0.w * scale, p1.y * scale, p1.w * scale);
      }
    }
  }
}

// ---- N = 16384 (B = 8192, long IRs): persistent, software-pipelined FFT workgroups ------------------------------
// One 16384-point complex FFT fills 144 KB of LDS, so only ONE workgroup fits on a CU and the plain kernel above
// runs load -> FFT -> store strictly one after the other: the memory system idles while the CU computes and vice
// versa (it reached ~2.5 TB/s).  Here a workgroup walks a run of consecutive blocks of one (pair, channel):
//   * the NEXT block's input is requested into registers before the current FFT starts, so its HBM latency is
//     hidden behind the butterflies; the spectrum / output stores of the current block are fire-and-forget;
//   * forward: consecutive overlap-save windows share half their samples — the shared half stays in registers and
//     the input is read once instead of twice;
//   * every twiddle a thread needs (the forward kernel re-reads the eight q = 4096 ones per block) is
//     loaded once per workgroup and lives in registers: no global loads inside the butterfly stages.
// The arithmetic and its order are those of fft_dif_padded / fft_dit_inv_padded for n = 16384 (which butterfly a thread
// executes does not change any value): results are bit-identical to the plain kernel (WAA_CONV_FFT_PLAIN=1 selects it; tests compare the two).
// Cache policy of the streamed buffers (compile-time, -DWAA_CONV_POL=n for an A/B library, tools/ab_lib.py): bit 0 =
// non-temporal loads, bit 1 = non-temporal stores.  Measured on T1 (three alternations of all four builds on one box): no
// policy moves any of the three kernels outside the run-to-run spread (fwd 2.71-2.86, product 3.08-3.48, inv 3.09-3.26 ms),
// so the plain forms stay.  (The streaming biquad does gain 1.4 % from non-temporal loads, waa_biquad_stream.hip.)
#ifndef WAA_CONV_POL
#define WAA_CONV_POL 0
#endif
template <class T>
__device__ __forceinline__ T ld_pol(const T* p) {
  if constexpr (WAA_CONV_POL & 1)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}
template <class T>
__device__ __forceinline__ void st_pol(T* p, T v) {
  if constexpr (WAA_CONV_POL & 2)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
constexpr int PIPE_N = 16384, PIPE_NT = 512, PIPE_B = PIPE_N / 2;
constexpr int PIPE_IT = PIPE_N / 4 / PIPE_NT;     // radix-4 butterflies per thread in the q = 16 stage (8)
constexpr int PIPE_ROWS = PIPE_N / 16 / PIPE_NT;  // 16-element register rows per thread in the tail (2)
// Two radix-4 stages per LDS round trip: the 16 elements e0 + m * stride (m = 0..15) are closed under the stage
// with q = 4 * stride (butterflies over m, m+4, m+8, m+12) and under the stage with q = stride (butterflies over
// 4g .. 4g+3), so a thread runs both on registers — the values and their order of operations are exactly those
// of two separate passes of fft_dif_padded, one barrier and 256 KB of LDS traffic less per pair of stages.
// Stages (q = 4096, 1024) and (256, 64) are paired; q = 16 stays a radix-4 pass; strides 4 and 1 are the register
// tail on 16 contiguous elements.  Four LDS round trips per FFT instead of six.
struct PipeTw {
                  // pass 1, q = 4096: tw[j' + r * 1024], j' = tid + set * 512: parked in LDS (pipe_park_tw0)
  Cplx b0[2];     // pass 1, q = 1024: tw[4 j']
  Cplx a1[4];     // pass 2, q = 256: tw[16 (j'' + r * 64)], j'' = tid % 64 (the same for both sets)
  Cplx b1;        // pass 2, q = 64: tw[64 j'']
  Cplx s16;       // q = 16: tw[256 (tid % 16)]
};
// register tail: exp(-2 pi i j / 16), j = 0..3 — the table's entries tw[j * 1024] ((float)cos / (float)sin of the f64 angle)
// as literals: uniform constants cost no vector registers (the forward kernel was 21 registers over its budget)
#define PIPE_T(j)                                                                                                          \
  ((j) == 0 ? Cplx{0x1p+0f, -0x0p+0f}                                                                                       \
            : (j) == 1 ? Cplx{0x1.d906bcp-1f, -0x1.87de2ap-2f}                                                             \
                       : (j) == 2 ? Cplx{0x1.6a09e6p-1f, -0x1.6a09e6p-1f} : Cplx{0x1.87de2ap-2f, -0x1.d906bcp-1f})
template <bool KEEP0>
__device__ __forceinline__ PipeTw pipe_twiddles(const Cplx* tw, int tid) {
  PipeTw r;
#pragma unroll
  for (int set = 0; set < 2; set++) {
    const int jp = tid + set * PIPE_NT;
    r.b0[set] = tw[jp * 4];
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; q4++) r.a1[q4] = tw[((tid % 64) + q4 * 64) * 16];
  r.b1 = tw[(tid % 64) * 64];
  r.s16 = tw[(tid % 16) * 256];
  return r;
}
// (keeps the LDS address arithmetic of a pass from being hoisted above the previous pass, where it would only
// occupy registers: everything below is pure arithmetic on the thread index)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <bool INVERSE>
__device__ __forceinline__ void pipe_radix16(Cplx* a, int e0, int stride, const Cplx (&twa)[4], Cplx twb) {
  Cplx x[16];
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = a[pad(e0 + m * stride)];
  if (!INVERSE) {
#pragma unroll
    for (int r = 0; r < 4; r++) radix4_dif(x[r], x[r + 4], x[r + 8], x[r + 12], twa[r], true);
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dif(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], twb, true);
  } else {
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dit(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], conj(twb), true);
#pragma unroll
    for (int r = 0; r < 4; r++) radix4_dit(x[r], x[r + 4], x[r + 8], x[r + 12], conj(twa[r]), true);
  }
#pragma unroll
  for (int m = 0; m < 16; m++) a[pad(e0 + m * stride)] = x[m];
}
// KEEP0: the q = 4096 twiddles come from w.a0 or, when `tws` is given, from LDS — set 0 from the 16 KB behind the
// transform buffer (tws[r * 512 + tid]), set 1 from the two padding slots of rows 2 tid and 2 tid + 1 of the buffer itself
// (pad() never maps an element there and the register tail touches 16 of a row's 18 slots): 2 x 16 KB that the
// 16384-point transform leaves unused, exactly 8 twiddles for each of the 512 threads
template <bool INVERSE, bool KEEP0>
__device__ __forceinline__ void pipe_pass1(Cplx* a, const PipeTw& w, const Cplx* twg, int tid, const Cplx* tws = nullptr) {
#pragma unroll
  for (int set = 0; set < 2; set++) {
    const int jp = opaque(tid) + set * PIPE_NT;
    Cplx twa[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
      twa[r] = !KEEP0 ? twg[jp + r * 1024]
                      : set == 0 ? tws[r * PIPE_NT + opaque(tid)]
                                 : a[18 * (2 * opaque(tid) + (r >> 1)) + 16 + (r & 1)];  // (the rows' two padding slots)
    pipe_radix16<INVERSE>(a, jp, 1024, twa, w.b0[set]);
    __builtin_amdgcn_sched_barrier(0);  // one 16-element set at a time: the registers hold the prefetched block
  }
}
template <bool INVERSE>
__device__ __forceinline__ void pipe_pass2(Cplx* a, const PipeTw& w, int tid) {
#pragma unroll
  for (int set = 0; set < 2; set++) {
    const int t2 = opaque(tid) + set * PIPE_NT;
    pipe_radix16<INVERSE>(a, (t2 / 64) * 1024 + (t2 % 64), 64, w.a1, w.b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the q = 16 stage: eight radix-4 butterflies per thread (index arithmetic of fft_dif_padded)
template <bool INVERSE>
__device__ __forceinline__ void pipe_stage16(Cplx* a, const PipeTw& w, int tid) {
  constexpr int q = 16;
#pragma unroll
  for (int it = 0; it < PIPE_IT; it++) {
    const int b = tid + it * PIPE_NT;
    const int j = b % q, base = (b / q) * 4 * q + j;
    const int i0 = pad(base), i1 = pad(base + q), i2 = pad(base + 2 * q), i3 = pad(base + 3 * q);
    Cplx x0 = a[i0], x1 = a[i1], x2 = a[i2], x3 = a[i3];
    if (INVERSE)
      radix4_dit(x0, x1, x2, x3, conj(w.s16), true);
    else
      radix4_dif(x0, x1, x2, x3, w.s16, true);
    a[i0] = x0;
    a[i1] = x1;
    a[i2] = x2;
    a[i3] = x3;
  }
}
// The q = 4096 twiddles of a block (8 per thread, L2 hits).  They are requested BEFORE the next block's input: the
// hardware counts returning loads in order, so the wait for the twiddles inside pass 1 then leaves the younger input
// prefetch in flight.  Requested inside pass 1 (behind the prefetch) they fenced it: every block waited for its
// successor's input a few hundred cycles after asking for it.
// The q = 4096 twiddles (8 per thread) of the persistent kernels are parked in LDS once per workgroup.  Re-read from
// the table in every block (round 1: L2 hits, "free") they were loads that sit BEHIND the next block's input prefetch in
// the in-order return queue: the wait for them inside pass 1 of the forward transform was a wait for the prefetch, a few
// hundred cycles after it had been issued — the prefetch overlapped nothing.
// Workgroup barrier for LDS hand-offs.  __syncthreads() is a workgroup-scope fence plus s_barrier, and the release half
// of that fence waits for every outstanding GLOBAL store (vmcnt(0)): inside these persistent loops each barrier behind
// a block's 128 KB of spectrum / output stores stalled until they had reached memory, i.e. the stores overlapped
// nothing.  Here: this wave's LDS traffic has completed (DS operations of a wave finish in order), then the barrier.
__device__ __forceinline__ void pipe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void pipe_park_tw0(Cplx* a, Cplx* tws, const Cplx* twg, int tid) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    tws[r * PIPE_NT + tid] = twg[tid + r * 1024];
    a[18 * (2 * tid + (r >> 1)) + 16 + (r & 1)] = twg[tid + PIPE_NT + r * 1024];
  }
}
__device__ __forceinline__ void pipe_fft_dif(Cplx* a, const PipeTw& w, const Cplx* twg, int tid, const Cplx* tws) {
  pipe_pass1<false, true>(a, w, twg, opaque(tid), tws);
  pipe_barrier();
  pipe_pass2<false>(a, w, opaque(tid));
  pipe_barrier();
  pipe_stage16<false>(a, w, opaque(tid));
  pipe_barrier();
#pragma unroll
  for (int rr = 0; rr < PIPE_ROWS; rr++) {
    float4* row = reinterpret_cast<float4*>(a + 18 * (opaque(tid) + rr * PIPE_NT));  // elements [16t, 16t+16)
    Cplx x[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 v = row[k];
      x[2 * k] = Cplx{v.x, v.y};
      x[2 * k + 1] = Cplx{v.z, v.w};
    }
#pragma unroll
    for (int j = 0; j < 4; j++) radix4_dif(x[j], x[j + 4], x[j + 8], x[j + 12], PIPE_T(j), true);
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dif(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], Cplx{1.f, 0.f}, false);
#pragma unroll
    for (int k = 0; k < 8; k++) row[k] = make_float4(x[2 * k].re, x[2 * k].im, x[2 * k + 1].re, x[2 * k + 1].im);
  }
  pipe_barrier();
}
__device__ __forceinline__ void pipe_fft_dit_inv(Cplx* a, const PipeTw& w, const Cplx* twg, int tid, const Cplx* tws) {
#pragma unroll
  for (int rr = 0; rr < PIPE_ROWS; rr++) {
    float4* row = reinterpret_cast<float4*>(a + 18 * (opaque(tid) + rr * PIPE_NT));
    Cplx x[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 v = row[k];
      x[2 * k] = Cplx{v.x, v.y};
      x[2 * k + 1] = Cplx{v.z, v.w};
    }
#pragma unroll
    for (int g = 0; g < 4; g++) radix4_dit(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], Cplx{1.f, 0.f}, false);
#pragma unroll
    for (int j = 0; j < 4; j++) radix4_dit(x[j], x[j + 4], x[j + 8], x[j + 12], conj(PIPE_T(j)), true);
#pragma unroll
    for (int k = 0; k < 8; k++) row[k] = make_float4(x[2 * k].re, x[2 * k].im, x[2 * k + 1].re, x[2 * k + 1].im);
  }
  pipe_barrier();
  pipe_stage16<true>(a, w, opaque(tid));
  pipe_barrier();
  pipe_pass2<true>(a, w, opaque(tid));
  pipe_barrier();
  pipe_pass1<true, true>(a, w, twg, opaque(tid), tws);
  pipe_barrier();
}

constexpr int PIPE_H = PIPE_B / 4 / PIPE_NT;  // float4 groups per thread in half a window (4)
constexpr int PIPE_S = PIPE_N / 2 / PIPE_NT;  // float4 (two complex bins) per thread in a spectrum (16)
// half a window = B frames of both instances: PIPE_H x (16 B of a, 16 B of b) per thread
// The loads are UNCONDITIONAL (a frame group outside [0, frames) reads frame 0 instead and is zeroed when it is staged):
// with the load inside an `if` the compiler merged the loaded value and the zero default in copy instructions right
// behind the load — i.e. waited for every prefetch the moment it was issued, and the "prefetch" overlapped nothing.
__device__ __forceinline__ void pipe_load_half(const float* pa, const float* pb, bool has_b, int64_t f0, uint64_t frames, int tid,
                                               f4v (&va)[PIPE_H], f4v (&vb)[PIPE_H]) {
  (void)has_b;  // (pb == pa without a second instance)
#pragma unroll
  for (int r = 0; r < PIPE_H; r++) {
    const int64_t f = f0 + 4 * (int64_t)(tid + r * PIPE_NT);
    const bool ok = f >= 0 && (uint64_t)f + 3 < frames;
    const int64_t fc = ok ? f : 0;
    va[r] = ld_pol(reinterpret_cast<const f4v*>(pa + fc));
    vb[r] = ld_pol(reinterpret_cast<const f4v*>(pb + fc));
  }
}
__device__ __forceinline__ void pipe_stage_half(Cplx* a, int e0, int tid, const f4v (&va)[PIPE_H], const f4v (&vb)[PIPE_H], bool has_b,
                                                int64_t f0, uint64_t frames) {
#pragma unroll
  for (int r = 0; r < PIPE_H; r++) {
    const int64_t f = f0 + 4 * (int64_t)(tid + r * PIPE_NT);
    const bool ok = f >= 0 && (uint64_t)f + 3 < frames;
    const f4v z = {0.f, 0.f, 0.f, 0.f};
    const f4v xa = ok ? va[r] : z, xb = ok && has_b ? vb[r] : z;
    f4v* dst4 = reinterpret_cast<f4v*>(a + pad(e0 + 4 * (tid + r * PIPE_NT)));
    dst4[0] = f4v{xa.x, xb.x, xa.y, xb.y};
    dst4[1] = f4v{xa.z, xb.z, xa.w, xb.w};
  }
}
// Forces the wait for a prefetch to sit HERE.  gfx9 counts loads and stores in one counter and the compiler has to
// assume they complete out of order, so a wait for loads that is placed behind newer stores becomes "wait for
// everything": placed in front of the stores of a block, it costs nothing (the loads are a whole FFT old) and the
// stores then drain behind the next block's work.
template <int N>
__device__ __forceinline__ void pipe_settle(f4v (&v)[N]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < N; r++) asm volatile("" : "+v"(v[r])::"memory");
  __builtin_amdgcn_sched_barrier(0);
}

// DBG (WAA_CONV_PIPE_DEBUG, measurement aid, results wrong by construction): 1 = no global stores, 2 = no global loads
template <int MODE, int DBG = 0>
__global__ __launch_bounds__(PIPE_NT) void conv_fft_pipe_kernel(const ConvDesc d, int blocks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  Cplx* a = reinterpret_cast<Cplx*>(lds_raw);
  const int tid = threadIdx.x;
  const int c = blockIdx.y;
  const uint32_t pair = blockIdx.z;
  const int k0 = d.kb0 + blockIdx.x * blocks_per_wg;
  const int k1 = k0 + blocks_per_wg < d.kb1 ? k0 + blocks_per_wg : d.kb1;
  if (k0 >= k1) return;
  const PipeTw w = pipe_twiddles<false>(d.tw, tid);
  Cplx* tws = a + (PIPE_N + PIPE_N / 8);  // behind the padded transform buffer
  pipe_park_tw0(a, tws, d.tw, tid);
  const uint32_t ia = pair * 2, ib = pair * 2 + 1;
  const bool has_b = ib < d.n_inst;
  if (MODE == MODE_FWD) {
    const float* pa = d.in.base + (uint64_t)ia * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
    const float* pb = d.in.base + (uint64_t)(has_b ? ib : ia) * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
    f4v oa[PIPE_H], ob[PIPE_H], na[PIPE_H], nb[PIPE_H];
    pipe_load_half(pa, pb, has_b, ((int64_t)k0 - 1) * PIPE_B, d.in_valid, tid, oa, ob);
    pipe_load_half(pa, pb, has_b, (int64_t)k0 * PIPE_B, d.in_valid, tid, na, nb);
    // (no load may be pending on loop entry: the wait counts the compiler computes for the loop header are the merge of
    // this path and the back edge, where the previous block's stores are in flight — a count that is right for loads
    // pending from here is a wait for those stores on every later iteration)
    pipe_settle(oa);
    pipe_settle(ob);
    pipe_settle(na);
    pipe_settle(nb);
    for (int k = k0; k < k1; k++) {
      // (opaque copy of the thread index: otherwise every LDS address of all five stages is loop-invariant, gets
      // hoisted out of the block loop and ~160 address registers stay live across it)
      int tid_k = tid;
      asm volatile("" : "+v"(tid_k));
      pipe_stage_half(a, 0, tid_k, oa, ob, has_b, ((int64_t)k - 1) * PIPE_B, d.in_valid);  // window [(k-1)B, (k+1)B)
      pipe_stage_half(a, PIPE_B, tid_k, na, nb, has_b, (int64_t)k * PIPE_B, d.in_valid);
      pipe_barrier();
#pragma unroll
      for (int r = 0; r < PIPE_H; r++) {
        oa[r] = na[r];
        ob[r] = nb[r];
      }
      // the next block's new half: in flight during the FFT below (all-zero past the end of the stream)
      if (DBG != 2)
        pipe_load_half(pa, pb, has_b, k + 1 < k1 ? ((int64_t)k + 1) * PIPE_B : (int64_t)d.in_valid, d.in_valid, tid_k, na, nb);
      pipe_fft_dif(a, w, d.tw, tid_k, tws);
      pipe_settle(na);
      pipe_settle(nb);
      f4v* dst = reinterpret_cast<f4v*>(d.X + (((uint64_t)pair * d.cin + c) * d.nb + k) * PIPE_N);
#pragma unroll
      for (int r = 0; r < PIPE_S; r++) {
        const int i = tid_k + r * PIPE_NT;
        const f4v v = *reinterpret_cast<const f4v*>(a + pad(2 * i));
        if (DBG == 1)
          asm volatile("" ::"v"(v));
        else
          st_pol(dst + i, v);
      }
      pipe_barrier();  // LDS is rewritten by the next window
    }
  } else {
    float* pa = d.out.base + (uint64_t)ia * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
    float* pb = d.out.base + (uint64_t)(has_b ? ib : ia) * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
    const float scale = 1.f / (float)PIPE_N;
    const f4v* ybase = reinterpret_cast<const f4v*>(d.Y + ((uint64_t)pair * d.cout + c) * d.nb * PIPE_N);
    f4v y[PIPE_S];
#pragma unroll
    for (int r = 0; r < PIPE_S; r++) y[r] = ld_pol(ybase + (uint64_t)k0 * (PIPE_N / 2) + tid + r * PIPE_NT);
    pipe_settle(y);  // (as in the forward kernel)
    for (int k = k0; k < k1; k++) {
      int tid_k = tid;
      asm volatile("" : "+v"(tid_k));
#pragma unroll
      for (int r = 0; r < PIPE_S; r++) *reinterpret_cast<f4v*>(a + pad(2 * (tid_k + r * PIPE_NT))) = y[r];
      pipe_barrier();
      // the next spectrum: in flight during the inverse FFT below (the last block re-reads itself: harmless)
      const int kn = k + 1 < k1 ? k + 1 : k;
#pragma unroll
      for (int r = 0; r < PIPE_S; r++)
        if (DBG != 2) y[r] = ld_pol(ybase + (uint64_t)kn * (PIPE_N / 2) + tid_k + r * PIPE_NT);
      pipe_fft_dit_inv(a, w, d.tw, tid_k, tws);
      pipe_settle(y);
#pragma unroll
      for (int r = 0; r < PIPE_H; r++) {
        const int i4 = tid_k + r * PIPE_NT;
        const uint64_t f = (uint64_t)k * PIPE_B + 4 * (uint64_t)i4;
        if (f + 3 < d.frames) {
          // overlap-save: the last B samples are the linear convolution; re -> instance a, im -> instance b
          const f4v p0 = reinterpret_cast<const f4v*>(a + pad(PIPE_B + 4 * i4))[0];
          const f4v p1 = reinterpret_cast<const f4v*>(a + pad(PIPE_B + 4 * i4))[1];
          const f4v va = f4v{p0.x * scale, p0.z * scale, p1.x * scale, p1.z * scale};
          const f4v vb = f4v{p0.y * scale, p0.w * scale, p1.y * scale, p1.w * scale};
          if (DBG == 1) {
            asm volatile("" ::"v"(va), "v"(vb));
          } else {
            st_pol(reinterpret_cast<f4v*>(pa + f), va);
            if (has_b) st_pol(reinterpret_cast<f4v*>(pb + f), vb);
          }
        }
      }
      pipe_barrier();
    }
  }
}

// Y_k = sum_terms sum_p H_p X_{k-p}, KT output blocks per register tile, partitions in chunks of PC.
// A thread owns one spectral position and walks ALL k-tiles of its (pair, output channel) in order, so the
// (PC - 1) input spectra a tile shares with its predecessor were read by the same CU a moment ago (L2 / MALL
// hits instead of HBM re-reads).
template <int KT, int PC>
__global__ __launch_bounds__(256) void conv_mac_kernel(const ConvDesc d) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const uint32_t pair = blockIdx.y / (uint32_t)d.cout;
  const int co = (int)(blockIdx.y % (uint32_t)d.cout);
  const int n = d.n, nb = d.nb, P = d.parts;
  Cplx* Yc = d.Y + ((uint64_t)pair * d.cout + co) * nb * n + pos;
  for (int k0 = d.kb0; k0 < d.kb1; k0 += KT) {
    Cplx acc[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) acc[i] = Cplx{0.f, 0.f};
    for (int t = 0; t < d.n_terms; t++) {
      if (d.terms[t].out_ch != co) continue;
      const Cplx* Hc = d.H + (uint64_t)d.terms[t].ir_ch * P * n + pos;
      const Cplx* Xc = d.X + ((uint64_t)pair * d.cin + d.terms[t].in_ch) * nb * n + pos;
      for (int pc0 = 0; pc0 < P; pc0 += PC) {
        Cplx h[PC];
#pragma unroll
        for (int i = 0; i < PC; i++) h[i] = (pc0 + i < P) ? Hc[(uint64_t)(pc0 + i) * n] : Cplx{0.f, 0.f};
#pragma clang loop unroll(full)
        for (int jj = 0; jj < KT + PC - 1; jj++) {
          const int j = k0 - pc0 - (PC - 1) + jj;
          Cplx x = Cplx{0.f, 0.f};
          if (j >= 0 && j < nb) x = Xc[(uint64_t)j * n];
#pragma clang loop unroll(full)
          for (int i = 0; i < KT; i++) {
            const int pl = i + (PC - 1) - jj;  // local partition index, compile-time after unrolling
            if (pl >= 0 && pl < PC) {
              // (two v_pk_fma_f32 per complex multiply-add instead of these four scalar FMAs were measured: 5.9 ms
              // against 4.5 ms — the swapped (im, re) operand costs extra moves and the packed op is not faster)
              acc[i].re = __builtin_fmaf(h[pl].re, x.re, acc[i].re);
              acc[i].re = __builtin_fmaf(-h[pl].im, x.im, acc[i].re);
              acc[i].im = __builtin_fmaf(h[pl].re, x.im, acc[i].im);
              acc[i].im = __builtin_fmaf(h[pl].im, x.re, acc[i].im);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < KT; i++)
      if (k0 + i < d.kb1) Yc[(uint64_t)(k0 + i) * n] = acc[i];
  }
}

// The same product with a SLIDING register window: consecutive k-tiles of one spectral position share PC - 1 input
// spectra; conv_mac_kernel re-reads them (2.2x the X bytes from HBM, profiles/r02a_t1_fetch.txt), here they stay in
// registers, the IR column h[0..PC) is loaded once per thread, and every X value is read exactly once.  One term per
// output channel and P <= PC (every routing but the true-stereo 4-channel IR; the launcher picks).
typedef float c2v __attribute__((ext_vector_type(2)));  // one complex value in a 64-bit register pair: (re, im)
// acc += h * x as two packed FMAs (v_pk_fma_f32: two f32 FMAs per lane and issue slot).  The operand selects do what the
// compiler cannot express from C: (re, im) += h.re * (x.re, x.im), then (re, im) += h.im * (-x.im, x.re) with x's halves
// SWAPPED by op_sel and the sign on the low half by neg_lo — no moves.  Same four FMAs in the same order as
//   re = fma(h.re, x.re, re); re = fma(-h.im, x.im, re); im = fma(h.re, x.im, im); im = fma(h.im, x.re, im)
// hence bit-identical to the scalar form (conv_mac_kernel keeps it).  Same-box A/B against the four scalar FMAs on T1 / C3:
// 3.12-3.35 ms against 3.19-3.32 ms — the kernel waits for memory (5 TB/s), not for the VALU; the packed form halves the
// VALU instructions (704 instead of 1408 per k-tile) and a third of the registers (132 against 206) for the same time.
__device__ __forceinline__ void cmac_pk(c2v& acc, const c2v h, const c2v x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(h), "v"(x));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(h), "v"(x));
}
template <int KT, int PC, bool PREFETCH = true>
__global__ __launch_bounds__(256) void conv_mac_win_kernel(const ConvDesc d) {
  // Workgroup order.  Every thread reads its position's IR column (P values) once; all (pair, channel) workgroups of one
  // position block read the same 45 KB.  In grid order (position block fastest) a CU's neighbours in time are OTHER position
  // blocks, and by the time the same block comes round again (one workgroup lifetime, ~100 us, 60 MB of spectra through the
  // XCD's L2 later) the column is gone: half of the 2.95 GB of column reads were trips to memory (9.5 GB fetched for 7.9 GB
  // of X, profiles/r03r_t1_fetch.txt).  Dispatch order id -> XCD id % 8; with  id = x_lo + 8 * (y + gridDim.y * x_hi)
  // an XCD works through ALL (pair, channel) of one position block before the next: its IR columns stay in L2, and the
  // eight XCDs still read eight adjacent position blocks (16 KB contiguous per spectrum) at a time.
  uint32_t bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.x % 8 == 0 && !d.mac_grid_order) {
    const uint32_t id = blockIdx.x + gridDim.x * blockIdx.y;
    const uint32_t rest = id >> 3;
    by = rest % gridDim.y;
    bx = (rest / gridDim.y) * 8 + (id & 7);
  }
  const int pos = bx * 256 + threadIdx.x;
  const uint32_t pair = by / (uint32_t)d.cout;
  const int co = (int)(by % (uint32_t)d.cout);
  const int n = d.n, nb = d.nb, P = d.parts;
  int term = 0;
  for (int t = 0; t < d.n_terms; t++)
    if (d.terms[t].out_ch == co) term = t;
  c2v* Yc = reinterpret_cast<c2v*>(d.Y + ((uint64_t)pair * d.cout + co) * nb * n + pos);
  const c2v* Hc = reinterpret_cast<const c2v*>(d.H + (uint64_t)d.terms[term].ir_ch * P * n + pos);
  const c2v* Xc = reinterpret_cast<const c2v*>(d.X + ((uint64_t)pair * d.cin + d.terms[term].in_ch) * nb * n + pos);
  const c2v zero = {0.f, 0.f};
  // Loads are unconditional (clamped index, the zero selected afterwards): behind `i < P ? load : 0` the compiler
  // branched around every load and waited for it on the spot — 22 memory latencies in a row before the first product
  // (about half of a workgroup's life), and again one full latency per k-tile before any of its arithmetic.
  c2v h[PC];
#pragma unroll
  for (int i = 0; i < PC; i++) h[i] = Hc[(uint64_t)(i < P ? i : 0) * n];
#pragma unroll
  for (int i = 0; i < PC; i++)
    if (i >= P) h[i] = zero;
  c2v win[PC - 1];  // X_{k0 - (PC-1)} .. X_{k0 - 1}
#pragma unroll
  for (int i = 0; i < PC - 1; i++) win[i] = zero;
  if (d.kb0 > 0) {  // a later block range of a block-scheduled loop: the window's history comes back from X
#pragma unroll
    for (int i = 0; i < PC - 1; i++) {
      const int j = d.kb0 - (PC - 1) + i;
      win[i] = Xc[(uint64_t)(j >= 0 ? j : 0) * n];
      if (j < 0) win[i] = zero;
    }
  }
  // The k-tiles are software-pipelined: tile k + 1's sixteen spectra are requested before tile k's 352 complex
  // multiply-adds start and are settled right in front of tile k's stores (loads and stores share one counter: a wait
  // placed behind the stores would wait for them too) — the memory pipe no longer idles while a wave computes.
  c2v xq[KT];
  const int kend = d.kb1;
#pragma unroll
  for (int i = 0; i < KT; i++) xq[i] = ld_pol(Xc + (uint64_t)(d.kb0 + i < kend ? d.kb0 + i : kend - 1) * n);
  for (int k0 = d.kb0; k0 < kend; k0 += KT) {
    c2v xn[KT];  // X_{k0} .. X_{k0 + KT - 1}
#pragma unroll
    for (int i = 0; i < KT; i++) xn[i] = k0 + i >= kend ? zero : xq[i];
    if (PREFETCH) {
      const int kn = k0 + KT;  // (past the end: the last block again — an L2 hit nobody uses)
#pragma unroll
      for (int i = 0; i < KT; i++) xq[i] = ld_pol(Xc + (uint64_t)(kn + i < kend ? kn + i : kend - 1) * n);
    }
    c2v acc[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) acc[i] = zero;
#pragma clang loop unroll(full)
    for (int jj = 0; jj < KT + PC - 1; jj++) {
      const c2v x = jj < PC - 1 ? win[jj < PC - 1 ? jj : 0] : xn[jj >= PC - 1 ? jj - (PC - 1) : 0];
#pragma clang loop unroll(full)
      for (int i = 0; i < KT; i++) {
        const int pl = i + (PC - 1) - jj;  // partition index, compile-time after unrolling
        if (pl >= 0 && pl < PC) {
          cmac_pk(acc[i], h[pl], x);
        }
      }
    }
    if (PREFETCH) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < KT; i++) asm volatile("" : "+v"(xq[i])::"memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < KT; i++)
      if (k0 + i < kend) st_pol(Yc + (uint64_t)(k0 + i) * n, acc[i]);
    if (!PREFETCH) {
      const int kn = k0 + KT;
#pragma unroll
      for (int i = 0; i < KT; i++) xq[i] = ld_pol(Xc + (uint64_t)(kn + i < kend ? kn + i : kend - 1) * n);
    }
    // slide the window by KT blocks
#pragma unroll
    for (int w = 0; w < PC - 1; w++) {
      const int src = w + KT;  // index into the concatenation [win | xn]
      win[w] = src < PC - 1 ? win[src < PC - 1 ? src : 0] : xn[src >= PC - 1 ? src - (PC - 1) : 0];
    }
  }
}

// Short impulse responses (<= DIRECT_MAX_TAPS after trimming): direct time-domain FIR with f64 accumulation.
// This is the linear convolution itself, so it is at least as close to the exact result as the reference's
// f32 FFT convolver (the reference's own delta-IR tests ask for 1e-7).  Per term the f64 sum is rounded to f32
// and the terms are added in f32, like `o_left += o_2` in convolver.rs:430-441.
constexpr int DIRECT_TILE = 1024;
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvDesc d) {
  __shared__ float xs[2][DIRECT_TILE + DIRECT_MAX_TAPS];
  __shared__ float hs[4][DIRECT_MAX_TAPS];
  const int tid = threadIdx.x;
  const uint64_t f0 = ((uint64_t)blockIdx.x + (uint64_t)d.kb0) * DIRECT_TILE;  // (kb0 / kb1: 1024-frame pieces here)
  const int co = blockIdx.y;
  const uint32_t inst = blockIdx.z;
  const int taps = (int)d.ir_len;
  for (int c = 0; c < d.cin; c++) {
    const float* p = d.in.base + (uint64_t)inst * d.in.inst_stride + (uint64_t)c * d.in.ch_stride;
    for (int i = tid; i < DIRECT_TILE + DIRECT_MAX_TAPS; i += 256) {
      const int64_t f = (int64_t)f0 - DIRECT_MAX_TAPS + i;
      xs[c][i] = (f >= 0 && (uint64_t)f < d.in_valid) ? p[f] : 0.f;
    }
  }
  for (int t = 0; t < d.n_terms; t++)
    for (int i = tid; i < DIRECT_MAX_TAPS; i += 256) hs[t][i] = i < taps ? d.ir[(uint64_t)d.terms[t].ir_ch * d.ir_len + i] : 0.f;
  __syncthreads();
  float* o = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)co * d.out.ch_stride;
  for (int i = tid; i < DIRECT_TILE; i += 256) {
    float sum = 0.f;
    bool first = true;
    for (int t = 0; t < d.n_terms; t++) {
      if (d.terms[t].out_ch != co) continue;
      const float* x = xs[d.terms[t].in_ch] + DIRECT_MAX_TAPS + i;
      double acc = 0.;
      for (int k = 0; k < taps; k++) acc = __builtin_fma((double)hs[t][k], (double)x[-k], acc);
      sum = first ? (float)acc : sum + (float)acc;
      first = false;
    }
    const uint64_t f = f0 + i;
    if (f < d.frames) o[f] = sum;
  }
}

// AnalyserNode control side (analysis.rs:261-401): most recent fft_size frames of the mono down-mix, Blackman window, real
// FFT via a packed complex FFT of half the size, |X[k]| / N, smoothing against the previous (zero) spectrum, dB and byte
// conversion.  One workgroup per INSTANCE: a pull for the whole batch is one launch (BASELINE config 4: one pull per
// context; as 4096 single launches with a stream sync each the pulls cost more than the render).
__global__ void analyser_kernel(const AnalyserDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  Cplx* a = reinterpret_cast<Cplx*>(lds_raw);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = d.fft_size, M = N >> 1;
  const uint32_t inst = blockIdx.x;
  const float* p0 = d.sig.base + (uint64_t)inst * d.sig.inst_stride;
  float* time_out = d.time_out + (uint64_t)inst * N;
  const int64_t first = (int64_t)d.frames_written - N;  // ring_buffer.read: the last N frames written
  for (int i = tid; i < N; i += nt) {
    const int64_t f = first + i;
    float v = 0.f;
    if (f >= 0) {
      // mono down-mix of the analyser input (analyser.rs:277-280, quantum.rs:387-397)
      const uint64_t cs = d.sig.ch_stride;
      int nch = d.sig.nch;
      if (d.code) {  // (dynamic-count plans: the count of this frame's quantum)
        const uint32_t c = d.code[(uint64_t)inst * d.code_stride + (uint64_t)(f >> 7)];
        nch = (c & 0x80u) ? 0 : (int)(c & 63u);
      }
      switch (nch) {  // quantum.rs:387-429 speaker down-mix to mono
        case 0: v = 0.f; break;  // (a silent quantum)
        case 1: v = p0[f]; break;
        case 2: v = 0.5f * (p0[f] + p0[cs + f]); break;
        case 4: v = 0.25f * (p0[f] + p0[cs + f] + p0[2 * cs + f] + p0[3 * cs + f]); break;
        case 6:
          v = __builtin_fmaf(0.70710678118654752440f, p0[f] + p0[cs + f],
                             __builtin_fmaf(0.5f, p0[4 * cs + f] + p0[5 * cs + f], p0[2 * cs + f]));
          break;
        default: v = p0[f]; break;  // other layouts: truncate
      }
    }
    time_out[i] = v;
    const float wv = v * d.window[i];
    reinterpret_cast<float*>(a)[i] = wv;  // z[n] = x[2n] + i x[2n+1]
  }
  __syncthreads();
  fft_dif(a, d.tw, M, tid, nt);
  const int lg = 31 - __builtin_clz(M);
  const float nf = 1.f / (float)N;
  const float tau = d.smoothing;
  const float bscale = 255.f / (d.max_db - d.min_db);
  for (int k = tid; k < M; k += nt) {
    const int k2 = (M - k) & (M - 1);
    const Cplx z = a[__brev((unsigned)k) >> (32 - lg)];
    const Cplx zc = conj(a[__brev((unsigned)k2) >> (32 - lg)]);
    const Cplx e = Cplx{0.5f * (z.re + zc.re), 0.5f * (z.im + zc.im)};
    const Cplx o = mul_negi(Cplx{0.5f * (z.re - zc.re), 0.5f * (z.im - zc.im)});
    const Cplx w = d.tw_full[k];  // exp(-2 pi i k / N)
    const Cplx x = cadd(e, cmul(o, w));
    const float norm = hypotf(x.re, x.im) * nf;
    float value = tau * d.prev[k] + (1.f - tau) * norm;
    value = isfinite(value) ? value : 0.f;
    const float db = 20.f * log10f(value);  // analysis.rs:365-368
    d.db_out[(uint64_t)inst * M + k] = db;
    // analysis.rs:388-400
    const float scaled = bscale * (db - d.min_db);
    const float clamped = scaled < 0.f ? 0.f : scaled > 255.f ? 255.f : scaled;
    d.byte_out[(uint64_t)inst * M + k] = isnan(scaled) ? (uint8_t)0 : (uint8_t)clamped;
  }
}

int fft_threads(int n) {
  int t = n / 16;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

}  // namespace

static void allow_big_lds(size_t bytes) {
  if (bytes <= 64 * 1024) return;
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_kernel<MODE_IR>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_kernel<MODE_FWD>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_kernel<MODE_INV>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_FWD, 0>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_INV, 0>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_FWD, 1>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_INV, 1>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_FWD, 2>));
  raise_lds_limit(reinterpret_cast<const void*>(conv_fft_pipe_kernel<MODE_INV, 2>));
}

void launch_conv_ir_spectra(const ConvDesc& d, void* stream) {
  if (d.fft3) return launch_conv3_ir_spectra(d, stream);
  allow_big_lds((size_t)d.n * sizeof(Cplx));
  hipLaunchKernelGGL(conv_fft_kernel<MODE_IR>, dim3(d.parts, d.ir_nch, 1), dim3(fft_threads(d.n)), (size_t)(d.n + d.n / 8) * sizeof(Cplx),
                     (hipStream_t)stream, d);
}
// blocks per persistent workgroup: whole (pair, channel) streams when there are enough of them to fill the chip,
// shorter runs otherwise
static int pipe_blocks_per_wg(const ConvDesc& d, int channels) {
  const int streams = (int)d.n_pairs * channels, nbl = d.kb1 - d.kb0;
  int segs = streams >= 512 ? 1 : (512 + streams - 1) / streams;
  if (segs > nbl) segs = nbl;
  return (nbl + segs - 1) / segs;
}
static bool use_pipe(const ConvDesc& d) { return d.n == PIPE_N && !measure_switch("WAA_CONV_FFT_PLAIN"); }

void launch_conv_forward(const ConvDesc& d, void* stream) {
  if (d.fft3) return launch_conv3_forward(d, stream);
  allow_big_lds((size_t)d.n * sizeof(Cplx));
  if (use_pipe(d)) {
    const int bpw = pipe_blocks_per_wg(d, d.cin);
    const dim3 grid((d.kb1 - d.kb0 + bpw - 1) / bpw, d.cin, d.n_pairs);
    const size_t lds = (size_t)(d.n + d.n / 8) * sizeof(Cplx) + (size_t)PIPE_NT * 4 * sizeof(Cplx);
    const char* dbg = measure_switch("WAA_CONV_PIPE_DEBUG");
    if (dbg && dbg[0] == '1')
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_FWD, 1>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    else if (dbg && dbg[0] == '2')
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_FWD, 2>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    else
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_FWD, 0>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    return;
  }
  hipLaunchKernelGGL(conv_fft_kernel<MODE_FWD>, dim3(d.kb1 - d.kb0, d.cin, d.n_pairs), dim3(fft_threads(d.n)), (size_t)(d.n + d.n / 8) * sizeof(Cplx),
                     (hipStream_t)stream, d);
}
void launch_conv_inverse(const ConvDesc& d, void* stream) {
  if (d.fft3) return launch_conv3_inverse(d, stream);
  allow_big_lds((size_t)d.n * sizeof(Cplx));
  if (use_pipe(d)) {
    const int bpw = pipe_blocks_per_wg(d, d.cout);
    const dim3 grid((d.kb1 - d.kb0 + bpw - 1) / bpw, d.cout, d.n_pairs);
    const size_t lds = (size_t)(d.n + d.n / 8) * sizeof(Cplx) + (size_t)PIPE_NT * 4 * sizeof(Cplx);
    const char* dbg = measure_switch("WAA_CONV_PIPE_DEBUG");
    if (dbg && dbg[0] == '1')
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_INV, 1>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    else if (dbg && dbg[0] == '2')
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_INV, 2>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    else
      hipLaunchKernelGGL((conv_fft_pipe_kernel<MODE_INV, 0>), grid, dim3(PIPE_NT), lds, (hipStream_t)stream, d, bpw);
    return;
  }
  hipLaunchKernelGGL(conv_fft_kernel<MODE_INV>, dim3(d.kb1 - d.kb0, d.cout, d.n_pairs), dim3(fft_threads(d.n)), (size_t)(d.n + d.n / 8) * sizeof(Cplx),
                     (hipStream_t)stream, d);
}
void launch_conv_direct(const ConvDesc& d, void* stream) {
  dim3 grid((unsigned)(d.kb1 - d.kb0), d.cout, d.n_inst);
  hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
}
void launch_analyser(const AnalyserDesc& d, void* stream) {
  const int M = d.fft_size / 2;
  if ((size_t)M * sizeof(Cplx) > 64 * 1024) raise_lds_limit(reinterpret_cast<const void*>(analyser_kernel));
  hipLaunchKernelGGL(analyser_kernel, dim3(d.n_inst), dim3(fft_threads(M)), (size_t)M * sizeof(Cplx), (hipStream_t)stream, d);
}
void launch_conv_mac(const ConvDesc& d0, void* stream) {
  dim3 grid(d0.n / 256, d0.n_pairs * (uint32_t)d0.cout);
  ConvDesc dm = d0;
  dm.mac_grid_order = measure_switch("WAA_CONV_MAC_GRID_ORDER") ? 1 : 0;
  const ConvDesc& d = dm;
  bool one_term = true;
  for (int co = 0; co < d.cout; co++) {
    int cnt = 0;
    for (int t = 0; t < d.n_terms; t++) cnt += d.terms[t].out_ch == co;
    one_term &= cnt == 1;
  }
  if (one_term && d.parts > 8 && d.parts <= 24 && !measure_switch("WAA_CONV_MAC_REREAD")) {  // (switch: A/B against conv_mac_kernel)
    if (d.parts <= 12)
      hipLaunchKernelGGL((conv_mac_win_kernel<16, 12>), grid, dim3(256), 0, (hipStream_t)stream, d);
    else if (d.parts <= 16)
      hipLaunchKernelGGL((conv_mac_win_kernel<16, 16>), grid, dim3(256), 0, (hipStream_t)stream, d);
    else if (d.parts <= 22) {  // (8 output blocks per tile were measured: 220 registers all the same, 3.78 ms against 3.59)
      if (measure_switch("WAA_CONV_MAC_NO_PREFETCH"))  // (switch: same-box A/B of the software pipeline)
        hipLaunchKernelGGL((conv_mac_win_kernel<16, 22, false>), grid, dim3(256), 0, (hipStream_t)stream, d);
      else
        hipLaunchKernelGGL((conv_mac_win_kernel<16, 22>), grid, dim3(256), 0, (hipStream_t)stream, d);
    }
    else  // (with the prefetch buffer 24 partitions need 172 registers: two waves per SIMD instead of three)
      hipLaunchKernelGGL((conv_mac_win_kernel<16, 24, false>), grid, dim3(256), 0, (hipStream_t)stream, d);
    return;
  }
  if (d.parts <= 8)
    hipLaunchKernelGGL((conv_mac_kernel<16, 8>), grid, dim3(256), 0, (hipStream_t)stream, d);
  else
    // (32 and 30 output blocks per register tile were measured: fewer re-reads of X on paper, but 280 registers
    // leave one wave per SIMD (5.5 ms) and capping at 256 spills (4.7 ms) against 4.6 ms for this one — the
    // re-reads are L2 / MALL hits already)
    hipLaunchKernelGGL((conv_mac_kernel<16, 24>), grid, dim3(256), 0, (hipStream_t)stream, d);
}

}  // namespace waa
