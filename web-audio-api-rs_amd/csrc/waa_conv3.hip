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
  const int k0 = blockIdx.x * blocks_per_wg;
  const int k1 = k0 + blocks_per_wg < d.nb ? k0 + blocks_per_wg : d.nb;
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
  const int k0 = blockIdx.x * blocks_per_wg;
  const int k1 = k0 + blocks_per_wg < d.nb ? k0 + blocks_per_wg : d.nb;
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
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fft3_fwd_kernel<F3_FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fft3_fwd_kernel<F3_IR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fft3_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}
// blocks per persistent workgroup: whole (pair, channel) streams when there are enough of them to fill the chip, shorter
// runs otherwise
static int f3_blocks_per_wg(const ConvDesc& d, int channels) {
  const int streams = (int)d.n_pairs * channels;
  int segs = streams >= 512 ? 1 : (512 + streams - 1) / streams;
  if (segs > d.nb) segs = d.nb;
  return (d.nb + segs - 1) / segs;
}

void launch_conv3_ir_spectra(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  hipLaunchKernelGGL(conv_fft3_fwd_kernel<F3_IR>, dim3(d.parts, d.ir_nch, 1), dim3(NT), (size_t)LDS_BYTES, (hipStream_t)stream, d, 1);
}
void launch_conv3_forward(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  const int bpw = f3_blocks_per_wg(d, d.cin);
  hipLaunchKernelGGL(conv_fft3_fwd_kernel<F3_FWD>, dim3((d.nb + bpw - 1) / bpw, d.cin, d.n_pairs), dim3(NT), (size_t)LDS_BYTES,
                     (hipStream_t)stream, d, bpw);
}
void launch_conv3_inverse(const ConvDesc& d, void* stream) {
  f3_allow_lds();
  const int bpw = f3_blocks_per_wg(d, d.cout);
  hipLaunchKernelGGL(conv_fft3_inv_kernel, dim3((d.nb + bpw - 1) / bpw, d.cout, d.n_pairs), dim3(NT), (size_t)LDS_BYTES,
                     (hipStream_t)stream, d, bpw);
}

}  // namespace waa
