// waa_echo.hip — the echo loop  line = inputs + g * delayed(line)  with its delay line in LDS.
//
// A block-scheduled feedback loop whose body is ONE element-wise launch per block (DelayNode <-> GainNode, the delay read from
// its line by the summing stage: DESIGN.md 3.1d) costs, per block, a read of the source, a read of the delayed line and a write
// of the line — and 47 launches of 250 MB for a 10 s render.  Here one workgroup per instance walks the render in chunks and
// keeps the last 16384 frames of the line in LDS: the delayed read never goes to memory, the line is written once (its
// consumers outside the loop read it from HBM as before).  Same arithmetic in the same order as chain_kernel's input stage
// (waa_kernels.hip: load -> edge gain with gain.rs' mute / pass-through cases -> up-mix -> sum in edge order; the delayed
// sample is fma(1 - k, x[i], k * x[i + 1]), delay.rs:560-590): bit-identical, which tests/test_cycles.py asserts.
// Qualification: echo_ring_applicable() below; everything else keeps the launch-per-block form.
//
// The tail.  The usual echo graph sends  dry + delayed(line)  to the destination: as a launch of its own that stage reads the
// source and the line again and writes the output (3 x 3.9 GB of the fb workload's 19.7 GB).  When the line has exactly one
// reader outside the loop and that reader is such a sum of the delayed line and signals the loop reads anyway
// (echo_tail_applicable), this kernel renders it from the values it already holds — the delayed samples out of the ring, the
// source out of registers — and the line itself is never written to memory: the loop costs its compulsory traffic, the source
// read once and the output written once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "waa_internal.hpp"

namespace waa {

constexpr int ECHO_RING = 16384;  // frames per channel kept in LDS (two channels: 128 KB)
constexpr int ECHO_EXT = 2;        // inputs from outside the loop (held in registers two chunks ahead)

namespace {
__device__ __forceinline__ float echo_delay_value(const ParamRef& p, uint32_t inst) {
  return p.mode == 3 ? __uint_as_float((uint32_t)p.stride) : load_global(p.base + inst);
}

// gain.rs:163-179 on an input edge (mode 0 / 1: one value for the quantum), then quantum.rs' up-mix 1 -> 2: copy (speakers) /
// silence (discrete)
template <int CM>
__device__ __forceinline__ void echo_edge(const InputRef& in, uint32_t inst, uint32_t qc, int to_nch, int interp, float (&u)[CM][4]) {
  if (in.has_gain) {
    const float g = in.gain.mode == 0 ? load_global(in.gain.base + inst) : load_global(in.gain.base + (uint64_t)inst * in.gain.stride + qc);
    const bool mute = fabsf(g) <= 1e-6f, pass = fabsf(1.f - g) <= 1e-6f;
#pragma unroll
    for (int c = 0; c < CM; c++)
      if (c < in.nch) {
#pragma unroll
        for (int e = 0; e < 4; e++) u[c][e] = mute ? 0.f : (pass ? u[c][e] : u[c][e] * g);
      }
  }
  if (CM == 2 && in.nch == 1 && to_nch == 2) {
#pragma unroll
    for (int e = 0; e < 4; e++) u[CM - 1][e] = interp == 1 ? 0.f : u[0][e];
  }
}

// C: channels of the line (the ring); CT: channels of the fused tail stage (0: none)
template <int C, int CT>
__global__ __launch_bounds__(1024) void echo_ring_kernel(const ChainDesc d, int fb, int chunk_subtiles, const EchoTail t) {
  constexpr int CM = C > CT ? C : CT;
  extern __shared__ __attribute__((aligned(16))) float ring[];  // [C][ECHO_RING]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t inst = blockIdx.x;
  for (int i = tid; i < C * ECHO_RING; i += blockDim.x) ring[i] = 0.f;
  __syncthreads();
  const InputRef& fbin = d.in[fb];
  // DelayReader's position arithmetic (delay.rs:560-569), one delayTime per instance
  const float dv = echo_delay_value(fbin.offset, inst);
  const double position = 0. - (double)dv * fbin.sample_rate;
  const double fl = floor(position);
  const int64_t pf0 = (int64_t)fl;
  const float kf = (float)(position - fl);
  const uint32_t total_sub = (d.tile1 - d.tile0) * (TILE / 256);
  const uint64_t f_first = (uint64_t)d.tile0 * TILE;
  // The inputs from outside the loop (at most ECHO_EXT of them, slot = their order among the non-feedback inputs) do not
  // depend on the ring: they are requested TWO chunks ahead.  One workgroup per CU and 16 KB per chunk and channel: with the
  // loads of one chunk in flight the chip holds 8 MB of reads, 4 TB/s at the loaded latency; two chunks cover it.
  float nx1[ECHO_EXT][C][4], nx2[ECHO_EXT][C][4];
  auto fetch_ext = [&](uint32_t sub_n, float (&dst)[ECHO_EXT][C][4]) __attribute__((always_inline)) {
    const uint64_t fn = f_first + (uint64_t)sub_n * 256 + (uint64_t)lane * 4;
#pragma unroll
    for (int sl = 0; sl < ECHO_EXT; sl++) {
#pragma unroll
      for (int c = 0; c < C; c++) dst[sl][c][0] = dst[sl][c][1] = dst[sl][c][2] = dst[sl][c][3] = 0.f;
      const int k = sl + (sl >= fb ? 1 : 0);  // (input index of slot sl)
      if (k < d.n_inputs && sub_n < total_sub) {
        const InputRef& in = d.in[k];
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < in.nch) {
            const float* p = in.sig.base + (uint64_t)inst * in.sig.inst_stride + (uint64_t)c * in.sig.ch_stride;
            const bool inside = in.valid == 0 || fn + 3 < in.valid;
            const f4v t4 = load_global_f4(inside ? p + fn : p);  // (unconditional load, the zero selected below)
            dst[sl][c][0] = inside ? t4.x : 0.f;
            dst[sl][c][1] = inside ? t4.y : 0.f;
            dst[sl][c][2] = inside ? t4.z : 0.f;
            dst[sl][c][3] = inside ? t4.w : 0.f;
          }
      }
    }
  };
  fetch_ext((uint32_t)wave, nx1);
  fetch_ext((uint32_t)(wave + chunk_subtiles), nx2);
  for (uint32_t s0 = 0; s0 < total_sub; s0 += (uint32_t)chunk_subtiles) {
    const uint32_t sub = s0 + (uint32_t)wave;
    float cur[ECHO_EXT][C][4];
#pragma unroll
    for (int sl = 0; sl < ECHO_EXT; sl++)
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          cur[sl][c][e] = nx1[sl][c][e];
          nx1[sl][c][e] = nx2[sl][c][e];
        }
    fetch_ext(sub + 2u * (uint32_t)chunk_subtiles, nx2);
    if (wave < chunk_subtiles && sub < total_sub) {
      const uint64_t f = f_first + (uint64_t)sub * 256 + (uint64_t)lane * 4;
      const uint32_t q = (uint32_t)(f / RQ);
      const uint32_t qc = q < d.n_quanta ? q : d.n_quanta - 1;
      float v[C][4], xd[C][4];
#pragma unroll
      for (int c = 0; c < C; c++) {
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
        xd[c][0] = xd[c][1] = xd[c][2] = xd[c][3] = 0.f;
      }
      if (q < d.n_quanta) {  // the delayed samples of this group, out of the ring
#pragma unroll
        for (int c = 0; c < C; c++) {
          float x[5];
#pragma unroll
          for (int e = 0; e < 5; e++) {
            const int64_t idx = (int64_t)f + pf0 + e;
            x[e] = idx < 0 ? 0.f : ring[c * ECHO_RING + (int)(idx & (ECHO_RING - 1))];
          }
#pragma unroll
          for (int e = 0; e < 4; e++) xd[c][e] = __builtin_fmaf(1.f - kf, x[e], kf * x[e + 1]);
        }
      }
#pragma unroll
      for (int k = 0; k < MAX_INPUTS; k++) {
        if (k >= d.n_inputs) continue;
        const InputRef& in = d.in[k];
        float u[CM][4];
#pragma unroll
        for (int c = 0; c < CM; c++) u[c][0] = u[c][1] = u[c][2] = u[c][3] = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < in.nch) {
#pragma unroll
            for (int e = 0; e < 4; e++) u[c][e] = k == fb ? xd[c][e] : (k - (k > fb ? 1 : 0) == 0 ? cur[0][c][e] : cur[1][c][e]);
          }
        echo_edge<CM>(in, inst, qc, d.in_nch, d.in_interp, u);
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < d.in_nch) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[c][e] = k == 0 ? u[c][e] : v[c][e] + u[c][e];
          }
      }
#pragma unroll
      for (int c = 0; c < C; c++)
        if (c < d.out.nch) {
          if (CT == 0 || t.store_line) {
            float* po = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride + f;
            *reinterpret_cast<float4*>(po) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
          }
          *reinterpret_cast<f4v*>(ring + c * ECHO_RING + (int)(f & (ECHO_RING - 1))) = f4v{v[c][0], v[c][1], v[c][2], v[c][3]};
        }
      if (CT > 0) {  // the tail stage: same input arithmetic, its operands taken from xd / cur
        float w[CM][4];
#pragma unroll
        for (int c = 0; c < CM; c++) w[c][0] = w[c][1] = w[c][2] = w[c][3] = 0.f;
#pragma unroll
        for (int k = 0; k < MAX_INPUTS; k++) {
          if (k >= t.n_inputs) continue;
          const InputRef& in = t.in[k];
          const int al = t.alias[k];
          float u[CM][4];
#pragma unroll
          for (int c = 0; c < CM; c++) u[c][0] = u[c][1] = u[c][2] = u[c][3] = 0.f;
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) {
#pragma unroll
              for (int e = 0; e < 4; e++) {
                u[c][e] = al == 0 ? cur[0][c][e] : (al == 1 ? cur[1][c][e] : xd[c][e]);  // (alias -2: the delayed line)
              }
            }
          echo_edge<CM>(in, inst, qc, t.in_nch, t.in_interp, u);
#pragma unroll
          for (int c = 0; c < CM; c++)
            if (c < t.in_nch) {
#pragma unroll
              for (int e = 0; e < 4; e++) w[c][e] = k == 0 ? u[c][e] : w[c][e] + u[c][e];
            }
        }
#pragma unroll
        for (int c = 0; c < CM; c++)
          if (c < t.out.nch) {
            float* po = t.out.base + (uint64_t)inst * t.out.inst_stride + (uint64_t)c * t.out.ch_stride + f;
            *reinterpret_cast<float4*>(po) = make_float4(w[c][0], w[c][1], w[c][2], w[c][3]);
          }
      }
    }
    __syncthreads();  // the chunk is in the ring before the next one reads behind it
  }
}
}  // namespace

// The loop step `d` (one element-wise launch per block, ChainDesc::tile0 / tile1 = the whole render) as the LDS-ring kernel?
// Returns the index of the feedback input and the chunk size (sub-tiles of 256 frames), or -1.
int echo_ring_applicable(const ChainDesc& d, const float* delay_min_max_frames, int* chunk_subtiles) {
  if (d.n_ops != 0 || d.in_nch < 1 || d.in_nch > 2 || d.out.nch != d.in_nch || d.n_inputs < 2 || d.n_inputs > 1 + ECHO_EXT) return -1;
  int fb = -1;
  for (int k = 0; k < d.n_inputs; k++) {
    const InputRef& in = d.in[k];
    if (in.has_gain && !(in.gain.mode == 0 || in.gain.mode == 1)) return -1;
    if (in.nch != d.in_nch && !(in.nch == 1 && d.in_nch == 2)) return -1;
    if (in.kind == IN_DELAYED) {
      // the line this launch writes, read back by itself; one delayTime per instance
      if (fb >= 0 || !in.feedback || in.sig.base != d.out.base || in.sig.inst_stride != d.out.inst_stride ||
          in.sig.ch_stride != d.out.ch_stride || !(in.offset.mode == 0 || in.offset.mode == 3))
        return -1;
      fb = k;
    } else if (in.kind != IN_SIGNAL) {
      return -1;
    } else if (in.sig.base == d.out.base || ((uintptr_t)in.sig.base & 15) || (in.sig.ch_stride & 3) || (in.sig.inst_stride & 3)) {
      return -1;
    }
  }
  if (fb < 0 || ((uintptr_t)d.out.base & 15) || (d.out.ch_stride & 3) || (d.out.inst_stride & 3)) return -1;
  // the frames a chunk reads must lie BEHIND the chunk (delay > chunk) and still be in the ring (delay + chunk < ring)
  const float dmin = delay_min_max_frames[0], dmax = delay_min_max_frames[1];
  int ch = 0;
  for (int cand : {16, 8, 4})
    if ((float)(cand * 256 + 8) <= dmin) {
      ch = cand;
      break;
    }
  if (!ch || dmax > (float)(ECHO_RING - ch * 256 - 8)) return -1;
  *chunk_subtiles = ch;
  return fb;
}

// The chain step `tail` (outside the loop) as the tail stage of the ring kernel: a plain sum, to at least the line's channel
// count, of delayed(line) — the loop's own delayTime — and of signals the loop step reads too.
int echo_tail_applicable(const ChainDesc& d, int fb, const ChainDesc& tail, EchoTail* t) {
  if (tail.n_ops != 0 || tail.in_nch < d.in_nch || tail.in_nch > 2 || tail.out.nch != tail.in_nch || tail.n_inputs < 1 ||
      tail.n_inputs > MAX_INPUTS || tail.n_inst != d.n_inst || tail.n_quanta != d.n_quanta)
    return 0;
  if (((uintptr_t)tail.out.base & 15) || (tail.out.ch_stride & 3) || (tail.out.inst_stride & 3) || tail.out.base == d.out.base) return 0;
  const InputRef& fbin = d.in[fb];
  for (int j = 0; j < d.n_inputs; j++)
    if (d.in[j].sig.base == tail.out.base) return 0;  // (rendered in place over something the loop still reads)
  EchoTail r{};
  r.n_inputs = tail.n_inputs;
  r.in_nch = tail.in_nch;
  r.in_interp = tail.in_interp;
  r.out = tail.out;
  bool reads_line = false;
  for (int k = 0; k < tail.n_inputs; k++) {
    const InputRef& in = tail.in[k];
    if (in.has_gain && !(in.gain.mode == 0 || in.gain.mode == 1)) return 0;
    if (in.nch != tail.in_nch && !(in.nch == 1 && tail.in_nch == 2)) return 0;
    r.in[k] = in;
    r.alias[k] = -1;
    if (in.kind == IN_DELAYED) {
      if (in.sig.base != d.out.base || in.sig.inst_stride != d.out.inst_stride || in.sig.ch_stride != d.out.ch_stride ||
          in.nch != d.in_nch || !(in.offset.mode == 0 || in.offset.mode == 3) || in.sample_rate != fbin.sample_rate)
        return 0;  // (a line belongs to ONE DelayNode: every delayed read of it is by that node's delayTime, the loop's own)
      r.alias[k] = -2;
      reads_line = true;
    } else if (in.kind == IN_SIGNAL) {
      for (int j = 0; j < d.n_inputs; j++) {
        const InputRef& lj = d.in[j];
        if (j != fb && lj.kind == IN_SIGNAL && lj.sig.base == in.sig.base && lj.sig.inst_stride == in.sig.inst_stride &&
            lj.sig.ch_stride == in.sig.ch_stride && lj.nch == in.nch && lj.valid == in.valid)
          r.alias[k] = j - (j > fb ? 1 : 0);  // (its register slot)
      }
      if (r.alias[k] < 0) return 0;
    } else {
      return 0;
    }
  }
  if (!reads_line) return 0;
  *t = r;
  return 1;
}

void launch_echo_ring(const ChainDesc& d, int fb, int chunk_subtiles, const EchoTail* tail, void* stream) {
  const size_t lds = (size_t)d.in_nch * ECHO_RING * sizeof(float);
  const int ct = tail ? tail->in_nch : 0;
  EchoTail t{};
  if (tail) t = *tail;
  const dim3 block((unsigned)chunk_subtiles * 64), grid(d.n_inst);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, d, fb, chunk_subtiles, t);
  };
  if (d.in_nch == 1) {
    if (ct == 0) go(echo_ring_kernel<1, 0>);
    else if (ct == 1) go(echo_ring_kernel<1, 1>);
    else go(echo_ring_kernel<1, 2>);
  } else {
    if (ct == 0) go(echo_ring_kernel<2, 0>);
    else go(echo_ring_kernel<2, 2>);
  }
}

}  // namespace waa
