// waa_delay.hip — DelayNode outside a cycle (src/node/delay.rs:428-745): the writer/reader pair over a ring of
// render quanta is, in absolute time, a gather from the node's own input signal:
//     position = i - delay * sample_rate            (i = frame within the quantum; f64, delay.rs:697-701)
//     prev     = quantum_start + floor(position),   k = (float)(position - floor(position))
//     out      = fma(1 - k, in[prev], k * in[prev + 1])   in f32 (delay.rs:642), frames before 0 read silence.
// Node-major like the convolver: the input is a materialised signal, one thread renders 4 consecutive frames
// (16 B store), reads come from L2/HBM at the delayed offset (coalesced: neighbouring lanes read neighbouring
// frames for k-rate delays).  HBM-bound: 4 B read (+ overlap) and 4 B written per frame-channel.
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {

__global__ __launch_bounds__(256) void delay_kernel(const DelayDesc d) {
  const uint32_t sid = blockIdx.y;  // instance * nch + channel
  const uint32_t inst = sid / (uint32_t)d.nch;
  const int ch = (int)(sid % (uint32_t)d.nch);
  const uint64_t f0 = (uint64_t)d.tile0 * TILE + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (f0 >= (uint64_t)d.tile1 * TILE) return;
  const float* in = d.in.base + (uint64_t)inst * d.in.inst_stride + (uint64_t)ch * d.in.ch_stride;
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride;
  const uint32_t q = (uint32_t)(f0 / RQ);
  const int i0 = (int)(f0 % RQ);
  const int64_t qstart = (int64_t)q * RQ;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (q < d.n_quanta && d.delay.mode != 2) {
    // one value per quantum (constant or k-rate)
    const float dv = d.delay.mode == 0 ? d.delay.base[inst] : d.delay.base[(uint64_t)inst * d.delay.stride + q];
    delay_read4(in, d.frames, dv, d.sample_rate, d.num_quanta, d.in_cycle != 0, d.quantum_duration, q, i0, r);
  } else if (q < d.n_quanta) {
    // a-rate: every frame from its own value (delay.rs:591-606)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int i = i0 + e;
      const float dv = d.delay.base[(uint64_t)inst * d.delay.stride + (uint64_t)q * RQ + i];
      double dd = (double)dv;
      if (d.in_cycle) dd = fmax(dd, d.quantum_duration);
      const double position = (double)i - dd * d.sample_rate;
      const double fl = floor(position);
      const int64_t pf = (int64_t)fl;
      const float k = (float)(position - fl);
      const int64_t prev = qstart + pf;
      // the sample after frame 127 of the newest block is frame 0 of the OLDEST ring block (delay.rs:622-626);
      // only reachable with a zero delay, where k == 0
      int64_t next = pf == RQ - 1 ? (int64_t)(q - (int64_t)d.num_quanta) * RQ : prev + 1;
      if (d.in_cycle && next >= qstart) next -= ((int64_t)d.num_quanta + 1) * RQ;
      const float ps = prev >= 0 ? in[prev] : 0.f;
      const float nsamp = next >= 0 ? in[next] : 0.f;
      r[e] = __builtin_fmaf(1.f - k, ps, k * nsamp);
    }
  }
  *reinterpret_cast<float4*>(out + f0) = make_float4(r[0], r[1], r[2], r[3]);
}

void launch_delay(const DelayDesc& d, void* stream) {
  const dim3 grid((uint32_t)(((uint64_t)(d.tile1 - d.tile0) * TILE + 1023) / 1024), d.n_inst * (uint32_t)d.nch), block(256);
  hipLaunchKernelGGL(delay_kernel, grid, block, 0, (hipStream_t)stream, d);
}

}  // namespace waa
