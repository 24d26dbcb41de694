// waa_decode.hip — input preparation on the device: the sample conversion of decoded 16-bit PCM (what the reference's
// decoder delivers for WAV input, decoding.rs:15-54: symphonia's i16 -> f32, sample / 32768) followed by
// AudioBuffer::resample to the context's rate (buffer.rs:311-363: linear interpolation that keeps the first and the
// last sample, f64 index arithmetic, f32 weights, `k_inv * prev + k * next` unfused).  The caller hands over interleaved
// i16 frames — half the PCIe bytes of f32 planes, which is what bounds the boundary — and the
// planes the render kernels read are produced here.  Bit-identical to the host path (waa_buffer_resample).
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {

__global__ __launch_bounds__(256) void pcm16_resample_kernel(const DecodeDesc d) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t item = blockIdx.y;  // one AudioBuffer of the batch
  if (i >= d.target_frames) return;
  const int16_t* src = d.pcm + (uint64_t)item * d.frames * d.nch;
  float* dst = d.out + (uint64_t)item * d.out_item_stride;
  uint64_t prev = i, next = i;
  float k = 0.f;
  if (d.resample) {
    const double position = (double)i / (double)(d.target_frames - 1);  // [0, 1]
    const double playhead = position * (double)(d.frames - 1);
    const double pf = floor(playhead);
    prev = (uint64_t)pf;
    next = prev + 1 < d.frames - 1 ? prev + 1 : d.frames - 1;
    k = (float)(playhead - pf);
  }
  const float k_inv = 1.f - k;
  for (uint32_t c = 0; c < d.nch; c++) {
    const float a = (float)src[prev * d.nch + c] / 32768.f;
    float v = a;
    if (d.resample) {
      const float bnext = (float)src[next * d.nch + c] / 32768.f;
      v = k_inv * a + k * bnext;
    }
    dst[(uint64_t)c * d.out_ch_stride + i] = v * d.scale;
  }
}
void launch_pcm16_resample(const DecodeDesc& d, void* stream) {
  if (d.target_frames == 0 || d.n_items == 0) return;
  dim3 grid((unsigned)((d.target_frames + 255) / 256), d.n_items);
  hipLaunchKernelGGL(pcm16_resample_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
}

}  // namespace waa
