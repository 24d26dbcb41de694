// waa_decode.hip — input preparation on the device: the sample conversion of decoded 16-bit PCM (what the reference's
// decoder delivers for WAV input, decoding.rs:15-54: symphonia's i16 -> f32, sample / 32768) followed by
// AudioBuffer::resample to the context's rate (buffer.rs:311-363: linear interpolation that keeps the first and the
// last sample, f64 index arithmetic, f32 weights, `k_inv * prev + k * next` unfused).  The caller hands over interleaved
// i16 frames — half the PCIe bytes of f32 planes, which is what bounds the boundary — and the
// planes the render kernels read are produced here.  Bit-identical to the host path (waa_buffer_resample).
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {

constexpr uint32_t MAX_GRID_ROWS = 65535;  // gridDim.y limit

__device__ __forceinline__ void pcm16_resample_item(const DecodeDesc& d, uint32_t item, uint64_t i);
__global__ __launch_bounds__(256) void pcm16_resample_kernel(const DecodeDesc d) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= d.target_frames) return;
  // one AudioBuffer of the batch per blockIdx.y; batches above the grid's 65535 rows are walked in strides
  for (uint32_t item = blockIdx.y; item < d.n_items; item += gridDim.y) pcm16_resample_item(d, item, i);
}
__device__ __forceinline__ void pcm16_resample_item(const DecodeDesc& d, uint32_t item, uint64_t i) {
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
  dim3 grid((unsigned)((d.target_frames + 255) / 256), d.n_items < MAX_GRID_ROWS ? d.n_items : MAX_GRID_ROWS);
  hipLaunchKernelGGL(pcm16_resample_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
}

// Rendered AudioBuffers -> interleaved 16-bit PCM on the device (waa_download_all_pcm16): the inverse of the decoder's
// sample / 32768, rounded to nearest and saturated (what a WAV writer does with an AudioBuffer — the reference itself has no
// such step: an OfflineAudioContext hands back f32 planes).  Channels the destination's signal does not carry are zero.
__global__ __launch_bounds__(256) void pcm16_pack_kernel(const EncodeDesc d) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= d.frames) return;
  for (uint32_t item = blockIdx.y; item < d.n_items; item += gridDim.y) {  // (more than 65535 contexts: strides of the grid's rows)
    int16_t* dst = d.pcm + ((uint64_t)item * d.frames + i) * d.nch_out;
    for (uint32_t c = 0; c < d.nch_out; c++) {
      float v = c < d.nch_in ? load_global(d.in + (uint64_t)item * d.in_item_stride + (uint64_t)c * d.in_ch_stride + i) : 0.f;
      v = v * 32768.f;
      v = v < -32768.f ? -32768.f : (v > 32767.f ? 32767.f : v);
      dst[c] = (int16_t)__float2int_rn(v != v ? 0.f : v);
    }
  }
}
void launch_pcm16_pack(const EncodeDesc& d, void* stream) {
  if (d.frames == 0 || d.n_items == 0) return;
  dim3 grid((unsigned)((d.frames + 255) / 256), d.n_items < MAX_GRID_ROWS ? d.n_items : MAX_GRID_ROWS);
  hipLaunchKernelGGL(pcm16_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
}

}  // namespace waa
