// alloc_probe — is the "slow kind / fast kind" of C2 a property of WHICH allocation a buffer is, and does the access shape matter
// on a slow one?  (measurement tool, not part of the product; standalone HIP, no torch)
//
// Rounds 2-5 found the streaming Biquad of C2 at 1.33-1.39 or 1.53-1.60 ms "depending on where hipMalloc puts the output", with
// the first batch of a process always of the slow kind, and round 5's review showed product kernels beating the "copy floor" of
// the same box.  This probe allocates NB buffers of C2's footprint (3.94 GB) one after the other and, per buffer:
//   memset        hipMemsetAsync over it (write side alone)
//   w:stream      one-wave-per-stream copy (C2's shape: 2048 waves x 8 KB tiles) FROM the last buffer INTO this one
//   r:stream      the same FROM this one INTO the last buffer
//   w:wg16 / wg8  one workgroup per context (both channels), 16 / 8 waves x 1 KB per channel and chunk (the ring kernel's shape)
//   w:quad        four waves per stream, tile t of the stream to wave t % 4 (32 KB of one stream in flight per workgroup)
//   w:linear      grid-stride float4 copy
// then frees every second buffer, allocates them again and repeats (allocation after free traffic), then times eighths of two
// buffers (is a slow buffer slow everywhere?).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/alloc_probe tools/alloc_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                     \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void linear_copy(const f4v* __restrict__ in, f4v* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i] * 0.5f;
}

// one wave per stream, 8 KB tiles, the next tile in registers before the stores (biquad_stream_kernel's pipeline)
template <int NV4, bool NT>
__global__ __launch_bounds__(64) void stream_copy(const float* __restrict__ in, float* __restrict__ out, size_t stream_len) {
  const int lane = threadIdx.x;
  constexpr int TILE = 64 * NV4 * 4;
  const size_t tiles = stream_len / TILE;
  const float* ip = in + (size_t)blockIdx.x * stream_len;
  float* op = out + (size_t)blockIdx.x * stream_len;
  f4v cur[NV4], nxt[NV4];
#pragma unroll
  for (int j = 0; j < NV4; j++) cur[j] = *(const f4v*)(ip + j * 256 + lane * 4);
  for (size_t t = 0; t < tiles; t++) {
    const size_t tn = t + 1 < tiles ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < NV4; j++)
      nxt[j] = NT ? __builtin_nontemporal_load((const f4v*)(ip + tn * TILE + j * 256 + lane * 4)) : *(const f4v*)(ip + tn * TILE + j * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const f4v v = cur[j] * 0.5f;
      if (NT)
        __builtin_nontemporal_store(v, (f4v*)(op + t * TILE + j * 256 + lane * 4));
      else
        *(f4v*)(op + t * TILE + j * 256 + lane * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < NV4; j++) cur[j] = nxt[j];
  }
}

// W waves per stream: tile t belongs to wave t % W (a workgroup consumes W x 8 KB of ONE stream at a time)
template <int W>
__global__ __launch_bounds__(64 * W) void quad_copy(const float* __restrict__ in, float* __restrict__ out, size_t stream_len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NV4 = 8, TILE = 64 * NV4 * 4;
  const size_t tiles = stream_len / TILE;
  const float* ip = in + (size_t)blockIdx.x * stream_len;
  float* op = out + (size_t)blockIdx.x * stream_len;
  f4v cur[NV4], nxt[NV4];
  if ((size_t)wave >= tiles) return;
#pragma unroll
  for (int j = 0; j < NV4; j++) cur[j] = __builtin_nontemporal_load((const f4v*)(ip + (size_t)wave * TILE + j * 256 + lane * 4));
  for (size_t t = wave; t < tiles; t += W) {
    const size_t tn = t + W < tiles ? t + W : t;
#pragma unroll
    for (int j = 0; j < NV4; j++) nxt[j] = __builtin_nontemporal_load((const f4v*)(ip + tn * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) __builtin_nontemporal_store(cur[j] * 0.5f, (f4v*)(op + t * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) cur[j] = nxt[j];
  }
}

// one workgroup per context: W waves, per chunk wave w copies 256 frames (1 KB) of each of the two channels, two chunks ahead
template <int W>
__global__ __launch_bounds__(64 * W) void wg_copy(const float* __restrict__ in, float* __restrict__ out, size_t stream_len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t chunks = stream_len / (256 * W);
  const float* ip = in + (size_t)blockIdx.x * 2 * stream_len + (size_t)wave * 256 + lane * 4;
  float* op = out + (size_t)blockIdx.x * 2 * stream_len + (size_t)wave * 256 + lane * 4;
  f4v a[2], b[2], c[2];
  auto ld = [&](size_t k, f4v (&v)[2]) {
    const size_t kk = k < chunks ? k : chunks - 1;
    v[0] = *(const f4v*)(ip + kk * 256 * W);
    v[1] = *(const f4v*)(ip + stream_len + kk * 256 * W);
  };
  auto st = [&](size_t k, const f4v (&v)[2]) {
    if (k >= chunks) return;
    *(f4v*)(op + k * 256 * W) = v[0] * 0.5f;
    *(f4v*)(op + stream_len + k * 256 * W) = v[1] * 0.5f;
  };
  ld(0, a);
  ld(1, b);
  for (size_t k = 0; k < chunks; k += 3) {
    ld(k + 2, c);
    st(k, a);
    ld(k + 3, a);
    st(k + 1, b);
    ld(k + 4, b);
    st(k + 2, c);
  }
}

static hipEvent_t e0, e1;
template <class F>
static float timeit(F launch, int reps = 4) {
  launch();
  std::vector<float> v;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    v.push_back(ms);
  }
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 18;
  const size_t n_streams = 2048, stream_len = 235 * 2048, n = n_streams * stream_len, bytes = n * sizeof(float);
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<float*> buf(NB, nullptr);
  for (int i = 0; i < NB; i++) {
    CHECK(hipMalloc(&buf[i], bytes));
    CHECK(hipMemset(buf[i], 0x3F, bytes));
  }
  CHECK(hipDeviceSynchronize());
  auto table = [&](const char* title) {
    printf("## %s  (ms; copy = 2 x %.2f GB: 1.35 ms = 5.8 TB/s, 1.57 ms = 5.0 TB/s)\n", title, bytes / 1e9);
    printf("%3s %16s %7s | %8s %8s | %8s %8s | %7s %7s %7s %7s %7s\n", "i", "ptr", "memset", "w:stream", "r:stream", "w:strmNT", "r:strmNT",
           "w:wg16", "w:wg8", "w:quad4", "w:quad2", "w:lin");
    float* ref = buf[NB - 1];
    for (int i = 0; i < NB; i++) {
      float* b = buf[i];
      float* other = i == NB - 1 ? buf[NB - 2] : ref;
      const float t_set = timeit([&] { CHECK(hipMemsetAsync(b, 0x3F, bytes, 0)); });
      const float t_ws = timeit([&] { hipLaunchKernelGGL((stream_copy<8, false>), dim3(n_streams), dim3(64), 0, 0, other, b, stream_len); });
      const float t_rs = timeit([&] { hipLaunchKernelGGL((stream_copy<8, false>), dim3(n_streams), dim3(64), 0, 0, b, other, stream_len); });
      const float t_wn = timeit([&] { hipLaunchKernelGGL((stream_copy<8, true>), dim3(n_streams), dim3(64), 0, 0, other, b, stream_len); });
      const float t_rn = timeit([&] { hipLaunchKernelGGL((stream_copy<8, true>), dim3(n_streams), dim3(64), 0, 0, b, other, stream_len); });
      const float t_w16 = timeit([&] { hipLaunchKernelGGL((wg_copy<16>), dim3(n_streams / 2), dim3(1024), 0, 0, other, b, stream_len); });
      const float t_w8 = timeit([&] { hipLaunchKernelGGL((wg_copy<8>), dim3(n_streams / 2), dim3(512), 0, 0, other, b, stream_len); });
      const float t_q4 = timeit([&] { hipLaunchKernelGGL((quad_copy<4>), dim3(n_streams), dim3(256), 0, 0, other, b, stream_len); });
      const float t_q2 = timeit([&] { hipLaunchKernelGGL((quad_copy<2>), dim3(n_streams), dim3(128), 0, 0, other, b, stream_len); });
      const float t_l = timeit([&] { hipLaunchKernelGGL(linear_copy, dim3(65536), dim3(256), 0, 0, (const f4v*)other, (f4v*)b, n / 4); });
      printf("%3d %16p %7.3f | %8.3f %8.3f | %8.3f %8.3f | %7.3f %7.3f %7.3f %7.3f %7.3f\n", i, (void*)b, t_set, t_ws, t_rs, t_wn, t_rn, t_w16, t_w8,
             t_q4, t_q2, t_l);
      fflush(stdout);
    }
  };
  table("allocation order, fresh process");
  // pairs (i -> i+1): what a batch sees (input and output allocated one after the other)
  printf("## neighbours: in = buf[i], out = buf[i+1], C2's shape (non-temporal)\n");
  for (int i = 0; i + 1 < NB; i++) {
    const float t = timeit([&] { hipLaunchKernelGGL((stream_copy<8, true>), dim3(n_streams), dim3(64), 0, 0, buf[i], buf[i + 1], stream_len); });
    printf("  %2d -> %2d  %7.3f\n", i, i + 1, t);
  }
  // eighths of the first and of a middle buffer (write side, from the last buffer)
  printf("## eighths (linear copy of 0.49 GB pieces, ms; the whole = 8 x)\n");
  for (int i : {0, 1, NB / 2}) {
    printf("  buf %2d:", i);
    for (int p = 0; p < 8; p++) {
      const size_t off = (n / 8) * p;
      const float t = timeit([&] { hipLaunchKernelGGL(linear_copy, dim3(16384), dim3(256), 0, 0, (const f4v*)(buf[NB - 1] + off), (f4v*)(buf[i] + off), n / 32); });
      printf(" %6.4f", t);
    }
    printf("\n");
  }
  for (int i = 0; i < NB; i += 2) CHECK(hipFree(buf[i]));
  for (int i = 0; i < NB; i += 2) {
    CHECK(hipMalloc(&buf[i], bytes));
    CHECK(hipMemset(buf[i], 0x3F, bytes));
  }
  CHECK(hipDeviceSynchronize());
  table("every second buffer freed and allocated again");
  return 0;
}
