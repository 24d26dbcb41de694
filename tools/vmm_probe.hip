// vmm_probe — a write / read grade of the device's memory, 1 GiB of PHYSICAL memory at a time (measurement tool, standalone HIP)
//
// tools/alloc_probe showed (profiles/r06a_alloc_probe.txt) that the "slow kind" of C2 is a property of the REGION a buffer comes
// from: the first 8 x 3.94 GB a fresh process allocates take 1.49 ms as the destination of C2's copy shape and 1.31 ms as its
// source, the regions behind them 1.30-1.37 ms either way.  This probe asks what a library can do about it with the virtual
// memory API: it creates NC physical chunks of 1 GiB (hipMemCreate), maps them, grades every chunk as destination and as source of
// the one-wave-per-stream copy, then COMPOSES buffers of C2's size from chosen chunks (hipMemMap into one address range) and times
// C2's shape on them: best-read -> best-write, worst -> worst, creation order.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe tools/vmm_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
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

template <int NV4>
__global__ __launch_bounds__(64) void stream_copy(const float* __restrict__ in, float* __restrict__ out, size_t stream_len) {
  const int lane = threadIdx.x;
  constexpr int TILE = 64 * NV4 * 4;
  const size_t tiles = stream_len / TILE;
  const float* ip = in + (size_t)blockIdx.x * stream_len;
  float* op = out + (size_t)blockIdx.x * stream_len;
  f4v cur[NV4], nxt[NV4];
#pragma unroll
  for (int j = 0; j < NV4; j++) cur[j] = __builtin_nontemporal_load((const f4v*)(ip + j * 256 + lane * 4));
  for (size_t t = 0; t < tiles; t++) {
    const size_t tn = t + 1 < tiles ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < NV4; j++) nxt[j] = __builtin_nontemporal_load((const f4v*)(ip + tn * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) __builtin_nontemporal_store(cur[j] * 0.5f, (f4v*)(op + t * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) cur[j] = nxt[j];
  }
}
__global__ __launch_bounds__(256) void fill(f4v* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = f4v{0.7f, 0.7f, 0.7f, 0.7f};
}
__global__ __launch_bounds__(256) void read_only(const f4v* in, float* sink, size_t n4) {
  f4v a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) a += __builtin_nontemporal_load(in + i);
  if (a.x + a.y + a.z + a.w == 12345.f) *sink = a.x;
}

static hipEvent_t e0, e1;
template <class F>
static float timeit(F launch, int reps = 3) {
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

// hipMalloc'ed buffers of C2's size, one after the other; C2's shape from each into the next
static void malloc_pairs(const char* title, int nb, bool keep) {
  const size_t stream_len = 235 * 2048, bytes = 2048 * stream_len * 4;
  std::vector<float*> b(nb, nullptr);
  for (int i = 0; i < nb; i++) {
    CHECK(hipMalloc(&b[i], bytes));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)b[i], bytes / 16);
  }
  CHECK(hipDeviceSynchronize());
  printf("## hipMalloc pairs, %s (ms):", title);
  for (int i = 0; i + 1 < nb; i++) {
    const float t = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, b[i], b[i + 1], stream_len); }, 5);
    printf(" %d->%d %.3f", i, i + 1, t);
  }
  printf("\n");
  fflush(stdout);
  if (!keep)
    for (int i = 0; i < nb; i++) CHECK(hipFree(b[i]));
}

static void malloc_pairs(const char* title, int nb, bool keep);
// mode 4: is it the page-table fragment?  The same eight physical chunks of 1 GiB (in = 4, out = 4) mapped at virtual addresses of
// different alignment: a mapping whose virtual and physical addresses agree modulo 2^k can be described by fragments of up to 2^k.
static int align_sweep() {
  const size_t CH = 1ull << 30;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  hipMemGenericAllocationHandle_t h[8];
  for (int k = 0; k < 8; k++) CHECK(hipMemCreate(&h[k], CH, &prop, 0));
  void* va = nullptr;
  CHECK(hipMemAddressReserve(&va, 12 * CH, CH, nullptr, 0));
  const uintptr_t base = ((uintptr_t)va + CH - 1) / CH * CH;  // 1 GiB-aligned whatever the reservation's own alignment is
  printf("# reservation %p (its own alignment: %d bits), working base %#zx\n", va, __builtin_ctzll((unsigned long long)(uintptr_t)va), (size_t)base);
  const size_t stream_len = 235 * 2048;
  auto run = [&](const char* name, size_t shift_in, size_t shift_out) {
    char* pin = (char*)base + shift_in;
    char* pout = (char*)base + 5 * CH + shift_out;
    for (int k = 0; k < 4; k++) {
      CHECK(hipMemMap(pin + (size_t)k * CH, CH, 0, h[k], 0));
      CHECK(hipMemMap(pout + (size_t)k * CH, CH, 0, h[4 + k], 0));
    }
    CHECK(hipMemSetAccess(pin, 4 * CH, &acc, 1));
    CHECK(hipMemSetAccess(pout, 4 * CH, &acc, 1));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)pin, 4 * CH / 16);
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)pout, 4 * CH / 16);
    const float t = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, (const float*)pin, (float*)pout, stream_len); }, 5);
    const float tf = timeit([&] { hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)pout, 4 * CH / 16); }, 5);
    printf("  %-40s in +%-9zu out +%-9zu  C2 shape %7.3f ms  %6.0f GB/s   fill of 4 GiB %7.3f ms\n", name, shift_in, shift_out, t,
           2.0 * 2048 * stream_len * 4 / (t * 1e-3) / 1e9, tf);
    fflush(stdout);
    CHECK(hipDeviceSynchronize());
    for (int k = 0; k < 4; k++) {
      CHECK(hipMemUnmap(pin + (size_t)k * CH, CH));
      CHECK(hipMemUnmap(pout + (size_t)k * CH, CH));
    }
  };
  for (int rep = 0; rep < 2; rep++) {
    run("both 1 GiB-aligned", 0, 0);
    run("both +32 MiB", 32u << 20, 32u << 20);
    run("both +2 MiB", 2u << 20, 2u << 20);
    run("both +64 KiB", 64u << 10, 64u << 10);
    run("both +4 KiB", 4096, 4096);
    run("out +2 MiB only", 0, 2u << 20);
    run("out +64 KiB only", 0, 64u << 10);
    run("out +4 KiB only", 0, 4096);
    run("in +4 KiB only", 4096, 0);
  }
  return 0;
}

// mode 5: virtual or physical?  E1: the SAME eight chunks mapped at six different address windows; E2: ONE window, eight disjoint
// sets of chunks; E3: hipMalloc'ed pairs in between (the same process, the same minute).
static int window_vs_chunks(int NC) {
  const size_t CH = 1ull << 30;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NC);
  for (int k = 0; k < NC; k++) CHECK(hipMemCreate(&h[k], CH, &prop, 0));
  const size_t stream_len = 235 * 2048;
  const int NW = 6;
  void* win[NW];
  for (int w = 0; w < NW; w++) {
    CHECK(hipMemAddressReserve(&win[w], 8 * CH, 0, nullptr, 0));
    void* spacer = nullptr;  // (the windows are not neighbours)
    CHECK(hipMemAddressReserve(&spacer, (size_t)(3 + 5 * w) * CH + ((size_t)w << 21), 0, nullptr, 0));
  }
  auto run = [&](void* w, int first) {
    for (int k = 0; k < 8; k++) CHECK(hipMemMap((char*)w + (size_t)k * CH, CH, 0, h[first + k], 0));
    CHECK(hipMemSetAccess(w, 8 * CH, &acc, 1));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)w, 8 * CH / 16);
    const float t = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, (const float*)w, (float*)((char*)w + 4 * CH), stream_len); }, 5);
    const float tr = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, (const float*)((char*)w + 4 * CH), (float*)w, stream_len); }, 5);
    CHECK(hipDeviceSynchronize());
    for (int k = 0; k < 8; k++) CHECK(hipMemUnmap((char*)w + (size_t)k * CH, CH));
    printf("   window %p chunks %2d..%2d  lo->hi %7.3f ms  hi->lo %7.3f ms\n", w, first, first + 7, t, tr);
    fflush(stdout);
  };
  printf("## E1: the same chunks 0..7 at six windows\n");
  for (int w = 0; w < NW; w++) run(win[w], 0);
  printf("## E2: window 0, disjoint sets of chunks\n");
  for (int f = 0; f + 8 <= NC; f += 8) run(win[0], f);
  printf("## E1 again\n");
  for (int w = 0; w < NW; w++) run(win[w], 0);
  malloc_pairs("same process", 8, false);
  printf("## E2 again, window 3\n");
  for (int f = 0; f + 8 <= NC; f += 8) run(win[3], f);
  return 0;
}

// mode 6: does the DISTANCE between what a wavefront reads and what it writes matter?  16 chunks mapped side by side; the input at
// the window's start (+ a), the output 8 GiB further (+ b): b alone moves the output relative to the input, a = b moves both.
static int delta_sweep() {
  const size_t CH = 1ull << 30;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  void* w = nullptr;
  CHECK(hipMemAddressReserve(&w, 16 * CH, 0, nullptr, 0));
  for (int k = 0; k < 16; k++) {
    hipMemGenericAllocationHandle_t h;
    CHECK(hipMemCreate(&h, CH, &prop, 0));
    CHECK(hipMemMap((char*)w + (size_t)k * CH, CH, 0, h, 0));
  }
  CHECK(hipMemSetAccess(w, 16 * CH, &acc, 1));
  hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)w, 16 * CH / 16);
  CHECK(hipDeviceSynchronize());
  const size_t stream_len = 235 * 2048;
  auto run = [&](size_t a, size_t b) {
    const float* in = (const float*)((char*)w + a);
    float* out = (float*)((char*)w + 8 * CH + b);
    return timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, in, out, stream_len); }, 5);
  };
  std::vector<size_t> ds = {0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1u << 20, 2u << 20, 4u << 20, 8u << 20,
                            16u << 20, 32u << 20, 64u << 20, 128u << 20, 256u << 20, 512u << 20, 1u << 30, 3u << 12, (1u << 20) + 4096, 5u << 20, 37u << 20, 1000u << 20};
  printf("## delta (bytes): output moved / both moved / input moved   (ms)\n");
  for (size_t d : ds) {
    printf("  %11zu  %7.3f  %7.3f  %7.3f\n", d, run(0, d), run(d, d), run(d, 0));
    fflush(stdout);
  }
  // the output BELOW the input (hi -> lo), and both in the same GiB-aligned slots exchanged
  auto run2 = [&](size_t in_off, size_t out_off) {
    return timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, (const float*)((char*)w + in_off), (float*)((char*)w + out_off), stream_len); }, 5);
  };
  printf("## slots (GiB): in -> out\n");
  for (int i : {0, 4, 8, 12})
    for (int o : {0, 4, 8, 12})
      if (i != o) printf("  %2d -> %2d  %7.3f\n", i, o, run2((size_t)i * CH, (size_t)o * CH));
  return 0;
}

int main(int argc, char** argv) {
  int NC = argc > 1 ? atoi(argv[1]) : 160;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;  // 1: hipMalloc pairs first (freed), chunks, pairs again; 2: chunks first, then pairs; 3: pairs KEPT, then chunks
  const size_t CH = 1ull << 30;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  if (mode == 4) return align_sweep();
  if (mode == 5) return window_vs_chunks(NC);
  if (mode == 6) return delta_sweep();
  if (mode == 1) malloc_pairs("first thing in the process, freed afterwards", 6, false);
  if (mode == 3) malloc_pairs("first thing in the process, KEPT", 6, true);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  printf("# granularity %zu, free %.1f GiB of %.1f GiB\n", gran, free_b / 1073741824.0, total_b / 1073741824.0);
  std::vector<hipMemGenericAllocationHandle_t> h;
  for (int k = 0; k < NC; k++) {
    hipMemGenericAllocationHandle_t hk;
    if (hipMemCreate(&hk, CH, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    h.push_back(hk);
  }
  NC = (int)h.size();
  printf("# %d chunks of 1 GiB created\n", NC);
  void* win = nullptr;
  CHECK(hipMemAddressReserve(&win, (size_t)NC * CH, 0, nullptr, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int k = 0; k < NC; k++) CHECK(hipMemMap((char*)win + (size_t)k * CH, CH, 0, h[k], 0));
  CHECK(hipMemSetAccess(win, (size_t)NC * CH, &acc, 1));
  auto chunk = [&](int k) { return (float*)((char*)win + (size_t)k * CH); };
  for (int k = 0; k < NC; k++) hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)chunk(k), CH / 16);
  CHECK(hipDeviceSynchronize());
  float* sink = nullptr;
  CHECK(hipMalloc(&sink, 256));
  // grades: 2048 streams x 128 Ki floats = 1 GiB
  const size_t gs = CH / 4 / 2048;
  const int ref = NC - 1;
  std::vector<float> gw(NC), gr(NC), gf(NC), go(NC);
  for (int k = 0; k < NC; k++) {
    const int other = k == ref ? ref - 1 : ref;
    gw[k] = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, chunk(other), chunk(k), gs); });
    gr[k] = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, chunk(k), chunk(other), gs); });
    gf[k] = timeit([&] { hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (f4v*)chunk(k), CH / 16); });
    go[k] = timeit([&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, (const f4v*)chunk(k), sink, CH / 16); });
  }
  printf("## grades per chunk, creation order (us): copy INTO it from the last chunk / copy FROM it / plain fill / plain read\n");
  for (int k = 0; k < NC; k++) {
    printf("%4d w %5.0f r %5.0f f %5.0f o %5.0f%s", k, gw[k] * 1e3, gr[k] * 1e3, gf[k] * 1e3, go[k] * 1e3, k % 4 == 3 ? "\n" : "   |");
  }
  printf("\n");
  // second pass of the write grade (is a grade stable?)
  std::vector<float> gw2(NC);
  for (int k = 0; k < NC; k++) {
    const int other = k == ref ? ref - 1 : ref;
    gw2[k] = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, chunk(other), chunk(k), gs); });
  }
  double d = 0;
  for (int k = 0; k < NC; k++) d = std::max(d, (double)fabsf(gw2[k] - gw[k]) / gw[k]);
  printf("## write grade, second pass: largest relative change %.3f\n", d);
  // compose buffers of C2's size: 4 chunks each (3.67 GiB = 2048 streams x 235 x 2048 floats fits 4 GiB)
  std::vector<int> by_w(NC), by_r(NC);
  std::iota(by_w.begin(), by_w.end(), 0);
  std::iota(by_r.begin(), by_r.end(), 0);
  std::sort(by_w.begin(), by_w.end(), [&](int a, int b) { return gw[a] < gw[b]; });
  std::sort(by_r.begin(), by_r.end(), [&](int a, int b) { return gr[a] < gr[b]; });
  void* comp = nullptr;
  CHECK(hipMemAddressReserve(&comp, 8 * CH, 0, nullptr, 0));
  auto compose = [&](const char* name, std::vector<int> src, std::vector<int> dst) {
    // chunks are taken out of the probe window and mapped side by side: in = comp[0, 4 GiB), out = comp[4, 8 GiB)
    std::vector<int> all = src;
    all.insert(all.end(), dst.begin(), dst.end());
    for (int i = 0; i < 8; i++) {
      CHECK(hipMemUnmap((char*)win + (size_t)all[i] * CH, CH));
      CHECK(hipMemMap((char*)comp + (size_t)i * CH, CH, 0, h[all[i]], 0));
    }
    CHECK(hipMemSetAccess(comp, 8 * CH, &acc, 1));
    const size_t stream_len = 235 * 2048;
    const float t = timeit([&] { hipLaunchKernelGGL((stream_copy<8>), dim3(2048), dim3(64), 0, 0, (const float*)comp, (float*)((char*)comp + 4 * CH), stream_len); }, 5);
    printf("  %-44s in {%d %d %d %d} out {%d %d %d %d}  %7.3f ms  %6.0f GB/s\n", name, src[0], src[1], src[2], src[3], dst[0], dst[1], dst[2], dst[3], t,
           2.0 * 2048 * stream_len * 4 / (t * 1e-3) / 1e9);
    fflush(stdout);
    CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 8; i++) {
      CHECK(hipMemUnmap((char*)comp + (size_t)i * CH, CH));
      CHECK(hipMemMap((char*)win + (size_t)all[i] * CH, CH, 0, h[all[i]], 0));
    }
    CHECK(hipMemSetAccess(win, (size_t)NC * CH, &acc, 1));
  };
  printf("## composed buffers, C2's shape (2 x 3.94 GB)\n");
  auto pick = [&](const std::vector<int>& order, int from, std::vector<int> avoid) {
    std::vector<int> r;
    for (int i = from; i < NC && r.size() < 4; i++)
      if (std::find(avoid.begin(), avoid.end(), order[i]) == avoid.end() && order[i] != ref) r.push_back(order[i]);
    return r;
  };
  {
    std::vector<int> bw = pick(by_w, 0, {});
    std::vector<int> br = pick(by_r, 0, bw);
    compose("best read -> best write", br, bw);
    std::vector<int> rw(by_w.rbegin(), by_w.rend());
    std::vector<int> ww = pick(rw, 0, {});
    std::vector<int> wr = pick(rw, 4, ww);
    compose("worst write -> worst write", wr, ww);
    compose("worst write -> best write", ww, bw);
    compose("best write -> best write (next four)", pick(by_w, 4, bw), bw);
    compose("creation order 0-3 -> 4-7", {0, 1, 2, 3}, {4, 5, 6, 7});
    if (NC >= 48) compose("creation order 40-43 -> 44-47", {40, 41, 42, 43}, {44, 45, 46, 47});
    // interleaved classes: does one slow chunk in four cost a quarter of the difference, or all of it?
    compose("3 best + 1 worst as destination", br, {bw[0], bw[1], bw[2], ww[0]});
  }
  if (mode) malloc_pairs("after the chunks (chunks alive)", 6, false);
  return 0;
}
