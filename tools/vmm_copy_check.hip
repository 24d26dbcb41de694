// vmm_copy_check — which runtime copies work on memory mapped from several hipMemCreate handles side by side?  (measurement tool)
// The graded arena (csrc/waa_arena.cpp) maps physical units next to each other; kernels see plain addresses, but hipMemcpy* /
// hipMemset* resolve a pointer to the runtime's memory object first.  Checks, within one unit and across a unit boundary:
// hipMemcpy H2D / D2H, hipMemcpyAsync, hipMemcpy2DAsync D2H / H2D, hipMemsetAsync, hipMemcpyAsync D2D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void iota(unsigned* p, size_t n, unsigned salt) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned)i * 2654435761u + salt;
}
__global__ void check(const unsigned* p, size_t n, unsigned salt, unsigned* bad) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (p[i] != (unsigned)i * 2654435761u + salt) atomicAdd(bad, 1u);
}
int main(int argc, char** argv) {
  const size_t U = 64u << 20;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  char* w = nullptr;
  const int NU = 12;
  const bool remap = argc > 1;  // the graded arena's flow: mapped in a probe window first, then unmapped and mapped elsewhere in another order
  hipMemGenericAllocationHandle_t h[NU];
  for (int k = 0; k < NU; k++) CHECK(hipMemCreate(&h[k], U, &prop, 0));
  if (remap) {
    char* probe = nullptr;
    CHECK(hipMemAddressReserve((void**)&probe, NU * U, 0, nullptr, 0));
    for (int k = 0; k < NU; k++) CHECK(hipMemMap(probe + k * U, U, 0, h[k], 0));
    CHECK(hipMemSetAccess(probe, NU * U, &acc, 1));
    CHECK(hipMemset(probe, 1, NU * U));
    CHECK(hipDeviceSynchronize());
    for (int k = 0; k < NU; k++) CHECK(hipMemUnmap(probe + k * U, U));
    CHECK(hipMemAddressFree(probe, NU * U));
    printf("# remapped flow\n");
  }
  CHECK(hipMemAddressReserve((void**)&w, NU * U, 0, nullptr, 0));
  for (int k = 0; k < NU; k++) CHECK(hipMemMap(w + k * U, U, 0, h[remap ? (k * 5) % NU : k], 0));
  CHECK(hipMemSetAccess(w, NU * U, &acc, 1));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* bad = nullptr;
  CHECK(hipMalloc(&bad, 4));
  const size_t N = 24u << 20;  // bytes per test region
  char *hp = nullptr, *hq = nullptr;
  CHECK(hipHostMalloc(&hp, N));
  CHECK(hipHostMalloc(&hq, N));
  std::vector<char> pageable(N);
  auto dev_matches = [&](const char* d, unsigned salt) {
    CHECK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, (const unsigned*)d, N / 4, salt, bad);
    unsigned b = 0;
    CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
    return b == 0;
  };
  auto host_fill = [&](char* h, unsigned salt) { for (size_t i = 0; i < N / 4; i++) ((unsigned*)h)[i] = (unsigned)i * 2654435761u + salt; };
  auto host_matches = [&](const char* h, unsigned salt) { for (size_t i = 0; i < N / 4; i++) if (((const unsigned*)h)[i] != (unsigned)i * 2654435761u + salt) return false; return true; };
  struct Place { const char* name; size_t off; } places[] = {{"inside unit 0", 8u << 20}, {"across units 0|1", U - (12u << 20)}, {"across 1|2, odd offset", 2 * U - (5u << 20) - 4096}};
  unsigned salt = 1;
  for (auto& pl : places) {
    char* d = w + pl.off;
    auto report = [&](const char* what, hipError_t e, bool ok) { printf("  %-26s %-34s %s%s\n", pl.name, what, e != hipSuccess ? hipGetErrorString(e) : (ok ? "ok" : "WRONG DATA"), ""); (void)hipGetLastError(); };
    hipError_t e;
    // H2D sync, pageable
    salt++; host_fill(pageable.data(), salt);
    e = hipMemcpy(d, pageable.data(), N, hipMemcpyHostToDevice); CHECK(hipDeviceSynchronize());
    report("hipMemcpy H2D pageable", e, e == hipSuccess && dev_matches(d, salt));
    // H2D async pinned
    salt++; host_fill(hp, salt);
    e = hipMemcpyAsync(d, hp, N, hipMemcpyHostToDevice, s); CHECK(hipStreamSynchronize(s));
    report("hipMemcpyAsync H2D pinned", e, e == hipSuccess && dev_matches(d, salt));
    // D2H sync / async
    salt++; hipLaunchKernelGGL(iota, dim3(1024), dim3(256), 0, 0, (unsigned*)d, N / 4, salt); CHECK(hipDeviceSynchronize());
    memset(hq, 0, N);
    e = hipMemcpyAsync(hq, d, N, hipMemcpyDeviceToHost, s); CHECK(hipStreamSynchronize(s));
    report("hipMemcpyAsync D2H pinned", e, e == hipSuccess && host_matches(hq, salt));
    memset(pageable.data(), 0, N);
    e = hipMemcpy(pageable.data(), d, N, hipMemcpyDeviceToHost);
    report("hipMemcpy D2H pageable", e, e == hipSuccess && host_matches(pageable.data(), salt));
    // 2D D2H: rows of 1 MB out of a pitch of 1.5 MB -> dense host rows (16 rows)
    {
      const size_t row = 1u << 20, pitch = 3u << 19, rows = 16;
      memset(hq, 0, N);
      e = hipMemcpy2DAsync(hq, row, d, pitch, row, rows, hipMemcpyDeviceToHost, s); CHECK(hipStreamSynchronize(s));
      bool ok = e == hipSuccess;
      for (size_t r = 0; ok && r < rows; r++)
        for (size_t i = 0; i < row / 4; i += 997) ok = ok && ((unsigned*)(hq + r * row))[i] == (unsigned)(r * pitch / 4 + i) * 2654435761u + salt;
      report("hipMemcpy2DAsync D2H", e, ok);
      // 2D H2D
      salt++; host_fill(hp, salt);
      e = hipMemcpy2DAsync(d, pitch, hp, row, row, rows, hipMemcpyHostToDevice, s); CHECK(hipStreamSynchronize(s));
      std::vector<unsigned> back(N / 4);
      CHECK(hipMemcpy(back.data(), d, N, hipMemcpyDeviceToHost));
      ok = e == hipSuccess;
      for (size_t r = 0; ok && r < rows; r++)
        for (size_t i = 0; i < row / 4; i += 997) ok = ok && back[r * pitch / 4 + i] == (unsigned)(r * row / 4 + i) * 2654435761u + salt;
      report("hipMemcpy2DAsync H2D", e, ok);
    }
    // memset async
    e = hipMemsetAsync(d, 0x5A, N, s); CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(pageable.data(), d, N, hipMemcpyDeviceToHost));
    { bool ok = true; for (size_t i = 0; i < N; i += 4093) ok = ok && (unsigned char)pageable[i] == 0x5A; report("hipMemsetAsync", e, e == hipSuccess && ok && (unsigned char)pageable[N - 1] == 0x5A); }
    // D2D from plain hipMalloc memory and back
    {
      char* plain = nullptr; CHECK(hipMalloc(&plain, N));
      salt++; hipLaunchKernelGGL(iota, dim3(1024), dim3(256), 0, 0, (unsigned*)plain, N / 4, salt); CHECK(hipDeviceSynchronize());
      e = hipMemcpyAsync(d, plain, N, hipMemcpyDeviceToDevice, s); CHECK(hipStreamSynchronize(s));
      report("hipMemcpyAsync D2D into it", e, e == hipSuccess && dev_matches(d, salt));
      CHECK(hipMemset(plain, 0, N));
      e = hipMemcpyAsync(plain, d, N, hipMemcpyDeviceToDevice, s); CHECK(hipStreamSynchronize(s));
      report("hipMemcpyAsync D2D out of it", e, e == hipSuccess && dev_matches(plain, salt));
      CHECK(hipFree(plain));
    }
    hipPointerAttribute_t at{};
    e = hipPointerGetAttributes(&at, d);
    printf("  %-26s hipPointerGetAttributes: %s type %d\n", pl.name, hipGetErrorString(e), (int)at.type); (void)hipGetLastError();
  }
  // the download of a sub-batch: 256 rows of 1 920 000 bytes out of a pitch of 1 925 120, across many units, on a stream
  {
    const size_t row = 1920000, pitch = 1925120, rows = 256;
    char* d = w + (3u << 20);
    char* hb = nullptr;
    CHECK(hipHostMalloc(&hb, row * rows));
    hipLaunchKernelGGL(iota, dim3(4096), dim3(256), 0, 0, (unsigned*)d, pitch * rows / 4, 77u);
    CHECK(hipDeviceSynchronize());
    hipError_t e = hipMemcpy2DAsync(hb, row, d, pitch, row, rows, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    bool ok = e == hipSuccess && e2 == hipSuccess;
    for (size_t r = 0; ok && r < rows; r++)
      for (size_t i = 0; i < row / 4; i += 9973) ok = ok && ((unsigned*)(hb + r * row))[i] == (unsigned)(r * pitch / 4 + i) * 2654435761u + 77u;
    printf("  big 2D D2H (493 MB over %zu units): %s / %s %s\n", (pitch * rows + U - 1) / U, hipGetErrorString(e), hipGetErrorString(e2), ok ? "ok" : "WRONG DATA");
    (void)hipGetLastError();
    e = hipMemcpy2DAsync(d, pitch, hb, row, row, rows, hipMemcpyHostToDevice, s);
    e2 = hipStreamSynchronize(s);
    printf("  big 2D H2D: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
    (void)hipGetLastError();
    // the same download in pieces whose extent stays below half a unit
    {
      memset(hb, 0, row * rows);
      const size_t per = (U / 2) / pitch;
      hipError_t ec = hipSuccess;
      for (size_t r0 = 0; r0 < rows && ec == hipSuccess; r0 += per)
        ec = hipMemcpy2DAsync(hb + r0 * row, row, d + r0 * pitch, pitch, row, std::min(per, rows - r0), hipMemcpyDeviceToHost, s);
      e2 = hipStreamSynchronize(s);
      ok = ec == hipSuccess && e2 == hipSuccess;
      for (size_t r = 0; ok && r < rows; r++)
        for (size_t i = 0; i < row / 4; i += 9973) ok = ok && ((unsigned*)(hb + r * row))[i] == (unsigned)(r * pitch / 4 + i) * 2654435761u + 77u;
      printf("  big 2D D2H in pieces of %zu rows: %s / %s %s\n", per, hipGetErrorString(ec), hipGetErrorString(e2), ok ? "ok" : "WRONG DATA");
      (void)hipGetLastError();
    }
    // 1D copies and a memset larger than a unit
    {
      const size_t big = 3 * U + (5u << 20);
      char* hbig = nullptr;
      CHECK(hipHostMalloc(&hbig, big));
      for (size_t i = 0; i < big / 4; i++) ((unsigned*)hbig)[i] = (unsigned)i * 2654435761u + 99u;
      e = hipMemcpyAsync(d, hbig, big, hipMemcpyHostToDevice, s);
      e2 = hipStreamSynchronize(s);
      CHECK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, (const unsigned*)d, big / 4, 99u, bad);
      unsigned nb = 0;
      CHECK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      printf("  1D H2D of 3 units + 5 MB: %s / %s %s\n", hipGetErrorString(e), hipGetErrorString(e2), nb == 0 ? "ok" : "WRONG DATA");
      (void)hipGetLastError();
      memset(hbig, 0, big);
      e = hipMemcpyAsync(hbig, d, big, hipMemcpyDeviceToHost, s);
      e2 = hipStreamSynchronize(s);
      bool okb = true;
      for (size_t i = 0; i < big / 4; i += 1013) okb = okb && ((unsigned*)hbig)[i] == (unsigned)i * 2654435761u + 99u;
      printf("  1D D2H of 3 units + 5 MB: %s / %s %s\n", hipGetErrorString(e), hipGetErrorString(e2), okb ? "ok" : "WRONG DATA");
      (void)hipGetLastError();
      e = hipMemsetAsync(d, 0x33, big, s);
      e2 = hipStreamSynchronize(s);
      CHECK(hipMemcpy(hbig, d, big, hipMemcpyDeviceToHost));
      okb = true;
      for (size_t i = 0; i < big; i += 4093) okb = okb && (unsigned char)hbig[i] == 0x33;
      printf("  memset of 3 units + 5 MB: %s / %s %s\n", hipGetErrorString(e), hipGetErrorString(e2), okb && (unsigned char)hbig[big - 1] == 0x33 ? "ok" : "WRONG DATA");
      (void)hipGetLastError();
      std::vector<char> pg(big);
      for (size_t i = 0; i < big / 4; i++) ((unsigned*)pg.data())[i] = (unsigned)i * 2654435761u + 55u;
      e = hipMemcpy(d, pg.data(), big, hipMemcpyHostToDevice);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, (const unsigned*)d, big / 4, 55u, bad);
      CHECK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      printf("  1D pageable H2D of 3 units + 5 MB: %s %s\n", hipGetErrorString(e), nb == 0 ? "ok" : "WRONG DATA");
      (void)hipGetLastError();
    }
    // the last row ends exactly at the end of the mapping (pitch * rows would reach beyond it)
    char* dend = w + NU * U - ((rows - 1) * pitch + row);
    e = hipMemcpy2DAsync(hb, row, dend, pitch, row, rows, hipMemcpyDeviceToHost, s);
    e2 = hipStreamSynchronize(s);
    printf("  big 2D D2H ending at the end of the range: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
    (void)hipGetLastError();
  }
  return 0;
}
