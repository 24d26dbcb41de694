// waa_arena.cpp — the optional per-device slab the big buffers of every batch are carved from (waa_device_arena_reserve,
// waa_device_arena_reserve_graded, include/waa_hip.h; dev_alloc in waa_host.hpp).
//
// The graded arena (round 6).  Which PHYSICAL memory a buffer lies in decides how fast a streaming kernel writes it: measured on
// MI355X with standalone probes (tools/alloc_probe.hip, tools/vmm_probe.hip; profiles/r06a...r06f): the one-wave-per-stream copy
// of C2's footprint takes 1.30-1.37 ms into some regions and 1.49-1.52 ms into others (which read slightly FASTER: 1.31 against
// 1.36 ms), whatever the virtual address (the same chunks at six address windows, any shift of either buffer: the same time),
// whatever the kernel shape (linear copy, 8 / 16-wave workgroups: the same two kinds), stable for the life of the allocation;
// regions are tens of GB wide and a fresh process is usually served from a slow-to-write one first — the "slow kind" of C2 that
// rounds 2-5 chased.  hipMalloc gives no say in it; the virtual-memory API does: the graded arena creates
// physical units (hipMemCreate), times the copy INTO each unit, and maps the units side by side sorted fastest-to-write first.
// Buffers a batch writes (signals, spectra, outputs) are carved from the bottom, buffers it only reads (source AudioBuffers)
// from the top (FreeList::alloc_top) — the slow-to-write units are the fast-to-read ones.  Surplus candidates are released.
#include <algorithm>
#include <chrono>
#include <iterator>
#include <map>
#include <mutex>
#include <numeric>
#include <vector>

#include "waa_freelist.hpp"
#include "waa_host.hpp"

// ---- device arena -------------------------------------------------------------------------------------------------
// A first-fit free list with coalescing (waa_freelist.hpp; round 5): the bump allocator of round 4 rewound only
// when NO piece was live, so a process whose batch lifetimes overlap (waa_render_sharded's pipeline, a server) never got a
// byte back and every later batch silently fell back to hipMalloc / hipFree — the device-wide synchronisation the arena exists
// to avoid.  Pieces are 2 MB-aligned; a freed piece merges with its free neighbours.  Misses (a request >= 1 MB the slab could
// not serve) are counted and readable through waa_device_arena_stats.
namespace waa {
namespace host {
namespace {
struct Arena {
  char* base = nullptr;
  size_t size = 0;
  FreeList list;
  // graded form: the slab is an address range with physical units mapped into it (empty: one hipMalloc)
  std::vector<hipMemGenericAllocationHandle_t> units;
  std::vector<float> grade_ms;  // per mapped unit, in address order (ascending = slower to write)
  size_t unit_bytes = 0;
  uint32_t candidates = 0;
  float worst_candidate_ms = 0.f, grading_ms = 0.f;
};
std::mutex g_arena_lock;
std::map<int, Arena> g_arenas;
constexpr size_t ARENA_ALIGN = 2u << 20;
}  // namespace
void* arena_alloc(int device, size_t bytes, bool read_only) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end() || !it->second.base) return nullptr;
  // (a plain slab has no better or worse end: everything from the bottom, as before)
  const size_t off = read_only && !it->second.units.empty() ? it->second.list.alloc_top(bytes) : it->second.list.alloc(bytes);
  return off == FreeList::npos ? nullptr : it->second.base + off;  // npos: does not fit (any more), the caller takes hipMalloc
}
size_t arena_unit_of(int device, const void* p) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end() || it->second.units.empty()) return 0;
  const Arena& a = it->second;
  const char* c = static_cast<const char*>(p);
  return c >= a.base && c < a.base + a.size ? a.unit_bytes : 0;
}
// ---- WAA_GUARD_ALLOC=1 (testing aid, round 6): every device buffer of a batch gets physical memory of its own, mapped so that the
// buffer ENDS where the mapping ends — the address behind it is reserved but unmapped, and a kernel that reads or writes one element
// past a buffer faults on the spot instead of reading its neighbour (hipMalloc sub-allocates: out-of-bounds reads are silent, and a
// fault only ever happened with several processes on the device, never twice in the same place).  tools/fuzz_crash_probe.py then names
// the graph.  16-byte placement granularity (what the kernels' vector loads need); 2 MiB of memory per buffer: small batches only.
namespace {
struct GuardPiece {
  void* va;
  size_t va_size, mapped;
  hipMemGenericAllocationHandle_t handle;
};
std::map<void*, GuardPiece> g_guard;
}  // namespace
void* guard_alloc(int device, size_t bytes) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) return nullptr;
  const size_t mapped = (bytes + gran - 1) / gran * gran;
  GuardPiece g{};
  g.mapped = mapped;
  g.va_size = mapped + gran;  // (one granule behind the mapping stays unmapped)
  if (hipMemAddressReserve(&g.va, g.va_size, gran, nullptr, 0) != hipSuccess) return nullptr;
  if (hipMemCreate(&g.handle, mapped, &prop, 0) != hipSuccess) {
    (void)hipMemAddressFree(g.va, g.va_size);
    return nullptr;
  }
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemMap(g.va, mapped, 0, g.handle, 0) != hipSuccess || hipMemSetAccess(g.va, mapped, &acc, 1) != hipSuccess) {
    (void)hipMemRelease(g.handle);
    (void)hipMemAddressFree(g.va, g.va_size);
    return nullptr;
  }
  char* p = static_cast<char*>(g.va) + mapped - (bytes + 15) / 16 * 16;
  (void)hipMemset(g.va, 0xFF, mapped);  // (what lies in FRONT of the buffer is poison)
  (void)hipDeviceSynchronize();
  g_guard[p] = g;
  if (getenv("WAA_ARENA_TRACE")) fprintf(stderr, "[waa guard] %zu bytes at %p, mapping ends at %p\n", bytes, (void*)p, (void*)(static_cast<char*>(g.va) + mapped));
  return p;
}
bool guard_free(void* p) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_guard.find(p);
  if (it == g_guard.end()) return false;
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(it->second.va, it->second.mapped);
  (void)hipMemRelease(it->second.handle);
  (void)hipMemAddressFree(it->second.va, it->second.va_size);
  g_guard.erase(it);
  return true;
}
bool arena_free(int device, void* p) {
  if (guard_free(p)) return true;
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end()) return false;
  Arena& a = it->second;
  if (!a.base || static_cast<char*>(p) < a.base || static_cast<char*>(p) >= a.base + a.size) return false;
  return a.list.release((size_t)(static_cast<char*>(p) - a.base));
}
}  // namespace host
}  // namespace waa

namespace {
void release_arena(waa::host::Arena& a) {
  if (!a.base) return;
  if (a.units.empty()) {
    (void)hipFree(a.base);
  } else {
    for (size_t k = 0; k < a.units.size(); k++) {
      (void)hipMemUnmap(a.base + k * a.unit_bytes, a.unit_bytes);
      (void)hipMemRelease(a.units[k]);
    }
    (void)hipMemAddressFree(a.base, a.size);
  }
  a = waa::host::Arena{};
}

// The grade: C2's access shape — one wavefront per stream, 8 KB tiles, the next tile requested before the stores of the one at
// hand, non-temporal both ways (biquad_stream_kernel_t's pipeline without its arithmetic) — from a fixed source unit into the
// unit under test.  2048 streams fill the device the way the render's kernels do.
typedef float arena_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void arena_grade_kernel(const float* __restrict__ in, float* __restrict__ out, size_t stream_len) {
  const int lane = threadIdx.x;
  constexpr int NV4 = 8, TILE = 64 * NV4 * 4;
  const size_t tiles = stream_len / TILE;
  const float* ip = in + (size_t)blockIdx.x * stream_len;
  float* op = out + (size_t)blockIdx.x * stream_len;
  arena_f4 cur[NV4], nxt[NV4];
#pragma unroll
  for (int j = 0; j < NV4; j++) cur[j] = __builtin_nontemporal_load((const arena_f4*)(ip + j * 256 + lane * 4));
  for (size_t t = 0; t < tiles; t++) {
    const size_t tn = t + 1 < tiles ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < NV4; j++) nxt[j] = __builtin_nontemporal_load((const arena_f4*)(ip + tn * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) __builtin_nontemporal_store(cur[j], (arena_f4*)(op + t * TILE + j * 256 + lane * 4));
#pragma unroll
    for (int j = 0; j < NV4; j++) cur[j] = nxt[j];
  }
}

// the caller's current HIP device is left as it was found
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace

extern "C" waa_status waa_device_arena_reserve(int32_t device, uint64_t bytes) {
  DeviceGuard guard;
  int dev = device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipSetDevice(dev));
  std::lock_guard<std::mutex> l(waa::host::g_arena_lock);
  auto& a = waa::host::g_arenas[dev];
  if (a.base) {
    if (a.list.live())
      return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - the arena of device %d is in use by %zu allocation(s)", dev, a.list.live());
    release_arena(a);
  }
  if (bytes == 0) {
    waa::host::g_arenas.erase(dev);
    return WAA_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    waa::host::g_arenas.erase(dev);
    return fail(WAA_ERR_DEVICE, "hipMalloc(%llu bytes) for the device arena failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
  }
  a.base = static_cast<char*>(p);
  a.size = bytes;
  a.list.reset(bytes, waa::host::ARENA_ALIGN);
  // WAA_POISON_ALLOC=1 (testing aid, see dev_alloc): the WHOLE slab starts as 0xFF bytes — what lies BEHIND a piece is then poison
  // as well, so that a kernel reading past the end of its buffer shows (round 6: a source read 56 frames behind its AudioBuffer for
  // three rounds; only stale memory behind the LAST plane of a batch, transformed by a convolver, ever made it visible)
  if (getenv("WAA_POISON_ALLOC")) {
    (void)hipMemset(a.base, 0xFF, bytes);
    (void)hipDeviceSynchronize();
  }
  return WAA_OK;
}

extern "C" waa_status waa_device_arena_reserve_graded(int32_t device, uint64_t bytes, uint64_t candidate_bytes) {
  using waa::host::Arena;
  DeviceGuard guard;
  int dev = device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipSetDevice(dev));
  std::lock_guard<std::mutex> l(waa::host::g_arena_lock);
  auto& a = waa::host::g_arenas[dev];
  if (a.base) {
    if (a.list.live())
      return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - the arena of device %d is in use by %zu allocation(s)", dev, a.list.live());
    release_arena(a);
  }
  if (bytes == 0) {
    waa::host::g_arenas.erase(dev);
    return WAA_OK;
  }
  const auto t_start = std::chrono::steady_clock::now();
  // units: 4 GiB — the footprint of C2's output (2048 streams x 235 tiles of 8 KB = 3.94 GB), so that a unit is graded with
  // EXACTLY the shape whose two kinds were measured (the regions seen are tens of GB wide; 2 GiB units graded with half-length
  // streams did not separate the kinds: profiles/r06l_c2_variants.txt) — smaller for small arenas (tests)
  size_t unit = 4ull << 30;
  while (unit > (32u << 20) && bytes < 4 * unit) unit >>= 1;
  const size_t n_keep = (size_t)((bytes + unit - 1) / unit);
  size_t n_cand = std::max<size_t>(n_keep, (size_t)(candidate_bytes / unit));
  {
    // never more than what is free minus a margin for everything that is not carved from the arena
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t margin = std::min<size_t>(free_b / 8, 8ull << 30);
      n_cand = std::min(n_cand, free_b > margin ? (free_b - margin) / unit : 0);
    }
  }
  if (n_cand < n_keep) {
    waa::host::g_arenas.erase(dev);
    return fail(WAA_ERR_DEVICE, "the device does not have %llu bytes free for the arena", (unsigned long long)bytes);
  }
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h;
  char* probe = nullptr;
  char* slab = nullptr;
  size_t mapped_probe = 0, mapped_slab = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto cleanup = [&](bool keep_slab) {
    for (size_t k = 0; k < mapped_probe; k++) (void)hipMemUnmap(probe + k * unit, unit);
    if (probe) (void)hipMemAddressFree(probe, n_cand * unit);
    if (!keep_slab) {
      for (size_t k = 0; k < mapped_slab; k++) (void)hipMemUnmap(slab + k * unit, unit);
      if (slab) (void)hipMemAddressFree(slab, n_keep * unit);
      for (auto hk : h) (void)hipMemRelease(hk);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  };
  auto bail = [&](const char* what, hipError_t e) {
    cleanup(false);
    (void)hipGetLastError();
    waa::host::g_arenas.erase(dev);
    return fail(WAA_ERR_DEVICE, "graded arena: %s failed: %s", what, hipGetErrorString(e));
  };
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < n_cand; k++) {
    hipMemGenericAllocationHandle_t hk;
    if ((e = hipMemCreate(&hk, unit, &prop, 0)) != hipSuccess) {
      (void)hipGetLastError();
      break;  // (what there is: fewer candidates)
    }
    h.push_back(hk);
  }
  if (h.size() < n_keep || h.size() < 2) return bail("hipMemCreate", e != hipSuccess ? e : hipErrorOutOfMemory);
  n_cand = h.size();
  if ((e = hipMemAddressReserve((void**)&probe, n_cand * unit, 0, nullptr, 0)) != hipSuccess) return bail("hipMemAddressReserve", e);
  for (size_t k = 0; k < n_cand; k++) {
    if ((e = hipMemMap(probe + k * unit, unit, 0, h[k], 0)) != hipSuccess) return bail("hipMemMap", e);
    mapped_probe++;
  }
  if ((e = hipMemSetAccess(probe, n_cand * unit, &acc, 1)) != hipSuccess) return bail("hipMemSetAccess", e);
  if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return bail("hipEventCreate", e);
  // every unit as the destination of the copy, from the last candidate (the last one itself: from the one before it)
  std::vector<float> grade(n_cand, 0.f);
  {
    // 2048 streams of an ODD number of 8 KB tiles: a power-of-two distance between the streams aliases them onto the same memory
    // channels and that pattern's own conflicts hide the region's (the first form of this grading used unit / 2048 = 1 MiB per
    // stream and saw 0.755-0.807 ms everywhere — profiles/r06g...r06k — while the full-size shape on the same memory split into
    // 1.30 and 1.56 ms); C2's own distance is 235 tiles
    constexpr size_t TILE = 2048;
    size_t tiles = unit / sizeof(float) / 2048 / TILE;
    if (tiles >= 256) tiles = 235;  // (a 4 GiB unit: C2's own stream length)
    if (tiles > 1 && tiles % 2 == 0) tiles--;
    const size_t stream_len = tiles * TILE;
    auto once = [&](size_t k) -> float {
      const size_t src = k + 1 == n_cand ? n_cand - 2 : n_cand - 1;
      (void)hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(arena_grade_kernel, dim3(2048), dim3(64), 0, nullptr, (const float*)(probe + src * unit), (float*)(probe + k * unit), stream_len);
      (void)hipEventRecord(e1, nullptr);
      if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      return ms;
    };
    (void)hipMemsetAsync(probe + (n_cand - 1) * unit, 0, unit, nullptr);  // (the sources hold defined values)
    (void)hipMemsetAsync(probe + (n_cand - 2) * unit, 0, unit, nullptr);
    for (size_t k = 0; k < n_cand; k++) {
      float best = once(k);  // (first touch of the unit: not counted unless it is all there is)
      for (int r = 0; r < 3 && best >= 0.f; r++) {
        const float t = once(k);
        best = r == 0 || (t >= 0.f && t < best) ? t : best;
      }
      if (best < 0.f) return bail("the grading launch", hipGetLastError());
      grade[k] = best;
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail("the grading launches", e);
  }
  if (getenv("WAA_ARENA_TRACE")) {  // (the candidates' grades in creation order, on stderr: where in the device's memory the kinds lie)
    fprintf(stderr, "[arena] %zu candidates of %zu MiB, copy into each (ms), creation order:", n_cand, unit >> 20);
    for (size_t k = 0; k < n_cand; k++) fprintf(stderr, " %.3f", grade[k]);
    fprintf(stderr, "\n");
  }
  std::vector<size_t> order(n_cand);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return grade[x] < grade[y]; });
  for (size_t k = 0; k < n_cand; k++) (void)hipMemUnmap(probe + k * unit, unit);
  mapped_probe = 0;
  if ((e = hipMemAddressReserve((void**)&slab, n_keep * unit, 0, nullptr, 0)) != hipSuccess) return bail("hipMemAddressReserve", e);
  for (size_t k = 0; k < n_keep; k++) {
    if ((e = hipMemMap(slab + k * unit, unit, 0, h[order[k]], 0)) != hipSuccess) return bail("hipMemMap", e);
    mapped_slab++;
  }
  if ((e = hipMemSetAccess(slab, n_keep * unit, &acc, 1)) != hipSuccess) return bail("hipMemSetAccess", e);
  a = Arena{};
  a.base = slab;
  a.size = n_keep * unit;
  a.unit_bytes = unit;
  a.candidates = (uint32_t)n_cand;
  a.worst_candidate_ms = grade[order[n_cand - 1]];
  for (size_t k = 0; k < n_cand; k++) {
    if (k < n_keep) {
      a.units.push_back(h[order[k]]);
      a.grade_ms.push_back(grade[order[k]]);
    } else {
      (void)hipMemRelease(h[order[k]]);
    }
  }
  h.clear();
  cleanup(true);
  a.list.reset(a.size, waa::host::ARENA_ALIGN);
  a.grading_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  return WAA_OK;
}

extern "C" waa_status waa_device_arena_grades(int32_t device, waa_arena_grades* out, float* unit_ms, uint32_t capacity) {
  if (!out) return fail(WAA_ERR_INVALID_ARGUMENT, "null grades");
  int dev = device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  *out = waa_arena_grades{};
  std::lock_guard<std::mutex> l(waa::host::g_arena_lock);
  auto it = waa::host::g_arenas.find(dev);
  if (it == waa::host::g_arenas.end() || it->second.units.empty()) return WAA_OK;  // no graded arena: all zero
  const auto& a = it->second;
  out->unit_bytes = a.unit_bytes;
  out->n_units = (uint32_t)a.units.size();
  out->n_candidates = a.candidates;
  out->best_ms = a.grade_ms.front();
  out->worst_kept_ms = a.grade_ms.back();
  out->worst_candidate_ms = a.worst_candidate_ms;
  out->grading_ms = a.grading_ms;
  if (unit_ms)
    for (uint32_t k = 0; k < capacity && k < a.grade_ms.size(); k++) unit_ms[k] = a.grade_ms[k];
  return WAA_OK;
}

extern "C" waa_status waa_device_arena_stats(int32_t device, waa_arena_stats* out) {
  if (!out) return fail(WAA_ERR_INVALID_ARGUMENT, "null stats");
  int dev = device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  *out = waa_arena_stats{};
  std::lock_guard<std::mutex> l(waa::host::g_arena_lock);
  auto it = waa::host::g_arenas.find(dev);
  if (it == waa::host::g_arenas.end()) return WAA_OK;  // no arena reserved: all zero
  const auto& a = it->second;
  out->reserved_bytes = a.size;
  out->in_use_bytes = a.list.in_use();
  out->peak_bytes = a.list.peak();
  out->largest_free_bytes = a.list.largest_free();
  out->served = a.list.served();
  out->misses = a.list.misses();
  out->miss_bytes = a.list.miss_bytes();
  return WAA_OK;
}
