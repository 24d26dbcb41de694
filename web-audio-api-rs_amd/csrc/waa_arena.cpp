// waa_arena.cpp — the optional per-device slab the big buffers of every batch are carved from (waa_device_arena_reserve,
// include/waa_hip.h; dev_alloc in waa_host.hpp).
#include <algorithm>
#include <iterator>
#include <map>
#include <mutex>

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
};
std::mutex g_arena_lock;
std::map<int, Arena> g_arenas;
constexpr size_t ARENA_ALIGN = 2u << 20;
}  // namespace
void* arena_alloc(int device, size_t bytes) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end() || !it->second.base) return nullptr;
  const size_t off = it->second.list.alloc(bytes);
  return off == FreeList::npos ? nullptr : it->second.base + off;  // npos: does not fit (any more), the caller takes hipMalloc
}
bool arena_free(int device, void* p) {
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
    (void)hipFree(a.base);
    a = waa::host::Arena{};
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
