// waa_arena.cpp — the optional per-device slab the big buffers of every batch are carved from (waa_device_arena_reserve,
// include/waa_hip.h; dev_alloc in waa_host.hpp).
#include <map>
#include <mutex>

#include "waa_host.hpp"

// ---- device arena -------------------------------------------------------------------------------------------------
namespace waa {
namespace host {
namespace {
struct Arena {
  char* base = nullptr;
  size_t size = 0, top = 0;
  int live = 0;
};
std::mutex g_arena_lock;
std::map<int, Arena> g_arenas;
constexpr size_t ARENA_ALIGN = 2u << 20;
}  // namespace
void* arena_alloc(int device, size_t bytes) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end()) return nullptr;
  Arena& a = it->second;
  const size_t need = (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  if (a.top + need > a.size) return nullptr;  // does not fit (any more): the caller takes hipMalloc
  void* p = a.base + a.top;
  a.top += need;
  a.live++;
  return p;
}
bool arena_free(int device, void* p) {
  std::lock_guard<std::mutex> l(g_arena_lock);
  auto it = g_arenas.find(device);
  if (it == g_arenas.end()) return false;
  Arena& a = it->second;
  if (static_cast<char*>(p) < a.base || static_cast<char*>(p) >= a.base + a.size) return false;
  if (--a.live == 0) a.top = 0;  // pieces are handed back when the last batch that holds one is gone
  return true;
}
}  // namespace host
}  // namespace waa

extern "C" waa_status waa_device_arena_reserve(int32_t device, uint64_t bytes) {
  int dev = device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipSetDevice(dev));
  std::lock_guard<std::mutex> l(waa::host::g_arena_lock);
  auto& a = waa::host::g_arenas[dev];
  if (a.base) {
    if (a.live) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - the arena of device %d is in use by %d allocation(s)", dev, a.live);
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
  return WAA_OK;
}

