// freelist_check.cpp — random alloc / release sequences on waa::host::FreeList (csrc/waa_freelist.hpp, the device arena's
// bookkeeping) against a byte-map model: no two live pieces overlap, everything stays inside the slab and aligned, a released
// piece is reusable at once (coalescing: after releasing everything ONE free block of the whole slab remains), misses are
// counted.  g++ only; run by tests/test_arena_freelist.py.  Prints "ok <ops> <misses>" or a diagnosis and exits 1.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../web-audio-api-rs_amd/csrc/waa_freelist.hpp"

using waa::host::FreeList;

int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
  const size_t align = 64, slab = 64 * 997 + 13;  // (the tail below one alignment unit is never handed out)
  FreeList fl;
  fl.reset(slab, align);
  if (fl.size() != 64 * 997) return printf("size %zu\n", fl.size()), 1;
  std::vector<int> owner(fl.size() / align, -1);
  struct Piece { size_t off, units; };
  std::vector<Piece> live;
  std::mt19937 rng(seed);
  uint64_t ops = 0, misses = 0;
  for (int it = 0; it < 200000; it++) {
    const bool do_alloc = live.empty() || (rng() % 100) < 55;
    if (do_alloc) {
      const size_t bytes = 1 + rng() % (align * 40);
      const size_t units = (bytes + align - 1) / align;
      const bool top = (rng() % 3) == 0;  // (a third of the requests from the top end, as payloads of a graded arena)
      const size_t off = top ? fl.alloc_top(bytes) : fl.alloc(bytes);
      ops++;
      if (off == FreeList::npos) {
        // a miss is only legal when no run of `units` free units exists (first fit finds any)
        size_t run = 0, best = 0;
        for (int o : owner) { run = o < 0 ? run + 1 : 0; best = std::max(best, run); }
        if (best >= units) return printf("miss of %zu units with a free run of %zu\n", units, best), 1;
        misses++;
        continue;
      }
      if (off % align || off + units * align > fl.size()) return printf("bad offset %zu\n", off), 1;
      for (size_t u = 0; u < units; u++) {
        if (owner[off / align + u] >= 0) return printf("overlap at unit %zu\n", off / align + u), 1;
        owner[off / align + u] = it;
      }
      live.push_back({off, units});
    } else {
      const size_t k = rng() % live.size();
      if (!fl.release(live[k].off)) return printf("release of a live piece refused\n"), 1;
      if (fl.release(live[k].off)) return printf("double release accepted\n"), 1;
      for (size_t u = 0; u < live[k].units; u++) owner[live[k].off / align + u] = -1;
      live[k] = live.back();
      live.pop_back();
      ops++;
    }
    size_t used = 0;
    for (const Piece& p : live) used += p.units * align;
    if (used != fl.in_use() || live.size() != fl.live()) return printf("accounting: %zu vs %zu\n", used, fl.in_use()), 1;
  }
  if (fl.misses() != misses) return printf("miss count %llu vs %llu\n", (unsigned long long)fl.misses(), (unsigned long long)misses), 1;
  for (const Piece& p : live) fl.release(p.off);
  if (fl.in_use() != 0 || fl.free_blocks() != 1 || fl.largest_free() != fl.size()) return printf("not coalesced: %zu blocks\n", fl.free_blocks()), 1;
  if (fl.release(0)) return printf("release of a free offset accepted\n"), 1;
  // the round-4 failure mode: overlapping lifetimes (a pipeline) never exhaust the slab
  fl.reset(10 * align, align);
  size_t prev = fl.alloc(4 * align);
  for (int i = 0; i < 1000; i++) {
    const size_t cur = fl.alloc(4 * align);  // (the next batch is created before the previous one is destroyed)
    if (cur == FreeList::npos) return printf("pipeline step %d missed\n", i), 1;
    fl.release(prev);
    prev = cur;
  }
  if (fl.misses()) return printf("pipeline misses\n"), 1;
  printf("ok %llu %llu\n", (unsigned long long)ops, (unsigned long long)misses);
  return 0;
}
