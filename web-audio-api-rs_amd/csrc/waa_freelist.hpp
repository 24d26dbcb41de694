// waa_freelist.hpp — first-fit free list with coalescing over one slab, offsets only (no HIP): the bookkeeping of the device
// arena (waa_arena.cpp).  Two directions: alloc() takes the LOWEST block that fits and carves from its start, alloc_top() the
// HIGHEST and carves from its end — a graded arena (waa_device_arena_reserve_graded) is sorted best-for-writing first, so what a
// batch writes comes from the bottom and what it only reads (source AudioBuffers) from the top.  Plain C++ so that tools/freelist_check.cpp exercises it on a box without a GPU
// (tests/test_arena_freelist.py).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <map>

namespace waa {
namespace host {

class FreeList {
 public:
  static constexpr size_t npos = ~(size_t)0;
  void reset(size_t bytes, size_t align) {
    align_ = align;
    size_ = bytes / align * align;
    free_.clear();
    used_.clear();
    in_use_ = peak_ = 0;
    served_ = misses_ = miss_bytes_ = 0;
    if (size_) free_[0] = size_;
  }
  size_t round(size_t bytes) const { return (bytes + align_ - 1) / align_ * align_; }
  // offset of a piece of round(bytes), or npos (counted as a miss)
  size_t alloc(size_t bytes) {
    const size_t need = std::max(round(bytes), align_);
    for (auto fb = free_.begin(); fb != free_.end(); ++fb) {
      if (fb->second < need) continue;
      const size_t off = fb->first, rest = fb->second - need;
      free_.erase(fb);
      if (rest) free_[off + need] = rest;
      used_[off] = need;
      in_use_ += need;
      peak_ = std::max(peak_, in_use_);
      served_++;
      return off;
    }
    misses_++;
    miss_bytes_ += need;
    return npos;
  }
  // the same from the other end: the highest free block that fits, its LAST round(bytes)
  size_t alloc_top(size_t bytes) {
    const size_t need = std::max(round(bytes), align_);
    for (auto fb = free_.rbegin(); fb != free_.rend(); ++fb) {
      if (fb->second < need) continue;
      const size_t start = fb->first, rest = fb->second - need, off = start + rest;
      if (rest)
        fb->second = rest;
      else
        free_.erase(std::next(fb).base());
      used_[off] = need;
      in_use_ += need;
      peak_ = std::max(peak_, in_use_);
      served_++;
      return off;
    }
    misses_++;
    miss_bytes_ += need;
    return npos;
  }
  // false if `off` is not the start of a live piece
  bool release(size_t off) {
    auto u = used_.find(off);
    if (u == used_.end()) return false;
    size_t lo = off, len = u->second;
    in_use_ -= len;
    used_.erase(u);
    auto next = free_.lower_bound(lo);
    if (next != free_.end() && next->first == lo + len) {  // merge with the free block behind
      len += next->second;
      next = free_.erase(next);
    }
    if (next != free_.begin()) {  // ... and with the one in front
      auto prev = std::prev(next);
      if (prev->first + prev->second == lo) {
        lo = prev->first;
        len += prev->second;
        free_.erase(prev);
      }
    }
    free_[lo] = len;
    return true;
  }
  size_t size() const { return size_; }
  size_t in_use() const { return in_use_; }
  size_t peak() const { return peak_; }
  size_t live() const { return used_.size(); }
  size_t free_blocks() const { return free_.size(); }
  size_t largest_free() const {
    size_t m = 0;
    for (const auto& fb : free_) m = std::max(m, fb.second);
    return m;
  }
  uint64_t served() const { return served_; }
  uint64_t misses() const { return misses_; }
  uint64_t miss_bytes() const { return miss_bytes_; }

 private:
  size_t size_ = 0, align_ = 1, in_use_ = 0, peak_ = 0;
  uint64_t served_ = 0, misses_ = 0, miss_bytes_ = 0;
  std::map<size_t, size_t> free_, used_;  // offset -> bytes
};

}  // namespace host
}  // namespace waa
