// waa_sharded.cpp — the N-device render component behind the C ABI (include/waa_hip.h: waa_render_sharded,
// waa_download_all_pcm16).  SURVEY.md section 8(e): the OfflineAudioContexts of a job are independent — contiguous instance
// ranges per device, no collective — and a caller that holds every context's AudioBuffer on the HOST is bound by the link,
// not by the render: each device's range is cut into sub-batches, one host thread per sub-batch,
//     upload(k + 1)  ||  render(k)  ||  download(k - 1)
// on each device, all devices in parallel; one transfer per direction and device at a time (the link is full duplex), handed
// on in sub-batch order by two turn counters per device.  This is what web-audio-api-rs_amd/sharding.py did in Python in
// round 3; that module is now a thin caller of this function, and a Rust or C host calls it directly.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <map>

#include "waa_host.hpp"
extern "C" int waa_internal_xfer_d2h(waa_batch* b, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);  // waa_abi.cpp

extern "C" int waa_settle_loops(waa_batch* b);  // waa_abi.cpp

namespace waa {
namespace host {
int fill_pending_uploads(waa_batch* b);
}
}  // namespace waa

namespace {

// sub-batches of one device take a direction of the link in index order
struct Turn {
  std::mutex m;
  std::condition_variable cv;
  uint32_t next = 0;
  void wait(uint32_t k) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return next == k; });
  }
  void done() {
    {
      std::lock_guard<std::mutex> l(m);
      next++;
    }
    cv.notify_all();
  }
};

struct Shard {
  uint32_t slot, k, lo, hi;
  int32_t device;
};

// At most `window` sub-batches of one device exist at a time (every sub-batch thread used to create, allocate and
// plan its batch at once, so the whole device range plus its PCM staging was resident together and a job that only fits when
// pipelined ran out of memory).  Sub-batch k is admitted once k - window + 1 of its predecessors are destroyed.
struct Window {
  std::mutex m;
  std::condition_variable cv;
  uint32_t finished = 0, limit = 4;
  void enter(uint32_t k) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return k < finished + limit; });
  }
  void leave() {
    {
      std::lock_guard<std::mutex> l(m);
      finished++;
    }
    cv.notify_all();
  }
};

// A callback reports through its return value only.  This thread's message is cleared before the call: if the callback failed
// inside a library call, that call's text is kept (prefixed); otherwise a message is written here — report() reads it from this
// thread, and a C or Rust host used to get a status with an empty (or stale) text.
int call_back(waa_shard_setup_fn fn, const char* which, waa_batch* b, const Shard& sh, void* user) {
  fail(0, "%s", "");
  const int st = fn(b, sh.lo, sh.hi - sh.lo, sh.device, user);
  if (!st) return 0;
  const std::string inner = waa_last_error();
  if (inner.empty()) return fail(st, "the %s callback of sub-batch [%u, %u) on device %d returned status %d", which, sh.lo, sh.hi, sh.device, st);
  return fail(st, "%s (in the %s callback of sub-batch [%u, %u) on device %d)", inner.c_str(), which, sh.lo, sh.hi, sh.device);
}

// sub-batches in flight per device: upload(k + 1) || render(k) || download(k - 1) needs three; the fourth prepares (creates,
// sets up, plans) while the third waits for its upload turn.  waa_sharded_in_flight changes it (0 = no bound).
std::atomic<uint32_t> g_in_flight{4};

// items lo .. hi of n split into `parts` contiguous ranges that differ by at most one (tests/test_multi_rank.py pins the rule)
void split_range(uint32_t n, uint32_t part, uint32_t parts, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = n / parts, rem = n % parts;
  *lo = part * base + std::min(part, rem);
  *hi = *lo + base + (part < rem ? 1u : 0u);
}

}  // namespace

extern "C" {

waa_status waa_shard_range(uint32_t n_total, uint32_t part, uint32_t n_parts, uint32_t* first, uint32_t* end) {
  if (!first || !end || n_parts == 0 || part >= n_parts) return fail(WAA_ERR_INVALID_ARGUMENT, "bad shard index %u of %u", part, n_parts);
  split_range(n_total, part, n_parts, first, end);
  return WAA_OK;
}

waa_status waa_sharded_in_flight(uint32_t max_sub_batches_per_device) {
  g_in_flight.store(max_sub_batches_per_device);
  return WAA_OK;
}

waa_status waa_download_all_pcm16(waa_batch* b, int16_t* dst) {
  if (!b || !dst) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch / destination");
  if (b->dry) return fail(WAA_ERR_DEVICE, "plan-only batch has no device");
  if (!b->planned || !b->rendered) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - nothing rendered yet");
  if (b->length == 0) return WAA_OK;
  HIP_TRY(hipSetDevice(b->device));
  if (int es = waa_settle_loops(b)) return es;
  const size_t count = (size_t)b->n_inst * b->length * b->n_out;
  if (!b->pcm_out) {  // (the batch's size never changes; owned by the batch like every other buffer: the device arena serves it when reserved)
    int e = dev_alloc(b, &b->pcm_out, count, true);
    if (e) return e;
    b->pcm_out_count = count;
  }
  const waa::SignalRef& s = b->nodes[0].sig;
  waa::EncodeDesc d{};
  d.in = s.base;
  d.pcm = b->pcm_out;
  d.frames = b->length;
  d.in_item_stride = s.inst_stride;
  d.in_ch_stride = s.ch_stride;
  d.nch_in = (uint32_t)s.nch;
  d.nch_out = b->n_out;
  d.n_items = b->n_inst;
  waa::launch_pcm16_pack(d, b->stream);
  HIP_TRY(hipGetLastError());
  if (int e = waa_internal_xfer_d2h(b, dst, count * sizeof(int16_t), b->pcm_out, count * sizeof(int16_t), count * sizeof(int16_t), 1)) return e;
  HIP_TRY(hipStreamSynchronize(b->stream));
  return WAA_OK;
}

waa_status waa_render_sharded(const waa_sharded_job* job, double* seconds) {
  if (!job || !job->graph) return fail(WAA_ERR_INVALID_ARGUMENT, "null job / graph");
  if (job->n_instances == 0) return fail(WAA_ERR_INVALID_ARGUMENT, "a sharded job needs at least one context");
  if (!job->devices || job->n_devices == 0) return fail(WAA_ERR_INVALID_ARGUMENT, "a sharded job needs at least one device");
  if (!job->host_out) return fail(WAA_ERR_INVALID_ARGUMENT, "null output buffer");
  const bool streamed = job->source_node != WAA_NO_NODE;
  if (streamed && (!job->host_in || job->in_channels == 0 || job->in_channels > WAA_MAX_CHANNELS))
    return fail(WAA_ERR_INVALID_ARGUMENT, "the streamed source needs host_in and 1..%d channels", WAA_MAX_CHANNELS);
  // contiguous instance ranges per entry of `devices` (a device may be listed twice), each cut into sub-batches
  std::vector<Shard> shards;
  const uint32_t sub = std::max<uint32_t>(1, job->sub_batches);
  for (uint32_t di = 0; di < job->n_devices; di++) {
    uint32_t lo, hi;
    split_range(job->n_instances, di, job->n_devices, &lo, &hi);
    const uint32_t parts = std::max<uint32_t>(1, std::min(sub, hi - lo));
    uint32_t k = 0;
    for (uint32_t p = 0; p < parts; p++) {
      uint32_t a, b2;
      split_range(hi - lo, p, parts, &a, &b2);
      if (b2 > a) shards.push_back(Shard{di, k++, lo + a, lo + b2, job->devices[di]});
    }
  }
  std::vector<Turn> up(job->n_devices), down(job->n_devices);
  std::vector<Window> window(job->n_devices);
  {
    const uint32_t lim = g_in_flight.load();
    for (auto& w : window) w.limit = lim ? lim : UINT32_MAX;
  }
  const size_t row_in = (size_t)job->in_channels * job->in_frames * (job->in_pcm16 ? sizeof(int16_t) : sizeof(float));
  const size_t row_out = (size_t)job->n_channels_out * job->length_frames * (job->out_pcm16 ? sizeof(int16_t) : sizeof(float));
  std::mutex err_lock;
  int first_status = WAA_OK;
  std::string first_error;
  auto report = [&](int st) {
    std::lock_guard<std::mutex> l(err_lock);
    if (first_status == WAA_OK) {
      first_status = st;
      first_error = waa_last_error();
    }
  };
  // WAA_SHARD_TRACE=1 (runtime switch, stderr): when every phase of every sub-batch started and ended, in ms from the call
  static const bool trace = getenv("WAA_SHARD_TRACE") != nullptr;
  const auto t00 = std::chrono::steady_clock::now();
  std::mutex trace_lock;
  auto stamp = [&](const Shard& sh, const char* what) {
    if (!trace) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t00).count();
    std::lock_guard<std::mutex> l(trace_lock);
    fprintf(stderr, "[shard %u.%u] %8.2f ms  %s\n", sh.slot, sh.k, ms, what);
  };
  // reuse_batches: downloaded sub-batches wait here, by (slot, contexts), for a later sub-batch of the same size (waa_batch_rearm)
  std::mutex pool_lock;
  std::map<std::pair<uint32_t, uint32_t>, std::vector<waa_batch*>> pool;
  auto run = [&](const Shard& sh) {
    waa_batch* b = nullptr;
    bool took_up = false, took_down = false;
    window[sh.slot].enter(sh.k);
    stamp(sh, "start");
    int st = WAA_OK;
    bool reused = false;
    if (job->reuse_batches) {
      std::lock_guard<std::mutex> l(pool_lock);
      auto& v = pool[{sh.slot, sh.hi - sh.lo}];
      if (!v.empty()) {
        b = v.back();
        v.pop_back();
        reused = true;
      }
    }
    if (reused) {
      st = waa_batch_rearm(b);
      stamp(sh, "re-armed");
    } else {
      st = waa_batch_create(job->graph, sh.hi - sh.lo, job->n_channels_out, job->length_frames, job->sample_rate, sh.device, &b);
      if (!st && job->setup) {
        st = call_back(job->setup, "setup", b, sh, job->user);
      }
      stamp(sh, "created + set up");
    }
    // The source's buffers are allocated and registered BEFORE the sub-batch's turn on the link, and the batch is planned: the
    // plan only needs their shape, and its small table uploads queue on the same DMA engine as the bulk upload of whichever
    // sub-batch holds the turn — planned after the upload, every render waited ~5 ms for its neighbour's transfer (WAA_SHARD_TRACE).
    // (Not when an AudioParam is modulated from the graph: that plan renders the modulating subgraph, which may read the source.)
    bool preplanned = false;
    if (!st && streamed) {
      bool modulated = false;
      for (uint32_t e = 0; e < job->graph->n_edges; e++) modulated |= (job->graph->edges[e].to_input & 0x80000000u) != 0;
      b->defer_fill = true;
      const char* src = static_cast<const char*>(job->host_in) + (size_t)sh.lo * row_in;
      st = job->in_pcm16 ? waa_source_set_buffer_pcm16_batch(b, job->source_node, reinterpret_cast<const int16_t*>(src), job->in_channels,
                                                             job->in_frames, job->in_sample_rate)
                         : waa_source_set_buffer_batch(b, job->source_node, reinterpret_cast<const float*>(src), job->in_channels,
                                                       job->in_frames, job->in_sample_rate);
      b->defer_fill = false;
      if (!st && !modulated && !reused) {
        st = waa_plan_describe(b, nullptr, 0, nullptr);
        preplanned = true;
      }
      stamp(sh, preplanned ? "buffers registered + planned" : "buffers registered");
    }
    if (!st) {
      up[sh.slot].wait(sh.k);
      took_up = true;
      stamp(sh, "upload turn");
      if (streamed) st = waa::host::fill_pending_uploads(b);
      up[sh.slot].done();
      stamp(sh, "uploaded");
    }
    if (!st) st = waa_render(b);
    if (!st) st = waa_sync(b);
    stamp(sh, "planned + rendered");
    if (!st && job->pull) {
      st = call_back(job->pull, "pull", b, sh, job->user);
    }
    if (!st) {
      down[sh.slot].wait(sh.k);
      took_down = true;
      stamp(sh, "download turn");
      char* dst = static_cast<char*>(job->host_out) + (size_t)sh.lo * row_out;
      st = job->out_pcm16 ? waa_download_all_pcm16(b, reinterpret_cast<int16_t*>(dst)) : waa_download_all(b, reinterpret_cast<float*>(dst));
      down[sh.slot].done();
      stamp(sh, "downloaded");
    }
    if (st) {
      report(st);
      // a failed sub-batch still takes and passes on its turns: the ones behind it must not wait forever
      if (!took_up) {
        up[sh.slot].wait(sh.k);
        up[sh.slot].done();
      }
      if (!took_down) {
        down[sh.slot].wait(sh.k);
        down[sh.slot].done();
      }
    }
    if (b && job->reuse_batches && !st) {
      std::lock_guard<std::mutex> l(pool_lock);
      pool[{sh.slot, sh.hi - sh.lo}].push_back(b);
      b = nullptr;
    }
    if (b) waa_batch_destroy(b);
    window[sh.slot].leave();
    stamp(sh, job->reuse_batches && !st ? "kept for re-use" : "destroyed");
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  threads.reserve(shards.size());
  for (const Shard& sh : shards) threads.emplace_back(run, std::cref(sh));
  for (auto& t : threads) t.join();
  for (auto& kv : pool)
    for (waa_batch* pb : kv.second) waa_batch_destroy(pb);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (first_status != WAA_OK) return fail(first_status, "%s", first_error.c_str());
  return WAA_OK;
}

}  // extern "C"
