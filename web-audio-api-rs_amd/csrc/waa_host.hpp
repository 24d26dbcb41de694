// waa_host.hpp — host-side data model of libwaa_hip.so shared by the scheduler (waa_schedule.cpp), the planner
// (waa_plan.cpp) and the C ABI (waa_abi.cpp).  Not part of the public interface (include/waa_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/waa_hip.h"
#include "waa_internal.hpp"

struct waa_batch;

namespace waa {
namespace host {

// last error of the calling thread (waa_last_error); returns `code`
int fail(int code, const char* fmt, ...);
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(WAA_ERR_DEVICE, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, \
                                      __LINE__, #expr);                                         \
  } while (0)

// AudioParam automation timeline (waa_automation.cpp): the reference's AudioParamProcessor on the host
class Timeline {
 public:
  Timeline(float default_value, float min_value, float max_value, bool a_rate);
  ~Timeline();
  // one automation method call (WAA_EVENT_*); returns a waa_status (the reference's panics)
  int schedule(int type, float value, double time, double aux, const float* curve, uint32_t n_curve);
  // the same call made while the render is suspended in front of quantum `arrival_q` (OfflineAudioContext::suspend_sync,
  // offline.rs:359-397): the render thread handles the control message right before it renders that quantum
  // (handle_control_messages, thread.rs:277-294) — the event enters the queue THEN, with the queue as it is then.  Argument
  // errors are reported at once, queue-dependent ones (an event inside a value curve) by apply_arrivals.
  int schedule_at(uint32_t arrival_q, int type, float value, double time, double aux, const float* curve, uint32_t n_curve);
  int apply_arrivals(uint32_t q);  // before compute() of quantum q
  bool has_arrivals() const { return !arrivals_.empty(); }
  // one block: 1 or `count` values into out (capacity >= count); returns how many
  uint32_t compute(double block_time, double dt, uint32_t count, float* out);
  float value() const;  // AudioParam::value()
  // the scheduled queue in the form timeline_kernel replays (waa_timeline.hip); `ev_off` / curve offsets are relative
  // to the arrays the events are appended to
  void export_queue(TlHeader* hdr, std::vector<TlEvent>* events, std::vector<float>* curves) const;

 private:
  struct Event;
  int insert(Event ev);
  float min_, max_, intrinsic_, current_;
  bool a_rate_;
  std::vector<Event> queue_;     // sorted by time (stable)
  std::unique_ptr<Event> last_;  // the last event that was consumed (start point of ramps and targets)
  struct Arrival {
    uint32_t q;
    int type;
    float value;
    double time, aux;
    std::vector<float> curve;
  };
  std::vector<Arrival> arrivals_;  // in call order (arrival quanta never decrease: the control clock only moves forward)
  size_t next_arrival_ = 0;
};

struct ParamBlock {
  uint32_t inst;
  uint64_t q0;
  uint32_t nq, vpq;
  std::vector<float> v;
};
struct ParamStore {
  std::vector<float> cst;
  std::vector<ParamBlock> blocks;
  std::vector<std::shared_ptr<Timeline>> timelines;  // [n_inst] or empty: scheduled automation (waa_param_schedule_event)
  bool k_rate = false;                               // AutomationRate::K (source playbackRate / detune)
  bool timelines_shared = true;                      // every event was scheduled for WAA_ALL_INSTANCES
  bool dev_tl = false;                               // the timelines are replayed on the device (per-instance automation)
  mutable bool dev_ready = false;                    // ... and their step has been planned:
  mutable ParamRef dev_ref{};                        //     per-frame values [n_inst][n_quanta * 128]
  mutable uint8_t* dev_lens = nullptr;               //     slice length per (instance, quantum)
  float defv = 0, minv = -FLT_MAX, maxv = FLT_MAX;
  void init(uint32_t n, float d, float lo, float hi) {
    cst.assign(n, d);
    defv = d;
    minv = lo;
    maxv = hi;
  }
  // AudioParamProcessor::mix_to_output clamp / NaN rule (param.rs:739-797)
  float fix(float x) const { return std::isnan(x) ? defv : std::fmin(std::fmax(x, minv), maxv); }
  int mode() const {
    if (dev_tl) return 2;
    int m = 0;
    for (auto& b : blocks) m = std::max(m, b.vpq == 1 ? 1 : 2);
    return m;
  }
};

// double-double (unevaluated sum hi + lo, ~106 bits) for the IIR transition-matrix powers
struct DD {
  double hi = 0., lo = 0.;
};
inline DD dd_add(DD a, DD b) {
  const double s = a.hi + b.hi, bb = s - a.hi;
  double e = (a.hi - (s - bb)) + (b.hi - bb);
  e += a.lo + b.lo;
  const double hi = s + e;
  return DD{hi, e - (hi - s)};
}
inline DD dd_mul(DD a, DD b) {
  const double p = a.hi * b.hi;
  double e = std::fma(a.hi, b.hi, -p);
  e += a.hi * b.lo + a.lo * b.hi;
  const double hi = p + e;
  return DD{hi, e - (hi - p)};
}

struct DeviceBuffer {  // an AudioBuffer resident in HBM
  float* base = nullptr;  // channel 0
  uint64_t ch_stride = 0;
  uint64_t frames = 0;
  uint32_t nch = 0;
  uint32_t nch_true = 0;  // != 0: the AudioBuffer's own channel count; `nch` is the widened copy's (widen_narrow_buffers)
  float sr = 0;
  bool valid = false;
  uint32_t count() const { return nch_true ? nch_true : nch; }
};

struct SourceSched {  // per instance scheduling parameters
  double start = DBL_MAX, stop = DBL_MAX, offset = 0, duration = DBL_MAX;
  int looping = 0;
  double loop_start = 0, loop_end = 0;
};

struct Node {
  waa_node_desc desc{};
  int cc = 2, mode = WAA_COUNT_MODE_MAX, interp = WAA_INTERP_SPEAKERS;
  std::vector<ParamStore> params;
  // sources
  std::vector<DeviceBuffer> bufs;   // [n_inst]
  std::vector<SourceSched> sched;   // [n_inst]
  // convolver
  std::vector<std::vector<float>> ir;  // host copy, scaled
  uint64_t ir_len = 0;
  int ir_nch = 0;
  bool has_ir = false;
  // the impulse response with the transfer function of the Biquad in front folded in (conv_fold_biquad_into_ir); empty = not folded
  std::vector<std::vector<float>> ir_lti;
  uint64_t ir_lti_len = 0;
  // waveshaper
  std::vector<float> curve;
  bool has_curve = false;
  float* d_curve = nullptr;
  // oscillator: custom PeriodicWave table (8192 points, periodic_wave.rs:76)
  std::vector<float> osc_wave;
  // iir filter: normalised coefficient pairs (iir_filter.rs:273-311)
  std::vector<double> iir_b, iir_a;
  // analyser (control side state): the pulls of ALL instances are computed by one launch and cached until the next render
  // (current_time after an offline render never changes: repeated pulls return the same data, analysis.rs:354-357)
  struct AnBatch {
    bool computed = false;              // device results are current
    bool have_db = false, have_bytes = false, have_time = false;  // ... and downloaded
    std::vector<float> db, time;        // [n_inst][fft_size / 2], [n_inst][fft_size]
    std::vector<uint8_t> bytes;         // [n_inst][fft_size / 2]
  } an;
  float* d_window = nullptr;
  Cplx *d_an_tw = nullptr, *d_an_twfull = nullptr;
  float *d_an_prev = nullptr, *d_an_db = nullptr, *d_an_time = nullptr;
  uint8_t* d_an_bytes = nullptr;
  // planning
  int in_nch = 1;      // computed input channel count
  int out_nch = 1;     // static output channel count
  bool live = false;
  bool materialized = false;
  SignalRef sig{};     // valid when materialized
  SignalRef hist{};    // DelayNode: the delay line (the node's mixed input, absolute time)
  bool hist_is_temp = false;
  std::vector<int> in_edges;   // indices into edges, in summing order
  // audio-rate inputs of this node's AudioParams (edges with to_input = WAA_PARAM_INPUT(k)), in summing order,
  // and the per-frame value signal planned for them (param.rs:686-795)
  std::vector<std::vector<int>> pin_edges;
  std::vector<ParamRef> pin_ref;
  std::vector<char> pin_ready;
  int n_consumers = 0;
  // a BufferSource that renders its AudioBuffer unchanged from frame 0 (fast track, rate 1, no loop, same layout for
  // every instance) and whose only consumer is a node-major step that takes a bounded view: it is not materialised,
  // the consumer reads the buffer in place (view_valid frames per channel, zeros beyond)
  bool is_view = false;
  // BiquadFilterNode with constant coefficients whose only consumer is a long ConvolverNode (three-pass transforms): the
  // forward transform's input stage renders it (fold_conv = the convolver; the convolver's pre_biquad = this node)
  int fold_conv = -1, pre_biquad = -1;
  // DelayNode outside a loop, constant / k-rate delayTime, consumed by chain input stages only: no reader pass, the
  // consumers gather from the delay line themselves (IN_DELAYED)
  bool delay_folded = false;
  int osc_step = -1;  // OscillatorNode: index of its launch in the plan (post ops may be folded into it later)
  uint64_t hist_valid = 0;  // frames of the delay line that may be read (zeros beyond): the padded length, or a source view's
  SignalRef view_sig{};
  uint64_t view_valid = 0;
  // dynamic plans (waa_dyn.hip): per-quantum codes of the published signal, and the quantum slot of channel 1 when the
  // signal feeds a mono-IR convolver whose second FFTConvolver only advances on stereo quanta
  uint8_t* code = nullptr;
  uint8_t* in_code = nullptr;  // ConvolverNode: codes of its mixed input
  uint32_t* remap = nullptr;
};

struct ProfileEntry {
  std::string name;
  uint64_t launches = 0;
  double total_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Step {
  int kind = 0;  // 15 link table of a frozen-state node, 16 one resampling stage of an oversampled WaveShaper (qgemm_kernel), 17 HRTF FIR; 0 chain (interpreter kernel), 1 streaming biquad kernel, 2 FFT convolver, 3 zero-fill, 4 direct FIR, 5 per-frame biquad coefficients, 6 streaming IIR kernel, 7 delay gather, 8 feedback loop, 9 oscillator, 10 dynamic-count group (dyn_kernel), 11 convolver codes, 12 digest of a shared per-frame coefficient table, 13 per-frame panner geometry, 14 automation timelines replayed on the device
  ChainDesc chain{};
  BiquadStreamDesc bq{};
  BiquadScanCtl scan{};   // kind 1 with scan.payload: the time-parallel form (waa_biquad_scan.hip)
  BiquadLanesDesc lanes{};  // kinds 18 (tile digests) and 19 (pass A + chain + pass B), waa_biquad_lanes.hip
  ConvDesc conv{};
  BiquadCoefDesc coef{};
  IirStreamDesc iir{};
  DelayDesc delay{};
  LoopDesc loop{};
  OscDesc osc{};
  DynDesc dyn{};
  ConvCodeDesc ccode{};
  BiquadHpDesc hp{};
  PannerGeomDesc geom{};
  TimelineDesc tl{};
  LinkDesc link{};
  QGemmDesc qgemm{};
  OsFftDesc osfft{};      // kind 20: the oversampled WaveShaper in one launch (waa_osfft.hip)
  HrtfDesc hrtf{};
  int slot_fwd = -1, slot_mac = -1, slot_inv = -1;
  void* zero_ptr = nullptr;
  size_t zero_bytes = 0;
  int cmax = 1;
  int profile_slot = -1;
  // plan validation (host only): signals and per-frame param tables this step reads / writes; a quantum-serial loop
  // step lists the external signals its items read and the signals they publish
  std::vector<const void*> loop_reads, loop_writes;
  int group = -1;         // >= 0: member of a block-scheduled feedback loop (launched block by block)
  int qgroup = -1;        // >= 0: member of a feedback loop that is cut at frozen-state nodes and launched QUANTUM BLOCK by quantum block
                          // (dynamic-count plans, round 5: ranged dyn_kernel / link / oversampling / HRTF launches; waa_batch::qgroup_quanta)
  bool prologue = false;  // inside a group: runs once over the full range before the blocks
  // the only body step of a block-scheduled loop, and the loop qualifies for the LDS-ring kernel (waa_echo.hip): index of
  // the feedback input (-1: no) and the chunk size in 256-frame sub-tiles
  int echo_fb = -1, echo_chunk = 0, echo_ring = 16384;  // (echo_ring: frames per channel of the LDS ring)
  // ... or the LAST of three body steps (delayed read -> streaming biquad -> sum) rendered as the ring kernel's BQ form
  // (echo_bq.coefs != nullptr); the first two are marked echo_fused
  EchoBq echo_bq{};
  // ... and the line's only reader outside the loop rendered by the same launch (EchoTail): on the loop step, index of that
  // step and its descriptor; on that step itself, echo_fused = it is not launched
  int echo_tail_step = -1;
  EchoTail echo_tail{};
  bool echo_fused = false;
  // a chain step outside any loop rendered by the ring kernel with nothing fed back (echo_feed_forward): the stand-in loop
  // stage; echo_tail / echo_chunk as above
  bool echo_ff = false;
  ChainDesc echo_line{};
};

struct SchedOut {
  std::vector<QRec> qrec;
  std::vector<SlowRec> slow;  // empty if no slow quantum
  std::vector<uint8_t> tile_fast;
  bool any_slow = false;
  int64_t ended_quantum = -1;  // quantum in which the renderer sends `ended` (-1: not during the render)
  bool ended_at_unload = false;  // ... or before_drop sends it after the last quantum
};
using SchedKey = std::tuple<double, double, double, double, int, double, double, uint64_t, float, float, float>;

}  // namespace host
}  // namespace waa

using namespace waa::host;  // (the opaque C handle lives in the global namespace)

constexpr uint32_t EDGE_NEVER = 0xFFFFFFFFu;
struct waa_batch {
  uint32_t n_inst = 0, n_out = 0;
  uint64_t length = 0;
  float sr = 0;
  uint32_t n_quanta = 0, n_tiles = 0;
  uint64_t lp = 0;  // padded frames per channel
  int device = 0;
  int n_cu = 256;  // compute units of the device (plan-only batches: an MI355X)
  hipStream_t stream = nullptr;
  void* stage = nullptr;  // pinned staging block for transfers from / to PAGEABLE caller memory (waa_internal_xfer_h2d / waa_internal_xfer_d2h, waa_abi.cpp)
  std::vector<Node> nodes;
  std::vector<waa_edge_desc> edges;
  // OfflineAudioContext::suspend_sync (offline.rs:359-397): the control clock — the render quantum in front of which the render
  // is "suspended" (waa_render_range moves it); control calls made at ctl_q > 0 take effect from that quantum on.  An edge is
  // live for quanta [edge_on, edge_off): waa_connect / waa_disconnect at ctl_q > 0 (EDGE_NEVER: no end).  At plan time an edge
  // that is not live for the whole render becomes a GainNode whose gain is 1 inside the window and 0 outside
  // (desugar_timed_edges, waa_abi.cpp): gain.rs' two fast paths — pass-through and silence — are exactly "connected" and "not".
  uint32_t ctl_q = 0;
  std::vector<uint32_t> edge_on, edge_off;
  bool ranged = false;              // waa_render_range has been called: waa_render would render from the start again
  uint32_t n_user_nodes = 0;        // nodes of the caller's graph (the gates of timed edges are appended behind them)
  bool timed_edges_done = false;
  std::string timed_note;
  std::vector<uint32_t> order;
  std::vector<uint8_t> cut;         // per DelayNode: writer->reader edge removed by the cycle breaker
  std::vector<uint32_t> group_tiles;  // block size (tiles) of every block-scheduled feedback loop
  std::vector<uint32_t> qgroup_quanta;  // block size (render quanta) of every quantum-blocked loop of a dynamic-count plan
  std::vector<int32_t*> loop_flags;     // device words a delay writer sets when a multi-quantum block was invalid (waa_dyn.hip)
  bool loops_one_quantum = false;       // ... after which the loops are rendered one quantum per block
  bool loops_unsettled = false;         // a render with multi-quantum blocks is in flight: its flags have not been looked at
  std::vector<void*> allocs;        // plan-owned device allocations
  std::vector<void*> payload_allocs;  // buffers uploaded through the API
  std::vector<std::pair<void*, size_t>> state_bufs;  // zeroed at the start of every render
  // host copies of the per-instance source tables handed to the kernels (key: the device table): a signal wider than six channels is
  // rendered in channel slices (push_chain_step), and a slice of a SOURCE is the same table with its base pointers moved on
  std::map<const waa::SrcInst*, std::vector<waa::SrcInst>> src_tables;
  std::vector<std::pair<void*, size_t>> ones_bufs;   // filled with 0xFF bytes at the start of every render
  std::vector<Step> steps;
  bool planned = false;
  // Graph-modulated playbackRate / detune of AudioBufferSourceNodes (k-rate params the HOST needs: the playhead replay).
  // prepass = true: build_plan plans only what feeds those params (the modulators and the params' summing chains); the steps
  // are run, one value per quantum is read back and installed as k-rate value blocks, the param edges are removed and the
  // real plan is built (waa_abi.cpp::resolve_source_rate_modulation).
  bool prepass = false;
  std::vector<std::pair<uint32_t, uint32_t>> prepass_params;  // (source node, param)
  std::vector<waa::ParamRef> prepass_refs;                          // per entry: the per-frame values the summing chain writes
  std::string prepass_note;
  bool no_short_ring = false;        // second planning pass: a short feedback loop tried as the LDS-ring kernel did not qualify
  std::set<uint32_t> short_ring_loops;  // DelayNodes of loops shorter than a tile planned for the ring kernel (loop_block_tiles)
  bool force_dynamic = false;        // second planning pass: a loop member the static loop kernel cannot render
  bool dynamic = false;              // the plan renders the reference's dynamic channel counts (dyn_kernel)
  uint64_t code_stride = 0;          // bytes per instance of a code table (n_quanta rounded up)
  bool rendered = false;
  // waa_render_sharded: buffers of the streamed source are allocated and registered first (the plan only needs their shape), filled
  // from the host when the sub-batch has its turn on the link (fill_pending_uploads, waa_abi.cpp)
  bool defer_fill = false;
  struct PendingFill {
    int kind;  // 0: f32 planes, 1: interleaved 16-bit PCM through the decode kernel
    const void* host;
    float* planes;
    int16_t* staging;
    uint32_t n_items, n_ch;
    uint64_t frames, stride, target;
    float src_sr;
  };
  std::vector<PendingFill> pending_fills;
  // waa_batch_rearm: the whole-batch source uploads by node, so that a planned batch can take the NEXT set of AudioBuffers of the
  // same shape into the same device buffers (a serving loop: same graph, new audio — no plan, no allocation)
  std::map<uint32_t, PendingFill> batch_fills;
  bool rearmed = false;
  int64_t batch_fill_node = -1;
  int16_t* pcm_out = nullptr;     // staging of waa_download_all_pcm16 (allocated on first use, freed with the batch)
  size_t pcm_out_count = 0;
  bool dry = false;                  // WAA_DEVICE_PLAN_ONLY: allocations are host memory, nothing is launched
  std::vector<std::string> plan_log;  // waa_plan_describe
  bool profiling = false;
  std::vector<ProfileEntry> prof;
  // where the first waa_render's host time goes (waa_plan_describe prints it: "timing: ..."): an offline context renders
  // once, so planning + allocation IS part of what a caller waits for
  double t_plan_ms = 0, t_alloc_ms = 0, t_upload_ms = 0;  // (alloc / upload: running totals of the batch)
  uint64_t n_alloc = 0, alloc_bytes = 0;
  uint32_t* scan_counter = nullptr;   // 8 unit counters (16 words apart) + error flag of the time-parallel biquad launches
  uint32_t scan_issued[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // host mirror of the counters during a render
  // one playhead replay per distinct (source node, schedule) and plan: the silence / count replay, the code rows and the source's
  // input tables all ask for it (a 10 s slow-track replay is milliseconds); cleared when build_plan returns
  std::map<std::pair<uint32_t, SchedKey>, std::shared_ptr<SchedOut>> sched_cache;
  double plan_alloc_ms = 0, plan_upload_ms = 0;          // ... and their part inside build_plan
  uint64_t plan_n_alloc = 0, plan_alloc_bytes = 0;
};

namespace waa {
namespace host {

// WAA_PLAN_TRACE=1 (measurement build only): how long the named sections of build_plan took, on stderr
struct PlanTrace {
  const char* what;
  std::chrono::steady_clock::time_point t0;
  bool on;
  explicit PlanTrace(const char* w) : what(w), t0(std::chrono::steady_clock::now()), on(measure_switch("WAA_PLAN_TRACE") != nullptr) {}
  ~PlanTrace() {
    if (on) fprintf(stderr, "[plan] %-40s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// Device arena (waa_device_arena_reserve, include/waa_hip.h): one slab per device, reserved once — typically at process start,
// before anything else fragments the device's memory — from which the big buffers of every batch are carved in 2 MB-aligned
// pieces; the slab is one mapping with the largest page-table fragments the driver gives, so WHERE a batch's output lands no
// longer changes from batch to batch (the same kernel ran at 1.33 or 1.55 ms depending on the
// hipMalloc that happened to serve its output).  Defined in waa_abi.cpp.
void* arena_alloc(int device, size_t bytes, bool read_only = false);  // (read_only: the top end of a graded arena)
bool arena_free(int device, void* p);
void* guard_alloc(int device, size_t bytes);  // WAA_GUARD_ALLOC=1 (testing aid, waa_arena.cpp)
// bytes of one physical unit when `p` lies in a GRADED arena of `device` (units mapped side by side), else 0
size_t arena_unit_of(int device, const void* p);
// hipMemcpy2DAsync whose device side may lie in a graded arena: the runtime refuses a pitched copy whose extent exceeds ONE mapped
// physical unit ("invalid argument"; 1-D copies, memsets and kernels are not affected — tools/vmm_copy_check.hip,
// profiles/r06h_vmm_copy_check*.txt), so such a copy is issued in row groups of at most half a unit.
inline hipError_t copy2d_async(int device, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                               hipMemcpyKind kind, hipStream_t stream) {
  const bool dev_is_dst = kind == hipMemcpyHostToDevice;
  const size_t unit = arena_unit_of(device, dev_is_dst ? dst : src), pitch = dev_is_dst ? dpitch : spitch;
  if (!unit || pitch * height <= unit / 2) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream);
  const size_t per = std::max<size_t>((unit / 2) / pitch, 1);
  for (size_t r = 0; r < height; r += per) {
    const hipError_t e = hipMemcpy2DAsync(static_cast<char*>(dst) + r * dpitch, dpitch, static_cast<const char*>(src) + r * spitch, spitch, width,
                                          std::min(per, height - r), kind, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <typename T>
int dev_alloc(waa_batch* b, T** out, size_t count, bool payload = false) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  if (b->dry) {
    // plan-only: big signal / spectrum buffers are never touched, so reserve address space lazily (calloc of a
    // huge block is not committed until written) — small tables are really filled by dev_upload
    p = std::calloc(1, bytes);
    if (!p) return fail(WAA_ERR_DEVICE, "plan-only allocation of %zu bytes failed", bytes);
    (payload ? b->payload_allocs : b->allocs).push_back(p);
    *out = reinterpret_cast<T*>(p);
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipSuccess;
  static const bool guard = getenv("WAA_GUARD_ALLOC") != nullptr;
  if (guard) p = guard_alloc(b->device, bytes);
  if (!p && bytes >= (1u << 20)) p = arena_alloc(b->device, bytes, payload);  // (small tables stay with hipMalloc; payloads are only read)
  if (!p) e = hipMalloc(&p, bytes);
  b->t_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  b->n_alloc++;
  b->alloc_bytes += bytes;
  if (e != hipSuccess) return fail(WAA_ERR_DEVICE, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  // WAA_POISON_ALLOC=1 (testing aid): fresh device memory is filled with 0xFF bytes (NaNs as f32 / f64, huge indices) —
  // a kernel whose OUTPUT depends on memory nobody wrote then fails loudly and every time, instead of once in 20 000
  // graphs when the pages happen to hold something else (hipMalloc does not clear; found that way: see DESIGN.md section 5)
  static const bool poison = getenv("WAA_POISON_ALLOC") != nullptr;
  if (poison) {  // (the fill must be over before anything else touches the buffer: it runs on the null stream, the batch on its own)
    (void)hipMemset(p, 0xFF, bytes);
    (void)hipDeviceSynchronize();
  }
  (payload ? b->payload_allocs : b->allocs).push_back(p);
  *out = reinterpret_cast<T*>(p);
  return 0;
}
template <typename T>
int dev_upload(waa_batch* b, T** out, const std::vector<T>& host) {
  int e = dev_alloc(b, out, host.size());
  if (e) return e;
  if (!host.empty()) {
    if (b->dry) {
      std::memcpy(*out, host.data(), host.size() * sizeof(T));
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      HIP_TRY(hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
      // A synchronous hipMemcpy runs on the NULL stream; the batch's kernels — also the plan-time ones, e.g. the impulse-
      // response spectra right after the IR's upload — run on a non-blocking stream of their own, which the null stream
      // does not order.  hipMemcpy returns when the host buffer may be reused, not necessarily when the last bytes have
      // landed in device memory: wait for the null stream, so that a table is THERE when dev_upload returns (a few us per
      // table at plan time).  Suspected cause of two load-dependent mismatches in 40 000 random graphs.
      HIP_TRY(hipStreamSynchronize(nullptr));
      b->t_upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  return 0;
}

// nodes whose render state freezes while they do not process (waa_frozen.hip): rendered node-major behind a link table
inline bool is_frozen_node(const Node& n) {
  return (n.desc.kind == WAA_NODE_WAVESHAPER && n.has_curve && n.desc.i[0] != WAA_OVERSAMPLE_NONE) ||
         (n.desc.kind == WAA_NODE_PANNER && n.desc.i[0] == WAA_PANNING_HRTF);
}

inline int check_node(waa_batch* b, uint32_t node, uint32_t kind) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (node >= b->nodes.size() || b->nodes[node].desc.kind != kind)
    return fail(WAA_ERR_INVALID_ARGUMENT, "node %u is not of the expected kind", node);
  return 0;
}
inline int check_inst(waa_batch* b, uint32_t inst) {
  if (inst != WAA_ALL_INSTANCES && inst >= b->n_inst) return fail(WAA_ERR_INVALID_ARGUMENT, "instance out of range");
  return 0;
}
inline int check_unplanned(waa_batch* b) {
  if (b->planned) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - the batch is frozen once rendering has started");
  return 0;
}

}  // namespace host
}  // namespace waa

namespace waa {
namespace host {

// ---- waa_schedule.cpp: host-side arithmetic of the reference's control/render split ---------------------
struct Coefs {
  double b0, b1, b2, a1, a2;
};
Coefs biquad_coefs(int type, double sample_rate, double f0, double gain, double q);   // biquad_filter.rs:28-373
float computed_freq(float freq, float detune);                                          // biquad_filter.rs:367-373
struct V3 {
  float x, y, z;
};
constexpr float PI_F = 3.14159265358979323846f;
void azimuth_elevation(V3 sp, V3 lp, V3 lf, V3 lu, float* az, float* el);               // spatial.rs / panner.rs
float spatial_angle(V3 sp, V3 so, V3 lp);
float cone_gain(const waa_node_desc& d, V3 sp, V3 so, V3 lp);
float dist_gain(const waa_node_desc& d, V3 sp, V3 lp);
// AudioBufferSourceNode scheduler: port of audio_buffer_source.rs:422-845
void schedule_source(const waa_batch* b, const SourceSched& cfg, uint64_t frames, float buf_sr, bool has_buffer,
                     const std::vector<float>& rate_q, const std::vector<float>& detune_q, SchedOut* out);
// the same through the batch's per-plan cache (params without per-quantum blocks only: the key holds their one value)
std::shared_ptr<SchedOut> schedule_source_cached(waa_batch* b, uint32_t node, const SchedKey& key, const SourceSched& cfg, uint64_t frames,
                                                 float buf_sr, bool has_buffer, const std::vector<float>& rate_q,
                                                 const std::vector<float>& detune_q);
// one value per quantum (or a single one) of a host-evaluated param, clamped like the reference
std::vector<float> param_per_quantum(const waa_batch* b, const ParamStore& p, uint32_t inst, bool* varies);

// ---- waa_plan.cpp: graph -> launch plan -----------------------------------------------------------------
int build_plan(waa_batch* b);
// waa_frozen_host.cpp: WaveShaper 2x / 4x and the HRTF panner as node-major steps; src_id >= 0: static plan, the node's
// only input is that source node (its host-known codes stand in for the codes a dynamic plan computes on the device)
int plan_oversampler(waa_batch* b, uint32_t id, int src_id);
int plan_hrtf(waa_batch* b, uint32_t id, int src_id);
bool hrtf_sphere_loaded();
void plan_note(waa_batch* b, const char* fmt, ...);
int slot_for(waa_batch* b, const char* name);
void default_channel_config(Node& n, uint32_t n_out);
void compute_order(const waa_batch* b, std::vector<uint8_t>* cut, std::vector<uint8_t>* muted, std::vector<uint32_t>* items);
int computed_in_nch(const Node& n, int maxc);  // quantum.rs:543-547

}  // namespace host
}  // namespace waa
