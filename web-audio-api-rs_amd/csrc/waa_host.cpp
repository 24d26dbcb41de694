// waa_host.cpp — host side of libwaa_hip.so: the C ABI (include/waa_hip.h), graph validation,
// the planner that fuses single-consumer node paths into chain-kernel launches, the host-side
// scheduler of AudioBufferSourceNode (port of the playhead state machine — scheduling stays on
// the host), AudioParam materialisation and coefficient pre-computation.
//
// All sample arithmetic happens in the HIP kernels (waa_kernels.hip, waa_conv.hip); there is no
// CPU fallback: without a HIP device every render call fails with WAA_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/waa_hip.h"
#include "waa_internal.hpp"

using namespace waa;

struct waa_batch;
static int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in);
namespace {
void plan_note(waa_batch* b, const char* fmt, ...);
}

namespace {

thread_local char g_err[768];
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(WAA_ERR_DEVICE, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, \
                                      __LINE__, #expr);                                         \
  } while (0)

struct ParamBlock {
  uint32_t inst;
  uint64_t q0;
  uint32_t nq, vpq;
  std::vector<float> v;
};
struct ParamStore {
  std::vector<float> cst;
  std::vector<ParamBlock> blocks;
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
  float sr = 0;
  bool valid = false;
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
  // waveshaper
  std::vector<float> curve;
  bool has_curve = false;
  float* d_curve = nullptr;
  // oscillator: custom PeriodicWave table (8192 points, periodic_wave.rs:76)
  std::vector<float> osc_wave;
  // iir filter: normalised coefficient pairs (iir_filter.rs:273-311)
  std::vector<double> iir_b, iir_a;
  // analyser (control side state)
  struct AnCache {
    std::vector<float> spec, time;
  };
  std::map<uint32_t, AnCache> an_cache;
  float* d_window = nullptr;
  Cplx *d_an_tw = nullptr, *d_an_twfull = nullptr;
  float *d_an_prev = nullptr, *d_an_spec = nullptr, *d_an_time = nullptr;
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
};

struct ProfileEntry {
  std::string name;
  uint64_t launches = 0;
  double total_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Step {
  int kind = 0;  // 0 chain (interpreter kernel), 1 streaming biquad kernel, 2 FFT convolver, 3 zero-fill, 4 direct FIR, 5 per-frame biquad coefficients, 6 streaming IIR kernel, 7 delay gather, 8 feedback loop, 9 oscillator
  ChainDesc chain{};
  BiquadStreamDesc bq{};
  ConvDesc conv{};
  BiquadCoefDesc coef{};
  IirStreamDesc iir{};
  DelayDesc delay{};
  LoopDesc loop{};
  OscDesc osc{};
  int slot_fwd = -1, slot_mac = -1, slot_inv = -1;
  void* zero_ptr = nullptr;
  size_t zero_bytes = 0;
  int cmax = 1;
  int profile_slot = -1;
  int group = -1;         // >= 0: member of a block-scheduled feedback loop (launched block by block)
  bool prologue = false;  // inside a group: runs once over the full range before the blocks
};

}  // namespace

struct waa_batch {
  uint32_t n_inst = 0, n_out = 0;
  uint64_t length = 0;
  float sr = 0;
  uint32_t n_quanta = 0, n_tiles = 0;
  uint64_t lp = 0;  // padded frames per channel
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<Node> nodes;
  std::vector<waa_edge_desc> edges;
  std::vector<uint32_t> order;
  std::vector<uint8_t> cut;         // per DelayNode: writer->reader edge removed by the cycle breaker
  std::vector<uint32_t> group_tiles;  // block size (tiles) of every block-scheduled feedback loop
  std::vector<void*> allocs;        // plan-owned device allocations
  std::vector<void*> payload_allocs;  // buffers uploaded through the API
  std::vector<std::pair<void*, size_t>> state_bufs;  // zeroed at the start of every render
  std::vector<Step> steps;
  bool planned = false;
  bool rendered = false;
  bool dry = false;                  // WAA_DEVICE_PLAN_ONLY: allocations are host memory, nothing is launched
  std::vector<std::string> plan_log;  // waa_plan_describe
  bool profiling = false;
  std::vector<ProfileEntry> prof;
};

namespace {

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
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(WAA_ERR_DEVICE, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  (payload ? b->payload_allocs : b->allocs).push_back(p);
  *out = reinterpret_cast<T*>(p);
  return 0;
}
template <typename T>
int dev_upload(waa_batch* b, T** out, const std::vector<T>& host) {
  int e = dev_alloc(b, out, host.size());
  if (e) return e;
  if (!host.empty()) {
    if (b->dry)
      std::memcpy(*out, host.data(), host.size() * sizeof(T));
    else
      HIP_TRY(hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  return 0;
}

int check_node(waa_batch* b, uint32_t node, uint32_t kind) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (node >= b->nodes.size() || b->nodes[node].desc.kind != kind)
    return fail(WAA_ERR_INVALID_ARGUMENT, "node %u is not of the expected kind", node);
  return 0;
}
int check_inst(waa_batch* b, uint32_t inst) {
  if (inst != WAA_ALL_INSTANCES && inst >= b->n_inst) return fail(WAA_ERR_INVALID_ARGUMENT, "instance out of range");
  return 0;
}
int check_unplanned(waa_batch* b) {
  if (b->planned) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - the batch is frozen once rendering has started");
  return 0;
}

// ---- almost crate 0.2 (see oracle header for the provenance note) ----------------------
const double ALMOST_TOL = 1.4901161193847656e-8;
bool almost_zero(double a) { return std::fabs(a) < ALMOST_TOL; }
bool almost_equal(double a, double b) {
  if (a == b) return true;
  if (!std::isfinite(a) || !std::isfinite(b)) return false;
  double scale = std::fmax(std::fabs(a), std::fabs(b));
  if (scale < 1.0) scale = 1.0;
  return std::fabs(a - b) < scale * ALMOST_TOL;
}

// ---- biquad coefficients (biquad_filter.rs:28-373), f64 ---------------------------------
struct Coefs {
  double b0, b1, b2, a1, a2;
};
Coefs norm(double b0, double b1, double b2, double a0, double a1, double a2) {
  double s = 1. / a0;
  return {b0 * s, b1 * s, b2 * s, a1 * s, a2 * s};
}
Coefs biquad_coefs(int type, double sample_rate, double f0, double gain, double q) {
  const double PI = 3.14159265358979323846;
  double nyq = sample_rate / 2.;
  double f = f0 / nyq;
  f = f < 0. ? 0. : f > 1. ? 1. : f;
  const Coefs wire{1., 0., 0., 0., 0.}, zero{0., 0., 0., 0., 0.};
  double A = std::pow(10., gain / 40.);
  switch (type) {
    case WAA_BIQUAD_LOWPASS: {
      if (f == 1.) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * std::pow(10., q / 20.)), cw = std::cos(w0), be = (1. - cw) / 2.;
      return norm(be, 2. * be, be, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_HIGHPASS: {
      if (f == 1.) return zero;
      if (f == 0.) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * std::pow(10., q / 20.)), cw = std::cos(w0), be = (1. + cw) / 2.;
      return norm(be, -2. * be, be, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_BANDPASS: {
      if (!(f > 0. && f < 1.)) return zero;
      if (!(q > 0.)) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(al, 0., -al, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_NOTCH: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return zero;
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1., -2. * cw, 1., 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_ALLPASS: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return Coefs{-1., 0., 0., 0., 0.};
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1. - al, -2. * cw, 1. + al, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_PEAKING: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return Coefs{A * A, 0., 0., 0., 0.};
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1. + al * A, -2. * cw, 1. - al * A, 1. + al / A, -2. * cw, 1. - al / A);
    }
    case WAA_BIQUAD_LOWSHELF: {
      if (f == 1.) return Coefs{A * A, 0., 0., 0., 0.};
      if (f == 0.) return wire;
      double w0 = PI * f, cw = std::cos(w0), as = std::sin(w0) / 2. * 1.41421356237309504880;
      double k = 2. * as * std::sqrt(A), ap = A + 1., am = A - 1.;
      return norm(A * (ap - am * cw + k), 2. * A * (am - ap * cw), A * (ap - am * cw - k), ap + am * cw + k,
                  -2. * (am + ap * cw), ap + am * cw - k);
    }
    default: {
      if (f == 1.) return wire;
      if (!(f > 0.)) return Coefs{A * A, 0., 0., 0., 0.};
      double w0 = PI * f, cw = std::cos(w0), as = std::sin(w0) / 2. * 1.41421356237309504880;
      double k = 2. * as * std::sqrt(A), ap = A + 1., am = A - 1.;
      return norm(A * (ap + am * cw + k), -2. * A * (am + ap * cw), A * (ap + am * cw - k), ap - am * cw + k,
                  2. * (am - ap * cw), ap - am * cw - k);
    }
  }
}
float computed_freq(float freq, float detune) { return detune != 0.f ? freq * exp2f(detune / 1200.f) : freq; }

// ---- spatial geometry (spatial.rs:205-299, panner.rs:927-985), f32 as the reference -----
struct V3 {
  float x, y, z;
};
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
V3 normalized(V3 a) { return scale(a, 1.f / std::sqrt(sqlen(a))); }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
const float PI_F = 3.14159265358979323846f;

void azimuth_elevation(V3 sp, V3 lp, V3 lf, V3 lu, float* az, float* el) {
  *az = 0.f;
  *el = 0.f;
  V3 rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return;
  V3 sl = normalized(rel);
  V3 right = cross(lf, lu);
  if (sqlen(right) == 0.f) return;
  V3 rn = normalized(right), fn = normalized(lf), up = cross(rn, fn);
  float elevation = 90.f - 180.f * acosf(dot(sl, up)) / PI_F;
  if (elevation > 90.f)
    elevation = 180.f - elevation;
  else if (elevation < -90.f)
    elevation = -180.f - elevation;
  float up_proj = dot(sl, up);
  V3 ps = sub(sl, scale(up, up_proj));
  *el = elevation;
  if (sqlen(ps) == 0.f) return;
  V3 psn = normalized(ps);
  float azimuth = 180.f * acosf(dot(psn, rn)) / PI_F;
  if (dot(psn, fn) < 0.f) azimuth = 360.f - azimuth;
  if (azimuth >= 0.f && azimuth <= 270.f)
    azimuth = 90.f - azimuth;
  else
    azimuth = 450.f - azimuth;
  *az = azimuth;
}
float spatial_angle(V3 sp, V3 so, V3 lp) {
  if (sqlen(so) == 0.f) return 0.f;
  V3 son = normalized(so), rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return 0.f;
  V3 sl = normalized(rel);
  return std::fabs(180.f * acosf(dot(sl, son)) / PI_F);
}
float cone_gain(const waa_node_desc& d, V3 sp, V3 so, V3 lp) {
  float in = (float)std::fabs(d.d[3]) / 2.f, out = (float)std::fabs(d.d[4]) / 2.f;
  if (in >= 180.f && out >= 180.f) return 1.f;
  float cog = (float)d.d[5];
  float a = spatial_angle(sp, so, lp);
  if (a < in) return 1.f;
  if (a >= out) return cog;
  float x = (a - in) / (out - in);
  return (1.f - x) + cog * x;
}
float dist_gain(const waa_node_desc& d, V3 sp, V3 lp) {
  double distance = (double)std::sqrt(sqlen(sub(sp, lp)));
  double ref = d.d[0], maxd = d.d[1], roll = d.d[2], g;
  switch (d.i[1]) {
    case WAA_DISTANCE_LINEAR: {
      double rf = roll < 0. ? 0. : roll > 1. ? 1. : roll;
      double lo = std::fmin(ref, maxd), hi = std::fmax(ref, maxd);
      double dc = distance < lo ? lo : distance > hi ? hi : distance;
      g = 1. - rf * (dc - lo) / (hi - lo);
      break;
    }
    case WAA_DISTANCE_INVERSE: {
      double rf = std::fmax(roll, 0.);
      g = distance > 0. ? ref / (ref + rf * (std::fmax(ref, distance) - ref)) : 1.;
      break;
    }
    default: {
      double rf = std::fmax(roll, 0.);
      g = std::pow(std::fmax(distance, ref) / ref, -rf);
    }
  }
  return (float)g;
}

// ---- AudioBufferSourceNode scheduler: port of audio_buffer_source.rs:422-845 -------------
struct SchedOut {
  std::vector<QRec> qrec;
  std::vector<SlowRec> slow;  // empty if no slow quantum
  std::vector<uint8_t> tile_fast;
  bool any_slow = false;
};
using SchedKey = std::tuple<double, double, double, double, int, double, double, uint64_t, float, float, float>;

void schedule_source(const waa_batch* b, const SourceSched& cfg, uint64_t frames, float buf_sr, bool has_buffer,
                     const std::vector<float>& rate_q, const std::vector<float>& detune_q, SchedOut* out) {
  const uint32_t nq = b->n_quanta;
  out->qrec.assign(nq, QRec{0, Q_SILENT, 0});
  out->slow.clear();
  out->any_slow = false;
  double start_time = cfg.start, stop_time = cfg.stop, offset = cfg.offset, duration = cfg.duration;
  const double sample_rate = (double)b->sr;
  const double dt = 1. / sample_rate;
  const double block_duration = dt * (double)RQ;
  const double buffer_duration = has_buffer ? (double)frames / (double)buf_sr : 0.;
  // clamp_loop_boundaries (:401-417)
  double loop_start = cfg.loop_start, loop_end = cfg.loop_end;
  if (has_buffer) {
    if (loop_start < 0.)
      loop_start = 0.;
    else if (loop_start > buffer_duration)
      loop_start = buffer_duration;
    if (loop_end <= 0. || loop_end > buffer_duration) loop_end = buffer_duration;
  }
  const bool is_looping = cfg.looping != 0;
  const double sampling_ratio = has_buffer ? (double)buf_sr / sample_rate : 1.;
  double buffer_time = 0., elapsed = 0.;
  bool started = false, entered_loop = false, is_aligned = false, ended = false;
  auto ensure_slow = [&]() {
    if (!out->any_slow) {
      out->slow.assign((size_t)nq * RQ, SlowRec{-1, -1, 0.});
      out->any_slow = true;
    }
  };
  for (uint32_t q = 0; q < nq; q++) {
    if (ended) break;
    const double block_time = (double)((uint64_t)q * RQ) / sample_rate;  // thread.rs:360
    const double next_block_time = block_time + block_duration;
    if (!has_buffer && start_time != DBL_MAX) break;  // ended
    if (start_time >= next_block_time) {
      if (stop_time <= next_block_time) break;
      continue;
    }
    if (!has_buffer) continue;
    const double detune = (double)detune_q[detune_q.size() == 1 ? 0 : q];
    const double playback_rate = (double)rate_q[rate_q.size() == 1 ? 0 : q];
    const double cpr = playback_rate * std::exp2(detune / 1200.);
    double actual_loop_start = 0., actual_loop_end = 0.;
    if (!started && start_time < block_time) start_time = block_time;
    if (start_time == block_time && offset == 0.) is_aligned = true;
    if (sampling_ratio != 1. || cpr != 1.) is_aligned = false;
    if (loop_start != 0. || loop_end != buffer_duration) is_aligned = false;
    if (buffer_time + block_duration > duration || block_time + block_duration > stop_time) is_aligned = false;
    if (is_aligned) {
      if (start_time == block_time) started = true;
      const int64_t start_index = (int64_t)std::llround(buffer_time * sample_rate);
      out->qrec[q] = QRec{start_index, is_looping ? (uint32_t)Q_FAST_LOOP : (uint32_t)Q_FAST, 0};
      if (buffer_time + block_duration > buffer_duration) {
        // did the playhead wrap inside this block?  (:568-607)
        int loop_point_index = -1;
        if (is_looping) {
          uint64_t si = (uint64_t)start_index, off = 0;
          for (int index = 0; index < RQ; index++) {
            uint64_t bi = si + (uint64_t)index - off;
            if (bi >= frames) {
              loop_point_index = index;
              si = 0;
              off = (uint64_t)index;
            }
          }
        }
        if (loop_point_index >= 0)
          buffer_time = std::fmod((double)(RQ - loop_point_index) / sample_rate, buffer_duration);
        else
          buffer_time += block_duration;
      } else {
        buffer_time += block_duration;
      }
      elapsed += block_duration;
    } else {
      if (is_looping) {
        if (loop_start >= 0. && loop_end > 0. && loop_start < loop_end) {
          actual_loop_start = loop_start;
          actual_loop_end = loop_end;
        } else {
          actual_loop_start = 0.;
          actual_loop_end = buffer_duration;
        }
      } else {
        entered_loop = false;
      }
      ensure_slow();
      out->qrec[q] = QRec{0, Q_SLOW, 0};
      SlowRec* rec = &out->slow[(size_t)q * RQ];
      for (int i = 0; i < RQ; i++) {
        rec[i] = SlowRec{-1, -1, 0.};
        const double current_time = block_time + (double)i * dt;
        if (!started && almost_equal(current_time, start_time)) start_time = current_time;
        if (almost_equal(elapsed, duration)) elapsed = duration;
        if (current_time < start_time || current_time >= stop_time || elapsed >= duration) continue;
        if (!started) {
          const double delta = current_time - start_time;
          offset += delta * cpr;
          offset = std::fmin(std::fmax(offset, 0.), buffer_duration);
          if (is_looping && cpr >= 0. && offset > actual_loop_end) offset = actual_loop_end;
          if (is_looping && cpr < 0. && offset < actual_loop_start) offset = actual_loop_start;
          buffer_time = offset;
          elapsed = std::fabs(delta * cpr);
          started = true;
        }
        if (is_looping) {
          if (almost_equal(buffer_time, actual_loop_end)) buffer_time = actual_loop_end;
          if (almost_equal(buffer_time, actual_loop_start)) buffer_time = actual_loop_start;
          if (!entered_loop) {
            if (offset < actual_loop_end && buffer_time >= actual_loop_start) entered_loop = true;
            if (offset >= actual_loop_end && buffer_time < actual_loop_end) entered_loop = true;
          }
          if (entered_loop) {
            while (buffer_time >= actual_loop_end) buffer_time -= actual_loop_end - actual_loop_start;
            while (buffer_time < actual_loop_start) buffer_time += actual_loop_end - actual_loop_start;
          }
        }
        if (almost_zero(buffer_time)) buffer_time = 0.;
        if (buffer_time >= 0. && buffer_time < buffer_duration) {
          const double position = buffer_time * sampling_ratio;
          const double playhead = position * sample_rate;
          const double pf = std::floor(playhead);
          const uint64_t prev = (uint64_t)pf;
          const double k = playhead - pf;
          if (prev < frames) {
            SlowRec r;
            r.prev = (int32_t)prev;
            r.k = k;
            if (prev + 1 < frames) {
              r.next = (int32_t)(prev + 1);
            } else if (is_looping) {
              if (playback_rate >= 0.) {
                const double sp = actual_loop_start * sample_rate;
                const uint64_t si = (std::floor(sp) == sp) ? (uint64_t)sp : (uint64_t)sp + 1;
                r.next = si < frames ? (int32_t)si : -1;
              } else {
                // the reference reads buffer_channel[end_index] (audio_buffer_source.rs:795-797): one past the
                // end when loop_end == duration (a Rust panic there); defined as a 0 sample here
                const double ep = actual_loop_end * sample_rate;
                const uint64_t ei = (uint64_t)ep;
                r.next = ei < frames ? (int32_t)ei : -1;
              }
            } else {
              r.next = (almost_equal(k, 1.) || prev == 0) ? -1 : -2;
            }
            rec[i] = r;
          }
        }
        const double time_incr = dt * cpr;
        buffer_time += time_incr;
        elapsed += std::fabs(time_incr);
      }
    }
    if (next_block_time >= stop_time || elapsed >= duration ||
        (!is_looping && ((cpr > 0. && buffer_time >= buffer_duration) || (cpr < 0. && buffer_time < 0.))))
      ended = true;
  }
  // per tile: can the whole tile be fetched as one aligned contiguous run?
  out->tile_fast.assign(b->n_tiles, 0);
  for (uint32_t t = 0; t < b->n_tiles; t++) {
    bool ok = true;
    int64_t s0 = 0;
    for (int k = 0; k < QUANTA_PER_TILE && ok; k++) {
      uint32_t q = t * QUANTA_PER_TILE + k;
      if (q >= nq) {
        ok = false;
        break;
      }
      const QRec& r = out->qrec[q];
      if (r.mode != Q_FAST && r.mode != Q_FAST_LOOP) ok = false;
      if (k == 0) s0 = r.start;
      if (r.start != s0 + (int64_t)k * RQ) ok = false;
    }
    if (ok && (s0 % 4 != 0 || (uint64_t)s0 + TILE > frames)) ok = false;
    out->tile_fast[t] = ok ? 1 : 0;
  }
}

// values of one param for one instance, one value per quantum (first sample of a len-128 slice)
std::vector<float> param_per_quantum(const waa_batch* b, const ParamStore& p, uint32_t inst, bool* varies) {
  std::vector<float> v(1, p.fix(p.cst[inst]));
  bool any = false;
  for (auto& blk : p.blocks)
    if (blk.inst == WAA_ALL_INSTANCES || blk.inst == inst) any = true;
  if (!any) {
    if (varies) *varies = false;
    return v;
  }
  v.assign(b->n_quanta, p.fix(p.cst[inst]));
  for (auto& blk : p.blocks) {
    if (!(blk.inst == WAA_ALL_INSTANCES || blk.inst == inst)) continue;
    for (uint32_t k = 0; k < blk.nq; k++) {
      uint64_t q = blk.q0 + k;
      if (q < b->n_quanta) v[q] = p.fix(blk.v[(size_t)k * blk.vpq]);
    }
  }
  if (varies) *varies = true;
  return v;
}

// Upload a param as a device ParamRef (mode 0 / 1 / 2), values clamped like the reference.
int upload_param(waa_batch* b, const ParamStore& p, ParamRef* ref) {
  const int mode = p.mode();
  std::vector<float> host;
  if (mode == 0) {
    host.resize(b->n_inst);
    for (uint32_t i = 0; i < b->n_inst; i++) host[i] = p.fix(p.cst[i]);
    ref->stride = 0;
  } else {
    const uint64_t per = mode == 1 ? b->n_quanta : (uint64_t)b->n_quanta * RQ;
    host.resize((size_t)b->n_inst * per);
    for (uint32_t i = 0; i < b->n_inst; i++) {
      float c = p.fix(p.cst[i]);
      std::fill(host.begin() + (size_t)i * per, host.begin() + (size_t)(i + 1) * per, c);
    }
    for (auto& blk : p.blocks) {
      uint32_t lo = blk.inst == WAA_ALL_INSTANCES ? 0 : blk.inst, hi = blk.inst == WAA_ALL_INSTANCES ? b->n_inst : blk.inst + 1;
      for (uint32_t i = lo; i < hi; i++)
        for (uint32_t k = 0; k < blk.nq; k++) {
          uint64_t q = blk.q0 + k;
          if (q >= b->n_quanta) continue;
          if (mode == 1) {
            host[(size_t)i * per + q] = p.fix(blk.v[k]);
          } else {
            for (int s = 0; s < RQ; s++)
              host[(size_t)i * per + q * RQ + s] = p.fix(blk.v[(size_t)k * blk.vpq + (blk.vpq == 1 ? 0 : s)]);
          }
        }
    }
    ref->stride = per;
  }
  float* d = nullptr;
  int e = dev_upload(b, &d, host);
  if (e) return e;
  ref->base = d;
  ref->mode = mode;
  ref->pad = 0;
  return 0;
}
// Value mode of param k of node `id` as the kernels will see it: a param with an audio-rate input is per-frame.
int param_mode(const Node& n, size_t k) {
  if (k < n.pin_edges.size() && !n.pin_edges[k].empty()) return 2;
  return n.params[k].mode();
}
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp,
                    const std::vector<OpDesc>& ops, const SignalRef& out);
int temp_signal(waa_batch* b, int nch, SignalRef* out);
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp);
// ParamRef of param k of node `id`.  Without an audio-rate input: the uploaded constants / value blocks.  With one
// (param.rs:686-795): a chain that sums the connected outputs mixed to ONE channel (count 1, explicit, discrete,
// param.rs:309-311), adds the intrinsic value and clamps, written to a one-channel signal the consumer reads
// as per-frame values.  Planned once, right before the first consumer (its producers are materialised and
// precede the owner in processing order).
int node_param(waa_batch* b, uint32_t id, size_t k, ParamRef* ref) {
  Node& n = b->nodes[id];
  if (k >= n.pin_edges.size() || n.pin_edges[k].empty()) return upload_param(b, n.params[k], ref);
  if (n.pin_ready[k]) {
    *ref = n.pin_ref[k];
    return 0;
  }
  std::vector<InputRef> ins;
  for (int ie : n.pin_edges[k]) {
    Node& pn = b->nodes[b->edges[ie].from];
    if (!pn.materialized || !pn.sig.base)
      return fail(WAA_ERR_INVALID_STATE, "internal: AudioParam input of node %u is not materialised yet", id);
    InputRef in{};
    in.kind = IN_SIGNAL;
    in.nch = pn.out_nch;
    in.sig = pn.sig;
    ins.push_back(in);
  }
  int e = reduce_fan_in(b, ins, 1, WAA_INTERP_DISCRETE);
  if (e) return e;
  OpDesc o{};
  o.kind = OP_PARAM_ADD;
  o.nch_in = o.nch_out = 1;
  if ((e = upload_param(b, n.params[k], &o.p0))) return e;
  auto bits = [](float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
  };
  o.i0 = bits(n.params[k].minv);
  o.i1 = bits(n.params[k].maxv);
  o.i2 = bits(n.params[k].defv);
  SignalRef sig;
  if ((e = temp_signal(b, 1, &sig))) return e;
  if ((e = push_chain_step(b, ins, 1, WAA_INTERP_DISCRETE, {o}, sig))) return e;
  ParamRef r{};
  r.base = sig.base;
  r.stride = sig.inst_stride;
  r.mode = 2;
  n.pin_ref[k] = r;
  n.pin_ready[k] = 1;
  *ref = r;
  return 0;
}

// Upload host-computed per-instance (mode 0) or per-(instance, quantum) (mode 1) values.
int upload_values(waa_batch* b, const std::vector<float>& host, int mode, ParamRef* ref) {
  float* d = nullptr;
  int e = dev_upload(b, &d, host);
  if (e) return e;
  ref->base = d;
  ref->mode = mode;
  ref->stride = mode == 0 ? 0 : b->n_quanta;
  ref->pad = 0;
  return 0;
}

int computed_in_nch(const Node& n, int maxc) {
  switch (n.mode) {
    case WAA_COUNT_MODE_MAX: return maxc;
    case WAA_COUNT_MODE_EXPLICIT: return n.cc;
    default: return std::min(maxc, n.cc);
  }
}

// graph.rs:323-487 order_nodes / visit.  A DelayNode is two graph nodes in the reference (delay.rs:283-366:
// writer, then reader, edge writer->reader); vertex `id` is the writer (or a plain node), `id | VTX_READER` the
// reader.  Cycles are broken at the first DelayNode writer on the detected loop (its writer->reader edge is
// cleared and the ordering restarts); nodes of a loop without one are dropped from the ordering (muted).
constexpr uint32_t VTX_READER = 0x80000000u;
struct OrderCtx {
  const waa_batch* b;
  std::vector<uint8_t> cut;
  std::vector<uint32_t> marked, temp, ordered, in_cycle;
  uint32_t breaker = 0;
};
bool is_delay(const waa_batch* b, uint32_t id);
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out);
bool order_visit(OrderCtx& c, uint32_t v) {
  auto pos = std::find(c.temp.begin(), c.temp.end(), v);
  if (pos != c.temp.end()) {
    for (auto it = pos; it != c.temp.end(); ++it)
      if (!(*it & VTX_READER) && is_delay(c.b, *it)) {
        c.breaker = *it;
        return true;
      }
    c.in_cycle.insert(c.in_cycle.end(), pos, c.temp.end());
    return false;
  }
  if (std::find(c.marked.begin(), c.marked.end(), v) != c.marked.end()) return false;
  c.marked.push_back(v);
  c.temp.push_back(v);
  std::vector<uint32_t> targets;
  vertex_targets(c.b, v, c.cut, targets);
  for (uint32_t t : targets)
    if (order_visit(c, t)) return true;
  c.ordered.push_back(v);
  c.temp.erase(std::remove(c.temp.begin(), c.temp.end(), v), c.temp.end());
  return false;
}

int plan_convolver(waa_batch* b, uint32_t id);
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp);

void plan_note(waa_batch* b, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  b->plan_log.emplace_back(buf);
}
const char* input_kind_name(int k) {
  switch (k) {
    case IN_SIGNAL: return "signal";
    case IN_SOURCE: return "source";
    case IN_CONSTANT: return "constant";
    default: return "silent";
  }
}
const char* op_name(int k) {
  switch (k) {
    case OP_GAIN: return "GAIN";
    case OP_BIQUAD: return "BIQUAD";
    case OP_WAVESHAPER: return "WAVESHAPER";
    case OP_STEREO_PAN: return "STEREO_PAN";
    case OP_PANNER: return "PANNER";
    case OP_MIX: return "MIX";
    case OP_IIR: return "IIR";
    case OP_PARAM_ADD: return "PARAM_ADD";
    default: return "?";
  }
}

int slot_for(waa_batch* b, const char* name) {
  for (size_t i = 0; i < b->prof.size(); i++)
    if (b->prof[i].name == name) return (int)i;
  b->prof.push_back(ProfileEntry{name});
  return (int)b->prof.size() - 1;
}

// ---------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch);

// One interpreter-kernel step: input(s) -> [mix to in_nch] -> ops -> out
int push_chain_step(waa_batch* b, const std::vector<InputRef>& inputs, int in_nch, int in_interp,
                    const std::vector<OpDesc>& ops, const SignalRef& out) {
  if (ops.size() > (size_t)MAX_OPS) return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d fused ops in one chain", MAX_OPS);
  Step st;
  ChainDesc& cd = st.chain;
  std::memset(&cd, 0, sizeof cd);
  int cmax = std::max(in_nch, out.nch);
  cd.n_inputs = (int)inputs.size();
  for (size_t k = 0; k < inputs.size(); k++) {
    cd.in[k] = inputs[k];
    cmax = std::max(cmax, inputs[k].nch);
  }
  cd.in_nch = in_nch;
  cd.in_interp = in_interp;
  for (auto& o : ops) cmax = std::max({cmax, o.nch_in, o.nch_out});
  for (auto& o : ops)
    if (o.kind == OP_IIR) return fail(WAA_ERR_DEVICE, "internal: IIR op reached the interpreter kernel");
  for (auto& o : ops)
    if (o.kind == OP_BIQUAD && cmax > 2)
      return fail(WAA_ERR_OUT_OF_SCOPE,
                  "chains with a BiquadFilter are limited to 2 channels per signal on the device path of this round (needs %d)",
                  cmax);
  cd.n_ops = (int)ops.size();
  for (size_t k = 0; k < ops.size(); k++) cd.ops[k] = ops[k];
  cd.out = out;
  cd.n_inst = b->n_inst;
  cd.n_tiles = b->n_tiles;
  cd.tile0 = 0;
  cd.tile1 = b->n_tiles;
  cd.n_quanta = b->n_quanta;
  st.cmax = cmax;
  st.profile_slot = slot_for(b, cmax <= 1 ? "chain_kernel<1>" : cmax <= 2 ? "chain_kernel<2>" : "chain_kernel<4+>");
  b->steps.push_back(st);
  {
    bool serial = false;
    std::string desc;
    for (auto& o : ops) {
      serial |= o.kind == OP_BIQUAD;
      char t[64];
      if (o.kind == OP_MIX)
        snprintf(t, sizeof t, "MIX(%d->%d)", o.nch_in, o.nch_out);
      else if (o.kind == OP_BIQUAD)
        snprintf(t, sizeof t, "BIQUAD(%s)", o.i0 == 0 ? "const" : o.i0 == 1 ? "k-rate" : "a-rate");
      else
        snprintf(t, sizeof t, "%s", op_name(o.kind));
      desc += desc.empty() ? t : std::string(",") + t;
    }
    std::string ins;
    for (auto& in : inputs) {
      char t[48];
      snprintf(t, sizeof t, "%s:%dch", input_kind_name(in.kind), in.nch);
      ins += ins.empty() ? t : std::string("+") + t;
    }
    plan_note(b, "chain %s C=%d in=[%s]->%dch ops=[%s] out=%dch", serial ? "serial" : "parallel", cmax, ins.c_str(), in_nch,
              desc.c_str(), out.nch);
  }
  return 0;
}

int temp_signal(waa_batch* b, int nch, SignalRef* out) {
  float* p = nullptr;
  int e = dev_alloc(b, &p, (size_t)b->n_inst * nch * b->lp);
  if (e) return e;
  *out = SignalRef{p, (uint64_t)nch * b->lp, b->lp, nch, 0};
  return 0;
}

// Turn one fused chain into kernel launches.  A Biquad with constant coefficients (plus up to two constant
// gains right behind it) goes to the streaming kernel (one wave per instance-channel, ~60 % of HBM peak); the
// ops around it run on the tile-parallel element-wise kernel.  Splitting costs one extra pass through HBM
// per cut but keeps every segment on a kernel that is 3-6x closer to the roofline than the serial interpreter,
// which remains the path for per-quantum / per-frame coefficient biquads.
int emit_segments(waa_batch* b, std::vector<InputRef> inputs, int in_nch, int in_interp, const std::vector<OpDesc>& ops,
                  const SignalRef& out) {
  bool any_stream = false;
  const int max_mode = getenv("WAA_NO_KRATE_STREAM") ? 0 : 1;  // debugging aid: force k-rate biquads onto the interpreter
  auto streams = [&](const OpDesc& o) { return o.kind == OP_IIR || (o.kind == OP_BIQUAD && o.i0 <= max_mode && o.nch_in <= 2); };
  for (auto& o : ops) any_stream |= streams(o);
  if (!any_stream) return push_chain_step(b, inputs, in_nch, in_interp, ops, out);
  std::vector<OpDesc> pending;
  int cur_nch = in_nch;
  size_t i = 0;
  while (i < ops.size()) {
    const OpDesc& o = ops[i];
    if (!streams(o)) {
      pending.push_back(o);
      cur_nch = o.nch_out;
      i++;
      continue;
    }
    // the streaming kernel wants ONE plain input (signal or source) of exactly the biquad's channel count
    const bool iir_exact = o.kind == OP_IIR && o.i0 < 0;  // the lane-per-stream kernel reads a materialised signal
    const bool plain = pending.empty() && inputs.size() == 1 &&
                       ((inputs[0].kind == IN_SOURCE && !iir_exact) || inputs[0].kind == IN_SIGNAL) && inputs[0].nch == cur_nch;
    if (!plain) {
      SignalRef tmp;
      int e = temp_signal(b, cur_nch, &tmp);
      if (e) return e;
      if ((e = push_chain_step(b, inputs, in_nch, in_interp, pending, tmp))) return e;
      pending.clear();
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = cur_nch;
      in.sig = tmp;
      inputs.assign(1, in);
      in_nch = cur_nch;
    }
    size_t j = i + 1;
    if (o.kind == OP_BIQUAD)  // the biquad kernel applies up to two constant gains on the way out
      while (j < ops.size() && j - i <= 2 && ops[j].kind == OP_GAIN && ops[j].p0.mode == 0 && ops[j].nch_in == cur_nch) j++;
    SignalRef seg_out = out;
    if (j < ops.size() || out.nch != cur_nch) {
      int e = temp_signal(b, cur_nch, &seg_out);
      if (e) return e;
    }
    if (o.kind == OP_IIR) {
      Step st;
      st.kind = 6;
      IirStreamDesc& q = st.iir;
      std::memset(&q, 0, sizeof q);
      q.in = inputs[0];
      q.coef = reinterpret_cast<const double*>(o.ptr0);
      q.pow = reinterpret_cast<const double*>(o.ptr2);
      q.state = reinterpret_cast<double*>(o.ptr1);
      q.ns = std::abs(o.i0);
      // exact kernels: one lane per stream pays off for low orders or very many streams, one DPP row per
      // stream otherwise (issue cycles per frame: ~12 ns + 32 per 64 streams vs ~64 per 4 streams, on 1024 SIMDs)
      q.exact = 0u;
      if (iir_exact) {
        const double streams = (double)b->n_inst * cur_nch;
        const double lane_cost = std::ceil(streams / 64. / 1024.) * (12. * q.ns + 32.);
        const double row_cost = std::ceil(streams / 4. / 1024.) * 64.;
        q.exact = (row_cost < lane_cost && !getenv("WAA_IIR_LANE")) || getenv("WAA_IIR_ROW") ? 2u : 1u;
      }
      q.nch = cur_nch;
      q.out = seg_out;
      q.n_inst = b->n_inst;
      q.n_tiles = b->n_tiles;
    q.tile0 = 0;
    q.tile1 = b->n_tiles;
      q.tile0 = 0;
      q.tile1 = b->n_tiles;
      q.n_quanta = b->n_quanta;
      char name[32];
      snprintf(name, sizeof name, "%s<%d>", q.exact == 2 ? "iir_row_kernel" : q.exact == 1 ? "iir_lane_kernel" : "iir_stream_kernel", q.ns);
      st.profile_slot = slot_for(b, name);
      b->steps.push_back(st);
      plan_note(b, "%s states=%d in=%s:%dch out=%s", q.exact == 2 ? "iir_exact(row)" : q.exact == 1 ? "iir_exact(lane)" : "iir_stream", q.ns,
                input_kind_name(inputs[0].kind), cur_nch, seg_out.base == out.base ? "final" : "temp");
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = cur_nch;
      in.sig = seg_out;
      inputs.assign(1, in);
      in_nch = cur_nch;
      i = j;
      if (i == ops.size() && seg_out.base == out.base) return 0;
      continue;
    }
    Step st;
    st.kind = 1;
    BiquadStreamDesc& q = st.bq;
    std::memset(&q, 0, sizeof q);
    q.in = inputs[0];
    q.coefs = reinterpret_cast<const double*>(o.ptr0);
    q.coef_stride = o.u0;
    q.vary = o.i0 == 1;
    q.state = reinterpret_cast<double*>(o.ptr1);
    q.n_gain = (int)(j - i - 1);
    for (size_t k = i + 1; k < j; k++) q.gain[k - i - 1] = ops[k].p0;
    q.nch = cur_nch;
    q.out = seg_out;
    q.n_inst = b->n_inst;
    q.n_tiles = b->n_tiles;
    q.tile0 = 0;
    q.tile1 = b->n_tiles;
    q.n_quanta = b->n_quanta;
    st.profile_slot = slot_for(b, "biquad_stream_kernel");
    b->steps.push_back(st);
    plan_note(b, "biquad_stream%s in=%s:%dch gains=%d out=%s", q.vary ? "(k-rate)" : "", input_kind_name(inputs[0].kind),
              cur_nch, q.n_gain, seg_out.base == out.base ? "final" : "temp");
    InputRef in{};
    in.kind = IN_SIGNAL;
    in.nch = cur_nch;
    in.sig = seg_out;
    inputs.assign(1, in);
    in_nch = cur_nch;
    i = j;
    if (i == ops.size() && seg_out.base == out.base) return 0;  // the streaming kernel wrote the final output
  }
  // trailing element-wise ops (or a pure channel-count change into `out`)
  return push_chain_step(b, inputs, in_nch, in_interp, pending, out);
}

bool is_delay(const waa_batch* b, uint32_t id) { return b->nodes[id].desc.kind == WAA_NODE_DELAY; }
// outgoing edges of a vertex of the expanded graph, in insertion order
void vertex_targets(const waa_batch* b, uint32_t v, const std::vector<uint8_t>& cut, std::vector<uint32_t>& out) {
  const uint32_t id = v & ~VTX_READER;
  if (!(v & VTX_READER) && is_delay(b, id)) {
    if (!cut[id]) out.push_back(id | VTX_READER);
    return;
  }
  for (auto& e : b->edges) {
    if (e.from != id) continue;
    // inputs go to the writer half; delayTime belongs to the reader half (delay.rs:316-318)
    out.push_back(is_delay(b, e.to) && (e.to_input & 0x80000000u) ? (e.to | VTX_READER) : e.to);
  }
}

int plan_loop(waa_batch* b, const std::vector<uint32_t>& loop_items);
int plan_delay_writer(waa_batch* b, uint32_t id);
int plan_oscillator(waa_batch* b, uint32_t id);
int plan_delay_reader(waa_batch* b, uint32_t id);
uint32_t loop_block_tiles(waa_batch* b, const std::vector<uint32_t>& loop_items);

int build_plan(waa_batch* b) {
  const uint32_t N = (uint32_t)b->nodes.size();
  for (uint32_t i = 0; i < N; i++)  // the reference takes the coefficients in the constructor
    if (b->nodes[i].desc.kind == WAA_NODE_IIR_FILTER && b->nodes[i].iir_b.empty())
      return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - IIRFilterNode %u has no coefficients", i);
  // processing order = reversed DFS post-order over outgoing edges in insertion order, cycle breakers applied
  // (graph.rs:323-487); `items` has two entries per DelayNode (writer, reader), none for muted nodes
  std::vector<uint32_t> items;
  std::vector<uint8_t> muted(N, 0);
  {
    OrderCtx c;
    c.b = b;
    c.cut.assign(N, 0);
    for (;;) {
      c.marked.clear();
      c.temp.clear();
      c.ordered.clear();
      c.in_cycle.clear();
      bool applied = false;
      for (uint32_t i = 0; i < N && !applied; i++) {
        applied = order_visit(c, i);
        if (!applied && is_delay(b, i)) applied = order_visit(c, i | VTX_READER);
      }
      if (!applied) break;
      c.cut[c.breaker] = 1;
    }
    for (uint32_t v : c.in_cycle) muted[v & ~VTX_READER] = 1;
    for (auto it = c.ordered.rbegin(); it != c.ordered.rend(); ++it)
      if (!muted[*it & ~VTX_READER]) items.push_back(*it);
    b->cut = c.cut;
  }
  // one entry per node, at the position where its OUTPUT is produced (the reader half of a DelayNode)
  b->order.clear();
  for (uint32_t v : items)
    if (!is_delay(b, v & ~VTX_READER) || (v & VTX_READER)) b->order.push_back(v & ~VTX_READER);
  for (uint32_t i = 0; i < N; i++)
    if (muted[i]) plan_note(b, "node %u is part of a cycle without a DelayNode: muted (graph.rs:362-368)", i);
  // feedback loops: strongly connected components of the graph with the writer->reader edges in place
  // (Tarjan); their members are rendered quantum by quantum by the loop kernel
  std::vector<int> scc_of(N, -1);
  int n_scc = 0;
  {
    const uint32_t V = 2 * N;
    auto vidx = [&](uint32_t v) { return (v & VTX_READER) ? N + (v & ~VTX_READER) : v; };
    std::vector<int> index(V, -1), low(V, 0), comp(V, -1);
    std::vector<uint8_t> on(V, 0);
    std::vector<uint32_t> stk;
    int counter = 0, n_comp = 0;
    const std::vector<uint8_t> no_cut(N, 0);
    std::vector<int> comp_size;
    std::function<void(uint32_t)> strong = [&](uint32_t v) {
      const uint32_t vi = vidx(v);
      index[vi] = low[vi] = counter++;
      stk.push_back(v);
      on[vi] = 1;
      std::vector<uint32_t> ts;
      vertex_targets(b, v, no_cut, ts);
      for (uint32_t t : ts) {
        if (muted[t & ~VTX_READER]) continue;
        const uint32_t ti = vidx(t);
        if (index[ti] < 0) {
          strong(t);
          low[vi] = std::min(low[vi], low[ti]);
        } else if (on[ti]) {
          low[vi] = std::min(low[vi], index[ti]);
        }
      }
      if (low[vi] == index[vi]) {
        int size = 0;
        for (;;) {
          const uint32_t w = stk.back();
          stk.pop_back();
          on[vidx(w)] = 0;
          comp[vidx(w)] = n_comp;
          size++;
          if (w == v) break;
        }
        comp_size.push_back(size);
        n_comp++;
      }
    };
    for (uint32_t i = 0; i < N; i++) {
      if (muted[i]) continue;
      if (index[i] < 0) strong(i);
      if (is_delay(b, i) && index[N + i] < 0) strong(i | VTX_READER);
    }
    std::map<int, int> renum;
    for (uint32_t i = 0; i < N; i++) {
      if (muted[i] || comp[i] < 0 || comp_size[comp[i]] < 2) continue;
      auto it = renum.find(comp[i]);
      if (it == renum.end()) it = renum.emplace(comp[i], n_scc++).first;
      scc_of[i] = it->second;
    }
  }
  std::vector<uint32_t> pos(N, 0xffffffffu);
  for (uint32_t i = 0; i < b->order.size(); i++) pos[b->order[i]] = i;
  // incoming edges in summing order: by processing position of the producer, then edge insertion order
  for (auto& n : b->nodes) {
    n.in_edges.clear();
    n.pin_edges.assign(n.params.size(), {});
    n.pin_ref.assign(n.params.size(), ParamRef{});
    n.pin_ready.assign(n.params.size(), 0);
    n.n_consumers = 0;
    n.live = n.materialized = false;
  }
  for (uint32_t e = 0; e < b->edges.size(); e++) {
    const waa_edge_desc& ed = b->edges[e];
    Node& to = b->nodes[ed.to];
    if (muted[ed.from] || muted[ed.to]) continue;  // a muted node renders nothing and contributes nothing
    if (ed.to_input & 0x80000000u) {
      const uint32_t pid = ed.to_input & 0x7fffffffu;
      if (pid >= to.params.size()) return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - node %u has no param %u", ed.to, pid);
      const uint32_t k = to.desc.kind;
      if (!(k == WAA_NODE_GAIN || k == WAA_NODE_BIQUAD || k == WAA_NODE_DELAY || k == WAA_NODE_STEREO_PANNER ||
            k == WAA_NODE_CONSTANT_SOURCE || k == WAA_NODE_OSCILLATOR))
        return fail(WAA_ERR_OUT_OF_SCOPE, "audio-rate modulation of a host-evaluated param (node %u) is out of scope", ed.to);
      to.pin_edges[pid].push_back((int)e);
    } else {
      to.in_edges.push_back((int)e);
    }
    b->nodes[ed.from].n_consumers++;
  }
  auto by_position = [&](int x, int y) { return pos[b->edges[x].from] < pos[b->edges[y].from]; };
  for (auto& n : b->nodes) {
    std::stable_sort(n.in_edges.begin(), n.in_edges.end(), by_position);
    for (auto& pe : n.pin_edges) std::stable_sort(pe.begin(), pe.end(), by_position);
  }
  // liveness: everything that reaches the destination or an analyser
  {
    std::vector<uint32_t> stack;
    for (uint32_t i = 0; i < N; i++)
      if (b->nodes[i].desc.kind == WAA_NODE_DESTINATION || b->nodes[i].desc.kind == WAA_NODE_ANALYSER) stack.push_back(i);
    while (!stack.empty()) {
      uint32_t id = stack.back();
      stack.pop_back();
      if (b->nodes[id].live) continue;
      b->nodes[id].live = true;
      for (int e : b->nodes[id].in_edges) stack.push_back(b->edges[e].from);
      for (auto& pe : b->nodes[id].pin_edges)
        for (int e : pe) stack.push_back(b->edges[e].from);
    }
  }
  // static channel counts (the reference's counts are dynamic: a silent quantum is mono; every case the
  // static count differs from the dynamic one carries zeros — see DESIGN.md "Silence and channel counts")
  // (inside a feedback loop a producer can come later in the order: iterate to the fixed point; counts only grow)
  for (auto& n : b->nodes) n.in_nch = n.out_nch = 1;
  for (int pass = 0; pass < 16; pass++) {
  bool changed = false;
  for (uint32_t id : b->order) {
    Node& n = b->nodes[id];
    if (!n.live) continue;
    const int old_in = n.in_nch, old_out = n.out_nch;
    int maxc = 1;
    for (int e : n.in_edges) maxc = std::max(maxc, b->nodes[b->edges[e].from].out_nch);
    n.in_nch = computed_in_nch(n, maxc);
    switch (n.desc.kind) {
      case WAA_NODE_BUFFER_SOURCE: {
        uint32_t nch = 0;
        for (auto& bf : n.bufs)
          if (bf.valid) {
            if (nch && bf.nch != nch)
              return fail(WAA_ERR_OUT_OF_SCOPE, "instances of one batch must use AudioBuffers with the same channel count");
            nch = bf.nch;
          }
        n.out_nch = nch ? (int)nch : 1;
        break;
      }
      case WAA_NODE_CONSTANT_SOURCE:
      case WAA_NODE_OSCILLATOR: n.out_nch = 1; break;
      case WAA_NODE_STEREO_PANNER:
      case WAA_NODE_PANNER: n.out_nch = 2; break;
      case WAA_NODE_CONVOLVER:
        if (!n.has_ir)
          n.out_nch = n.in_nch;
        else
          n.out_nch = (n.in_nch == 1 && n.ir_nch == 1) ? 1 : 2;
        break;
      default: n.out_nch = n.in_nch; break;
    }
    if (n.in_nch > 6 || n.out_nch > 6)
      return fail(WAA_ERR_OUT_OF_SCOPE, "the device path renders at most 6 channels per signal (node %u needs %d)", id,
                  std::max(n.in_nch, n.out_nch));
    changed |= n.in_nch != old_in || n.out_nch != old_out;
  }
  if (!changed || n_scc == 0) break;
  }
  // materialisation points
  for (uint32_t id = 0; id < N; id++) {
    Node& n = b->nodes[id];
    if (!n.live) continue;
    bool mat = false;
    const uint32_t kind = n.desc.kind;
    if (kind == WAA_NODE_DESTINATION || kind == WAA_NODE_ANALYSER || kind == WAA_NODE_CONVOLVER || kind == WAA_NODE_DELAY) mat = true;
    if (scc_of[id] >= 0) mat = true;  // loop members publish their own signal
    if (kind == WAA_NODE_OSCILLATOR) mat = true;  // rendered by its own (lane-per-instance) kernel
    int live_consumers = 0;
    for (auto& e : b->edges)
      if (e.from == id && b->nodes[e.to].live) {
        live_consumers++;
        const Node& c = b->nodes[e.to];
        if ((c.desc.kind == WAA_NODE_CONVOLVER && c.has_ir) || c.desc.kind == WAA_NODE_DELAY) mat = true;
        if (e.to_input & 0x80000000u) mat = true;  // feeds an AudioParam: read back as a per-frame value signal
        if (scc_of[e.to] >= 0) mat = true;         // feeds a feedback loop
        int live_in = 0;
        for (int ie : c.in_edges)
          if (b->nodes[b->edges[ie].from].live) live_in++;
        if (live_in > 1) mat = true;
      }
    if (live_consumers != 1) mat = true;
    n.materialized = mat;
  }
  auto alloc_signal = [&](Node& n) -> int {
    float* p = nullptr;
    int e = dev_alloc(b, &p, (size_t)b->n_inst * n.out_nch * b->lp);
    if (e) return e;
    n.sig = SignalRef{p, (uint64_t)n.out_nch * b->lp, b->lp, n.out_nch, 0};
    return 0;
  };
  // planning units: single nodes and whole feedback loops, producers first (the condensed graph is acyclic),
  // otherwise in processing order
  struct Unit {
    int scc;
    uint32_t id;
  };
  std::vector<Unit> units;
  {
    std::vector<uint8_t> node_done(N, 0), scc_done(n_scc, 0);
    std::function<void(uint32_t)> visit_unit = [&](uint32_t id) {
      const int sc = scc_of[id];
      if (sc >= 0 ? scc_done[sc] : node_done[id]) return;
      std::vector<uint32_t> members;
      if (sc >= 0) {
        scc_done[sc] = 1;
        for (uint32_t m : b->order)
          if (scc_of[m] == sc) members.push_back(m);
      } else {
        node_done[id] = 1;
        members.push_back(id);
      }
      for (uint32_t m : members) {
        for (int e : b->nodes[m].in_edges)
          if (sc < 0 || scc_of[b->edges[e].from] != sc) visit_unit(b->edges[e].from);
        for (auto& pe : b->nodes[m].pin_edges)
          for (int e : pe)
            if (sc < 0 || scc_of[b->edges[e].from] != sc) visit_unit(b->edges[e].from);
      }
      units.push_back(Unit{sc, id});
    };
    for (uint32_t id : b->order) visit_unit(id);
  }
  b->steps.clear();
  std::function<int(uint32_t)> plan_single = [&](uint32_t id) -> int {
    Node& term = b->nodes[id];
    if (!term.live || !term.materialized) return 0;
    if (term.desc.kind == WAA_NODE_CONVOLVER && term.has_ir) {
      if (scc_of[id] >= 0) return fail(WAA_ERR_OUT_OF_SCOPE, "a ConvolverNode inside a feedback loop is out of scope (node %u)", id);
      int e = alloc_signal(term);
      if (e) return e;
      if ((e = plan_convolver(b, id))) return e;
      return 0;
    }
    if (term.desc.kind == WAA_NODE_OSCILLATOR) {
      int e = alloc_signal(term);
      if (e) return e;
      return plan_oscillator(b, id);
    }
    if (term.desc.kind == WAA_NODE_DELAY) {  // outside a loop: writer and reader halves back to back
      int e = alloc_signal(term);
      if (e) return e;
      if ((e = plan_delay_writer(b, id))) return e;
      if ((e = plan_delay_reader(b, id))) return e;
      return 0;
    }
    // identity node on a materialised signal of the same layout (destination / analyser / passthrough right
    // behind a materialised producer): alias instead of copying 8 B per frame-channel through HBM
    if (term.in_edges.size() == 1) {
      Node& p = b->nodes[b->edges[term.in_edges[0]].from];
      const uint32_t k = term.desc.kind;
      const bool identity = k == WAA_NODE_DESTINATION || k == WAA_NODE_ANALYSER ||
                            (k == WAA_NODE_CONVOLVER && !term.has_ir) || (k == WAA_NODE_WAVESHAPER && !term.has_curve);
      if (identity && scc_of[id] < 0 && p.materialized && p.out_nch == term.in_nch && term.in_nch == term.out_nch) {
        term.sig = p.sig;
        plan_note(b, "alias node %u -> output of node %u", id, b->edges[term.in_edges[0]].from);
        return 0;
      }
    }
    if (!term.sig.base) {  // (members of a feedback loop are allocated up front)
      int e = alloc_signal(term);
      if (e) return e;
    }
    // walk back through fused single-input predecessors
    std::vector<uint32_t> path;  // terminal first
    uint32_t cur = id;
    Step st;
    ChainDesc& cd = st.chain;
    std::memset(&cd, 0, sizeof cd);
    for (;;) {
      path.push_back(cur);
      Node& n = b->nodes[cur];
      const uint32_t kind = n.desc.kind;
      if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE) break;  // chain input = this source
      if (n.in_edges.size() != 1) break;                                               // silent or fan-in head
      uint32_t p = b->edges[n.in_edges[0]].from;
      if (b->nodes[p].materialized) break;
      cur = p;
    }
    // inputs of the head node
    const uint32_t head = path.back();
    Node& hn = b->nodes[head];
    int cmax = 1;
    if (hn.desc.kind == WAA_NODE_BUFFER_SOURCE || hn.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
      cd.n_inputs = 1;
      cd.in_nch = hn.out_nch;
      cd.in_interp = 0;
      InputRef& in = cd.in[0];
      in.nch = hn.out_nch;
      if (hn.desc.kind == WAA_NODE_BUFFER_SOURCE) {
        in.kind = IN_SOURCE;
        // filled by prepare_source below
      } else {
        in.kind = IN_CONSTANT;
      }
    } else if (hn.in_edges.empty()) {
      cd.n_inputs = 1;
      cd.in[0].kind = IN_SILENT;
      cd.in[0].nch = 1;
      cd.in_nch = hn.in_nch;
      cd.in_interp = hn.interp;
    } else {
      cd.in_nch = hn.in_nch;
      cd.in_interp = hn.interp;
      std::vector<InputRef> ins;
      for (int ie : hn.in_edges) {
        Node& pn = b->nodes[b->edges[ie].from];
        if (!pn.materialized) return fail(WAA_ERR_INVALID_STATE, "internal: unmaterialised fan-in input");
        InputRef in{};
        in.kind = IN_SIGNAL;
        in.nch = pn.out_nch;
        in.sig = pn.sig;
        ins.push_back(in);
      }
      int e = reduce_fan_in(b, ins, hn.in_nch, hn.interp);
      if (e) return e;
      cd.n_inputs = (int)ins.size();
      for (int k = 0; k < cd.n_inputs; k++) {
        cd.in[k] = ins[k];
        cmax = std::max(cmax, ins[k].nch);
      }
    }
    cmax = std::max(cmax, cd.in_nch);
    // ops: head first
    std::vector<OpDesc> ops;
    int cur_nch = cd.in_nch;
    for (size_t k = path.size(); k-- > 0;) {
      uint32_t nid = path[k];
      const bool is_head = (k == path.size() - 1);
      int out_nch = cur_nch;
      int e = emit_node_ops(b, nid, cur_nch, is_head, ops, &out_nch);
      if (e) return e;
      cur_nch = out_nch;
    }
    // source inputs: schedules / buffer tables / constant ranges
    if (cd.in[0].kind == IN_SOURCE || cd.in[0].kind == IN_CONSTANT) {
      int e = prepare_source_input(b, head, &cd.in[0]);
      if (e) return e;
    }
    std::vector<InputRef> inputs(cd.in, cd.in + cd.n_inputs);
    int e = emit_segments(b, inputs, cd.in_nch, cd.in_interp, ops, term.sig);
    if (e) return e;
    return 0;
  };
  // chains, in processing order of their terminal node
  for (const Unit& unit : units) {
    const uint32_t id = unit.id;
    if (unit.scc >= 0) {
      std::vector<uint32_t> loop_items;
      bool any_live = false;
      for (uint32_t v : items)
        if (scc_of[v & ~VTX_READER] == unit.scc) {
          loop_items.push_back(v);
          any_live |= b->nodes[v & ~VTX_READER].live;
        }
      if (!any_live) continue;
      for (uint32_t v : loop_items) {
        Node& m = b->nodes[v & ~VTX_READER];
        if (!m.sig.base) {
          int e = alloc_signal(m);
          if (e) return e;
        }
      }
      const uint32_t bt = loop_block_tiles(b, loop_items);
      if (bt == 0) {  // short or modulated loop delay: quantum-serial loop kernel
        int e = plan_loop(b, loop_items);
        if (e) return e;
        continue;
      }
      // Block-scheduled loop: every delay that breaks the loop is longer than `bt` tiles, so a block of bt tiles
      // only reads loop history from earlier blocks: the members are planned as ordinary node-major steps (same
      // kernels as outside a loop) in the reference's processing order and launched block by block.
      const size_t first_step = b->steps.size();
      for (uint32_t v : loop_items) {
        const uint32_t mid = v & ~VTX_READER;
        int e = 0;
        if (is_delay(b, mid))
          e = (v & VTX_READER) ? plan_delay_reader(b, mid) : plan_delay_writer(b, mid);
        else
          e = plan_single(mid);
        if (e) return e;
      }
      const int group = (int)b->group_tiles.size();
      b->group_tiles.push_back(bt);
      for (size_t k = first_step; k < b->steps.size(); k++) {
        Step& st = b->steps[k];
        st.group = group;
        // steps that only depend on data from outside the loop run once, over the full range, before the blocks
        st.prologue = st.kind == 5 || st.kind == 3 || (st.kind == 0 && st.chain.n_ops == 1 && st.chain.ops[0].kind == OP_PARAM_ADD);
        if (st.kind == 2 || st.kind == 4)
          return fail(WAA_ERR_OUT_OF_SCOPE, "this node kind cannot be rendered inside a feedback loop");
      }
      plan_note(b, "feedback loop: block-scheduled, %u tile(s) = %u frames per block, %zu step(s) per block", bt, bt * TILE,
                b->steps.size() - first_step);
      continue;
    }
    int e = plan_single(id);
    if (e) return e;
  }
  b->planned = true;
  return 0;
}

}  // namespace

// Resolve a source node into an InputRef: schedules, per-instance buffer table, constant ranges.
static int prepare_source_input(waa_batch* b, uint32_t id, InputRef* in) {
  Node& n = b->nodes[id];
  if (n.desc.kind == WAA_NODE_CONSTANT_SOURCE) {
    int e = node_param(b, id, 0, &in->offset);
    if (e) return e;
    // active frame range per instance (constant_source.rs:203-258), found by replaying the quantum loop
    std::vector<int64_t> act((size_t)b->n_inst * 2);
    const double dt = 1. / (double)b->sr;
    for (uint32_t i = 0; i < b->n_inst; i++) {
      const double start = n.sched[i].start, stop = n.sched[i].stop;
      int64_t a0 = -1, a1 = -1;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const double ct = (double)((uint64_t)q * RQ) / (double)b->sr;
        const double nbt = ct + dt * (double)RQ;
        if (start >= nbt) continue;
        if (start <= ct && stop >= nbt) {
          if (a0 < 0) a0 = (int64_t)q * RQ;
          a1 = (int64_t)(q + 1) * RQ;
        } else {
          double t = ct;
          for (int s = 0; s < RQ; s++) {
            if (!(t < start || t >= stop)) {
              if (a0 < 0) a0 = (int64_t)q * RQ + s;
              a1 = (int64_t)q * RQ + s + 1;
            }
            t += dt;
          }
        }
        if (stop <= nbt) break;
      }
      act[(size_t)i * 2] = a0 < 0 ? 0 : a0;
      act[(size_t)i * 2 + 1] = a0 < 0 ? 0 : a1;
    }
    int64_t* d = nullptr;
    e = dev_upload(b, &d, act);
    if (e) return e;
    in->active = d;
    plan_note(b, "constant source node %u: active frames [%lld, %lld) for instance 0", id, (long long)act[0], (long long)act[1]);
    return 0;
  }
  // AudioBufferSourceNode
  std::vector<SrcInst> insts(b->n_inst);
  std::vector<SrcSchedule> scheds;
  std::map<SchedKey, uint32_t> dedup;
  const ParamStore& p_rate = n.params[WAA_PARAM_SOURCE_PLAYBACK_RATE];
  const ParamStore& p_det = n.params[WAA_PARAM_SOURCE_DETUNE];
  const bool automated = !p_rate.blocks.empty() || !p_det.blocks.empty();
  for (uint32_t i = 0; i < b->n_inst; i++) {
    const DeviceBuffer& bf = n.bufs[i];
    SrcInst& si = insts[i];
    si.base = bf.base;
    si.ch_stride = bf.ch_stride;
    si.frames = bf.frames;
    si.aligned = (bf.valid && ((uintptr_t)bf.base % 16 == 0) && (bf.ch_stride % 4 == 0)) ? 1 : 0;
    std::vector<float> rate_q = param_per_quantum(b, p_rate, i, nullptr);
    std::vector<float> det_q = param_per_quantum(b, p_det, i, nullptr);
    const SourceSched& ss = n.sched[i];
    const SchedKey key(ss.start, ss.stop, ss.offset, ss.duration, ss.looping, ss.loop_start, ss.loop_end,
                       bf.valid ? bf.frames : 0, bf.valid ? bf.sr : 0.f, rate_q[0], det_q[0]);
    if (!automated) {
      auto it = dedup.find(key);
      if (it != dedup.end()) {
        si.sched = it->second;
        continue;
      }
    }
    SchedOut so;
    schedule_source(b, n.sched[i], bf.frames, bf.sr, bf.valid, rate_q, det_q, &so);
    {
      uint32_t nf = 0, nl = 0, ns = 0, nt = 0;
      for (auto& r : so.qrec) {
        nf += r.mode == Q_FAST;
        nl += r.mode == Q_FAST_LOOP;
        ns += r.mode == Q_SLOW;
      }
      for (auto t : so.tile_fast) nt += t;
      plan_note(b, "source node %u schedule %zu: quanta fast=%u fast_loop=%u slow=%u silent=%u fast_tiles=%u/%u", id,
                scheds.size(), nf, nl, ns, (uint32_t)so.qrec.size() - nf - nl - ns, nt, b->n_tiles);
    }
    SrcSchedule ds{};
    QRec* dq = nullptr;
    int e = dev_upload(b, &dq, so.qrec);
    if (e) return e;
    ds.qrec = dq;
    if (so.any_slow) {
      SlowRec* dsr = nullptr;
      e = dev_upload(b, &dsr, so.slow);
      if (e) return e;
      ds.slow = dsr;
    }
    uint8_t* dtf = nullptr;
    e = dev_upload(b, &dtf, so.tile_fast);
    if (e) return e;
    ds.tile_fast = dtf;
    si.sched = (uint32_t)scheds.size();
    scheds.push_back(ds);
    if (!automated) dedup[key] = si.sched;
  }
  SrcInst* d_insts = nullptr;
  int e = dev_upload(b, &d_insts, insts);
  if (e) return e;
  SrcSchedule* d_scheds = nullptr;
  e = dev_upload(b, &d_scheds, scheds);
  if (e) return e;
  in->src = d_insts;
  in->sched = d_scheds;
  plan_note(b, "source node %u: %zu distinct schedule(s) for %u instance(s)", id, scheds.size(), b->n_inst);
  return 0;
}

namespace {

// Fan-in above MAX_INPUTS: sum the first MAX_INPUTS inputs (mixed to the receiver's channel count) into a
// temporary signal and continue; the left-to-right order of the f32 additions (graph.rs:524-535) is kept.
int reduce_fan_in(waa_batch* b, std::vector<InputRef>& ins, int in_nch, int interp) {
  while (ins.size() > (size_t)MAX_INPUTS) {
    float* ptr = nullptr;
    int e = dev_alloc(b, &ptr, (size_t)b->n_inst * in_nch * b->lp);
    if (e) return e;
    Step st;
    ChainDesc& cd = st.chain;
    std::memset(&cd, 0, sizeof cd);
    cd.n_inputs = MAX_INPUTS;
    for (int k = 0; k < MAX_INPUTS; k++) cd.in[k] = ins[k];
    cd.in_nch = in_nch;
    cd.in_interp = interp;
    cd.out = SignalRef{ptr, (uint64_t)in_nch * b->lp, b->lp, in_nch, 0};
    cd.n_inst = b->n_inst;
    cd.n_tiles = b->n_tiles;
    cd.tile0 = 0;
    cd.tile1 = b->n_tiles;
  cd.tile0 = 0;
  cd.tile1 = b->n_tiles;
    cd.n_quanta = b->n_quanta;
    int cmax = in_nch;
    for (int k = 0; k < MAX_INPUTS; k++) cmax = std::max(cmax, ins[k].nch);
    st.cmax = cmax;
    st.profile_slot = slot_for(b, cmax <= 1 ? "chain_kernel<1>" : "chain_kernel<2>");
    b->steps.push_back(st);
    InputRef partial{};
    partial.kind = IN_SIGNAL;
    partial.nch = in_nch;
    partial.sig = cd.out;
    ins.erase(ins.begin(), ins.begin() + MAX_INPUTS);
    ins.insert(ins.begin(), partial);
    plan_note(b, "fan-in partial sum of %d inputs -> %dch", MAX_INPUTS, in_nch);
  }
  return 0;
}

// ConvolverNode with an impulse response (convolver.rs:259-317, 343-490): input mix chain (if needed)
// + forward FFT / spectral MAC / inverse FFT steps.
// Input of a node-major step (convolver, delay): the single producer's signal if its channel count already
// matches, else a mixing chain into a temporary.
int node_input_signal(waa_batch* b, uint32_t id, SignalRef* out_sig, const SignalRef* target = nullptr) {
  Node& n = b->nodes[id];
  if (!target && n.in_edges.size() == 1) {
    Node& p = b->nodes[b->edges[n.in_edges[0]].from];
    if (p.materialized && p.out_nch == n.in_nch) {
      *out_sig = p.sig;
      return 0;
    }
  }
  SignalRef in_sig;
  int e = 0;
  if (target)
    in_sig = *target;  // mix into a signal somebody already reads from
  else
    e = temp_signal(b, n.in_nch, &in_sig);
  if (e) return e;
  std::vector<InputRef> ins;
  if (n.in_edges.empty()) {
    InputRef in{};
    in.kind = IN_SILENT;
    in.nch = 1;
    ins.push_back(in);
  } else {
    for (int ie : n.in_edges) {
      Node& pn = b->nodes[b->edges[ie].from];
      if (!pn.materialized) return fail(WAA_ERR_INVALID_STATE, "internal: unmaterialised input of a node-major step");
      InputRef in{};
      in.kind = IN_SIGNAL;
      in.nch = pn.out_nch;
      in.sig = pn.sig;
      ins.push_back(in);
    }
    if ((e = reduce_fan_in(b, ins, n.in_nch, n.interp))) return e;
  }
  if ((e = push_chain_step(b, ins, n.in_nch, n.interp, {}, in_sig))) return e;
  *out_sig = in_sig;
  return 0;
}

// OscillatorNode (oscillator.rs:323-660): one kernel, one lane per instance (the phase accumulator is serial)
int plan_oscillator(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  Step st;
  st.kind = 9;
  OscDesc& d = st.osc;
  std::memset(&d, 0, sizeof d);
  int e;
  if ((e = node_param(b, id, WAA_PARAM_OSCILLATOR_FREQUENCY, &d.frequency)) ||
      (e = node_param(b, id, WAA_PARAM_OSCILLATOR_DETUNE, &d.detune)))
    return e;
  std::vector<double> start(b->n_inst), stop(b->n_inst);
  for (uint32_t i = 0; i < b->n_inst; i++) {
    start[i] = n.sched[i].start;
    stop[i] = n.sched[i].stop;
  }
  double *d_start = nullptr, *d_stop = nullptr;
  if ((e = dev_upload(b, &d_start, start)) || (e = dev_upload(b, &d_stop, stop))) return e;
  d.start = d_start;
  d.stop = d_stop;
  d.type = n.osc_wave.empty() ? n.desc.i[0] : WAA_OSC_CUSTOM;
  if (d.type == WAA_OSC_CUSTOM && n.osc_wave.empty())
    return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - custom oscillator %u has no PeriodicWave", id);
  std::vector<float> table;
  if (d.type == WAA_OSC_CUSTOM) {
    table = n.osc_wave;
  } else {  // oscillator.rs:16-28 (same libm sinf as the reference's f32::sin)
    table.resize(2048);
    const float pi = 3.14159265358979323846f;
    for (int x = 0; x < 2048; x++) table[x] = std::sin(((float)x) * 2.0f * pi * (1.f / 2048.f));
  }
  float* d_table = nullptr;
  if ((e = dev_upload(b, &d_table, table))) return e;
  d.table = d_table;
  d.table_len = (int32_t)table.size();
  d.out = n.sig;
  d.frames = b->lp;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.sample_rate = (double)b->sr;
  const bool parallel = d.frequency.mode != 2 && d.detune.mode != 2 && !getenv("WAA_OSC_EXACT");
  if (parallel) {
    // host-known frequency: replay the per-quantum decisions of OscillatorRenderer::process (oscillator.rs:336-452)
    // and record the phase at the first active frame of every quantum
    std::vector<OscQuantum> tq((size_t)b->n_inst * b->n_quanta);
    const double sample_rate = (double)b->sr, dt = 1. / sample_rate, nyquist = sample_rate / 2.;
    auto frac = [](long double x) {
      long double r = x - floorl(x);
      return (double)(r >= 1.L ? r - 1.L : r);
    };
    for (uint32_t i = 0; i < b->n_inst; i++) {
      const auto fq = param_per_quantum(b, n.params[WAA_PARAM_OSCILLATOR_FREQUENCY], i, nullptr);
      const auto dq = param_per_quantum(b, n.params[WAA_PARAM_OSCILLATOR_DETUNE], i, nullptr);
      double start_time = start[i];
      const double stop_time = stop[i];
      long double phase = 0.L;
      bool started = false;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        OscQuantum& oq = tq[(size_t)i * b->n_quanta + q];
        oq = OscQuantum{0., 0., 0, 0, 0};
        const double block_time = (double)((uint64_t)q * RQ) / sample_rate;
        const double next_block_time = block_time + dt * (double)RQ;
        if (stop_time <= block_time || start_time >= next_block_time) continue;
        if (!started && start_time < block_time) start_time = block_time;
        const float f = fq[fq.size() == 1 ? 0 : q], det = dq[dq.size() == 1 ? 0 : q];
        const double computed_freq = (double)f * std::exp2((double)det / 1200.);
        const double incr = computed_freq / sample_rate;
        oq.incr = incr;
        oq.outside_nyquist = std::fabs(computed_freq) >= nyquist ? 1 : 0;
        // the reference advances current_time by repeated addition: replay it to find the active frame range
        double current_time = block_time;
        int first = -1, end = RQ;
        for (int k = 0; k < RQ; k++) {
          const bool active = !(current_time < start_time || current_time >= stop_time);
          if (active && first < 0) {
            first = k;
            if (!started) {
              if (current_time > start_time) phase = frac((long double)incr * (long double)((current_time - start_time) / dt));
              started = true;
            }
          }
          if (!active && first >= 0) {
            end = k;
            break;
          }
          current_time += dt;
        }
        if (first < 0) continue;
        oq.first = (int16_t)first;
        oq.end = (int16_t)end;
        oq.phase = (double)phase;
        phase = frac(phase + (long double)(end - first) * (long double)incr);
      }
    }
    OscQuantum* d_tq = nullptr;
    if ((e = dev_upload(b, &d_tq, tq))) return e;
    d.table_q = d_tq;
  }
  const bool scan = !parallel && !getenv("WAA_OSC_EXACT");
  if (scan) {
    // a-rate / graph-modulated frequency: the device forms the phase as a prefix sum of per-frame increments; the
    // host replays only the reference's clock (current_time += dt per frame, oscillator.rs:505-552) to find the
    // active frame range and the sub-sample start offset of every instance
    std::vector<int64_t> act((size_t)b->n_inst * 2, 0);
    std::vector<double> ratio(b->n_inst, 0.);
    const double sample_rate = (double)b->sr, dt = 1. / sample_rate;
    for (uint32_t i = 0; i < b->n_inst; i++) {
      double start_time = start[i];
      const double stop_time = stop[i];
      int64_t first = -1, end = -1;
      bool started = false;
      for (uint32_t q = 0; q < b->n_quanta; q++) {
        const double block_time = (double)((uint64_t)q * RQ) / sample_rate;
        const double next_block_time = block_time + dt * (double)RQ;
        if (stop_time <= block_time || start_time >= next_block_time) continue;
        if (!started && start_time < block_time) start_time = block_time;
        double current_time = block_time;
        for (int k = 0; k < RQ; k++) {
          const bool active = !(current_time < start_time || current_time >= stop_time);
          if (active) {
            if (first < 0) {
              first = (int64_t)q * RQ + k;
              if (current_time > start_time) ratio[i] = (current_time - start_time) / dt;
              started = true;
            }
            end = (int64_t)q * RQ + k + 1;
          }
          current_time += dt;
        }
      }
      act[(size_t)i * 2] = first < 0 ? 0 : first;
      act[(size_t)i * 2 + 1] = first < 0 ? 0 : end;
    }
    int64_t* d_act = nullptr;
    double* d_ratio = nullptr;
    if ((e = dev_upload(b, &d_act, act)) || (e = dev_upload(b, &d_ratio, ratio))) return e;
    d.active = d_act;
    d.start_ratio = d_ratio;
  }
  st.profile_slot = slot_for(b, parallel ? "osc_par_kernel" : scan ? "osc_scan_kernel" : "osc_kernel");
  b->steps.push_back(st);
  static const char* names[] = {"sine", "square", "sawtooth", "triangle", "custom"};
  plan_note(b, "oscillator node %u: %s (%s) frequency=%s detune=%s", id, names[d.type],
            parallel ? "time-parallel, closed-form phase" : scan ? "prefix-sum phase" : "lane per instance, serial phase",
            d.frequency.mode == 0 ? "const" : d.frequency.mode == 1 ? "k-rate" : "a-rate",
            d.detune.mode == 0 ? "const" : d.detune.mode == 1 ? "k-rate" : "a-rate");
  return 0;
}

// DelayNode (delay.rs:428-745).  Writer half: the node's mixed input becomes the delay line `hist` (an alias of the
// producer's signal when nothing has to be mixed).  Reader half: one gather kernel from the delay line.  Outside a
// loop the two are planned back to back; inside a block-scheduled loop each at its own place in the order.
int plan_delay_writer(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  if (n.hist.base) {  // the reader half was planned first (inside a loop) and chose the delay line
    if (!n.hist_is_temp) return 0;
    SignalRef same;
    return node_input_signal(b, id, &same, &n.hist);
  }
  return node_input_signal(b, id, &n.hist);
}
int plan_delay_reader(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  Step st;
  st.kind = 7;
  DelayDesc& d = st.delay;
  std::memset(&d, 0, sizeof d);
  const bool in_cycle = id < b->cut.size() && b->cut[id];
  if (in_cycle && !n.hist.base) {
    // the reader renders before its writer: the delay line is not planned yet.  It is the producer's signal when
    // there is exactly one materialised producer of the right layout, else a temporary the writer half fills.
    bool direct = false;
    if (n.in_edges.size() == 1) {
      Node& p = b->nodes[b->edges[n.in_edges[0]].from];
      direct = p.materialized && p.out_nch == n.in_nch && p.sig.base;
      if (direct) n.hist = p.sig;
    }
    if (!direct) {
      int e = temp_signal(b, n.in_nch, &n.hist);
      if (e) return e;
      n.hist_is_temp = true;
    }
  }
  if (!n.hist.base) return fail(WAA_ERR_INVALID_STATE, "internal: delay line of node %u not planned", id);
  d.in = n.hist;
  d.out = n.sig;
  int e = node_param(b, id, WAA_PARAM_DELAY_DELAY_TIME, &d.delay);
  if (e) return e;
  d.sample_rate = (double)b->sr;
  d.frames = b->lp;
  d.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);
  d.nch = n.in_nch;
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.tile0 = 0;
  d.tile1 = b->n_tiles;
  d.in_cycle = in_cycle ? 1 : 0;
  const double dt = 1. / (double)b->sr;
  d.quantum_duration = (double)RQ * dt;  // delay.rs:546-548
  st.profile_slot = slot_for(b, "delay_kernel");
  b->steps.push_back(st);
  plan_note(b, "delay node %u: %dch delayTime=%s ring=%d quanta%s", id, d.nch,
            d.delay.mode == 0 ? "const" : d.delay.mode == 1 ? "k-rate" : "a-rate", d.num_quanta + 1,
            in_cycle ? " (in a loop: clamped to one quantum)" : "");
  return 0;
}

// Block size (in 2048-frame tiles) for a block-scheduled feedback loop, 0 if the loop needs the quantum-serial
// kernel.  Every DelayNode whose writer->reader edge the cycle breaker removed must have a host-known delay
// (constant or k-rate blocks, not modulated from the graph) strictly longer than the block: then no frame of a
// block depends on loop history of the same block.
uint32_t loop_block_tiles(waa_batch* b, const std::vector<uint32_t>& loop_items) {
  if (getenv("WAA_LOOP_KERNEL")) return 0;  // debugging aid: force the quantum-serial kernel
  const double dt = 1. / (double)b->sr;
  const double quantum_duration = (double)RQ * dt;
  double dmin = 1e300;
  for (uint32_t v : loop_items) {
    const uint32_t id = v & ~VTX_READER;
    Node& n = b->nodes[id];
    if (!(v & VTX_READER)) {
      if (n.desc.kind == WAA_NODE_CONVOLVER && n.has_ir) return 0;
      continue;
    }
    if (!b->cut[id]) continue;  // keeps its writer->reader edge: reads the current block like any other node
    if (param_mode(n, WAA_PARAM_DELAY_DELAY_TIME) == 2) return 0;
    for (uint32_t i = 0; i < b->n_inst; i++)
      for (float dv : param_per_quantum(b, n.params[WAA_PARAM_DELAY_DELAY_TIME], i, nullptr))
        dmin = std::min(dmin, std::max((double)dv, quantum_duration) * (double)b->sr);
  }
  if (!(dmin < 1e300)) return 0;
  const double tiles = std::ceil(dmin / (double)TILE) - 1.;  // block < delay, strictly
  if (tiles < 1.) return 0;
  return (uint32_t)std::min(tiles, 64.);
}

// A feedback loop (strongly connected group around at least one DelayNode): one loop_kernel launch renders all
// members quantum by quantum in the reference's processing order.  `loop_items` = that order, two entries per
// DelayNode (writer / reader halves).
int plan_loop(waa_batch* b, const std::vector<uint32_t>& loop_items) {
  if (loop_items.size() > (size_t)LOOP_MAX_ITEMS)
    return fail(WAA_ERR_OUT_OF_SCOPE, "feedback loop with more than %d members", LOOP_MAX_ITEMS);
  std::map<uint32_t, int> out_item;  // node id -> item that produces its output
  std::map<uint32_t, int> writer_item;
  for (size_t k = 0; k < loop_items.size(); k++) {
    const uint32_t v = loop_items[k], id = v & ~VTX_READER;
    if (is_delay(b, id)) {
      if (v & VTX_READER)
        out_item[id] = (int)k;
      else
        writer_item[id] = (int)k;
    } else {
      out_item[id] = (int)k;
    }
  }
  std::vector<LoopItem> host(loop_items.size());
  std::string desc;
  for (size_t k = 0; k < loop_items.size(); k++) {
    const uint32_t v = loop_items[k], id = v & ~VTX_READER;
    Node& n = b->nodes[id];
    LoopItem& li = host[k];
    std::memset(&li, 0, sizeof li);
    if (n.in_nch > 2 || n.out_nch > 2)
      return fail(WAA_ERR_OUT_OF_SCOPE, "feedback loops render at most 2 channels per signal (node %u)", id);
    for (auto& pe : n.pin_edges)
      for (int e : pe)
        if (out_item.count(b->edges[e].from))
          return fail(WAA_ERR_OUT_OF_SCOPE, "an AudioParam of node %u is modulated from inside its own feedback loop", id);
    const bool reader = is_delay(b, id) && (v & VTX_READER);
    li.nch_in = n.in_nch;
    li.nch_out = n.out_nch;
    li.interp = n.interp;
    if (!reader) {
      // inputs of the node (of the writer half for a DelayNode), in summing order
      if (n.in_edges.size() > (size_t)MAX_INPUTS)
        return fail(WAA_ERR_OUT_OF_SCOPE, "more than %d inputs on node %u inside a feedback loop", MAX_INPUTS, id);
      li.n_in = (int)n.in_edges.size();
      for (int j = 0; j < li.n_in; j++) {
        const uint32_t pid = b->edges[n.in_edges[j]].from;
        Node& pn = b->nodes[pid];
        li.in_nch[j] = pn.out_nch;
        auto it = out_item.find(pid);
        if (it != out_item.end()) {
          if (it->second >= (int)k) return fail(WAA_ERR_INVALID_STATE, "internal: loop member order");
          li.in_item[j] = it->second;
        } else {
          if (!pn.materialized || !pn.sig.base) return fail(WAA_ERR_INVALID_STATE, "internal: loop input not planned");
          li.in_item[j] = -1;
          li.in_sig[j] = pn.sig;
        }
      }
    }
    char t[96];
    if (is_delay(b, id)) {
      if (!reader) {
        li.kind = LI_DELAY_W;
        int e = temp_signal(b, n.in_nch, &li.out);  // the delay line, in absolute time
        if (e) return e;
        snprintf(t, sizeof t, "delayW%u", id);
      } else {
        li.kind = LI_DELAY_R;
        li.out = n.sig;
        li.writer_item = writer_item.at(id);
        li.in_cycle = li.writer_item > (int)k ? 1 : 0;  // delay.rs:535-541: the writer has not rendered yet
        li.num_quanta = (int32_t)std::ceil(n.desc.d[0] * (double)b->sr / (double)RQ);
        int e = node_param(b, id, WAA_PARAM_DELAY_DELAY_TIME, &li.op.p0);
        if (e) return e;
        snprintf(t, sizeof t, "delayR%u%s", id, li.in_cycle ? "(clamped)" : "");
      }
    } else {
      li.kind = LI_NODE;
      li.out = n.sig;
      std::vector<OpDesc> ops;
      int out_nch = 0;
      int e = emit_node_ops(b, id, n.in_nch, true, ops, &out_nch);
      if (e) return e;
      if (ops.size() > 1) return fail(WAA_ERR_OUT_OF_SCOPE, "node %u cannot be rendered inside a feedback loop", id);
      if (!ops.empty()) {
        const OpDesc& o = ops[0];
        const bool ok = o.kind == OP_GAIN || o.kind == OP_BIQUAD || o.kind == OP_WAVESHAPER ||
                        (o.kind == OP_STEREO_PAN && o.p0.mode != 2);
        if (!ok)
          return fail(WAA_ERR_OUT_OF_SCOPE, "node %u (%s) cannot be rendered inside a feedback loop on the device path", id,
                      op_name(o.kind));
        li.op = o;
      }
      snprintf(t, sizeof t, "%s%u", ops.empty() ? "pass" : op_name(ops[0].kind), id);
    }
    desc += desc.empty() ? t : std::string(",") + t;
  }
  // fix up the reader items' writer outputs are read through host[writer_item].out on the device: same array
  LoopItem* dev = nullptr;
  int e = dev_upload(b, &dev, host);
  if (e) return e;
  Step st;
  st.kind = 8;
  LoopDesc& d = st.loop;
  std::memset(&d, 0, sizeof d);
  d.items = dev;
  d.n_items = (int32_t)host.size();
  d.n_inst = b->n_inst;
  d.n_quanta = b->n_quanta;
  d.sample_rate = (double)b->sr;
  const double dt = 1. / (double)b->sr;
  d.quantum_duration = (double)RQ * dt;  // delay.rs:546-548
  st.profile_slot = slot_for(b, "loop_kernel");
  b->steps.push_back(st);
  plan_note(b, "feedback loop: %d item(s) per quantum [%s]", d.n_items, desc.c_str());
  return 0;
}

int plan_convolver(waa_batch* b, uint32_t id) {
  Node& n = b->nodes[id];
  SignalRef in_sig{};
  {
    int e = node_input_signal(b, id, &in_sig);
    if (e) return e;
  }
  // one FFTConvolver per IR channel, at least two (convolver.rs:291-306); each trims its own trailing
  // |h| < 1e-6 samples (fft-convolver init) — only the longest trimmed length matters here
  const int ir_nch = n.ir_nch;
  uint64_t len = 0;
  for (int c = 0; c < ir_nch; c++) {
    uint64_t l = n.ir_len;
    while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;
    len = std::max(len, l);
  }
  Step st;
  st.kind = 2;
  ConvDesc& cv = st.conv;
  std::memset(&cv, 0, sizeof cv);
  if (len == 0) {
    // all-zero impulse response: FFTConvolver::process outputs zeros
    Step z;
    z.kind = 3;
    z.zero_ptr = n.sig.base;
    z.zero_bytes = (size_t)b->n_inst * n.out_nch * b->lp * sizeof(float);
    b->steps.push_back(z);
    plan_note(b, "convolver node %u: all-zero impulse response -> zero fill", id);
    return 0;
  }
  const bool direct_fir = len <= (uint64_t)DIRECT_MAX_TAPS;
  int B = 8192;
  for (int cand : {128, 512, 2048, 8192})
    if ((len + cand - 1) / cand <= 24) {
      B = cand;
      break;
    }
  cv.block = B;
  cv.n = 2 * B;
  cv.parts = (int)((len + B - 1) / B);
  cv.nb = (int)((b->lp + B - 1) / B);
  cv.cin = n.in_nch;
  cv.cout = n.out_nch;
  cv.in = in_sig;
  cv.out = n.sig;
  cv.frames = b->lp;
  cv.n_inst = b->n_inst;
  cv.n_pairs = (b->n_inst + 1) / 2;
  cv.ir_nch = ir_nch;
  cv.ir_len = len;
  // routing (convolver.rs:384-466)
  auto term = [&](int in_ch, int ir_ch, int out_ch) { cv.terms[cv.n_terms++] = ConvTerm{in_ch, ir_ch, out_ch, 0}; };
  if (n.in_nch == 1 && ir_nch == 1) {
    term(0, 0, 0);
  } else if (n.in_nch == 1 && ir_nch == 2) {
    term(0, 0, 0);
    term(0, 1, 1);
  } else if (n.in_nch == 2 && ir_nch == 1) {
    term(0, 0, 0);
    term(1, 0, 1);
  } else if (n.in_nch == 2 && ir_nch == 2) {
    term(0, 0, 0);
    term(1, 1, 1);
  } else if (n.in_nch == 2 && ir_nch == 4) {
    term(0, 0, 0);
    term(0, 1, 1);
    term(1, 2, 0);
    term(1, 3, 1);
  } else {
    term(0, 0, 0);
    term(0, 1, 1);
    term(0, 2, 0);
    term(0, 3, 1);
  }
  // device resources
  std::vector<float> irflat((size_t)ir_nch * len);
  for (int c = 0; c < ir_nch; c++) {
    uint64_t l = n.ir_len;
    while (l > 0 && std::fabs(n.ir[c][l - 1]) < 0.000001f) l--;  // samples past a channel's own trim are dropped
    for (uint64_t i = 0; i < len; i++) irflat[(size_t)c * len + i] = i < l ? n.ir[c][i] : 0.f;
  }
  float* d_ir = nullptr;
  int e = dev_upload(b, &d_ir, irflat);
  if (e) return e;
  cv.ir = d_ir;
  if (direct_fir) {
    st.kind = 4;
    st.slot_mac = slot_for(b, "conv_direct_kernel");
    b->steps.push_back(st);
    plan_note(b, "convolver node %u: direct FIR taps=%llu cin=%d cout=%d terms=%d", id, (unsigned long long)len, cv.cin,
              cv.cout, cv.n_terms);
    return 0;
  }
  std::vector<Cplx> tw(cv.n);
  for (int t = 0; t < cv.n; t++) {
    const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)cv.n;
    tw[t] = Cplx{(float)std::cos(a), (float)std::sin(a)};
  }
  Cplx* d_tw = nullptr;
  if ((e = dev_upload(b, &d_tw, tw))) return e;
  cv.tw = d_tw;
  Cplx *dH = nullptr, *dX = nullptr, *dY = nullptr;
  if ((e = dev_alloc(b, &dH, (size_t)ir_nch * cv.parts * cv.n))) return e;
  if ((e = dev_alloc(b, &dX, (size_t)cv.n_pairs * cv.cin * cv.nb * cv.n))) return e;
  if ((e = dev_alloc(b, &dY, (size_t)cv.n_pairs * cv.cout * cv.nb * cv.n))) return e;
  cv.H = dH;
  cv.X = dX;
  cv.Y = dY;
  if (!b->dry) {
    launch_conv_ir_spectra(cv, b->stream);  // control-side work of ConvolverNode::set_buffer, once
    HIP_TRY(hipGetLastError());
  }
  plan_note(b, "convolver node %u: fft B=%d N=%d P=%d blocks=%d pairs=%u cin=%d cout=%d terms=%d ir_len=%llu", id, cv.block,
            cv.n, cv.parts, cv.nb, cv.n_pairs, cv.cin, cv.cout, cv.n_terms, (unsigned long long)len);
  st.slot_fwd = slot_for(b, "conv_fft_kernel<fwd>");
  st.slot_mac = slot_for(b, "conv_mac_kernel");
  st.slot_inv = slot_for(b, "conv_fft_kernel<inv>");
  b->steps.push_back(st);
  return 0;
}

// Emit the fused ops of node `id` given the running channel count.
int emit_node_ops(waa_batch* b, uint32_t id, int cur_nch, bool head, std::vector<OpDesc>& ops, int* out_nch) {
  Node& n = b->nodes[id];
  const uint32_t kind = n.desc.kind;
  if (kind == WAA_NODE_BUFFER_SOURCE || kind == WAA_NODE_CONSTANT_SOURCE) {
    *out_nch = n.out_nch;
    return 0;
  }
  // input mixing to the node's computed channel count (quantum.rs:532-569); the chain head's inputs are
  // mixed by the input stage already
  if (!head && cur_nch != n.in_nch) {
    OpDesc m{};
    m.kind = OP_MIX;
    m.nch_in = cur_nch;
    m.nch_out = n.in_nch;
    m.i0 = n.interp;
    ops.push_back(m);
  }
  const int nch = n.in_nch;
  *out_nch = n.out_nch;
  switch (kind) {
    case WAA_NODE_GAIN: {
      OpDesc o{};
      o.kind = OP_GAIN;
      o.nch_in = o.nch_out = nch;
      int e = node_param(b, id, 0, &o.p0);
      if (e) return e;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_BIQUAD: {
      OpDesc o{};
      o.kind = OP_BIQUAD;
      o.nch_in = o.nch_out = nch;
      bool varies = false, a_rate = false;
      for (size_t k = 0; k < n.params.size(); k++) {
        if (param_mode(n, k) == 2) a_rate = true;
        if (param_mode(n, k) == 1) varies = true;
      }
      if (a_rate) {
        // a-rate params: coefficients per frame (biquad_filter.rs:837-855), computed on the device in f64 from
        // the per-frame param values into a table the chain kernel streams
        Step cs;
        cs.kind = 5;
        BiquadCoefDesc& cdsc = cs.coef;
        std::memset(&cdsc, 0, sizeof cdsc);
        int e;
        if ((e = node_param(b, id, WAA_PARAM_BIQUAD_FREQUENCY, &cdsc.frequency)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_DETUNE, &cdsc.detune)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_Q, &cdsc.q)) ||
            (e = node_param(b, id, WAA_PARAM_BIQUAD_GAIN, &cdsc.gain)))
          return e;
        cdsc.n_frames = (uint64_t)b->n_quanta * RQ;
        cdsc.n_inst = b->n_inst;
        cdsc.type = n.desc.i[0];
        cdsc.sample_rate = b->sr;
        double* dco = nullptr;
        if ((e = dev_alloc(b, &dco, (size_t)b->n_inst * cdsc.n_frames * 5))) return e;
        cdsc.coefs = dco;
        cs.profile_slot = slot_for(b, "biquad_coef_kernel");
        b->steps.push_back(cs);
        double* dst = nullptr;
        if ((e = dev_alloc(b, &dst, (size_t)b->n_inst * STATE_STRIDE))) return e;
        b->state_bufs.push_back({dst, (size_t)b->n_inst * STATE_STRIDE * sizeof(double)});
        o.i0 = 2;
        o.ptr0 = dco;
        o.ptr1 = dst;
        o.u0 = cdsc.n_frames * 5;
        ops.push_back(o);
        break;
      }
      const uint64_t per = varies ? (uint64_t)b->n_quanta * 5 : 5;
      std::vector<double> co((size_t)b->n_inst * per);
      for (uint32_t i = 0; i < b->n_inst; i++) {
        auto f = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_FREQUENCY], i, nullptr);
        auto d = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_DETUNE], i, nullptr);
        auto q = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_Q], i, nullptr);
        auto g = param_per_quantum(b, n.params[WAA_PARAM_BIQUAD_GAIN], i, nullptr);
        const uint32_t cnt = varies ? b->n_quanta : 1;
        for (uint32_t k = 0; k < cnt; k++) {
          auto at = [&](const std::vector<float>& v) { return v[v.size() == 1 ? 0 : k]; };
          Coefs c = biquad_coefs(n.desc.i[0], (double)b->sr, (double)computed_freq(at(f), at(d)), (double)at(g), (double)at(q));
          double* dst = &co[(size_t)i * per + (size_t)k * 5];
          dst[0] = c.b0;
          dst[1] = c.b1;
          dst[2] = c.b2;
          dst[3] = c.a1;
          dst[4] = c.a2;
        }
      }
      double* dco = nullptr;
      int e = dev_upload(b, &dco, co);
      if (e) return e;
      double* dst = nullptr;
      e = dev_alloc(b, &dst, (size_t)b->n_inst * STATE_STRIDE);
      if (e) return e;
      b->state_bufs.push_back({dst, (size_t)b->n_inst * STATE_STRIDE * sizeof(double)});
      o.i0 = varies ? 1 : 0;
      o.ptr0 = dco;
      o.ptr1 = dst;
      o.u0 = per;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_IIR_FILTER: {
      // iir_filter.rs:323-405.  N = len - 1 state variables, padded with zero coefficients to a kernel size.
      OpDesc o{};
      o.kind = OP_IIR;
      o.nch_in = o.nch_out = nch;
      const int len = (int)n.iir_b.size();
      const int ns = iir_padded_states(len - 1);
      if (ns < 0) return fail(WAA_ERR_DEVICE, "internal: IIR order");
      std::vector<double> co(2 * (size_t)(ns + 1), 0.);
      for (int k = 0; k < len; k++) {
        co[k] = n.iir_b[k];
        co[ns + 1 + k] = n.iir_a[k];
      }
      // zero-input state transition M: s_i' = -a_{i+1} s_0 + s_{i+1}; powers M^(32 * 2^k), k = 0..5, for the
      // lane scan of the kernel (double-double on the host, rounded once).  `growth` = largest entry of any power
      // the scan can form (intermediate squarings and all A^j, j <= 64): the scan's rounding error relative to
      // the state is about ns * growth * 2^-53, so ill-conditioned direct forms (clustered poles, high order)
      // and unstable filters go to the exact lane-per-stream kernel instead.
      // (double-double arithmetic, ~106 bits: repeated squaring of a matrix with large transient entries loses
      // growth^2 * eps per step, which long double cannot absorb for the filters that are still worth scanning)
      std::vector<DD> m((size_t)ns * ns), t((size_t)ns * ns);
      for (int i = 0; i < ns; i++) {
        m[(size_t)i * ns] = DD{-co[ns + 1 + i + 1], 0.};
        if (i + 1 < ns) m[(size_t)i * ns + i + 1] = dd_add(m[(size_t)i * ns + i + 1], DD{1., 0.});
      }
      double growth = 0.;
      auto note = [&](const std::vector<DD>& a) {
        for (const DD& v : a) growth = std::isfinite(v.hi) ? std::max(growth, std::fabs(v.hi)) : INFINITY;
      };
      auto mul = [&](const std::vector<DD>& x, const std::vector<DD>& y, std::vector<DD>& out) {
        for (int r = 0; r < ns; r++)
          for (int c = 0; c < ns; c++) {
            DD acc{0., 0.};
            for (int k = 0; k < ns; k++) acc = dd_add(acc, dd_mul(x[(size_t)r * ns + k], y[(size_t)k * ns + c]));
            out[(size_t)r * ns + c] = acc;
          }
      };
      for (int k = 0; k < 5; k++) {  // M^32
        mul(m, m, t);
        m.swap(t);
        note(m);
      }
      const std::vector<DD> A = m;
      std::vector<double> pw(6 * (size_t)ns * ns);
      for (int lvl = 0; lvl < 6; lvl++) {
        for (size_t k = 0; k < (size_t)ns * ns; k++) pw[lvl * (size_t)ns * ns + k] = m[k].hi + m[k].lo;
        mul(m, m, t);
        m.swap(t);
        note(m);
      }
      m = A;
      for (int j = 2; j <= 64 && std::isfinite(growth); j++) {  // every A^j a lane can see
        mul(m, A, t);
        m.swap(t);
        note(m);
      }
      const char* genv = getenv("WAA_IIR_GROWTH");  // experiments only
      const double growth_limit = genv ? atof(genv) : 1e4;
      const bool exact = !(growth <= growth_limit) || getenv("WAA_IIR_EXACT") != nullptr;  // env: debugging aid
      if (exact)
        for (auto& v : pw) v = 0.;  // unused
      double *dco = nullptr, *dpw = nullptr, *dst = nullptr;
      int e;
      if ((e = dev_upload(b, &dco, co)) || (e = dev_upload(b, &dpw, pw))) return e;
      const size_t n_state = (size_t)b->n_inst * nch * ns;
      if ((e = dev_alloc(b, &dst, n_state))) return e;
      b->state_bufs.push_back({dst, n_state * sizeof(double)});
      o.i0 = exact ? -ns : ns;
      o.ptr0 = dco;
      o.ptr1 = dst;
      o.ptr2 = dpw;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_WAVESHAPER: {
      if (n.has_curve) {
        OpDesc o{};
        o.kind = OP_WAVESHAPER;
        o.nch_in = o.nch_out = nch;
        if (!n.d_curve) {
          int e = dev_upload(b, &n.d_curve, n.curve);
          if (e) return e;
        }
        o.ptr0 = n.d_curve;
        o.i0 = (int)n.curve.size();
        ops.push_back(o);
      }
      break;
    }
    case WAA_NODE_STEREO_PANNER: {
      OpDesc o{};
      o.kind = OP_STEREO_PAN;
      o.nch_in = nch;
      o.nch_out = 2;
      const ParamStore& p = n.params[0];
      int e = node_param(b, id, 0, &o.p0);
      if (e) return e;
      if (param_mode(n, 0) != 2) {
        // gains on the host with the same libm sinf the reference's f32::sin resolves to (stereo_panner.rs:74-79)
        const uint32_t cnt = p.mode() == 1 ? b->n_quanta : 1;
        std::vector<float> gl((size_t)b->n_inst * cnt), gr((size_t)b->n_inst * cnt);
        for (uint32_t i = 0; i < b->n_inst; i++) {
          auto pv = param_per_quantum(b, p, i, nullptr);
          for (uint32_t k = 0; k < cnt; k++) {
            float pan = pv[pv.size() == 1 ? 0 : k];
            float x = nch == 1 ? (pan + 1.f) * 0.5f : (pan <= 0.f ? pan + 1.f : pan);
            gl[(size_t)i * cnt + k] = sinf((1.f - x) * PI_F / 2.f);
            gr[(size_t)i * cnt + k] = sinf(x * PI_F / 2.f);
          }
        }
        if ((e = upload_values(b, gl, p.mode(), &o.p1))) return e;
        if ((e = upload_values(b, gr, p.mode(), &o.p2))) return e;
      }
      ops.push_back(o);
      break;
    }
    case WAA_NODE_PANNER: {
      OpDesc o{};
      o.kind = OP_PANNER;
      o.nch_in = nch;
      o.nch_out = 2;
      int mode = 0;
      for (auto& p : n.params) mode = std::max(mode, p.mode());
      for (int k = 6; k < 15; k++)
        if (n.params[k].mode() == 2)
          return fail(WAA_ERR_OUT_OF_SCOPE, "a-rate AudioListener automation is out of scope for this round");
      // listener single-valued => the reference evaluates the geometry once per quantum from the first value of
      // every param (panner.rs:833-846)
      const uint32_t cnt = mode == 0 ? 1 : b->n_quanta;
      const int vmode = mode == 0 ? 0 : 1;
      std::vector<float> az((size_t)b->n_inst * cnt), gl(az.size()), gr(az.size()), dg(az.size()), cg(az.size());
      for (uint32_t i = 0; i < b->n_inst; i++) {
        std::vector<std::vector<float>> pv(15);
        for (int k = 0; k < 15; k++) pv[k] = param_per_quantum(b, n.params[k], i, nullptr);
        for (uint32_t k = 0; k < cnt; k++) {
          auto at = [&](int p) { return pv[p][pv[p].size() == 1 ? 0 : k]; };
          V3 sp{at(0), at(1), at(2)}, so{at(3), at(4), at(5)}, lp{at(6), at(7), at(8)}, lf{at(9), at(10), at(11)},
              lu{at(12), at(13), at(14)};
          float a, el;
          azimuth_elevation(sp, lp, lf, lu, &a, &el);
          // panner.rs:996-1004
          a = a < -180.f ? -180.f : a > 180.f ? 180.f : a;
          if (a < -90.f)
            a = -180.f - a;
          else if (a > 90.f)
            a = 180.f - a;
          float x = nch == 1 ? (a + 90.f) / 180.f : (a <= 0.f ? (a + 90.f) / 90.f : a / 90.f);
          const size_t ix = (size_t)i * cnt + k;
          az[ix] = a;
          gl[ix] = cosf(x * PI_F / 2.f);
          gr[ix] = sinf(x * PI_F / 2.f);
          dg[ix] = dist_gain(n.desc, sp, lp);
          cg[ix] = cone_gain(n.desc, sp, so, lp);
        }
      }
      int e;
      if ((e = upload_values(b, az, vmode, &o.p0)) || (e = upload_values(b, gl, vmode, &o.p1)) ||
          (e = upload_values(b, gr, vmode, &o.p2)) || (e = upload_values(b, dg, vmode, &o.p3)) ||
          (e = upload_values(b, cg, vmode, &o.p4)))
        return e;
      ops.push_back(o);
      break;
    }
    case WAA_NODE_CONVOLVER:
      // no buffer set: passthrough (convolver.rs:368-374)
      break;
    case WAA_NODE_ANALYSER:
    case WAA_NODE_DESTINATION:
    default:
      break;
  }
  return 0;
}

void default_config(Node& n, uint32_t n_out) {
  int cc = 2, mode = WAA_COUNT_MODE_MAX, interp = WAA_INTERP_SPEAKERS;
  switch (n.desc.kind) {
    case WAA_NODE_DESTINATION:
      cc = (int)n_out;
      mode = WAA_COUNT_MODE_EXPLICIT;
      break;
    case WAA_NODE_CONVOLVER:
    case WAA_NODE_STEREO_PANNER:
    case WAA_NODE_PANNER:
      mode = WAA_COUNT_MODE_CLAMPED_MAX;
      break;
    default: break;
  }
  if (n.desc.channel_count != 0) {
    cc = (int)n.desc.channel_count;
    mode = (int)n.desc.channel_count_mode;
    interp = (int)n.desc.channel_interpretation;
  }
  n.cc = cc;
  n.mode = mode;
  n.interp = interp;
}

}  // namespace

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

const char* waa_last_error(void) { return g_err; }

int32_t waa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

waa_status waa_batch_create(const waa_graph_desc* g, uint32_t n_inst, uint32_t n_out, uint64_t length, float sr,
                            int32_t device, waa_batch** out) {
  if (!g || !out || g->n_nodes == 0 || n_inst == 0) return fail(WAA_ERR_INVALID_ARGUMENT, "invalid arguments");
  if (g->nodes[0].kind != WAA_NODE_DESTINATION) return fail(WAA_ERR_INVALID_ARGUMENT, "node 0 must be the destination");
  if (n_out == 0 || n_out > WAA_MAX_CHANNELS)
    return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - Invalid number of channels: %u", n_out);
  if (!(sr >= 8000.f && sr <= 192000.f)) return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - Invalid sample rate: %f", sr);
  std::unique_ptr<waa_batch> b(new waa_batch);
  b->n_inst = n_inst;
  b->n_out = n_out;
  b->length = length;
  b->sr = sr;
  b->n_quanta = (uint32_t)((length + RQ - 1) / RQ);
  if (b->n_quanta == 0) b->n_quanta = 1;
  b->n_tiles = (b->n_quanta + QUANTA_PER_TILE - 1) / QUANTA_PER_TILE;
  b->lp = (uint64_t)b->n_tiles * TILE;
  for (uint32_t e = 0; e < g->n_edges; e++) {
    const waa_edge_desc& ed = g->edges[e];
    if (ed.from >= g->n_nodes || ed.to >= g->n_nodes || ed.from_output != 0 ||
        (ed.to_input != 0 && !(ed.to_input & 0x80000000u)))
      return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - invalid edge %u", e);
    b->edges.push_back(ed);
  }
  b->nodes.resize(g->n_nodes);
  for (uint32_t i = 0; i < g->n_nodes; i++) {
    Node& n = b->nodes[i];
    n.desc = g->nodes[i];
    if (n.desc.kind >= WAA_NODE_KIND_COUNT) return fail(WAA_ERR_INVALID_ARGUMENT, "unknown node kind");
    default_config(n, n_out);
    if (n.cc < 1 || n.cc > WAA_MAX_CHANNELS)
      return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - Invalid number of channels: %d", n.cc);
    auto P = [&](size_t k) -> ParamStore& {
      if (n.params.size() <= k) n.params.resize(k + 1);
      return n.params[k];
    };
    switch (n.desc.kind) {
      case WAA_NODE_BIQUAD:
        P(WAA_PARAM_BIQUAD_FREQUENCY).init(n_inst, 350.f, 0.f, sr / 2.f);
        P(WAA_PARAM_BIQUAD_DETUNE).init(n_inst, 0.f, -153600.f, 153600.f);
        P(WAA_PARAM_BIQUAD_Q).init(n_inst, 1.f, -FLT_MAX, FLT_MAX);
        P(WAA_PARAM_BIQUAD_GAIN).init(n_inst, 0.f, -FLT_MAX, 40.f * log10f(FLT_MAX));
        if (n.desc.i[0] < 0 || n.desc.i[0] > 7) return fail(WAA_ERR_INVALID_ARGUMENT, "bad biquad type");
        break;
      case WAA_NODE_GAIN: P(0).init(n_inst, 1.f, -FLT_MAX, FLT_MAX); break;
      case WAA_NODE_BUFFER_SOURCE:
        P(WAA_PARAM_SOURCE_PLAYBACK_RATE).init(n_inst, 1.f, -FLT_MAX, FLT_MAX);
        P(WAA_PARAM_SOURCE_DETUNE).init(n_inst, 0.f, -FLT_MAX, FLT_MAX);
        n.bufs.resize(n_inst);
        n.sched.resize(n_inst);
        break;
      case WAA_NODE_CONSTANT_SOURCE:
        P(0).init(n_inst, 1.f, -FLT_MAX, FLT_MAX);
        n.sched.resize(n_inst);
        break;
      case WAA_NODE_OSCILLATOR:  // oscillator.rs:210-262
        if (n.desc.i[0] < WAA_OSC_SINE || n.desc.i[0] > WAA_OSC_CUSTOM) return fail(WAA_ERR_INVALID_ARGUMENT, "bad oscillator type");
        P(WAA_PARAM_OSCILLATOR_FREQUENCY).init(n_inst, 440.f, -sr / 2.f, sr / 2.f);
        P(WAA_PARAM_OSCILLATOR_DETUNE).init(n_inst, 0.f, -153600.f, 153600.f);
        n.sched.resize(n_inst);
        break;
      case WAA_NODE_STEREO_PANNER:
        P(0).init(n_inst, 0.f, -1.f, 1.f);
        if (n.mode == WAA_COUNT_MODE_MAX)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - StereoPannerNode channel count mode cannot be set to max");
        if (n.cc > 2)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - StereoPannerNode channel count cannot be greater than two");
        break;
      case WAA_NODE_PANNER: {
        static const float defs[15] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, -1, 0, 1, 0};
        for (int p = 0; p < 15; p++) P(p).init(n_inst, defs[p], -FLT_MAX, FLT_MAX);
        if (n.desc.i[0] == WAA_PANNING_HRTF)
          return fail(WAA_ERR_OUT_OF_SCOPE, "HRTF panning is out of scope (third-party hrtf crate, parity unpinned)");
        if (n.mode == WAA_COUNT_MODE_MAX)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - PannerNode channel count mode cannot be set to max");
        if (n.cc > 2)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - PannerNode channel count cannot be greater than two");
        break;
      }
      case WAA_NODE_DELAY:  // delay.rs:283-335
        if (n.desc.d[0] == 0.) n.desc.d[0] = 1.;
        if (!(n.desc.d[0] > 0. && n.desc.d[0] < 180.))
          return fail(WAA_ERR_NOT_SUPPORTED,
                      "NotSupportedError - maxDelayTime MUST be greater than zero and less than three minutes");
        P(WAA_PARAM_DELAY_DELAY_TIME).init(n_inst, 0.f, 0.f, (float)n.desc.d[0]);
        break;
      case WAA_NODE_WAVESHAPER:
        if (n.desc.i[0] != WAA_OVERSAMPLE_NONE)
          return fail(WAA_ERR_OUT_OF_SCOPE, "WaveShaper oversampling is out of scope (third-party rubato, parity unpinned)");
        break;
      case WAA_NODE_CONVOLVER:
        if (n.cc > 2)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - ConvolverNode channel count cannot be greater than two");
        if (n.mode == WAA_COUNT_MODE_MAX)
          return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - ConvolverNode channel count mode cannot be set to max");
        break;
      case WAA_NODE_ANALYSER: {
        int fs = n.desc.i[0] ? n.desc.i[0] : 2048;
        if (fs < 32 || fs > 32768 || (fs & (fs - 1)))
          return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - Invalid fft size: %d is not a power of two", fs);
        n.desc.i[0] = fs;
        if (n.desc.d[0] == 0. && n.desc.d[1] == 0. && n.desc.d[2] == 0.) {
          n.desc.d[0] = 0.8;
          n.desc.d[1] = -100.;
          n.desc.d[2] = -30.;
        }
        if (n.desc.d[0] < 0. || n.desc.d[0] > 1.)
          return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - Invalid smoothing time constant");
        if (!(n.desc.d[1] < n.desc.d[2])) return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - Invalid min decibels");
        break;
      }
      default: break;
    }
  }
  if (device == WAA_DEVICE_PLAN_ONLY) {
    b->dry = true;
    b->device = -1;
    *out = b.release();
    return WAA_OK;
  }
  // the device is only touched from here on; a machine without a GPU still validates graphs above
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(WAA_ERR_DEVICE, "no HIP device available: libwaa_hip has no CPU fallback");
  if (device >= 0) {
    if (device >= ndev) return fail(WAA_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    b->device = device;
  } else {
    HIP_TRY(hipGetDevice(&b->device));
  }
  HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  *out = b.release();
  return WAA_OK;
}

void waa_batch_destroy(waa_batch* b) {
  if (!b) return;
  if (b->dry) {
    for (void* p : b->allocs) std::free(p);
    for (void* p : b->payload_allocs) std::free(p);
    delete b;
    return;
  }
  if (b->stream) {
    (void)hipStreamSynchronize(b->stream);
    for (auto& p : b->prof)
      for (auto& ev : p.pending) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
    for (void* p : b->allocs) (void)hipFree(p);
    for (void* p : b->payload_allocs) (void)hipFree(p);
    (void)hipStreamDestroy(b->stream);
  }
  delete b;
}

static int upload_buffer(waa_batch* b, const float* const* channels, uint32_t n_ch, uint64_t frames, float sr,
                         DeviceBuffer* out) {
  const uint64_t stride = (frames + 3) / 4 * 4;
  float* d = nullptr;
  int e = dev_alloc(b, &d, (size_t)n_ch * std::max<uint64_t>(stride, 4), true);
  if (e) return e;
  for (uint32_t c = 0; c < n_ch; c++)
    if (frames) {
      if (b->dry)
        std::memcpy(d + (size_t)c * stride, channels[c], frames * sizeof(float));
      else
        HIP_TRY(hipMemcpy(d + (size_t)c * stride, channels[c], frames * sizeof(float), hipMemcpyHostToDevice));
    }
  out->base = d;
  out->ch_stride = stride;
  out->frames = frames;
  out->nch = n_ch;
  out->sr = sr;
  out->valid = true;
  return 0;
}

waa_status waa_source_set_buffer(waa_batch* b, uint32_t node, uint32_t inst, const float* const* channels,
                                 uint32_t n_ch, uint64_t frames, float sr) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_BUFFER_SOURCE)) || (e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  if (n_ch == 0 || n_ch > WAA_MAX_CHANNELS) return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - Invalid number of channels");
  if (!b->dry) HIP_TRY(hipSetDevice(b->device));
  DeviceBuffer db;
  if ((e = upload_buffer(b, channels, n_ch, frames, sr, &db))) return e;
  Node& n = b->nodes[node];
  uint32_t lo = inst == WAA_ALL_INSTANCES ? 0 : inst, hi = inst == WAA_ALL_INSTANCES ? b->n_inst : inst + 1;
  for (uint32_t k = lo; k < hi; k++) n.bufs[k] = db;
  return WAA_OK;
}

waa_status waa_source_set_buffer_batch(waa_batch* b, uint32_t node, const float* data, uint32_t n_ch, uint64_t frames,
                                       float sr) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_BUFFER_SOURCE)) || (e = check_unplanned(b))) return e;
  if (n_ch == 0 || n_ch > WAA_MAX_CHANNELS) return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - Invalid number of channels");
  if (!b->dry) HIP_TRY(hipSetDevice(b->device));
  const uint64_t stride = (frames + 3) / 4 * 4;
  float* d = nullptr;
  if ((e = dev_alloc(b, &d, (size_t)b->n_inst * n_ch * std::max<uint64_t>(stride, 4), true))) return e;
  if (frames && !b->dry)
    HIP_TRY(hipMemcpy2D(d, stride * sizeof(float), data, frames * sizeof(float), frames * sizeof(float),
                        (size_t)b->n_inst * n_ch, hipMemcpyHostToDevice));
  Node& n = b->nodes[node];
  for (uint32_t k = 0; k < b->n_inst; k++) {
    DeviceBuffer db;
    db.base = d + (size_t)k * n_ch * stride;
    db.ch_stride = stride;
    db.frames = frames;
    db.nch = n_ch;
    db.sr = sr;
    db.valid = true;
    n.bufs[k] = db;
  }
  return WAA_OK;
}

waa_status waa_source_adopt_device(waa_batch* b, uint32_t node, const float* device_data, uint32_t n_ch, uint64_t frames,
                                   float sr) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_BUFFER_SOURCE)) || (e = check_unplanned(b))) return e;
  if (!device_data || n_ch == 0 || n_ch > WAA_MAX_CHANNELS) return fail(WAA_ERR_INVALID_ARGUMENT, "bad device buffer");
  Node& n = b->nodes[node];
  for (uint32_t k = 0; k < b->n_inst; k++) {
    DeviceBuffer db;
    db.base = const_cast<float*>(device_data) + (size_t)k * n_ch * frames;
    db.ch_stride = frames;
    db.frames = frames;
    db.nch = n_ch;
    db.sr = sr;
    db.valid = true;
    n.bufs[k] = db;
  }
  return WAA_OK;
}

waa_status waa_source_start(waa_batch* b, uint32_t node, uint32_t inst, double when, double offset, double duration) {
  int e;
  if (!b || node >= b->nodes.size()) return fail(WAA_ERR_INVALID_ARGUMENT, "bad node");
  const uint32_t kind = b->nodes[node].desc.kind;
  if (kind != WAA_NODE_BUFFER_SOURCE && kind != WAA_NODE_CONSTANT_SOURCE && kind != WAA_NODE_OSCILLATOR)
    return fail(WAA_ERR_INVALID_ARGUMENT, "node %u is not a scheduled source", node);
  if ((e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  if (!std::isfinite(when) || !std::isfinite(offset) || !std::isfinite(duration))
    return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided time value is non-finite.");
  if (when < 0. || offset < 0. || duration < 0.)
    return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - The provided time value cannot be negative");
  Node& n = b->nodes[node];
  uint32_t lo = inst == WAA_ALL_INSTANCES ? 0 : inst, hi = inst == WAA_ALL_INSTANCES ? b->n_inst : inst + 1;
  for (uint32_t k = lo; k < hi; k++) {
    if (n.sched[k].start != DBL_MAX) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - Cannot call `start` twice");
    n.sched[k].start = when;
    if (kind == WAA_NODE_BUFFER_SOURCE) {
      n.sched[k].offset = offset;
      n.sched[k].duration = duration;
    }
  }
  return WAA_OK;
}

waa_status waa_source_stop(waa_batch* b, uint32_t node, uint32_t inst, double when) {
  int e;
  if (!b || node >= b->nodes.size()) return fail(WAA_ERR_INVALID_ARGUMENT, "bad node");
  const uint32_t kind = b->nodes[node].desc.kind;
  if (kind != WAA_NODE_BUFFER_SOURCE && kind != WAA_NODE_CONSTANT_SOURCE && kind != WAA_NODE_OSCILLATOR)
    return fail(WAA_ERR_INVALID_ARGUMENT, "node %u is not a scheduled source", node);
  if ((e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  if (!std::isfinite(when)) return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided time value is non-finite.");
  if (when < 0.) return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - The provided time value cannot be negative");
  Node& n = b->nodes[node];
  uint32_t lo = inst == WAA_ALL_INSTANCES ? 0 : inst, hi = inst == WAA_ALL_INSTANCES ? b->n_inst : inst + 1;
  for (uint32_t k = lo; k < hi; k++) {
    if (n.sched[k].start == DBL_MAX) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - Cannot stop before start");
    n.sched[k].stop = when;
  }
  return WAA_OK;
}

waa_status waa_source_set_loop(waa_batch* b, uint32_t node, uint32_t inst, int32_t looping, double ls, double le) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_BUFFER_SOURCE)) || (e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  Node& n = b->nodes[node];
  uint32_t lo = inst == WAA_ALL_INSTANCES ? 0 : inst, hi = inst == WAA_ALL_INSTANCES ? b->n_inst : inst + 1;
  for (uint32_t k = lo; k < hi; k++) {
    n.sched[k].looping = looping;
    n.sched[k].loop_start = ls;
    n.sched[k].loop_end = le;
  }
  return WAA_OK;
}

// convolver.rs:16-53
static float normalize_buffer(const float* const* ch, uint32_t n_ch, uint64_t len, float sr) {
  const float gain_calibration = 0.00125f, gain_calibration_sample_rate = 44100.f, min_power = 0.000125f;
  float power = 0.f;
  for (uint32_t c = 0; c < n_ch; c++) {
    float s = 0.f;
    for (uint64_t i = 0; i < len; i++) s += ch[c][i] * ch[c][i];
    power += s;
  }
  power = std::sqrt(power / (float)(n_ch * len));
  if (!std::isfinite(power) || std::isnan(power) || power < min_power) power = min_power;
  float scale = 1.f / power;
  scale *= gain_calibration;
  scale *= gain_calibration_sample_rate / sr;
  if (n_ch == 4) scale *= 0.5f;
  return scale;
}

waa_status waa_convolver_set_buffer(waa_batch* b, uint32_t node, const float* const* channels, uint32_t n_ch,
                                    uint64_t frames, float sr) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_CONVOLVER)) || (e = check_unplanned(b))) return e;
  if (sr != b->sr)
    return fail(WAA_ERR_NOT_SUPPORTED,
                "NotSupportedError - sample rate of the convolution buffer must match the audio context");
  if (!(n_ch == 1 || n_ch == 2 || n_ch == 4))
    return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - the convolution buffer must consist of 1, 2 or 4 channels");
  Node& n = b->nodes[node];
  const float scale = n.desc.i[0] ? 1.f : normalize_buffer(channels, n_ch, frames, sr);
  n.ir.assign(n_ch, std::vector<float>(frames));
  for (uint32_t c = 0; c < n_ch; c++)
    for (uint64_t i = 0; i < frames; i++) n.ir[c][i] = channels[c][i] * scale;
  n.ir_len = frames;
  n.ir_nch = (int)n_ch;
  n.has_ir = true;
  return WAA_OK;
}

waa_status waa_waveshaper_set_curve(waa_batch* b, uint32_t node, const float* curve, uint32_t nn) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_WAVESHAPER)) || (e = check_unplanned(b))) return e;
  Node& n = b->nodes[node];
  if (n.has_curve) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - cannot assign curve twice");
  n.curve.assign(curve, curve + nn);
  n.has_curve = true;
  return WAA_OK;
}

// periodic_wave.rs:88-190 + oscillator.rs:318-321 (control side: the wavetable is generated on the host)
waa_status waa_oscillator_set_periodic_wave(waa_batch* b, uint32_t node, const float* real, const float* imag, uint32_t nn,
                                            int32_t disable_normalization) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_OSCILLATOR)) || (e = check_unplanned(b))) return e;
  if ((!real && !imag) || nn < 2) return fail(WAA_ERR_INVALID_ARGUMENT, "IndexSizeError - `real` and `imag` length should at least 2");
  const int size = 8192;
  std::vector<float> wavetable(size);
  const float pi_2 = 2.f * 3.14159265358979323846f;
  for (int i = 0; i < size; i++) {
    float sample = 0.f;
    const float phase = pi_2 * (float)i / (float)size;
    for (uint32_t j = 1; j < nn; j++) {
      const float freq = (float)j;
      const float re = real ? real[j] : 0.f, im = imag ? imag[j] : 0.f;
      const float rad = phase * freq;
      const float contrib = re * std::cos(rad) + im * std::sin(rad);
      sample += contrib;
    }
    wavetable[i] = sample;
  }
  if (!disable_normalization) {
    float max = 0.f;
    for (float v : wavetable) max = std::fabs(v) > max ? std::fabs(v) : max;
    if (max > 0.f) {
      const float norm_factor = 1.f / max;
      for (float& v : wavetable) v *= norm_factor;
    }
  }
  b->nodes[node].osc_wave.swap(wavetable);
  return WAA_OK;
}

// iir_filter.rs:17-46 (validation) and :273-311 (pad to equal length, normalise by a0)
static int check_iir_coefs(const double* ff, uint32_t nff, const double* fb, uint32_t nfb) {
  if (!ff || nff == 0 || nff > WAA_MAX_IIR_COEFFS)
    return fail(WAA_ERR_NOT_SUPPORTED,
                "NotSupportedError - IIR Filter feedforward coefficients should have length >= 0 and <= %d", WAA_MAX_IIR_COEFFS);
  bool all_zero = true;
  for (uint32_t i = 0; i < nff; i++) all_zero &= ff[i] == 0.;
  if (all_zero) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - IIR Filter feedforward coefficients cannot be all zeros");
  if (!fb || nfb == 0 || nfb > WAA_MAX_IIR_COEFFS)
    return fail(WAA_ERR_NOT_SUPPORTED,
                "NotSupportedError - IIR Filter feedback coefficients should have length >= 0 and <= %d", WAA_MAX_IIR_COEFFS);
  if (fb[0] == 0.) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - IIR Filter feedback first coefficient cannot be zero");
  return 0;
}

waa_status waa_iir_set_coefficients(waa_batch* b, uint32_t node, const double* ff, uint32_t nff, const double* fb,
                                    uint32_t nfb) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_IIR_FILTER)) || (e = check_unplanned(b))) return e;
  if ((e = check_iir_coefs(ff, nff, fb, nfb))) return e;
  Node& n = b->nodes[node];
  const uint32_t len = std::max(nff, nfb);
  const double a0 = fb[0];
  n.iir_b.assign(len, 0.);
  n.iir_a.assign(len, 0.);
  for (uint32_t i = 0; i < len; i++) {
    n.iir_b[i] = (i < nff ? ff[i] : 0.) / a0;
    n.iir_a[i] = (i < nfb ? fb[i] : 0.) / a0;
  }
  return WAA_OK;
}

waa_status waa_set_param_const(waa_batch* b, uint32_t node, uint32_t param, uint32_t inst, float value) {
  int e;
  if (!b || node >= b->nodes.size() || param >= b->nodes[node].params.size())
    return fail(WAA_ERR_INVALID_ARGUMENT, "no such param %u on node %u", param, node);
  if ((e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  ParamStore& p = b->nodes[node].params[param];
  uint32_t lo = inst == WAA_ALL_INSTANCES ? 0 : inst, hi = inst == WAA_ALL_INSTANCES ? b->n_inst : inst + 1;
  for (uint32_t k = lo; k < hi; k++) p.cst[k] = value;
  return WAA_OK;
}

waa_status waa_set_param_block(waa_batch* b, uint32_t node, uint32_t param, uint32_t inst, uint64_t q0, uint32_t nq,
                               uint32_t vpq, const float* values) {
  int e;
  if (!b || node >= b->nodes.size() || param >= b->nodes[node].params.size())
    return fail(WAA_ERR_INVALID_ARGUMENT, "no such param %u on node %u", param, node);
  if ((e = check_inst(b, inst)) || (e = check_unplanned(b))) return e;
  if (vpq != 1 && vpq != RQ) return fail(WAA_ERR_INVALID_ARGUMENT, "values_per_quantum must be 1 or 128");
  ParamBlock blk;
  blk.inst = inst;
  blk.q0 = q0;
  blk.nq = nq;
  blk.vpq = vpq;
  blk.v.assign(values, values + (size_t)nq * vpq);
  b->nodes[node].params[param].blocks.push_back(std::move(blk));
  return WAA_OK;
}

waa_status waa_plan_describe(waa_batch* b, char* buf, size_t cap, size_t* needed) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (!b->planned) {
    if (!b->dry) HIP_TRY(hipSetDevice(b->device));
    int e = build_plan(b);
    if (e) return e;
  }
  std::string text;
  char head[256];
  snprintf(head, sizeof head, "batch: %u instance(s) x %llu frames (%u quanta, %u tiles of %d) @ %g Hz, %u output channel(s)\n",
           b->n_inst, (unsigned long long)b->length, b->n_quanta, b->n_tiles, TILE, (double)b->sr, b->n_out);
  text += head;
  for (auto& l : b->plan_log) text += l + "\n";
  if (needed) *needed = text.size();
  if (buf && cap) {
    const size_t n = std::min(cap - 1, text.size());
    std::memcpy(buf, text.data(), n);
    buf[n] = 0;
  }
  return WAA_OK;
}

waa_status waa_render(waa_batch* b) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (b->dry) return fail(WAA_ERR_DEVICE, "plan-only batch (WAA_DEVICE_PLAN_ONLY) cannot render: there is no CPU fallback");
  HIP_TRY(hipSetDevice(b->device));
  if (!b->planned) {
    int e = build_plan(b);
    if (e) return e;
  }
  // every render starts from the initial state (offline contexts render exactly once; re-rendering the
  // same batch is what the benchmark loop does)
  for (auto& sb : b->state_bufs) HIP_TRY(hipMemsetAsync(sb.first, 0, sb.second, b->stream));
  for (auto& n : b->nodes) n.an_cache.clear();
  b->rendered = true;
  auto timed = [&](int slot, auto&& launch) -> int {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (b->profiling && slot >= 0) {
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, b->stream));
    }
    launch();
    HIP_TRY(hipGetLastError());
    if (b->profiling && slot >= 0) {
      HIP_TRY(hipEventRecord(e1, b->stream));
      b->prof[slot].pending.push_back({e0, e1});
    }
    return 0;
  };
  // one step over the tile range [t0, t1)
  auto run_step = [&](const Step& st, uint32_t t0, uint32_t t1) -> int {
    int e = 0;
    switch (st.kind) {
      case 1: {
        BiquadStreamDesc d = st.bq;
        d.tile0 = t0;
        d.tile1 = t1;
        e = timed(st.profile_slot, [&] { launch_biquad_stream(d, b->stream); });
        break;
      }
      case 2:
        if ((e = timed(st.slot_fwd, [&] { launch_conv_forward(st.conv, b->stream); }))) break;
        if ((e = timed(st.slot_mac, [&] { launch_conv_mac(st.conv, b->stream); }))) break;
        e = timed(st.slot_inv, [&] { launch_conv_inverse(st.conv, b->stream); });
        break;
      case 3: HIP_TRY(hipMemsetAsync(st.zero_ptr, 0, st.zero_bytes, b->stream)); break;
      case 4: e = timed(st.slot_mac, [&] { launch_conv_direct(st.conv, b->stream); }); break;
      case 5: e = timed(st.profile_slot, [&] { launch_biquad_coefs(st.coef, b->stream); }); break;
      case 6: {
        IirStreamDesc d = st.iir;
        d.tile0 = t0;
        d.tile1 = t1;
        e = timed(st.profile_slot, [&] { launch_iir_stream(d, b->stream); });
        break;
      }
      case 7: {
        DelayDesc d = st.delay;
        d.tile0 = t0;
        d.tile1 = t1;
        e = timed(st.profile_slot, [&] { launch_delay(d, b->stream); });
        break;
      }
      case 8: e = timed(st.profile_slot, [&] { launch_loop(st.loop, b->stream); }); break;
      case 9: e = timed(st.profile_slot, [&] { launch_osc(st.osc, b->stream); }); break;
      default: {
        ChainDesc d = st.chain;
        d.tile0 = t0;
        d.tile1 = t1;
        e = timed(st.profile_slot, [&] { launch_chain(d, st.cmax, b->stream); });
        break;
      }
    }
    return e;
  };
  for (size_t i = 0; i < b->steps.size();) {
    const Step& st = b->steps[i];
    if (st.group < 0) {
      int e = run_step(st, 0, b->n_tiles);
      if (e) return e;
      i++;
      continue;
    }
    // block-scheduled feedback loop: steps [i, j) block by block (graph.rs cycle breaker, see build_plan)
    size_t j = i;
    while (j < b->steps.size() && b->steps[j].group == st.group) j++;
    for (size_t k = i; k < j; k++)
      if (b->steps[k].prologue) {
        int e = run_step(b->steps[k], 0, b->n_tiles);
        if (e) return e;
      }
    const uint32_t bt = b->group_tiles[st.group];
    for (uint32_t t0 = 0; t0 < b->n_tiles; t0 += bt) {
      const uint32_t t1 = std::min(b->n_tiles, t0 + bt);
      for (size_t k = i; k < j; k++)
        if (!b->steps[k].prologue) {
          int e = run_step(b->steps[k], t0, t1);
          if (e) return e;
        }
    }
    i = j;
  }
  return WAA_OK;
}

static int drain_profile(waa_batch* b) {
  for (auto& p : b->prof) {
    for (auto& ev : p.pending) {
      float ms = 0.f;
      HIP_TRY(hipEventSynchronize(ev.second));
      HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
      p.total_ms += (double)ms;
      p.launches++;
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    p.pending.clear();
  }
  return 0;
}

waa_status waa_sync(waa_batch* b) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (b->dry) return fail(WAA_ERR_DEVICE, "plan-only batch has no device");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return drain_profile(b);
}

waa_status waa_download(waa_batch* b, uint32_t inst, uint32_t ch, float* dst, uint64_t frames) {
  if (!b || inst >= b->n_inst || ch >= b->n_out || frames > b->length)
    return fail(WAA_ERR_INVALID_ARGUMENT, "download out of range");
  if (!b->planned || !b->rendered) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - nothing rendered yet");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const SignalRef& s = b->nodes[0].sig;
  if ((int)ch < s.nch) {
    HIP_TRY(hipMemcpy(dst, s.base + (size_t)inst * s.inst_stride + (size_t)ch * s.ch_stride, frames * sizeof(float),
                      hipMemcpyDeviceToHost));
  } else {
    std::memset(dst, 0, frames * sizeof(float));
  }
  return WAA_OK;
}

waa_status waa_download_all(waa_batch* b, float* dst) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  if (!b->planned || !b->rendered) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - nothing rendered yet");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const SignalRef& s = b->nodes[0].sig;
  if (b->length == 0) return WAA_OK;
  if ((uint32_t)s.nch == b->n_out) {
    HIP_TRY(hipMemcpy2D(dst, b->length * sizeof(float), s.base, s.ch_stride * sizeof(float), b->length * sizeof(float),
                        (size_t)b->n_inst * b->n_out, hipMemcpyDeviceToHost));
  } else {
    for (uint32_t i = 0; i < b->n_inst; i++)
      for (uint32_t c = 0; c < b->n_out; c++) {
        int e = waa_download(b, i, c, dst + ((size_t)i * b->n_out + c) * b->length, b->length);
        if (e) return e;
      }
  }
  return WAA_OK;
}

waa_status waa_output_device(waa_batch* b, const float** p, uint64_t* is, uint64_t* cs) {
  if (!b || !b->planned) return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - nothing rendered yet");
  const SignalRef& s = b->nodes[0].sig;
  *p = s.base;
  *is = s.inst_stride;
  *cs = s.ch_stride;
  return WAA_OK;
}

// AnalyserNode pulls (analysis.rs:261-401).  current_time after an offline render never changes, so the
// spectrum is computed once per (node, instance) and repeated pulls return the same data (analysis.rs:354-357).
static int analyser_compute(waa_batch* b, uint32_t node, uint32_t inst, Node::AnCache** out) {
  int e;
  if ((e = check_node(b, node, WAA_NODE_ANALYSER))) return e;
  if (inst >= b->n_inst) return fail(WAA_ERR_INVALID_ARGUMENT, "instance out of range");
  Node& n = b->nodes[node];
  const int N = n.desc.i[0], M = N / 2;
  auto it = n.an_cache.find(inst);
  if (it != n.an_cache.end()) {
    *out = &it->second;
    return 0;
  }
  Node::AnCache cache;
  cache.spec.assign(M, 0.f);
  cache.time.assign(N, 0.f);
  if (b->planned && b->rendered && n.live) {
    HIP_TRY(hipSetDevice(b->device));
    if (!n.d_window) {
      // generate_blackman (analysis.rs:14-24), f32 with the host libm the reference's f32::cos resolves to
      std::vector<float> win(N);
      const float alpha = 0.16f, a0 = (1.f - alpha) / 2.f, a1 = 1.f / 2.f, a2 = alpha / 2.f;
      for (int i = 0; i < N; i++)
        win[i] = a0 - a1 * cosf(2.f * PI_F * (float)i / (float)N) + a2 * cosf(4.f * PI_F * (float)i / (float)N);
      std::vector<Cplx> tw(M), twf(M);
      for (int t = 0; t < M; t++) {
        const double x = -2.0 * 3.14159265358979323846 * (double)t / (double)M;
        const double y = -2.0 * 3.14159265358979323846 * (double)t / (double)N;
        tw[t] = Cplx{(float)std::cos(x), (float)std::sin(x)};
        twf[t] = Cplx{(float)std::cos(y), (float)std::sin(y)};
      }
      std::vector<float> zeros(M, 0.f);
      if ((e = dev_upload(b, &n.d_window, win)) || (e = dev_upload(b, &n.d_an_tw, tw)) ||
          (e = dev_upload(b, &n.d_an_twfull, twf)) || (e = dev_upload(b, &n.d_an_prev, zeros)) ||
          (e = dev_alloc(b, &n.d_an_spec, (size_t)M)) || (e = dev_alloc(b, &n.d_an_time, (size_t)N)))
        return e;
    }
    AnalyserDesc ad{};
    ad.sig = n.sig;
    ad.inst = inst;
    ad.fft_size = N;
    ad.frames_written = (uint64_t)b->n_quanta * RQ;
    ad.smoothing = (float)n.desc.d[0];
    ad.window = n.d_window;
    ad.tw = n.d_an_tw;
    ad.tw_full = n.d_an_twfull;
    ad.prev = n.d_an_prev;
    ad.spec_out = n.d_an_spec;
    ad.time_out = n.d_an_time;
    launch_analyser(ad, b->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpy(cache.spec.data(), n.d_an_spec, (size_t)M * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cache.time.data(), n.d_an_time, (size_t)N * sizeof(float), hipMemcpyDeviceToHost));
  }
  auto ins = n.an_cache.emplace(inst, std::move(cache));
  *out = &ins.first->second;
  return 0;
}

waa_status waa_analyser_get_float_frequency_data(waa_batch* b, uint32_t node, uint32_t inst, float* dst, uint32_t nn) {
  Node::AnCache* c;
  int e = analyser_compute(b, node, inst, &c);
  if (e) return e;
  const uint32_t len = std::min<uint32_t>(nn, (uint32_t)c->spec.size());
  for (uint32_t k = 0; k < len; k++) dst[k] = 20.f * log10f(c->spec[k]);  // analysis.rs:365-368
  return WAA_OK;
}
waa_status waa_analyser_get_byte_frequency_data(waa_batch* b, uint32_t node, uint32_t inst, uint8_t* dst, uint32_t nn) {
  Node::AnCache* c;
  int e = analyser_compute(b, node, inst, &c);
  if (e) return e;
  const Node& n = b->nodes[node];
  const float mind = (float)n.desc.d[1], maxd = (float)n.desc.d[2];
  const uint32_t len = std::min<uint32_t>(nn, (uint32_t)c->spec.size());
  for (uint32_t k = 0; k < len; k++) {  // analysis.rs:388-400
    const float db = 20.f * log10f(c->spec[k]);
    const float scaled = 255.f / (maxd - mind) * (db - mind);
    const float clamped = scaled < 0.f ? 0.f : scaled > 255.f ? 255.f : scaled;
    dst[k] = std::isnan(scaled) ? 0 : (uint8_t)clamped;
  }
  return WAA_OK;
}
waa_status waa_analyser_get_float_time_domain_data(waa_batch* b, uint32_t node, uint32_t inst, float* dst, uint32_t nn) {
  Node::AnCache* c;
  int e = analyser_compute(b, node, inst, &c);
  if (e) return e;
  const uint32_t N = (uint32_t)c->time.size();
  const uint32_t len = std::min(nn, N);  // ring_buffer.read: the most recent `len` frames (analysis.rs:114-127)
  for (uint32_t i = 0; i < len; i++) dst[i] = c->time[N - len + i];
  return WAA_OK;
}
waa_status waa_analyser_get_byte_time_domain_data(waa_batch* b, uint32_t node, uint32_t inst, uint8_t* dst, uint32_t nn) {
  Node::AnCache* c;
  int e = analyser_compute(b, node, inst, &c);
  if (e) return e;
  const uint32_t N = (uint32_t)c->time.size();
  const uint32_t len = std::min(nn, N);
  for (uint32_t i = 0; i < nn; i++) {  // analysis.rs:268-276 (elements past fft_size read a zeroed tmp)
    const float v = i < len ? c->time[N - len + i] : 0.f;
    const float scaled = 128.f * (1.f + v);
    const float clamped = scaled < 0.f ? 0.f : scaled > 255.f ? 255.f : scaled;
    dst[i] = (uint8_t)clamped;
  }
  return WAA_OK;
}

// buffer.rs:311-363 (input prep, host side)
uint64_t waa_buffer_resample(const float* src, uint64_t frames, float source_sr, float target_sr, float* dst, uint64_t cap) {
  if (std::fabs(source_sr - target_sr) <= 0.1f || frames == 0) {
    if (dst)
      for (uint64_t i = 0; i < frames && i < cap; i++) dst[i] = src[i];
    return frames;
  }
  const double ratio = (double)target_sr / (double)source_sr;
  const uint64_t tl = (uint64_t)std::ceil((double)frames * ratio);
  if (!dst) return tl;
  for (uint64_t i = 0; i < tl && i < cap; i++) {
    const double position = (double)i / (double)(tl - 1);
    const double playhead = position * (double)(frames - 1);
    const double pf = std::floor(playhead);
    const uint64_t prev = (uint64_t)pf;
    const uint64_t next = std::min<uint64_t>(prev + 1, frames - 1);
    const float k = (float)(playhead - pf), kinv = 1.f - k;
    dst[i] = kinv * src[prev] + k * src[next];
  }
  return tl;
}

// iir_filter.rs:218-262 (control side, host)
waa_status waa_iir_frequency_response(const double* ff, uint32_t nff, const double* fb, uint32_t nfb, float sample_rate,
                                      const float* hz, float* mag, float* phase, uint32_t n) {
  if (int e = check_iir_coefs(ff, nff, fb, nfb)) return e;
  if (n && (!hz || !mag || !phase)) return fail(WAA_ERR_INVALID_ARGUMENT, "null array");
  const double sr = (double)sample_rate, nyquist = sr / 2.;
  for (uint32_t i = 0; i < n; i++) {
    const double freq = (double)hz[i];
    if (freq < 0. || freq > nyquist) {
      mag[i] = std::nanf("");
      phase[i] = std::nanf("");
      continue;
    }
    const double z = -2.0 * 3.14159265358979323846 * freq / sr;
    std::complex<double> num(0., 0.), den(0., 0.);
    for (uint32_t k = 0; k < nff; k++) num += std::complex<double>(ff[k] * std::cos((double)k * z), ff[k] * std::sin((double)k * z));
    for (uint32_t k = 0; k < nfb; k++) den += std::complex<double>(fb[k] * std::cos((double)k * z), fb[k] * std::sin((double)k * z));
    const double ns = den.real() * den.real() + den.imag() * den.imag();
    const double rr = (num.real() * den.real() + num.imag() * den.imag()) / ns;
    const double ri = (num.imag() * den.real() - num.real() * den.imag()) / ns;
    mag[i] = (float)std::hypot(rr, ri);
    phase[i] = (float)std::atan2(ri, rr);
  }
  return WAA_OK;
}

// biquad_filter.rs:670-735 (control side, host)
waa_status waa_biquad_frequency_response(int32_t type, float sample_rate, float frequency, float detune, float q,
                                         float gain, const float* hz, float* mag, float* phase, uint32_t n) {
  if (type < 0 || type > 7) return fail(WAA_ERR_INVALID_ARGUMENT, "bad filter type");
  const double PI = 3.14159265358979323846;
  const float nyq = sample_rate / 2.f;
  const Coefs c = biquad_coefs(type, (double)sample_rate, (double)computed_freq(frequency, detune), (double)gain, (double)q);
  for (uint32_t i = 0; i < n; i++) {
    const float f = hz[i];
    if (f < 0.f || f > nyq) {
      mag[i] = NAN;
      phase[i] = NAN;
      continue;
    }
    const float fn = f / nyq;
    const double omega = -PI * (double)fn;
    const double zr = std::cos(omega), zi = std::sin(omega);
    const double tr = c.b1 + c.b2 * zr, ti = c.b2 * zi;
    const double nr = c.b0 + (tr * zr - ti * zi), ni = tr * zi + ti * zr;
    const double ur = c.a1 + c.a2 * zr, ui = c.a2 * zi;
    const double dr = 1. + (ur * zr - ui * zi), di = ur * zi + ui * zr;
    const double den = dr * dr + di * di;
    const double rr = (nr * dr + ni * di) / den, ri = (ni * dr - nr * di) / den;
    mag[i] = (float)std::hypot(rr, ri);
    phase[i] = (float)std::atan2(ri, rr);
  }
  return WAA_OK;
}

waa_status waa_profile_enable(waa_batch* b, int32_t on) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  b->profiling = on != 0;
  return WAA_OK;
}
int32_t waa_profile_count(waa_batch* b) { return b ? (int32_t)b->prof.size() : 0; }
waa_status waa_profile_get(waa_batch* b, int32_t i, const char** name, uint64_t* launches, double* ms) {
  if (!b || i < 0 || i >= (int32_t)b->prof.size()) return fail(WAA_ERR_INVALID_ARGUMENT, "profile index out of range");
  *name = b->prof[i].name.c_str();
  *launches = b->prof[i].launches;
  *ms = b->prof[i].total_ms;
  return WAA_OK;
}
waa_status waa_profile_reset(waa_batch* b) {
  if (!b) return fail(WAA_ERR_INVALID_ARGUMENT, "null batch");
  for (auto& p : b->prof) {
    p.launches = 0;
    p.total_ms = 0;
  }
  return WAA_OK;
}

}  // extern "C"
