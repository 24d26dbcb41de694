// waa_automation.cpp — AudioParam automation on the host (control-side work in the reference: the events are queued
// by the control thread and evaluated once per render quantum by AudioParamProcessor, src/param.rs:686-1584).
// `Timeline` is a C++ restatement of that processor: the event queue with its insertion rules
// (handle_incoming_event, param.rs:796-1047) and the per-block evaluation (compute_buffer and the five
// compute_*_automation methods, param.rs:1049-1584), same f32/f64 arithmetic and operation order.
// waa_param_schedule_event feeds per-(param, instance) timelines; the planner asks for all quanta at once and
// uploads the values like caller-provided blocks.  The stand-alone waa_timeline_* entry points expose the same
// object so that the reference's unit-test vectors can be checked without a device (tests/test_automation.py).
#include "waa_host.hpp"

namespace waa {
namespace host {

namespace {
constexpr float kSnapToTarget = 1e-10f;  // param.rs:22

float linear_sample(double t0, double duration, float v0, float diff, double t) {
  const double phase = (t - t0) / duration;
  return std::fma(diff, (float)phase, v0);
}
float exponential_sample(double t0, double duration, float v0, float ratio, double t) {
  const double phase = (t - t0) / duration;
  return v0 * std::pow(ratio, (float)phase);
}
float target_sample(double t0, double time_constant, float v1, float diff, double t) {
  const double exponent = -((t - t0) / time_constant);
  return std::fma(diff, (float)std::exp(exponent), v1);
}
float curve_sample(double t0, double duration, const std::vector<float>& values, double t) {
  if (t - t0 >= duration) return values.back();
  const double position = (double)(values.size() - 1) * (t - t0) / duration;
  // `position as usize` saturates in Rust: a sample time before the curve's start (the intrinsic value kept for
  // the NEXT block while the curve has not started yet, param.rs:1470-1478) reads segment 0 with the fractional
  // part of the negative position — reproduced, not repaired
  const size_t k = position > 0. ? (size_t)position : 0;
  const float phase = (float)(position - std::floor(position));
  return std::fma(values[k + 1] - values[k], phase, values[k]);
}
}  // namespace

struct Timeline::Event {
  int type = WAA_EVENT_SET_VALUE;
  float value = 0.f;
  double time = 0.;
  double time_constant = 0.;  // SetTarget
  bool cancelled = false;     // CancelAndHold rewrote the end of this event
  double cancel_time = 0.;
  double duration = 0.;       // SetValueCurve
  std::vector<float> values;  // SetValueCurve
};

Timeline::Timeline(float default_value, float min_value, float max_value, bool a_rate)
    : min_(min_value), max_(max_value), intrinsic_(default_value), current_(default_value), a_rate_(a_rate) {}
Timeline::~Timeline() = default;

float Timeline::value() const { return current_; }

void Timeline::export_queue(TlHeader* hdr, std::vector<TlEvent>* events, std::vector<float>* curves) const {
  hdr->minv = min_;
  hdr->maxv = max_;
  hdr->intrinsic = intrinsic_;
  hdr->a_rate = a_rate_ ? 1 : 0;
  hdr->ev_off = (int32_t)events->size();
  hdr->n_events = (int32_t)queue_.size();
  hdr->pad = 0;
  for (const Event& e : queue_) {
    TlEvent t{};
    t.type = e.type;
    t.value = e.value;
    t.time = e.time;
    t.time_constant = e.time_constant;
    t.cancel_time = e.cancel_time;
    t.duration = e.duration;
    t.cancelled = e.cancelled ? 1 : 0;
    t.curve_off = (int32_t)curves->size();
    t.curve_len = (int32_t)e.values.size();
    curves->insert(curves->end(), e.values.begin(), e.values.end());
    events->push_back(t);
  }
}

int Timeline::schedule(int type, float value, double time, double aux, const float* curve, uint32_t n_curve) {
  Event ev;
  ev.type = type;
  ev.value = value;
  ev.time = time;
  // ---- control side: the *_raw constructors and their assertions (param.rs:24-62, 399-596)
  const bool needs_value = type == WAA_EVENT_SET_VALUE || type == WAA_EVENT_SET_VALUE_AT_TIME || type == WAA_EVENT_LINEAR_RAMP ||
                           type == WAA_EVENT_EXPONENTIAL_RAMP || type == WAA_EVENT_SET_TARGET;
  if (type < WAA_EVENT_SET_VALUE || type > WAA_EVENT_SET_VALUE_CURVE) return fail(WAA_ERR_INVALID_ARGUMENT, "unknown automation event type %d", type);
  if (needs_value && !std::isfinite(value)) return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided value is non-finite.");
  if (type == WAA_EVENT_EXPONENTIAL_RAMP && value == 0.f)
    return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - value (0.0) should not be equal to zero");
  if (type != WAA_EVENT_SET_VALUE) {
    if (!std::isfinite(time)) return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided time value is non-finite.");
    if (time < 0.) return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - The provided time value cannot be negative");
  }
  switch (type) {
    case WAA_EVENT_SET_VALUE:
      current_ = std::fmin(std::fmax(value, min_), max_);
      ev.time = 0.;
      break;
    case WAA_EVENT_SET_TARGET:
      if (!std::isfinite(aux)) return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided time value is non-finite.");
      if (aux < 0.) return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - The provided time value cannot be negative");
      if (aux == 0.)
        ev.type = WAA_EVENT_SET_VALUE_AT_TIME;  // "the output value jumps immediately to the final value"
      else
        ev.time_constant = aux;
      break;
    case WAA_EVENT_CANCEL_SCHEDULED_VALUES:
    case WAA_EVENT_CANCEL_AND_HOLD: ev.value = 0.f; break;
    case WAA_EVENT_SET_VALUE_CURVE:
      if (!curve || n_curve < 2)
        return fail(WAA_ERR_INVALID_STATE, "InvalidStateError - sequence length (%u) should not be less than 2", n_curve);
      if (!std::isfinite(aux)) return fail(WAA_ERR_INVALID_ARGUMENT, "TypeError - The provided value is non-finite.");
      if (!(aux > 0.)) return fail(WAA_ERR_INVALID_ARGUMENT, "RangeError - duration (%g) should be strictly positive", aux);
      ev.value = 0.f;
      ev.duration = aux;
      ev.values.assign(curve, curve + n_curve);
      break;
    default: break;
  }
  return insert(std::move(ev));
}

int Timeline::schedule_at(uint32_t arrival_q, int type, float value, double time, double aux, const float* curve, uint32_t n_curve) {
  if (arrival_q == 0 && arrivals_.empty()) return schedule(type, value, time, aux, curve, n_curve);
  {
    // the control-side assertions now (a scratch timeline without a queue: only the argument checks can fire)
    Timeline probe(intrinsic_, min_, max_, a_rate_);
    if (int e = probe.schedule(type, value, time, aux, curve, n_curve)) return e;
  }
  Arrival a;
  a.q = arrival_q;
  a.type = type;
  a.value = value;
  a.time = time;
  a.aux = aux;
  if (curve && n_curve) a.curve.assign(curve, curve + n_curve);
  arrivals_.push_back(std::move(a));
  return 0;
}

int Timeline::apply_arrivals(uint32_t q) {
  while (next_arrival_ < arrivals_.size() && arrivals_[next_arrival_].q <= q) {
    const Arrival& a = arrivals_[next_arrival_++];
    if (int e = schedule(a.type, a.value, a.time, a.aux, a.curve.empty() ? nullptr : a.curve.data(), (uint32_t)a.curve.size())) return e;
  }
  return 0;
}

// handle_incoming_event, param.rs:796-1047
int Timeline::insert(Event ev) {
  auto sort_queue = [&] {
    std::stable_sort(queue_.begin(), queue_.end(), [](const Event& a, const Event& b) { return a.time < b.time; });
  };
  if (ev.type == WAA_EVENT_CANCEL_SCHEDULED_VALUES) {
    if (!queue_.empty()) {
      const Event& cur = queue_.front();
      const bool ramp = cur.type == WAA_EVENT_LINEAR_RAMP || cur.type == WAA_EVENT_EXPONENTIAL_RAMP;
      if (ramp && cur.time >= ev.time && last_) intrinsic_ = last_->value;  // in the middle of a ramp: restore
    }
    queue_.erase(std::remove_if(queue_.begin(), queue_.end(), [&](const Event& q) { return !(q.time < ev.time); }), queue_.end());
    return 0;
  }
  if (ev.type == WAA_EVENT_CANCEL_AND_HOLD) {
    sort_queue();
    Event *before = nullptr, *after = nullptr;  // E1: last event at or before t_c, E2: first event after it
    double t1 = -DBL_MAX, t2 = DBL_MAX;
    for (Event& q : queue_) {
      if (q.time >= t1 && q.time <= ev.time) {
        t1 = q.time;
        before = &q;
      } else if (q.time < t2 && q.time > ev.time) {
        t2 = q.time;
        after = &q;
      }
    }
    auto cancel = [&](Event* e) {
      e->cancelled = true;
      e->cancel_time = ev.time;
    };
    if (after) {
      if (after->type == WAA_EVENT_LINEAR_RAMP || after->type == WAA_EVENT_EXPONENTIAL_RAMP) cancel(after);
    } else if (before) {
      if (before->type == WAA_EVENT_SET_TARGET)
        cancel(before);
      else if (before->type == WAA_EVENT_SET_VALUE_CURVE && ev.time <= before->time + before->duration)
        cancel(before);
    }
    queue_.erase(std::remove_if(queue_.begin(), queue_.end(),
                                [&](const Event& q) { return !((q.cancelled ? q.cancel_time : q.time) <= ev.time); }),
                 queue_.end());
    return 0;
  }
  if (ev.type == WAA_EVENT_SET_VALUE_CURVE) {  // a curve may not span another event
    const double a = ev.time, b = ev.time + ev.duration;
    for (const Event& q : queue_)
      if (!(q.time <= a || q.time >= b))
        return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - scheduling SetValueCurveAtTime at time of another automation event");
  } else {  // ... and no automation method may fall inside a curve
    for (const Event& q : queue_)
      if (q.type == WAA_EVENT_SET_VALUE_CURVE && !(ev.time <= q.time || ev.time >= q.time + q.duration))
        return fail(WAA_ERR_NOT_SUPPORTED, "NotSupportedError - scheduling automation event during SetValueCurveAtTime");
  }
  if (ev.type == WAA_EVENT_SET_VALUE) intrinsic_ = ev.value;
  // a ramp (first event ever) or a SetTarget (empty queue) needs a start value: implicit SetValue at "now"
  const bool ramp = ev.type == WAA_EVENT_LINEAR_RAMP || ev.type == WAA_EVENT_EXPONENTIAL_RAMP;
  if (queue_.empty() && ((ramp && !last_) || ev.type == WAA_EVENT_SET_TARGET)) {
    Event init;
    init.type = WAA_EVENT_SET_VALUE;
    init.value = intrinsic_;
    init.time = 0.;
    queue_.push_back(init);
  }
  queue_.push_back(std::move(ev));
  sort_queue();
  return 0;
}

// compute_buffer, param.rs:1498-1584: writes 1 or `count` values
uint32_t Timeline::compute(double block_time, double dt, uint32_t count, float* out) {
  current_ = std::fmin(std::fmax(intrinsic_, min_), max_);
  const double next_block_time = std::fma(dt, (double)count, block_time);
  uint32_t len = 0;
  auto push = [&](float v) { out[len++] = v; };
  auto pop_to_last = [&](double time, float value) {
    Event e = std::move(queue_.front());
    queue_.erase(queue_.begin());
    e.time = time;
    e.value = value;
    e.values.clear();
    last_ = std::make_unique<Event>(std::move(e));
  };
  // index of the first frame at or after `t` (rounded like the reference), clipped to the block
  auto end_index = [&](double t) -> uint32_t {
    const double v = std::round(std::fmax(t - block_time, 0.) / dt);
    return v > (double)count ? count : (uint32_t)v;
  };
  bool constant_block = true;
  if (!queue_.empty()) {
    const Event& e = queue_.front();
    constant_block = (e.type != WAA_EVENT_LINEAR_RAMP && e.type != WAA_EVENT_EXPONENTIAL_RAMP) && e.time >= next_block_time;
  }
  if (!a_rate_ || constant_block) {
    push(intrinsic_);
    if (constant_block) return len;
  }
  for (;;) {
    if (queue_.empty()) {
      if (a_rate_)
        while (len < count) push(intrinsic_);
      break;
    }
    Event& ev = queue_.front();
    bool block_done = false;
    switch (ev.type) {
      case WAA_EVENT_SET_VALUE:
      case WAA_EVENT_SET_VALUE_AT_TIME: {  // param.rs:1049-1096
        const double time = ev.time == 0. ? block_time : ev.time;
        if (a_rate_) {
          const uint32_t end = end_index(time);
          while (len < end) push(intrinsic_);  // (push advances len)
        }
        if (time > next_block_time) {
          block_done = true;
          break;
        }
        intrinsic_ = ev.value;
        pop_to_last(time, ev.value);
        break;
      }
      case WAA_EVENT_LINEAR_RAMP:
      case WAA_EVENT_EXPONENTIAL_RAMP: {  // param.rs:1100-1278
        const bool linear = ev.type == WAA_EVENT_LINEAR_RAMP;
        // (a ramp or target at the head of the queue with no consumed event before it — possible when later calls
        // insert events EARLIER than an already queued first ramp — makes the reference panic, param.rs:1107 `unwrap`;
        // here the missing event reads as time 0 / value 0, like the oracle, instead of crashing the caller)
        const Event no_event{};
        const Event& last = last_ ? *last_ : no_event;
        const double t0 = last.time;
        const double duration = ev.time - t0;  // the declared slope survives a CancelAndHold
        const double t1 = ev.cancelled ? ev.cancel_time : ev.time;
        const float v0 = last.value, v1 = ev.value;
        const float k = linear ? v1 - v0 : v1 / v0;
        if (!linear && (v0 == 0.f || v0 * v1 < 0.f)) {  // v(t) = V0 until T1: behaves as a SetValueAtTime(T1)
          Event rep;
          rep.type = WAA_EVENT_SET_VALUE_AT_TIME;
          rep.value = v1;
          rep.time = t1;
          ev = rep;
          break;
        }
        auto sample = [&](double t) { return linear ? linear_sample(t0, duration, v0, k, t) : exponential_sample(t0, duration, v0, k, t); };
        if (a_rate_) {
          const uint32_t end = end_index(t1);
          if (end > len) {
            double t = std::fma((double)len, dt, block_time);
            float v = 0.f;
            while (len < end) {
              v = sample(t);
              push(v);
              t += dt;
            }
            intrinsic_ = v;
          }
        }
        if (t1 >= next_block_time) {  // continues in the next block
          intrinsic_ = sample(next_block_time);
          block_done = true;
          break;
        }
        if (ev.cancelled) {
          const float v = sample(t1);
          intrinsic_ = v;
          pop_to_last(t1, v);
        } else {
          intrinsic_ = v1;
          pop_to_last(ev.time, v1);
        }
        break;
      }
      case WAA_EVENT_SET_TARGET: {  // param.rs:1286-1420
        double t1 = next_block_time;
        bool ended = false;
        if (queue_.size() > 1) {
          const Event& nx = queue_[1];
          if (nx.type == WAA_EVENT_LINEAR_RAMP || nx.type == WAA_EVENT_EXPONENTIAL_RAMP) {
            t1 = block_time;  // the ramp replaces the SetTarget from "now"
            ended = true;
          } else if (nx.time < next_block_time) {
            t1 = nx.time;
            ended = true;
          }
        }
        if (ev.cancelled && ev.cancel_time < next_block_time) {
          t1 = ev.cancel_time;
          ended = true;
        }
        const double t0 = ev.time, tau = ev.time_constant;
        const float v0 = last_ ? last_->value : 0.f, v1 = ev.value, diff = v0 - v1;  // (no consumed event: see the ramps)
        if (a_rate_) {
          const uint32_t end = end_index(t1);
          if (end > len) {
            double t = std::fma((double)len, dt, block_time);
            float v = 0.f;
            while (len < end) {
              v = (t - t0 < 0.) ? intrinsic_ : target_sample(t0, tau, v1, diff, t);
              push(v);
              t += dt;
            }
            intrinsic_ = v;
          }
        }
        if (!ended) {
          const float v = target_sample(t0, tau, v1, diff, next_block_time);
          if (std::fabs(v1 - v) < kSnapToTarget) {  // close enough: becomes a SetValueAtTime(next block)
            intrinsic_ = v1;
            if (v1 == 0.f)
              for (uint32_t i = 0; i < len; i++)
                if (std::fpclassify(out[i]) == FP_SUBNORMAL) out[i] = 0.f;
            Event rep;
            rep.type = WAA_EVENT_SET_VALUE_AT_TIME;
            rep.value = v1;
            rep.time = next_block_time;
            ev = rep;
          } else {
            intrinsic_ = v;
          }
          block_done = true;
          break;
        }
        const float v = target_sample(t0, tau, v1, diff, t1);
        intrinsic_ = v;
        pop_to_last(t1, v);
        break;
      }
      case WAA_EVENT_SET_VALUE_CURVE: {  // param.rs:1422-1496
        const double t0 = ev.time, duration = ev.duration;
        const double t1 = ev.cancelled ? ev.cancel_time : t0 + duration;
        if (a_rate_) {
          const uint32_t end = end_index(t1);
          if (end > len) {
            double t = std::fma((double)len, dt, block_time);
            float v = 0.f;
            while (len < end) {
              v = t < t0 ? intrinsic_ : curve_sample(t0, duration, ev.values, t);
              push(v);
              t += dt;
            }
            intrinsic_ = v;
          }
        }
        if (t1 >= next_block_time) {
          intrinsic_ = curve_sample(t0, duration, ev.values, next_block_time);
          block_done = true;
          break;
        }
        const float v = ev.cancelled ? curve_sample(t0, duration, ev.values, t1) : ev.values.back();
        intrinsic_ = v;
        pop_to_last(t1, v);
        break;
      }
      default: block_done = true; break;
    }
    if (block_done) break;
  }
  return len;
}

}  // namespace host
}  // namespace waa

// ---- C ABI: the stand-alone timeline object ------------------------------------------------------------------
using waa::host::Timeline;
struct waa_timeline {
  Timeline tl;
  waa_timeline(float d, float lo, float hi, bool a) : tl(d, lo, hi, a) {}
};
extern "C" {
waa_timeline* waa_timeline_create(float default_value, float min_value, float max_value, int32_t a_rate) {
  return new waa_timeline(default_value, min_value, max_value, a_rate != 0);
}
void waa_timeline_destroy(waa_timeline* t) { delete t; }
waa_status waa_timeline_event(waa_timeline* t, int32_t type, float value, double time, double aux, const float* curve,
                              uint32_t n_curve) {
  if (!t) return waa::host::fail(WAA_ERR_INVALID_ARGUMENT, "null timeline");
  return t->tl.schedule(type, value, time, aux, curve, n_curve);
}
uint32_t waa_timeline_compute(waa_timeline* t, double block_time, double dt, uint32_t count, float* out) {
  return t->tl.compute(block_time, dt, count, out);
}
float waa_timeline_value(const waa_timeline* t) { return t->tl.value(); }
// The same timeline replayed by timeline_kernel on the current HIP device for `n_quanta` render quanta from time 0:
// out[n_quanta * 128] (single-valued slices replicated), lens[n_quanta] = 1 or 128.  The timeline object itself is
// not consumed.  Parity hook for tests/test_automation.py (device replay vs waa_timeline_compute).
waa_status waa_timeline_render_device(const waa_timeline* t, uint32_t n_quanta, float sample_rate, float* out, uint8_t* lens) {
  using namespace waa;
  if (!t || !out || !lens || n_quanta == 0) return waa::host::fail(WAA_ERR_INVALID_ARGUMENT, "bad arguments");
  TlHeader hdr{};
  std::vector<TlEvent> events;
  std::vector<float> curves;
  t->tl.export_queue(&hdr, &events, &curves);
  hdr.defv = 0.f;
  hdr.minv = -FLT_MAX;  // (the stand-alone object is compared before clamping, like waa_timeline_compute)
  hdr.maxv = FLT_MAX;
  TlHeader* d_hdr = nullptr;
  TlEvent *d_ev = nullptr, *d_work = nullptr;
  float *d_curves = nullptr, *d_out = nullptr;
  uint8_t* d_lens = nullptr;
  const size_t n_ev = std::max<size_t>(events.size(), 1), n_cv = std::max<size_t>(curves.size(), 1);
  auto cleanup = [&] {
    (void)hipFree(d_hdr);
    (void)hipFree(d_ev);
    (void)hipFree(d_work);
    (void)hipFree(d_curves);
    (void)hipFree(d_out);
    (void)hipFree(d_lens);
  };
#define TL_TRY(expr)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      cleanup();                                                                                              \
      return waa::host::fail(WAA_ERR_DEVICE, "HIP error %s (%s)", hipGetErrorString(e_), #expr);              \
    }                                                                                                         \
  } while (0)
  TL_TRY(hipMalloc(&d_hdr, sizeof hdr));
  TL_TRY(hipMalloc(&d_ev, n_ev * sizeof(TlEvent)));
  TL_TRY(hipMalloc(&d_work, n_ev * sizeof(TlEvent)));
  TL_TRY(hipMalloc(&d_curves, n_cv * sizeof(float)));
  TL_TRY(hipMalloc(&d_out, (size_t)n_quanta * RQ * sizeof(float)));
  TL_TRY(hipMalloc(&d_lens, n_quanta));
  TL_TRY(hipMemcpy(d_hdr, &hdr, sizeof hdr, hipMemcpyHostToDevice));
  if (!events.empty()) TL_TRY(hipMemcpy(d_ev, events.data(), events.size() * sizeof(TlEvent), hipMemcpyHostToDevice));
  if (!curves.empty()) TL_TRY(hipMemcpy(d_curves, curves.data(), curves.size() * sizeof(float), hipMemcpyHostToDevice));
  TimelineDesc d{};
  d.hdr = d_hdr;
  d.events = d_ev;
  d.work = d_work;
  d.curves = d_curves;
  d.out = d_out;
  d.lens = d_lens;
  d.out_stride = (uint64_t)n_quanta * RQ;
  d.rows = 1;
  d.n_quanta = n_quanta;
  d.sample_rate = (double)sample_rate;
  launch_timeline(d, nullptr);
  TL_TRY(hipGetLastError());
  TL_TRY(hipDeviceSynchronize());
  TL_TRY(hipMemcpy(out, d_out, (size_t)n_quanta * RQ * sizeof(float), hipMemcpyDeviceToHost));
  TL_TRY(hipMemcpy(lens, d_lens, n_quanta, hipMemcpyDeviceToHost));
#undef TL_TRY
  cleanup();
  return WAA_OK;
}
}
