// waa_timeline.hip — AudioParam automation evaluated on the device.
//
// The reference evaluates every AudioParam's event timeline once per render quantum on the render thread
// (AudioParamProcessor::compute_buffer and the compute_*_automation methods, src/param.rs:1049-1584).  The library's
// host restatement of that processor (waa_automation.cpp::Timeline) stays the way automation that is THE SAME for every
// instance is rendered: one evaluation for the whole batch, control-side work.  When the instances of a batch carry
// DIFFERENT event lists (one synth voice per context, each with its own envelope) the host would evaluate
// n_instances x n_quanta x 128 values and upload 512 B per instance-quantum; here the sorted event queues are uploaded
// instead (56 B per event) and one thread per (param, instance) replays compute_buffer for all quanta in order, writing
// the per-frame values the node kernels read as a-rate ParamRefs, and the length (1 or 128) the reference's slice would
// have had in each quantum.
//
// The code below is the device twin of Timeline::compute: same state (intrinsic value, the last consumed event, the
// head of the queue, events rewritten in place by CancelAndHold / SetTarget snapping), same f32 / f64 operations in
// the same order, the sample time advanced by repeated addition.  Linear ramps, value curves and set-value events are
// bit-identical to the host; exponential ramps (powf, evaluated in f64 and rounded once) and set-target (f64 exp) follow the
// device math library: identical to the host in all but rare last-bit cases.
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {

namespace {
constexpr float kSnapToTarget = 1e-10f;  // param.rs:22
enum { EV_SET_VALUE = 0, EV_SET_VALUE_AT_TIME = 1, EV_LINEAR_RAMP = 2, EV_EXPONENTIAL_RAMP = 3, EV_SET_TARGET = 5, EV_SET_VALUE_CURVE = 7 };

__device__ __forceinline__ float linear_sample(double t0, double duration, float v0, float diff, double t) {
  const double phase = (t - t0) / duration;
  return __builtin_fmaf(diff, (float)phase, v0);
}
__device__ __forceinline__ float exponential_sample(double t0, double duration, float v0, float ratio, double t) {
  const double phase = (t - t0) / duration;
  // f32::powf of the reference = glibc's powf (~0.52 ulp).  The device's powf is 1-2 ulp; evaluating in f64 and rounding
  // once gives the correctly rounded f32 power, which is what glibc returns in all but rare near-tie cases
  return v0 * (float)pow((double)ratio, (double)(float)phase);
}
__device__ __forceinline__ float target_sample(double t0, double time_constant, float v1, float diff, double t) {
  const double exponent = -((t - t0) / time_constant);
  return __builtin_fmaf(diff, (float)exp(exponent), v1);
}
__device__ __forceinline__ float curve_sample(double t0, double duration, const float* values, int n, double t) {
  if (t - t0 >= duration) return values[n - 1];
  const double position = (double)(n - 1) * (t - t0) / duration;
  const size_t k = position > 0. ? (size_t)position : 0;  // `as usize` saturates (see waa_automation.cpp)
  const float phase = (float)(position - floor(position));
  return __builtin_fmaf(values[k + 1] - values[k], phase, values[k]);
}
}  // namespace

__global__ __launch_bounds__(64) void timeline_kernel(const TimelineDesc d) {
  const uint32_t row = blockIdx.x * 64 + threadIdx.x;
  if (row >= d.rows) return;
  const TlHeader h = d.hdr[row];
  TlEvent* queue = d.work + h.ev_off;
  for (int i = 0; i < h.n_events; i++) queue[i] = d.events[h.ev_off + i];  // every render starts from the scheduled queue
  int head = 0;
  const int n_ev = h.n_events;
  float intrinsic = h.intrinsic;
  bool has_last = false;
  double last_time = 0.;
  float last_value = 0.f;
  const bool a_rate = h.a_rate != 0;
  const double dt = 1. / d.sample_rate;
  const uint32_t count = RQ;
  float* out_row = d.out + (uint64_t)row * d.out_stride;
  uint8_t* len_row = d.lens + (uint64_t)row * d.n_quanta;
  auto fix = [&](float x) { return x != x ? h.defv : fminf(fmaxf(x, h.minv), h.maxv); };  // param.rs:755-761

  for (uint32_t q = 0; q < d.n_quanta; q++) {
    const double block_time = (double)((uint64_t)q * RQ) / d.sample_rate;
    const double next_block_time = __builtin_fma(dt, (double)count, block_time);
    float* out = out_row + (uint64_t)q * RQ;
    uint32_t len = 0;
    auto push = [&](float v) { out[len++] = v; };
    auto pop_to_last = [&](double time, float value) {
      head++;
      has_last = true;
      last_time = time;
      last_value = value;
    };
    auto end_index = [&](double t) -> uint32_t {
      const double v = round(fmax(t - block_time, 0.) / dt);
      return v > (double)count ? count : (uint32_t)v;
    };
    bool constant_block = true;
    if (head < n_ev) {
      const TlEvent& e = queue[head];
      constant_block = (e.type != EV_LINEAR_RAMP && e.type != EV_EXPONENTIAL_RAMP) && e.time >= next_block_time;
    }
    bool finished = false;
    if (!a_rate || constant_block) {
      push(intrinsic);
      finished = constant_block;
    }
    while (!finished) {
      if (head >= n_ev) {
        if (a_rate)
          while (len < count) push(intrinsic);
        break;
      }
      TlEvent& ev = queue[head];
      bool block_done = false;
      switch (ev.type) {
        case EV_SET_VALUE:
        case EV_SET_VALUE_AT_TIME: {  // param.rs:1049-1096
          const double time = ev.time == 0. ? block_time : ev.time;
          if (a_rate) {
            const uint32_t end = end_index(time);
            while (len < end) push(intrinsic);
          }
          if (time > next_block_time) {
            block_done = true;
            break;
          }
          intrinsic = ev.value;
          pop_to_last(time, ev.value);
          break;
        }
        case EV_LINEAR_RAMP:
        case EV_EXPONENTIAL_RAMP: {  // param.rs:1100-1278
          const bool linear = ev.type == EV_LINEAR_RAMP;
          const double t0 = has_last ? last_time : 0.;
          const double duration = ev.time - t0;
          const double t1 = ev.cancelled ? ev.cancel_time : ev.time;
          const float v0 = has_last ? last_value : 0.f, v1 = ev.value;
          const float k = linear ? v1 - v0 : v1 / v0;
          if (!linear && (v0 == 0.f || v0 * v1 < 0.f)) {  // behaves as a SetValueAtTime(T1)
            TlEvent rep = ev;
            rep.type = EV_SET_VALUE_AT_TIME;
            rep.value = v1;
            rep.time = t1;
            rep.cancelled = 0;
            ev = rep;
            break;
          }
          if (a_rate) {
            const uint32_t end = end_index(t1);
            if (end > len) {
              double t = __builtin_fma((double)len, dt, block_time);
              float v = 0.f;
              while (len < end) {
                v = linear ? linear_sample(t0, duration, v0, k, t) : exponential_sample(t0, duration, v0, k, t);
                push(v);
                t += dt;
              }
              intrinsic = v;
            }
          }
          if (t1 >= next_block_time) {
            intrinsic = linear ? linear_sample(t0, duration, v0, k, next_block_time) : exponential_sample(t0, duration, v0, k, next_block_time);
            block_done = true;
            break;
          }
          if (ev.cancelled) {
            const float v = linear ? linear_sample(t0, duration, v0, k, t1) : exponential_sample(t0, duration, v0, k, t1);
            intrinsic = v;
            pop_to_last(t1, v);
          } else {
            intrinsic = v1;
            pop_to_last(ev.time, v1);
          }
          break;
        }
        case EV_SET_TARGET: {  // param.rs:1286-1420
          double t1 = next_block_time;
          bool ended = false;
          if (head + 1 < n_ev) {
            const TlEvent& nx = queue[head + 1];
            if (nx.type == EV_LINEAR_RAMP || nx.type == EV_EXPONENTIAL_RAMP) {
              t1 = block_time;
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
          const float v0 = has_last ? last_value : 0.f, v1 = ev.value, diff = v0 - v1;
          if (a_rate) {
            const uint32_t end = end_index(t1);
            if (end > len) {
              double t = __builtin_fma((double)len, dt, block_time);
              float v = 0.f;
              while (len < end) {
                v = (t - t0 < 0.) ? intrinsic : target_sample(t0, tau, v1, diff, t);
                push(v);
                t += dt;
              }
              intrinsic = v;
            }
          }
          if (!ended) {
            const float v = target_sample(t0, tau, v1, diff, next_block_time);
            if (fabsf(v1 - v) < kSnapToTarget) {
              intrinsic = v1;
              if (v1 == 0.f)
                for (uint32_t i = 0; i < len; i++) {
                  const uint32_t bits = __float_as_uint(out[i]);
                  if ((bits & 0x7f800000u) == 0u && (bits & 0x007fffffu) != 0u) out[i] = 0.f;  // subnormal -> 0
                }
              TlEvent rep = ev;
              rep.type = EV_SET_VALUE_AT_TIME;
              rep.value = v1;
              rep.time = next_block_time;
              rep.cancelled = 0;
              ev = rep;
            } else {
              intrinsic = v;
            }
            block_done = true;
            break;
          }
          const float v = target_sample(t0, tau, v1, diff, t1);
          intrinsic = v;
          pop_to_last(t1, v);
          break;
        }
        case EV_SET_VALUE_CURVE: {  // param.rs:1422-1496
          const double t0 = ev.time, duration = ev.duration;
          const double t1 = ev.cancelled ? ev.cancel_time : t0 + duration;
          const float* values = d.curves + ev.curve_off;
          if (a_rate) {
            const uint32_t end = end_index(t1);
            if (end > len) {
              double t = __builtin_fma((double)len, dt, block_time);
              float v = 0.f;
              while (len < end) {
                v = t < t0 ? intrinsic : curve_sample(t0, duration, values, ev.curve_len, t);
                push(v);
                t += dt;
              }
              intrinsic = v;
            }
          }
          if (t1 >= next_block_time) {
            intrinsic = curve_sample(t0, duration, values, ev.curve_len, next_block_time);
            block_done = true;
            break;
          }
          const float v = ev.cancelled ? curve_sample(t0, duration, values, ev.curve_len, t1) : values[ev.curve_len - 1];
          intrinsic = v;
          pop_to_last(t1, v);
          break;
        }
        default: block_done = true; break;
      }
      if (block_done) break;
    }
    // the consumers read 128 values per quantum: a single-valued slice is replicated; clamp like mix_to_output
    len_row[q] = (uint8_t)(len == 1 ? 1 : RQ);
    if (len == 1) {
      const float v = fix(out[0]);
      for (uint32_t i = 0; i < count; i++) out[i] = v;
    } else {
      for (uint32_t i = 0; i < count; i++) out[i] = fix(i < len ? out[i] : 0.f);
    }
  }
}

void launch_timeline(const TimelineDesc& d, void* stream) {
  hipLaunchKernelGGL(timeline_kernel, dim3((d.rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
}

}  // namespace waa
