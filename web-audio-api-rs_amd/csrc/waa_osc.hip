// waa_osc.hip — OscillatorNode (src/node/oscillator.rs:323-660) as an on-device source.
// The oscillator is a serial f64 phase accumulator with a conditional wrap per sample (`unroll_phase`), so its
// samples cannot be produced out of order without changing the rounding.  One LANE renders one instance, frame
// after frame, with the reference's arithmetic (table interpolation in f32 with mul_add, polyBLEP in f64,
// computed frequency in f64); 1024 contexts are 16 wavefronts.  Frequency and detune come as ParamRefs, so
// automation blocks and audio-rate modulation from the graph (FM) are the same code path.
// Latency-bound (about 100 cycles per frame and lane); stores are packed to 16 B per lane.
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"

namespace waa {

namespace {
__device__ __forceinline__ double unroll_phase(double phase) {  // oscillator.rs:646-654
  if (phase >= 1.) return phase - 1.;
  if (phase < 0.) return phase + 1.;
  return phase;
}
__device__ __forceinline__ double unroll_phase_unbounded(double phase) {  // rem_euclid(1.), :656-659
  const double r = fmod(phase, 1.);
  return r < 0. ? r + 1. : r;
}
__device__ __forceinline__ double poly_blep(double t, double dt) {  // :626-643 (production path)
  if (t < dt) {
    t /= dt;
    return t + t - t * t - 1.0;
  } else if (t > 1.0 - dt) {
    t = (t - 1.0) / dt;
    return __builtin_fma(t, t, t) + t + 1.0;
  }
  return 0.0;
}
__device__ __forceinline__ float table_sample(const float* table, int len, double phase) {  // :571-586, :604-619
  const double position = phase * (double)len;
  const double floored = floor(position);
  const int prev_index = (int)floored;
  int next_index = prev_index + 1;
  if (next_index == len) next_index = 0;
  const float k = (float)(position - floored);
  return __builtin_fmaf(table[prev_index], 1.f - k, table[next_index] * k);
}
__device__ __forceinline__ float waveform_sample(int type, const float* table, int table_len, double phase, double phase_incr) {  // :561-602
  switch (type) {
    case 1: {  // square
      double sample = phase < 0.5 ? 1.0 : -1.0;
      sample += poly_blep(phase, phase_incr);
      sample -= poly_blep(unroll_phase(phase + 0.5), phase_incr);
      return (float)sample;
    }
    case 2: {  // sawtooth
      const double ph = unroll_phase(phase + 0.5);
      double sample = 2.0 * ph - 1.0;
      sample -= poly_blep(ph, phase_incr);
      return (float)sample;
    }
    case 3: {  // triangle
      double sample = -4. * phase + 2.;
      if (sample > 1.)
        sample = 2. - sample;
      else if (sample < -1.)
        sample = -2. - sample;
      return (float)sample;
    }
    default: return table_sample(table, table_len, phase);  // sine (2048) or custom (8192)
  }
}
__device__ __forceinline__ float waveform_sample(const OscDesc& d, double phase, double phase_incr) {
  return waveform_sample(d.type, d.table, d.table_len, phase, phase_incr);
}
}  // namespace

namespace {
// four consecutive frames of the oscillator's output through the folded post ops (OscDesc::post_gain / post_dup)
__device__ __forceinline__ void osc_store(const OscDesc& d, uint32_t inst, float* out, uint64_t f0, float (&r)[4]) {
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (k < d.n_post) {
      const float g = d.post_gain[k].base[inst];
      const bool mute = fabsf(g) <= 1e-6f, pass = fabsf(1.f - g) <= 1e-6f;  // gain.rs:163-179
#pragma unroll
      for (int e = 0; e < 4; e++) r[e] = mute ? 0.f : (pass ? r[e] : r[e] * g);
    }
  if (d.pa_on) {  // AudioParamProcessor::mix_to_output on the way out (OP_PARAM_ADD of the chain kernel, the same arithmetic)
    const uint32_t q = (uint32_t)(f0 / RQ), qc = q < d.n_quanta ? q : d.n_quanta - 1;
    const float iv = d.pa_intrinsic.mode == 0 ? d.pa_intrinsic.base[inst] : d.pa_intrinsic.base[(uint64_t)inst * d.pa_intrinsic.stride + qc];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float o1 = r[e] + iv;
      r[e] = o1 != o1 ? d.pa_default : fminf(fmaxf(o1, d.pa_min), d.pa_max);
    }
  }
  *reinterpret_cast<float4*>(out + f0) = make_float4(r[0], r[1], r[2], r[3]);
  if (d.post_dup) *reinterpret_cast<float4*>(out + d.out.ch_stride + f0) = make_float4(r[0], r[1], r[2], r[3]);
}
}  // namespace

__global__ __launch_bounds__(64) void osc_kernel(const OscDesc d) {
  const uint32_t inst = blockIdx.x * 64 + threadIdx.x;
  if (inst >= d.n_inst) return;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride;
  double start_time = d.start[inst];
  const double stop_time = d.stop[inst];
  const double sample_rate = d.sample_rate, dt = 1. / sample_rate, nyquist = sample_rate / 2.;
  double phase = 0.;
  bool started = false;
  const bool per_frame = d.frequency.mode == 2 || d.detune.mode == 2;
  for (uint32_t q = 0; q < d.n_quanta; q++) {
    const uint64_t f0 = (uint64_t)q * RQ;
    const double block_time = (double)f0 / sample_rate;
    const double next_block_time = block_time + dt * (double)RQ;
    float4* o4 = reinterpret_cast<float4*>(out + f0);
    if (stop_time <= block_time || start_time >= next_block_time) {  // :349-375
      for (int j = 0; j < RQ / 4; j++) o4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    double current_time = block_time;
    if (!started && start_time < current_time) start_time = current_time;  // :386-393
    // the per-sample body of generate_sample, :505-553
    auto generate = [&](bool outside_nyquist, double phase_incr) __attribute__((always_inline)) -> float {
      float v = 0.f;
      if (!(current_time < start_time || current_time >= stop_time)) {
        if (!started) {
          if (current_time > start_time) {
            const double ratio = (current_time - start_time) / dt;
            phase = outside_nyquist ? unroll_phase_unbounded(phase_incr * ratio) : unroll_phase(phase_incr * ratio);
          }
          started = true;
        }
        v = outside_nyquist ? 0.f : waveform_sample(d, phase, phase_incr);
        phase = outside_nyquist ? unroll_phase_unbounded(phase + phase_incr) : unroll_phase(phase + phase_incr);
      }
      current_time += dt;
      return v;
    };
    if (!per_frame) {
      const float freq = d.frequency.mode == 0 ? d.frequency.base[inst] : d.frequency.base[(uint64_t)inst * d.frequency.stride + q];
      const float detune = d.detune.mode == 0 ? d.detune.base[inst] : d.detune.base[(uint64_t)inst * d.detune.stride + q];
      const double computed_freq = (double)freq * exp2((double)detune / 1200.);  // :30-32
      const double phase_incr = computed_freq / sample_rate;
      const bool outside_nyquist = fabs(computed_freq) >= nyquist;
      const bool fully_active = started && start_time <= block_time && stop_time >= next_block_time;
      if (fully_active && !outside_nyquist) {
        for (int j = 0; j < RQ / 4; j++) {
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            r[e] = waveform_sample(d, phase, phase_incr);
            phase = unroll_phase(phase + phase_incr);
          }
          o4[j] = make_float4(r[0], r[1], r[2], r[3]);
        }
      } else {
        for (int j = 0; j < RQ / 4; j++) {
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; e++) r[e] = generate(outside_nyquist, phase_incr);
          o4[j] = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    } else {
      for (int j = 0; j < RQ / 4; j++) {
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint64_t f = f0 + j * 4 + e;
          const float freq = d.frequency.mode == 0   ? d.frequency.base[inst]
                             : d.frequency.mode == 1 ? d.frequency.base[(uint64_t)inst * d.frequency.stride + q]
                                                     : d.frequency.base[(uint64_t)inst * d.frequency.stride + f];
          const float detune = d.detune.mode == 0   ? d.detune.base[inst]
                               : d.detune.mode == 1 ? d.detune.base[(uint64_t)inst * d.detune.stride + q]
                                                    : d.detune.base[(uint64_t)inst * d.detune.stride + f];
          const double computed_freq = (double)freq * exp2((double)detune / 1200.);
          const double phase_incr = computed_freq / sample_rate;
          r[e] = generate(fabs(computed_freq) >= nyquist, phase_incr);
        }
        o4[j] = make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  }
  // frames past the last quantum of the padded signal
  for (uint64_t f = (uint64_t)d.n_quanta * RQ; f < d.frames; f += 4) *reinterpret_cast<float4*>(out + f) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Time-parallel variant for host-known frequencies (constants or one value per quantum, no graph modulation):
// the host replays the scheduling decisions of the reference per quantum (start/stop inside a quantum, sub-sample
// start phase, Nyquist muting) and hands over the phase at the first active frame of every quantum; inside the
// quantum the phase is phase + i * incr in closed form instead of i rounded additions.  The two differ by the
// accumulated rounding of the reference's running sum (~1e-14 after 10 s), far below one f32 ulp of the output;
// the waveform arithmetic is the reference's.  HBM-bound: 4 B written per frame.
__global__ __launch_bounds__(256) void osc_par_kernel(const OscDesc d) {
  const uint32_t inst = blockIdx.y;
  const uint64_t f0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (f0 >= d.frames) return;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride;
  const uint32_t q = (uint32_t)(f0 / RQ);
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (q < d.n_quanta) {
    const OscQuantum oq = d.table_q[(uint64_t)d.tq_row[inst] * d.n_quanta + q];
    const int i0 = (int)(f0 % RQ);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int i = i0 + e;
      if (i >= oq.first && i < oq.end && !oq.outside_nyquist) {
        const double x = __builtin_fma((double)(i - oq.first), oq.incr, oq.phase);
        double ph = x - floor(x);
        if (ph >= 1.) ph -= 1.;
        r[e] = waveform_sample(d, ph, oq.incr);
      }
    }
  }
  osc_store(d, inst, out, f0, r);
}

// a-rate / graph-modulated frequency (FM): the phase is the running sum of per-frame increments.  One wavefront per
// instance walks 256-frame groups: every lane owns 4 consecutive frames (coalesced 16 B accesses), sums its four
// increments, a 6-step wavefront scan gives every lane the sum of everything before it in the group, and the group
// total is carried on (mod 1).  The order of the f64 additions differs from the reference's frame-by-frame sum —
// same ~1e-14 phase difference as the closed form above, far below one f32 ulp of the output; the per-frame
// decisions (active range from the host replay of the reference's clock, Nyquist muting, sub-sample start phase)
// and the waveform arithmetic are the reference's.
// The a-rate oscillator in TIME SEGMENTS: one wavefront per (instance, segment) instead of one per instance — with 1024
// instances that was one wave per SIMD walking 1875 groups of 256 frames one after the other, every exposed latency
// (the frequency signal, the f64 arithmetic, the shuffles of the scan) paid in full.  PASS 0 computes the phase advance
// of every segment (the same increments, no waveform), the render pass starts each segment at the running sum of the
// segments before it.  The phase is a sum either way (the group scan was already a tree): rounding differences of the
// order of 1e-16 in the phase, as between the group-serial form and the reference's frame-serial accumulation.
template <int PASS>
__global__ __launch_bounds__(64) void osc_scan_kernel(const OscDesc d) {
  const uint32_t inst = blockIdx.x;
  const uint32_t seg = blockIdx.y;
  const int lane = threadIdx.x;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride;
  const int64_t first = d.active[(uint64_t)inst * 2], end = d.active[(uint64_t)inst * 2 + 1];
  const double ratio = d.start_ratio[inst];  // (time of frame `first` - start_time) / dt, 0 when it starts on a frame
  const double sample_rate = d.sample_rate, nyquist = sample_rate / 2.;
  const uint64_t groups = (d.frames + 255) / 256, per_seg = (groups + OSC_SEGMENTS - 1) / OSC_SEGMENTS;
  const uint64_t g_begin = (uint64_t)seg * per_seg * 256, g_end = g_begin + per_seg * 256 < d.frames ? g_begin + per_seg * 256 : d.frames;
  double carry = 0.;  // phase at the first frame of the group
  if (PASS == 1) {
    for (uint32_t sgm = 0; sgm < seg; sgm++) {
      carry += d.seg_phase[(uint64_t)inst * OSC_SEGMENTS + sgm];
      carry -= floor(carry);
    }
  }
  // a constant detune: its factor once (the f64 exp2 per frame was most of the group's arithmetic)
  const bool det_const = d.detune.mode == 0;
  const double det_mul = det_const ? exp2((double)d.detune.base[inst] / 1200.) : 1.;
  for (uint64_t g0 = g_begin; g0 < g_end; g0 += 256) {
    const uint64_t f0 = g0 + (uint64_t)lane * 4;
    double incr[4];
    bool outside[4];
    OscQuantum mq4{};
    if (d.fm_q) {
      const uint64_t fq = f0 < (uint64_t)d.n_quanta * RQ ? f0 : (uint64_t)d.n_quanta * RQ - 1;
      mq4 = d.fm_q[(uint64_t)d.fm_row[inst] * d.n_quanta + fq / RQ];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint64_t f = f0 + e;
      const uint64_t fc = f < (uint64_t)d.n_quanta * RQ ? f : (uint64_t)d.n_quanta * RQ - 1;
      const uint32_t q = (uint32_t)(fc / RQ);
      float freq = d.frequency.mode == 0   ? d.frequency.base[inst]
                   : d.frequency.mode == 1 ? d.frequency.base[(uint64_t)inst * d.frequency.stride + q]
                                           : d.frequency.base[(uint64_t)inst * d.frequency.stride + fc];
      if (d.fm_q) {
        // the folded modulator (osc_par_kernel's arithmetic for frame fc), the edge gain (gain.rs:163-179), then mix_to_output
        const OscQuantum& mq = mq4;  // (the four frames of a lane lie in one render quantum)
        const int i = (int)(fc % RQ);
        float m = 0.f;
        if (i >= mq.first && i < mq.end && !mq.outside_nyquist) {
          const double x = __builtin_fma((double)(i - mq.first), mq.incr, mq.phase);
          double ph = x - floor(x);
          if (ph >= 1.) ph -= 1.;
          m = waveform_sample(d.fm_type, d.fm_table, d.fm_table_len, ph, mq.incr);
        }
        if (d.fm_has_gain) {
          const float g = d.fm_gain.mode == 0 ? d.fm_gain.base[inst] : d.fm_gain.base[(uint64_t)inst * d.fm_gain.stride + q];
          m = fabsf(g) <= 1e-6f ? 0.f : (fabsf(1.f - g) <= 1e-6f ? m : m * g);
        }
        float o1 = m + freq;
        freq = o1 != o1 ? d.fm_default : fminf(fmaxf(o1, d.fm_min), d.fm_max);
      }
      double computed_freq;
      if (det_const) {
        computed_freq = (double)freq * det_mul;
      } else {
        const float detune = d.detune.mode == 1 ? d.detune.base[(uint64_t)inst * d.detune.stride + q]
                                                : d.detune.base[(uint64_t)inst * d.detune.stride + fc];
        computed_freq = (double)freq * exp2((double)detune / 1200.);
      }
      incr[e] = computed_freq / sample_rate;
      outside[e] = fabs(computed_freq) >= nyquist;
    }
    // contribution of each frame to the phase of LATER frames: inactive frames do not advance the phase
    double adv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int64_t f = (int64_t)(f0 + e);
      adv[e] = (f >= first && f < end) ? incr[e] : 0.;
      if (f == first) adv[e] += incr[e] * ratio;  // sub-sample start: the first frame starts at incr * ratio (:516-528)
    }
    const double local = ((adv[0] + adv[1]) + adv[2]) + adv[3];
    double incl = local;  // inclusive scan over lanes
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const double t = __shfl_up(incl, sft, 64);
      if (lane >= sft) incl += t;
    }
    if (PASS == 1) {
      double ph = carry + (incl - local);  // phase before this lane's first frame (may exceed 1: reduced per frame)
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int64_t f = (int64_t)(f0 + e);
        double p = ph;
        if (f == first) p += incr[e] * ratio;
        p -= floor(p);
        if (p >= 1.) p -= 1.;
        r[e] = (f >= first && f < end && !outside[e]) ? waveform_sample(d, p, incr[e]) : 0.f;
        ph += adv[e];
      }
      osc_store(d, inst, out, f0, r);
    }
    double total = __shfl(incl, 63, 64) + carry;
    total -= floor(total);
    carry = total;
  }
  if (PASS == 0 && lane == 0) d.seg_phase[(uint64_t)inst * OSC_SEGMENTS + seg] = carry;
}

void launch_osc(const OscDesc& d, void* stream) {
  if (d.table_q) {
    hipLaunchKernelGGL(osc_par_kernel, dim3((unsigned)((d.frames + 1023) / 1024), d.n_inst), dim3(256), 0, (hipStream_t)stream, d);
    return;
  }
  if (d.active) {
    hipLaunchKernelGGL(osc_scan_kernel<0>, dim3(d.n_inst, OSC_SEGMENTS), dim3(64), 0, (hipStream_t)stream, d);
    hipLaunchKernelGGL(osc_scan_kernel<1>, dim3(d.n_inst, OSC_SEGMENTS), dim3(64), 0, (hipStream_t)stream, d);
    return;
  }
  hipLaunchKernelGGL(osc_kernel, dim3((d.n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
}

}  // namespace waa
