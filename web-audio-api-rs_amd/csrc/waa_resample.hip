// waa_resample.hip — AudioBufferSourceNode [-> WaveShaper] -> signal, the BASELINE config C5 shape
// (audio_buffer_source.rs:422-845 slow track + waveshaper.rs:555-573), without the op interpreter of chain_kernel.
//
// One wavefront renders one 256-frame sub-tile (two render quanta, 4 frames per lane) for GROUP consecutive
// instances.  What made the interpreter latency- and L1-bound (profiles/r01_c5_sq_counters.txt: 80 % of a wave's
// life in s_waitcnt, the vector L1 at 0.78 accesses per CU-cycle from 16 four-byte gathers per lane and instance):
//   * the per-frame playback records (prev, next, k — 16 B per frame, produced by the host's replay of the playhead
//     state machine and shared by every instance with the same schedule) are loaded ONCE per wave and reused for the
//     whole instance group; fast-track and silent quanta are turned into the same record form (k = 0 / prev = -1), so
//     there is one code path;
//   * the samples a sub-tile needs form one contiguous window of the AudioBuffer (256 * rate frames); the wave
//     fetches it with coalesced 16-byte loads into LDS and the lanes pick prev / next from there: a dozen L1 line
//     accesses per channel instead of ~300.  Sub-tiles whose window is too wide (a loop wrap inside the sub-tile,
//     extreme rates) fall back to per-lane gathers;
//   * the window of instance g + 1 is requested before instance g is interpolated (software pipeline across the
//     group), so the dependent chain record -> address -> sample is paid once per wave, not once per instance.
// The WaveShaper curve (<= 8192 points) is staged in LDS once per workgroup.  Arithmetic: exactly the reference's
// (f64 (1 - k).mul_add(prev, k * next) -> f32; curve lookup in f32, unfused).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"

namespace waa {

namespace {
constexpr int GROUP = 8;        // instances per wave
constexpr int WCAP = 512;       // window capacity in frames per channel (256 output frames at rates up to ~1.9)
constexpr int WAVES = 4;        // waves per workgroup

__device__ __forceinline__ float curve_lds(const float* curve, int nn, float input) {  // waveshaper.rs:555-573
  if (nn == 0) return 0.f;
  const float n = (float)nn;
  const float v = (n - 1.f) / 2.0f * (input + 1.f);
  if (v <= 0.f) return curve[0];
  if (v >= n - 1.f) return curve[nn - 1];
  const float k = floorf(v);
  const float f = v - k;
  const int ki = (int)k;
  return (1.f - f) * curve[ki] + f * curve[ki + 1];
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(v, d, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}
}  // namespace

// RECLOAD = true: the round-2 form of the window pipeline (WAA_RESAMPLE_RECORD_LOAD=1, same-box A/B), see `fetch` below
template <int C, bool RECLOAD = false>
__global__ __launch_bounds__(WAVES * 64) void resample_kernel(const ChainDesc d, int curve_op) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nn = curve_op >= 0 ? d.ops[curve_op].i0 : 0;
  float* curve = lds;                                      // [nn]
  const int curve_pad = (nn + 3) & ~3;
  const int wv = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  float* win = lds + curve_pad + wv * (C * WCAP);          // this wave's window, [C][WCAP]
  if (curve_op >= 0) {
    const float* src = reinterpret_cast<const float*>(d.ops[curve_op].ptr0);
    for (int i = threadIdx.x; i < nn; i += WAVES * 64) curve[i] = src[i];
  }
  __syncthreads();
  const uint32_t n_sub = (d.tile1 - d.tile0) * (TILE / 256);
  const uint32_t n_groups = (d.n_inst + GROUP - 1) / GROUP;
  const uint64_t wid = (uint64_t)blockIdx.x * WAVES + (uint32_t)__builtin_amdgcn_readfirstlane(wv);
  // neighbouring waves render the same sub-tile of different instance groups: the shared records stay in L2
  const uint32_t grp = (uint32_t)(wid % n_groups);
  const uint32_t sub = d.tile0 * (TILE / 256) + (uint32_t)(wid / n_groups);
  if (wid / n_groups >= n_sub) return;
  const InputRef& in = d.in[0];
  const uint32_t inst0 = grp * GROUP;
  const uint32_t n_here = d.n_inst - inst0 < (uint32_t)GROUP ? d.n_inst - inst0 : (uint32_t)GROUP;
  const uint32_t q = sub * 2 + (lane >> 5);                // this lane's render quantum
  const uint32_t i0 = (lane & 31) * 4;                     // its first frame within the quantum
  const uint64_t f_out = (uint64_t)sub * 256 + lane * 4;   // its first output frame
  auto wave_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };

  int32_t rp[4], rn[4];   // prev / next buffer index per frame (-1: none, next == -2: extrapolate from prev - 1)
  double rk[4];
  int wlo = 0, whi = -1;  // window of buffer frames [wlo, whi] needed by the wave, wlo aligned down to 4 frames
  // ---- records of one schedule -> (rp, rn, rk) and the window
  auto load_records = [&](const SrcInst& si) __attribute__((always_inline)) {
    const SrcSchedule sc = si.sc;
    const bool valid_q = q < d.n_quanta;
    const QRec r = load_global(sc.qrec + (valid_q ? q : 0));
    const uint32_t mode = valid_q ? r.mode : (uint32_t)Q_SILENT;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      rp[e] = -1;
      rn[e] = -1;
      rk[e] = 0.;
    }
    if (mode == Q_SLOW) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const SlowRec s = load_global(sc.slow + (uint64_t)q * RQ + i0 + e);
        rp[e] = s.prev;
        rn[e] = s.next;
        rk[e] = s.k;
      }
    } else if (mode == Q_FAST || mode == Q_FAST_LOOP) {
      // audio_buffer_source.rs:562-607 as records: index start + i, nothing past the end, wrap when looping; k = 0
      // makes the interpolation formula return the sample itself (fma(1, prev, 0 * 0))
#pragma unroll
      for (int e = 0; e < 4; e++) {
        uint64_t bi = (uint64_t)r.start + i0 + e;
        bool ok = true;
        if (bi >= si.frames) {
          if (mode == Q_FAST_LOOP)
            bi = bi % si.frames;
          else
            ok = false;
        }
        rp[e] = ok ? (int32_t)bi : -1;
      }
    }
    int lo = 0x7fffffff, hi = -1;
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (rp[e] >= 0) {
        const int a = rn[e] == -2 ? rp[e] - 1 : rp[e];
        const int b2 = rn[e] >= 0 ? rn[e] : rp[e];
        const int mn = a < b2 ? a : b2, mx = a > b2 ? a : b2;
        lo = mn < lo ? mn : lo;
        hi = mx > hi ? mx : hi;
        hi = rp[e] > hi ? rp[e] : hi;
      }
    lo = wave_min(lo);
    hi = wave_max(hi);
    wlo = hi >= 0 ? (lo & ~3) : 0;
    whi = hi;
  };
  // ---- interpolation (audio_buffer_source.rs:754-822) with samples fetched by `at(c, index)`, curve, store
  auto finish = [&](uint32_t inst, auto&& at) __attribute__((always_inline)) {
    float v[C][4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
      for (int c = 0; c < C; c++) {
        float o = 0.f;
        if (rp[e] >= 0) {
          const double prev_sample = (double)at(c, rp[e]);
          double next_sample;
          if (rn[e] >= 0)
            next_sample = (double)at(c, rn[e]);
          else if (rn[e] == -1)
            next_sample = 0.;
          else
            next_sample = 2. * prev_sample - (double)at(c, rp[e] - 1);
          o = (float)__builtin_fma(1. - rk[e], prev_sample, rk[e] * next_sample);
        }
        v[c][e] = o;
      }
    }
    if (curve_op >= 0) {
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) v[c][e] = curve_lds(curve, nn, v[c][e]);
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
      float* p = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride + f_out;
      *reinterpret_cast<float4*>(p) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    }
  };

  // Do all instances of the group share one schedule and have aligned buffers?  (uniform: scalar loads)
  const SrcInst si0 = in.src[inst0];
  bool uniform = si0.aligned != 0;
  for (uint32_t g = 1; g < n_here; g++) {
    const SrcInst& sg = in.src[inst0 + g];
    uniform = uniform && sg.sched == si0.sched && sg.aligned && sg.frames == si0.frames;
  }
  load_records(si0);
  const int span = whi - wlo + 1;
  if (uniform && whi >= 0 && span <= WCAP) {
    // ---- pipelined window path: the window of instance g + 1 is in flight while instance g is interpolated
    constexpr int NV = WCAP / 256;
    f4v nx[C][NV];
    if constexpr (RECLOAD) {
      auto fetch = [&](uint32_t g) __attribute__((always_inline)) {
        const SrcInst& sg = in.src[inst0 + g];
        const float* base = sg.base + wlo;
        const uint64_t cs = sg.ch_stride;
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
          for (int j = 0; j < NV; j++) {
            const int o = j * 256 + lane * 4;
            f4v t = {0.f, 0.f, 0.f, 0.f};
            if (o < span) {
              const float* pch = base + (uint64_t)c * cs + o;
              if ((uint64_t)(wlo + o + 3) < si0.frames) {
                t = load_global_f4(pch);
              } else {  // the buffer ends inside this vector
                t.x = (uint64_t)(wlo + o) < si0.frames ? load_global(pch) : 0.f;
                t.y = (uint64_t)(wlo + o + 1) < si0.frames ? load_global(pch + 1) : 0.f;
                t.z = (uint64_t)(wlo + o + 2) < si0.frames ? load_global(pch + 2) : 0.f;
              }
            }
            nx[c][j] = t;
          }
      };
      fetch(0);
#pragma unroll 1
      for (uint32_t g = 0; g < n_here; g++) {
        wave_sync();  // the previous instance's LDS reads are done
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
          for (int j = 0; j < NV; j++) {
            const int o = j * 256 + lane * 4;
            if (o < span) *reinterpret_cast<f4v*>(win + c * WCAP + o) = nx[c][j];
          }
        if (g + 1 < n_here) fetch(g + 1);
        wave_sync();
        finish(inst0 + g, [&](int c, int idx) __attribute__((always_inline)) { return win[c * WCAP + idx - wlo]; });
      }
      return;
    }
    // Round 3 (tools/isa_waits.py, profiles/r03z_c5_sq1.txt: 72 % of the kernel's wave-cycles were waits).  The form above
    // (a) reads the next instance's buffer base / channel stride from its record inside the loop: a dependent load in front
    // of every window request, (b) requests the window under conditions (inside the span? does the buffer end inside the
    // vector?), so the compiler cannot count what is in flight and every wait is vmcnt(0) — which, vmcnt being ONE in-order
    // counter, also waits for the stores of the instance just finished.  Here lane g fetches instance g's base and stride
    // once (readlane picks them), every window vector is loaded unconditionally from a clamped offset (aligned rows: a
    // vector that straddles the buffer's end stays inside the row's padding) and looked at only when it is written to LDS,
    // where the frames past the end are zeroed.
    const uint64_t* rec = reinterpret_cast<const uint64_t*>(in.src + inst0 + ((uint32_t)lane < n_here ? (uint32_t)lane : 0u));
    const uint64_t my_base = load_global(rec);      // SrcInst::base
    const uint64_t my_cs = load_global(rec + 1);    // SrcInst::ch_stride
    auto fetch = [&](uint32_t g) __attribute__((always_inline)) {
      const int gl = (int)__builtin_amdgcn_readfirstlane((int)g);
      const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_base >> 32), gl) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_base, gl);
      const uint64_t cs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_cs >> 32), gl) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_cs, gl);
      const float* base = reinterpret_cast<const float*>(b) + wlo;
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int j = 0; j < NV; j++) {
          const int o = j * 256 + lane * 4;
          nx[c][j] = load_global_f4(base + (uint64_t)c * cs + (o < span ? o : 0));
        }
    };
    fetch(0);
#pragma unroll 1
    for (uint32_t g = 0; g < n_here; g++) {
      wave_sync();  // the previous instance's LDS reads are done
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int j = 0; j < NV; j++) {
          const int o = j * 256 + lane * 4;
          if (o < span) {
            const uint64_t a = (uint64_t)(wlo + o);  // (a < frames: the window ends at a frame some record names)
            f4v t = nx[c][j];
            t.y = a + 1 < si0.frames ? t.y : 0.f;
            t.z = a + 2 < si0.frames ? t.z : 0.f;
            t.w = a + 3 < si0.frames ? t.w : 0.f;
            *reinterpret_cast<f4v*>(win + c * WCAP + o) = t;
          }
        }
      fetch(g + 1 < n_here ? g + 1 : g);  // (past the group: the same window again, never used — the count stays fixed)
      wave_sync();
      finish(inst0 + g, [&](int c, int idx) __attribute__((always_inline)) { return win[c * WCAP + idx - wlo]; });
    }
    return;
  }
  // ---- general path: per-lane gathers, records reloaded when the schedule changes (loop wrap inside the sub-tile,
  // very high rates, unaligned buffers, mixed schedules, silent sub-tiles)
  uint32_t sched_prev = si0.sched;
#pragma unroll 1
  for (uint32_t g = 0; g < n_here; g++) {
    const SrcInst si = in.src[inst0 + g];
    if (si.sched != sched_prev) {
      load_records(si);
      sched_prev = si.sched;
    }
    const float* base = si.base;
    const uint64_t cs = si.ch_stride;
    finish(inst0 + g, [&](int c, int idx) __attribute__((always_inline)) { return load_global(base + (uint64_t)c * cs + idx); });
  }
}

// Does this chain have the shape the kernel covers?  One AudioBufferSource input without an edge gain, at most one op
// (a WaveShaper whose curve fits in LDS), no channel-count change anywhere.
bool resample_shape(const ChainDesc& d, int* curve_op) {
  if (d.n_inputs != 1 || d.in[0].kind != IN_SOURCE || d.in[0].has_gain) return false;
  if (d.in[0].nch != d.in_nch || d.in_nch != d.out.nch || d.in_nch > 2) return false;
  if (d.n_ops > 1) return false;
  *curve_op = -1;
  if (d.n_ops == 1) {
    const OpDesc& o = d.ops[0];
    if (o.kind != OP_WAVESHAPER || o.i0 <= 0 || o.i0 > 8192 || o.nch_in != d.in_nch) return false;
    *curve_op = 0;
  }
  return measure_switch("WAA_NO_RESAMPLE_KERNEL") == nullptr;  // (switch: A/B against the interpreter)
}
void launch_resample(const ChainDesc& d, int curve_op, void* stream) {
  const int nn = curve_op >= 0 ? d.ops[curve_op].i0 : 0;
  const int C = d.in_nch;
  const size_t lds = ((size_t)((nn + 3) & ~3) + (size_t)WAVES * C * WCAP) * sizeof(float);
  const uint64_t n_sub = (uint64_t)(d.tile1 - d.tile0) * (TILE / 256);
  const uint64_t waves = n_sub * ((d.n_inst + GROUP - 1) / GROUP);
  const dim3 grid((unsigned)((waves + WAVES - 1) / WAVES)), block(WAVES * 64);
  if (measure_switch("WAA_RESAMPLE_RECORD_LOAD")) {  // (measurement switch: the round-2 window pipeline)
    if (C == 1)
      hipLaunchKernelGGL((resample_kernel<1, true>), grid, block, lds, (hipStream_t)stream, d, curve_op);
    else
      hipLaunchKernelGGL((resample_kernel<2, true>), grid, block, lds, (hipStream_t)stream, d, curve_op);
    return;
  }
  if (C == 1)
    hipLaunchKernelGGL((resample_kernel<1>), grid, block, lds, (hipStream_t)stream, d, curve_op);
  else
    hipLaunchKernelGGL((resample_kernel<2>), grid, block, lds, (hipStream_t)stream, d, curve_op);
}

}  // namespace waa
