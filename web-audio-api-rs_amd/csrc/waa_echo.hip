// waa_echo.hip — the echo loop  line = inputs + g * delayed(line)  with its delay line in LDS.
//
// A block-scheduled feedback loop whose body is ONE element-wise launch per block (DelayNode <-> GainNode, the delay read from
// its line by the summing stage: DESIGN.md 3.1d) costs, per block, a read of the source, a read of the delayed line and a write
// of the line — and 47 launches of 250 MB for a 10 s render.  Here one workgroup per instance walks the render in chunks and
// keeps the last 16384 frames of the line in LDS: the delayed read never goes to memory, the line is written once for its
// consumers outside the loop (or not at all: "the tail" below).  Same arithmetic in the same order as chain_kernel's input stage
// (waa_kernels.hip: load -> edge gain with gain.rs' mute / pass-through cases -> up-mix -> sum in edge order; the delayed
// sample is fma(1 - k, x[i], k * x[i + 1]), delay.rs:560-590): bit-identical, which tests/test_cycles.py asserts.
// Qualification: echo_ring_applicable() below; everything else keeps the launch-per-block form.
//
// The tail.  The usual echo graph sends  dry + delayed(line)  to the destination: as a launch of its own that stage reads the
// source and the line again and writes the output (3 x 3.9 GB of the fb workload's 19.7 GB).  When the line has exactly one
// reader outside the loop and that reader is such a sum of the delayed line and signals the loop reads anyway
// (echo_tail_applicable), this kernel renders it from the values it already holds — the delayed samples out of the ring, the
// source out of registers — and the line itself is never written to memory: the loop costs its compulsory traffic, the source
// read once and the output written once.
//
// Nothing fed back.  out = X + g * delayed(X) outside any loop (echo_feed_forward) is the same walk with the stand-in loop
// stage line = X: X goes through the ring instead of being read twice by two million short-lived wavefronts of the
// tile-parallel chain kernel (1.5 against 2.1 ms on the echo workload; taken when there is an instance per CU to walk).
//
// A Biquad in the loop (round 4).  The classic filtered echo  line = X + g * biquad(delayed(line))  was three launches per block
// (the delayed read, the streaming biquad, the sum) with the delayed samples and the filter's output going through memory in
// between.  The BQ form of the kernel filters the delayed samples on their way from the ring into the sum: per chunk every
// lane has 4 consecutive frames of every channel; zero-state response of those frames (f64), the lanes' incoming y states
// from a scan over the wavefront with the uniform powers of A = M^4 (M the one-frame transition), the waves' incoming states
// from the chunk's starting state and the waves' zero-state end states (one LDS hand-over), then the reference's evaluation
// order from the true incoming state (biquad_filter.rs:877-883) — the scheme of waa_biquad_stream.hip on the ring kernel's data
// layout.  Constant coefficients only (one set per instance); anything else keeps the three launches.  The filter's output takes
// the place of the delayed line as the operand of the loop stage and of the tail; it is stored only for other readers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "waa_internal.hpp"

namespace waa {

void raise_lds_limit(const void* kernel) {
  static std::mutex lock;
  static std::set<std::pair<int, const void*>> raised;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(lock);
  if (!raised.insert({dev, kernel}).second) return;
  // the limit covers DYNAMIC memory only; a kernel that also has static __shared__ arrays (dyn_kernel: 336 bytes) must ask
  // for 160 KB minus those — asking for the full 160 KB fails, silently until the launch that needed it is refused
  // ("invalid argument": fuzz seed 903488 of round 4, a 13-item dynamic group on 4-channel signals = 66.6 KB)
  hipFuncAttributes fa{};
  size_t stat = 0;
  if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) stat = fa.sharedSizeBytes;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - stat)) != hipSuccess) {
    (void)hipGetLastError();  // (the launch that follows reports with the kernel's name)
    raised.erase({dev, kernel});
  }
}

constexpr int ECHO_RING = 16384;  // frames per channel kept in LDS AT MOST (two channels: 128 KB); a launch takes the power of two its delays need
constexpr int ECHO_EXT = 2;        // inputs from outside the loop (held in registers two chunks ahead)
constexpr int ECHO_TAIL_IN = 3;    // inputs of the fused tail stage: the delayed line and those

namespace {
__device__ __forceinline__ float echo_delay_value(const ParamRef& p, uint32_t inst) {
  return p.mode == 3 ? __uint_as_float((uint32_t)p.stride) : load_global(p.base + inst);
}

// One chunk's operands from outside the loop, requested two chunks before they are used: the samples of the (at most
// ECHO_EXT) inputs that are not the feedback input — slot = their order among those — and one edge gain per input of the
// loop stage and of the tail stage.  RAW values: nothing may look at them before the chunk that uses them (a select on a
// loaded value is a wait for the load, the prefetch would be none); every load is unconditional, so that the compiler can
// count what is in flight behind it (vmcnt is in order: an unknown number of younger accesses turns every wait into "all").
template <int C, int NG>
struct EchoOperands {
  float x[ECHO_EXT][C][4];
  float g[NG];
};

// C: channels of the line (the ring); CT: channels of the fused tail stage (0: none); STORE: the line goes to memory too.
// Everything that does not change from chunk to chunk — pointers, which operand an edge takes, whether it has a gain or
// up-mixes — is worked out ONCE, into wave-uniform registers, before the walk: read from the descriptors inside the chunk it
// was 1500 instructions per chunk and wave (a third of them scalar-register spills), as long as the chunk's memory time.
struct M2e {
  double a, b, c, d;
};
__device__ __forceinline__ M2e mme(const M2e& x, const M2e& y) {
  M2e r;
  r.a = __builtin_fma(x.a, y.a, x.b * y.c);
  r.b = __builtin_fma(x.a, y.b, x.b * y.d);
  r.c = __builtin_fma(x.c, y.a, x.d * y.c);
  r.d = __builtin_fma(x.c, y.b, x.d * y.d);
  return r;
}
__device__ __forceinline__ double uniform_d(double v) {  // a wave-uniform double into scalar registers
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
constexpr int ECHO_BQ_WAVES = 8;  // the BQ form: at most eight waves (256 registers each)

template <int C, int CT, bool STORE, bool BQ = false>
__global__ __launch_bounds__(BQ ? ECHO_BQ_WAVES * 64 : 1024) void echo_ring_kernel(const ChainDesc d, int fb, int chunk_subtiles, const EchoTail t,
                                                                                 const EchoBq bq) {
  constexpr int CM = C > CT ? C : CT;
  constexpr int NL = 1 + ECHO_EXT, NT = CT > 0 ? ECHO_TAIL_IN : 0, NG = NL + NT;
  extern __shared__ __attribute__((aligned(16))) float ring[];  // [C][RF]
  const int RF = __builtin_amdgcn_readfirstlane(t.ring_frames), RM = RF - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t inst = blockIdx.x;
  for (int i = tid; i < C * RF; i += blockDim.x) ring[i] = 0.f;
  __syncthreads();
  // ---- BQ: the filter's constants (one coefficient set per instance) and what the scan needs of them
  __shared__ double wz[BQ ? ECHO_BQ_WAVES : 1][C][2];  // zero-state end states (y1, y2) of the waves' sub-tiles of this chunk
  [[maybe_unused]] double b0 = 0., b1 = 0., b2 = 0., a1 = 0., a2 = 0.;
  [[maybe_unused]] M2e P[6], A64{}, PL{};   // A^(2^k), A^64 (A = M^4) — uniform; A^lane — per lane
  __shared__ double ystate[2][C][2];        // y state (y1, y2) in front of the chunk at hand, by chunk parity (the last wave writes the next one's)
  [[maybe_unused]] uint32_t parity = 0;
  [[maybe_unused]] float* py[C];
  if constexpr (BQ) {
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (FTZ/DAZ render scope, thread.rs:374-382)
    const double* cf = bq.coefs + (uint64_t)inst * bq.coef_stride;
    b0 = uniform_d(load_global(cf));
    b1 = uniform_d(load_global(cf + 1));
    b2 = uniform_d(load_global(cf + 2));
    a1 = uniform_d(load_global(cf + 3));
    a2 = uniform_d(load_global(cf + 4));
    M2e m = {-a1, -a2, 1., 0.};
    m = mme(m, m);
    m = mme(m, m);  // A = M^4
    PL = M2e{1., 0., 0., 1.};
#pragma unroll
    for (int k = 0; k < 6; k++) {
      P[k] = m;
      if (lane & (1 << k)) PL = mme(m, PL);
      m = mme(m, m);
    }
    // (a wavefront's sub-tile is 64 lanes = 256 frames, or 32 lanes = 128 frames with half-wave chunks: the power that carries a
    // state across ONE sub-tile — A^32 there; with A^64 a filter whose memory outlasts 128 frames was off by 4e-4, fuzz seed 661385)
    A64 = t.sub_frames == 128 ? P[5] : m;
#pragma unroll
    for (int c = 0; c < C; c++) py[c] = bq.y.base + (uint64_t)inst * bq.y.inst_stride + (uint64_t)c * bq.y.ch_stride;
    if (tid < 2 * C * 2) (&ystate[0][0][0])[tid] = 0.;
    __syncthreads();
  }
  // DelayReader's position arithmetic (delay.rs:560-569), one delayTime per instance
  const float dv = echo_delay_value(t.delay, inst);
  const double position = 0. - (double)dv * t.sample_rate;
  const double fl = floor(position);
  const int32_t pf0 = (int32_t)fl;  // (-RF < pf0 < 0: echo_ring_applicable / echo_ring_frames)
  const float kf = (float)(position - fl);
  // frames per wavefront and chunk: 256 (4 per lane), or 128 for delays below 264 frames (the upper half of the wavefront idles)
  const uint32_t SUBF = (uint32_t)__builtin_amdgcn_readfirstlane(t.sub_frames);
  const bool lane_on = (uint32_t)lane * 4u < SUBF;
  const int last_lane = (int)(SUBF / 4u) - 1;
  const uint32_t total_sub = (d.tile1 - d.tile0) * ((uint32_t)TILE / SUBF);
  const uint32_t f_first = d.tile0 * TILE;  // (frames fit 31 bits: echo_ring_applicable)
  const uint32_t cs = (uint32_t)chunk_subtiles;
  const uint32_t last_q = d.n_quanta - 1;

  // ---- the walk's constants
  // (BQ: the pointer tables are kept in VECTOR registers — an opaque per-lane zero is added to each: the 8-wave form has 256 of
  // those and the same ~100 scalar registers, which the tables plus the filter's constants overflowed: 130 spill reloads per chunk)
  uint32_t vz32 = 0;
  if constexpr (BQ) asm volatile("v_mov_b32 %0, 0" : "=v"(vz32));
  const uint64_t vz = vz32;
  const float* px[ECHO_EXT][C];  // slot -> channel rows of this instance
  uint32_t vlim[ECHO_EXT];       // ... and the frames that may be read (zeros beyond)
#pragma unroll
  for (int sl = 0; sl < ECHO_EXT; sl++) {
    int k = sl + (sl >= fb ? 1 : 0);                 // (input index of slot sl)
    if (k >= d.n_inputs) k = fb == 0 ? 1 : 0;        // (no such input: the slot re-reads another one, unused)
    const InputRef& in = d.in[k];
#pragma unroll
    for (int c = 0; c < C; c++)
      px[sl][c] = in.sig.base + (uint64_t)inst * in.sig.inst_stride + (uint64_t)(c < in.nch ? c : 0) * in.sig.ch_stride + vz;
    vlim[sl] = in.valid == 0 || in.valid > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)in.valid;
  }
  const float* pg[NG];   // edge -> its gain values (an edge without a gain: a word of its signal, unused)
  uint32_t gstep[NG];    // 1: one value per quantum, 0: one per instance
  // per-edge flags as bit masks (bit k = edge k): is_d / is_a = operand of the edge: the delayed line / slot 0 (neither: slot 1).
  // (as bool arrays every flag was a 64-bit lane mask in two scalar registers: 60 of them, most of them spilled)
  uint32_t m_is_d = 0, m_is_a = 0, m_present = 0, m_has_gain = 0, m_up = 0;
#pragma unroll
  for (int k = 0; k < NG; k++) {
    const bool tail = k >= NL;
    const int kk = tail ? k - NL : k;
    const bool pres = kk < (tail ? t.n_inputs : d.n_inputs);
    const InputRef& in = tail ? t.in[pres ? kk : 0] : d.in[pres ? kk : 0];
    const bool hg = pres && in.has_gain;
    pg[k] = (!hg ? in.sig.base : in.gain.mode == 0 ? in.gain.base + inst : in.gain.base + (uint64_t)inst * in.gain.stride) + vz;
    gstep[k] = (uint32_t)__builtin_amdgcn_readfirstlane(hg && in.gain.mode != 0 ? -1 : 0);  // (a 32-bit mask, not a lane mask)
    m_present |= (pres ? 1u : 0u) << k;
    m_has_gain |= (hg ? 1u : 0u) << k;
    const int slot = tail ? t.alias[kk < MAX_INPUTS ? kk : 0] : (kk == fb ? -2 : kk - (kk > fb ? 1 : 0));
    m_is_d |= (slot < 0 ? 1u : 0u) << k;
    m_is_a |= (slot == 0 ? 1u : 0u) << k;
    m_up |= (in.nch == 1 && (tail ? t.in_nch : d.in_nch) == 2 ? 1u : 0u) << k;
  }
  m_is_d = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_is_d);
  m_is_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_is_a);
  m_present = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_present);
  m_has_gain = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_has_gain);
  m_up = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_up);
  auto bit = [](uint32_t m, int k) { return ((m >> k) & 1u) != 0; };
  const bool l_discrete = d.in_interp == 1, t_discrete = t.in_interp == 1;
  float* po[C];
  float* pt[CT > 0 ? CT : 1];
#pragma unroll
  for (int c = 0; c < C; c++) po[c] = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride;
#pragma unroll
  for (int c = 0; c < CT; c++) pt[c] = t.out.base + (uint64_t)inst * t.out.inst_stride + (uint64_t)c * t.out.ch_stride + vz;

  auto fetch = [&](uint32_t sub_n, EchoOperands<C, NG>& o) __attribute__((always_inline)) {
    const uint32_t fn = f_first + sub_n * SUBF + (uint32_t)lane * 4u;
    const uint32_t qn = fn / RQ;
    const uint32_t qcn = qn < last_q ? qn : last_q;
#pragma unroll
    for (int sl = 0; sl < ECHO_EXT; sl++) {
      const bool ok = sub_n < total_sub && fn + 3u < vlim[sl] && lane_on;
      const uint32_t off = ok ? fn : 0u;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const f4v t4 = load_global_f4(px[sl][c] + off);
        o.x[sl][c][0] = t4.x;
        o.x[sl][c][1] = t4.y;
        o.x[sl][c][2] = t4.z;
        o.x[sl][c][3] = t4.w;
      }
    }
#pragma unroll
    for (int k = 0; k < NG; k++) o.g[k] = load_global(pg[k] + (qcn & gstep[k]));
  };

  // gain.rs:163-179 on an input edge (one value for the quantum), then quantum.rs' up-mix 1 -> 2: copy (speakers) /
  // silence (discrete); the edge's operand picked from the three the chunk has
  // (three separate operand arrays and two flags per edge: ONE array indexed by the edge's selector is put in scratch memory)
  auto edge = [&](int k, float g, bool discrete, const float (&xd)[C][4], const float (&xa)[C][4], const float (&xb)[C][4],
                  float (&u)[CM][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CM; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) u[c][e] = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) u[c][e] = bit(m_is_d, k) ? xd[c][e] : (bit(m_is_a, k) ? xa[c][e] : xb[c][e]);
    if (bit(m_has_gain, k)) {
      const bool mute = fabsf(g) <= 1e-6f, pass = fabsf(1.f - g) <= 1e-6f;
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) u[c][e] = mute ? 0.f : (pass ? u[c][e] : u[c][e] * g);
    }
    if (CM == 2) {
      if (bit(m_up, k)) {
#pragma unroll
        for (int e = 0; e < 4; e++) u[CM - 1][e] = discrete ? 0.f : u[0][e];
      }
    }
  };

  // The sum stage(s) of one group of 4 frames: `xd` = the operand the feedback edge takes (the delayed samples; BQ: the filter's
  // output), the loop stage into the ring (and to memory), the tail stage to memory.
  auto finish = [&](const EchoOperands<C, NG>& o, uint32_t f, const float (&xd)[C][4]) __attribute__((always_inline)) {
    float xa[C][4], xb[C][4];  // the inputs of slot 0 / 1
    {
      const bool in_a = f + 3u < vlim[0], in_b = f + 3u < vlim[1];
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          xa[c][e] = in_a ? o.x[0][c][e] : 0.f;
          xb[c][e] = in_b ? o.x[1][c][e] : 0.f;
        }
    }
    float v[C][4];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      if (!bit(m_present, k)) continue;
      float u[CM][4];
      edge(k, o.g[k], l_discrete, xd, xa, xb, u);
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) v[c][e] = k == 0 ? u[c][e] : v[c][e] + u[c][e];
    }
#pragma unroll
    for (int c = 0; c < C; c++) {  // (C == in_nch == out.nch: echo_ring_applicable)
      if (STORE) *reinterpret_cast<float4*>(po[c] + f) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
      *reinterpret_cast<f4v*>(ring + c * RF + (int)(f & (uint32_t)RM)) = f4v{v[c][0], v[c][1], v[c][2], v[c][3]};
    }
    if (CT > 0) {  // the tail stage: same input arithmetic, its operands the ones the loop stage had
      float w[CM][4];
#pragma unroll
      for (int k = 0; k < NT; k++) {
        if (!bit(m_present, NL + k)) continue;
        float u[CM][4];
        edge(NL + k, o.g[NL + k], t_discrete, xd, xa, xb, u);
#pragma unroll
        for (int c = 0; c < CM; c++)
#pragma unroll
          for (int e = 0; e < 4; e++) w[c][e] = k == 0 ? u[c][e] : w[c][e] + u[c][e];
      }
#pragma unroll
      for (int c = 0; c < CT; c++)  // (CT == the tail's in_nch == out.nch: echo_tail_applicable)
        *reinterpret_cast<float4*>(pt[c] + f) = make_float4(w[c][0], w[c][1], w[c][2], w[c][3]);
    }
  };

  // One chunk: wave w renders sub-tile s0 + w (256 frames, 4 per lane).  GUARD: the last, partial chunk.
  auto chunk = [&](const EchoOperands<C, NG>& o, uint32_t s0, auto guard) __attribute__((always_inline)) {
    constexpr bool GUARD = decltype(guard)::value;
    const uint32_t sub = s0 + wave;
    const bool live = (!GUARD || sub < total_sub) && lane_on;
    const uint32_t f = f_first + sub * SUBF + (uint32_t)lane * 4u;
    const bool dead = f / RQ > last_q;
    if constexpr (!BQ) {
      if (live) {
        float xd[C][4];  // the delayed samples of this group, out of the ring
#pragma unroll
        for (int c = 0; c < C; c++) {
          float x[5];
#pragma unroll
          for (int e = 0; e < 5; e++) {
            const int32_t idx = (int32_t)f + pf0 + e;
            const float r = ring[c * RF + (idx & RM)];
            x[e] = idx < 0 || dead ? 0.f : r;
          }
#pragma unroll
          for (int e = 0; e < 4; e++) xd[c][e] = __builtin_fmaf(1.f - kf, x[e], kf * x[e + 1]);
        }
        finish(o, f, xd);
      }
    } else {
      // ---- the delayed samples of the group AND of the two frames in front of it (the filter's x history: the same values
      // the lane in front computed), the zero-state response of the four frames, the scan over the wavefront
      double xq[C][6], r1[C], r2[C];
      if (live) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          float x[7];
#pragma unroll
          for (int e = 0; e < 7; e++) {
            const int32_t idx = (int32_t)f + pf0 + e - 2;
            const float r = ring[c * RF + (idx & RM)];
            x[e] = idx < 0 || dead ? 0.f : r;
          }
#pragma unroll
          for (int e = 0; e < 6; e++) xq[c][e] = (double)__builtin_fmaf(1.f - kf, x[e], kf * x[e + 1]);
          const double w0 = __builtin_fma(b2, xq[c][0], __builtin_fma(b1, xq[c][1], b0 * xq[c][2]));
          const double w1 = __builtin_fma(b2, xq[c][1], __builtin_fma(b1, xq[c][2], b0 * xq[c][3]));
          const double w2 = __builtin_fma(b2, xq[c][2], __builtin_fma(b1, xq[c][3], b0 * xq[c][4]));
          const double w3 = __builtin_fma(b2, xq[c][3], __builtin_fma(b1, xq[c][4], b0 * xq[c][5]));
          const double z0 = w0, z1 = __builtin_fma(-a1, z0, w1), z2 = __builtin_fma(-a2, z0, __builtin_fma(-a1, z1, w2)),
                       z3 = __builtin_fma(-a2, z1, __builtin_fma(-a1, z2, w3));
          r1[c] = z3;
          r2[c] = z2;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const int dd = 1 << k;
#pragma unroll
          for (int c = 0; c < C; c++) {
            const double q1 = __shfl_up(r1[c], dd, 64), q2 = __shfl_up(r2[c], dd, 64);
            if (lane >= dd) {  // (a real branch on the execution mask: as selects it was 8 more instructions per step and channel)
              r1[c] = __builtin_fma(P[k].a, q1, __builtin_fma(P[k].b, q2, r1[c]));
              r2[c] = __builtin_fma(P[k].c, q1, __builtin_fma(P[k].d, q2, r2[c]));
              asm volatile("" : "+v"(r1[c]), "+v"(r2[c]));
            }
          }
        }
        if (lane == last_lane) {
#pragma unroll
          for (int c = 0; c < C; c++) {
            wz[wave][c][0] = r1[c];
            wz[wave][c][1] = r2[c];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // ---- this wave's incoming state: the chunk's starting state pushed through the sub-tiles in front of this wave's (a uniform
      // branch per step: wave w runs w of them); the last wave goes one further and leaves the next chunk's starting state
      double s1[C], s2[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        double t1 = ystate[parity][c][0], t2 = ystate[parity][c][1];
        double z1[ECHO_BQ_WAVES], z2[ECHO_BQ_WAVES];
#pragma unroll
        for (int j = 0; j < ECHO_BQ_WAVES; j++) {
          z1[j] = wz[j][c][0];
          z2[j] = wz[j][c][1];
        }
#pragma unroll
        for (int j = 0; j < ECHO_BQ_WAVES - 1; j++) {
          if ((uint32_t)j < wave) {
            const double n1 = __builtin_fma(A64.a, t1, __builtin_fma(A64.b, t2, z1[j]));
            const double n2 = __builtin_fma(A64.c, t1, __builtin_fma(A64.d, t2, z2[j]));
            t1 = n1;
            t2 = n2;
            asm volatile("" : "+v"(t1), "+v"(t2));
          }
        }
        s1[c] = t1;
        s2[c] = t2;
        if (wave + 1u == cs && lane == last_lane && live) {  // (the last lane holds the wave's own end state in r1 / r2)
          ystate[parity ^ 1u][c][0] = __builtin_fma(A64.a, t1, __builtin_fma(A64.b, t2, r1[c]));
          ystate[parity ^ 1u][c][1] = __builtin_fma(A64.c, t1, __builtin_fma(A64.d, t2, r2[c]));
        }
      }
      parity ^= 1u;
      if (live) {
        float yo[C][4];
#pragma unroll
        for (int c = 0; c < C; c++) {
          double y1 = __shfl_up(r1[c], 1, 64), y2 = __shfl_up(r2[c], 1, 64);
          if (lane == 0) y1 = y2 = 0.;
          y1 = __builtin_fma(PL.a, s1[c], __builtin_fma(PL.b, s2[c], y1));
          y2 = __builtin_fma(PL.c, s1[c], __builtin_fma(PL.d, s2[c], y2));
          // the reference's evaluation order from the true incoming state (biquad_filter.rs:877-883)
          double p1 = xq[c][1], p2 = xq[c][0];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const double x = xq[c][e + 2];
            double y = b0 * x + b1 * p1 + b2 * p2 - a1 * y1 - a2 * y2;
            if (!__builtin_isnormal(y)) y = 0.;
            p2 = p1;
            p1 = x;
            y2 = y1;
            y1 = y;
            yo[c][e] = (float)y;
          }
          if (bq.store_y) *reinterpret_cast<float4*>(py[c] + f) = make_float4(yo[c][0], yo[c][1], yo[c][2], yo[c][3]);
        }
        finish(o, f, yo);
      }
    }
    // the chunk is in the ring before the next one reads behind it.  (NOT __syncthreads(): its release fence waits for every
    // outstanding global access, vmcnt(0) — the two chunks of loads in flight and this chunk's stores — at every chunk)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // Three operand sets in rotation (chunk i uses set i % 3 and requests chunk i + 2 into the set chunk i - 1 used): the
  // loop is unrolled by three so that no set is ever copied — a register copy of a pending load is a wait for it.
  EchoOperands<C, NG> o0, o1, o2;
  const uint32_t n_chunks = (total_sub + cs - 1) / cs, full_chunks = total_sub / cs;
  fetch(wave, o0);
  fetch(wave + cs, o1);
  uint32_t i = 0;
  const std::false_type plain{};
  const std::true_type guarded{};
  for (; i + 3 <= full_chunks; i += 3) {
    fetch((i + 2) * cs + wave, o2);
    chunk(o0, i * cs, plain);
    fetch((i + 3) * cs + wave, o0);
    chunk(o1, (i + 1) * cs, plain);
    fetch((i + 4) * cs + wave, o1);
    chunk(o2, (i + 2) * cs, plain);
  }
  // (at most two full chunks and a partial one are left; what would be requested now lies past the end)
  if (i < n_chunks) {
    fetch((i + 2) * cs + wave, o2);
    chunk(o0, i * cs, guarded);
    i++;
  }
  if (i < n_chunks) {
    chunk(o1, i * cs, guarded);
    i++;
  }
  if (i < n_chunks) chunk(o2, i * cs, guarded);
}
}  // namespace

static int echo_chunk_for(float dmin, float dmax);

// The loop step `d` (one element-wise launch per block, ChainDesc::tile0 / tile1 = the whole render) as the LDS-ring kernel?
// Returns the index of the feedback input and the chunk size (frames), or -1.
int echo_ring_applicable(const ChainDesc& d, const float* delay_min_max_frames, int* chunk_subtiles) {
  if (d.n_ops != 0 || d.in_nch < 1 || d.in_nch > 2 || d.out.nch != d.in_nch || d.n_inputs < 2 || d.n_inputs > 1 + ECHO_EXT) return -1;
  int fb = -1;
  for (int k = 0; k < d.n_inputs; k++) {
    const InputRef& in = d.in[k];
    if (in.has_gain && !(in.gain.mode == 0 || in.gain.mode == 1)) return -1;
    if (in.nch != d.in_nch && !(in.nch == 1 && d.in_nch == 2)) return -1;
    if (in.kind == IN_DELAYED) {
      // the line this launch writes, read back by itself; one delayTime per instance
      if (fb >= 0 || !in.feedback || in.sig.base != d.out.base || in.sig.inst_stride != d.out.inst_stride ||
          in.sig.ch_stride != d.out.ch_stride || !(in.offset.mode == 0 || in.offset.mode == 3))
        return -1;
      fb = k;
    } else if (in.kind != IN_SIGNAL) {
      return -1;
    } else if (in.sig.base == d.out.base || ((uintptr_t)in.sig.base & 15) || (in.sig.ch_stride & 3) || (in.sig.inst_stride & 3)) {
      return -1;
    }
  }
  if ((uint64_t)d.n_tiles * TILE >= (1ull << 31)) return -1;  // (32-bit frame arithmetic)
  if (fb < 0 || ((uintptr_t)d.out.base & 15) || (d.out.ch_stride & 3) || (d.out.inst_stride & 3)) return -1;
  const int ch = echo_chunk_for(delay_min_max_frames[0], delay_min_max_frames[1]);
  if (!ch) return -1;
  *chunk_subtiles = ch;
  return fb;
}

// The chain step `tail` (outside the loop) as the tail stage of the ring kernel: a plain sum, to at least the line's channel
// count, of delayed(line) — the loop's own delayTime — and of signals the loop step reads too.
int echo_tail_applicable(const ChainDesc& d, int fb, const ChainDesc& tail, EchoTail* t, const char** why, const EchoBq* bq) {
  auto no = [&](const char* reason) {
    *why = reason;
    return 0;
  };
  if (tail.n_ops != 0) return no("it has ops of its own");
  if (tail.in_nch < d.in_nch || tail.in_nch > 2 || tail.out.nch != tail.in_nch) return no("it mixes the line down, or to more than two channels");
  if (tail.n_inputs < 1 || tail.n_inputs > ECHO_TAIL_IN) return no("it sums more than three inputs");
  if (tail.n_inst != d.n_inst || tail.n_quanta != d.n_quanta) return no("it covers another range");
  if (((uintptr_t)tail.out.base & 15) || (tail.out.ch_stride & 3) || (tail.out.inst_stride & 3) || tail.out.base == d.out.base)
    return no("its output is not 16-byte aligned");
  // (frames that may be read: 0 = no limit, and so is any limit past the rendered range)
  auto limit = [&](uint64_t valid) { return valid >= (uint64_t)d.n_tiles * TILE ? (uint64_t)0 : valid; };
  const double line_rate = fb < d.n_inputs ? d.in[fb].sample_rate : 0.;  // (0: nothing fed back, echo_feed_forward)
  for (int j = 0; j < d.n_inputs; j++)
    if (d.in[j].sig.base == tail.out.base) return no("it renders in place over a signal the loop reads");
  EchoTail r{};
  r.n_inputs = tail.n_inputs;
  r.in_nch = tail.in_nch;
  r.in_interp = tail.in_interp;
  r.out = tail.out;
  bool reads_line = false;
  for (int k = 0; k < tail.n_inputs; k++) {
    const InputRef& in = tail.in[k];
    if (in.has_gain && !(in.gain.mode == 0 || in.gain.mode == 1)) return no("an edge gain of it has a value per frame");
    if (in.nch != tail.in_nch && !(in.nch == 1 && tail.in_nch == 2)) return no("an input of it is mixed by a computed rule");
    r.in[k] = in;
    r.alias[k] = -1;
    if (bq && in.kind == IN_SIGNAL && in.sig.base == bq->y.base) {
      // the BQ form: the filter's output is the operand the kernel holds in place of the delayed line
      if (in.sig.inst_stride != bq->y.inst_stride || in.sig.ch_stride != bq->y.ch_stride || in.nch != d.in_nch || limit(in.valid) != 0)
        return no("it reads the filter's output in another layout");
      r.alias[k] = -2;
      reads_line = true;
    } else if (bq && in.kind == IN_DELAYED) {
      return no("it reads the delay line itself, which the filtered loop does not hold as an operand");
    } else if (in.kind == IN_DELAYED) {
      // (a line belongs to ONE DelayNode: every delayed read of it is by that node's delayTime, the loop's own)
      if (in.sig.base != d.out.base || in.sig.inst_stride != d.out.inst_stride || in.sig.ch_stride != d.out.ch_stride ||
          in.nch != d.in_nch || !(in.offset.mode == 0 || in.offset.mode == 3) || (line_rate != 0. && in.sample_rate != line_rate))
        return no("it reads another delay line too");
      r.alias[k] = -2;
      reads_line = true;
    } else if (in.kind == IN_SIGNAL) {
      for (int j = 0; j < d.n_inputs; j++) {
        const InputRef& lj = d.in[j];
        if (j != fb && lj.kind == IN_SIGNAL && lj.sig.base == in.sig.base && lj.sig.inst_stride == in.sig.inst_stride &&
            lj.sig.ch_stride == in.sig.ch_stride && lj.nch == in.nch && limit(lj.valid) == limit(in.valid))
          r.alias[k] = j - (j > fb ? 1 : 0);  // (its register slot)
      }
      if (r.alias[k] < 0) return no("it sums a signal the loop does not read");
    } else {
      return no("it has a source or constant input");
    }
  }
  if (!reads_line) return no(bq ? "it does not read the filter's output" : "it does not read the line");
  *t = r;
  return 1;
}

int echo_ring_frames(float dmax_frames, int chunk_frames) {
  int need = (int)std::ceil(dmax_frames) + chunk_frames + 8, rf = 1024;
  while (rf < need && rf < ECHO_RING) rf <<= 1;
  return rf;
}

static int echo_chunk_for(float dmin, float dmax) {
  // the frames a chunk reads must lie BEHIND the chunk (delay > chunk) and still be in the ring (delay + chunk < ring)
  // (round 4: down to ONE sub-tile — a comb filter's / a plucked string's feedback delay of a few hundred frames walks in chunks
  // of 256 frames, one wavefront per instance, with a ring as small as its delay needs: many instances per CU)
  // (the largest chunk that fits BOTH ends: a batch whose delays span 4200 .. 13000 frames walks in chunks of 2048, not 4096)
  // (128: half a wavefront — the plucked string above ~180 Hz has a delay below 264 frames)
  for (int cand : {4096, 2048, 1024, 512, 256, 128})
    if ((float)(cand + 8) <= dmin && dmax <= (float)(ECHO_RING - cand - 8)) return cand;
  return 0;
}

int echo_bq_applicable(const ChainDesc& rd, const BiquadStreamDesc& f, const ChainDesc& sum, const float* delay_min_max_frames,
                       int* chunk_subtiles, EchoBq* bq) {
  const int nch = sum.in_nch;
  // the delayed read: the line the sum writes, by one delayTime per instance, nothing else
  if (rd.n_ops != 0 || rd.n_inputs != 1 || rd.in_nch != nch || rd.out.nch != nch || nch < 1 || nch > 2) return -1;
  const InputRef& D = rd.in[0];
  if (D.kind != IN_DELAYED || !D.feedback || D.has_gain || D.nch != nch || D.sig.base != sum.out.base || D.sig.inst_stride != sum.out.inst_stride ||
      D.sig.ch_stride != sum.out.ch_stride || !(D.offset.mode == 0 || D.offset.mode == 3))
    return -1;
  // the filter: constant coefficients, the delayed read in, no gains behind it
  if (f.in.kind != IN_SIGNAL || f.in.has_gain || f.in.sig.base != rd.out.base || f.in.nch != nch || f.nch != nch || f.vary != 0 || f.n_gain != 0 ||
      f.dup_out || f.out.nch != nch || !f.coefs || ((uintptr_t)f.out.base & 15) || (f.out.ch_stride & 3) || (f.out.inst_stride & 3))
    return -1;
  // the sum: the filter's output (once) plus signals from outside the loop
  if (sum.n_ops != 0 || sum.out.nch != nch || sum.n_inputs < 2 || sum.n_inputs > 1 + ECHO_EXT) return -1;
  int fb = -1;
  for (int k = 0; k < sum.n_inputs; k++) {
    const InputRef& in = sum.in[k];
    if (in.kind != IN_SIGNAL) return -1;
    if (in.has_gain && !(in.gain.mode == 0 || in.gain.mode == 1)) return -1;
    if (in.nch != nch && !(in.nch == 1 && nch == 2)) return -1;
    if (in.sig.base == f.out.base) {
      if (fb >= 0 || in.nch != nch || in.sig.inst_stride != f.out.inst_stride || in.sig.ch_stride != f.out.ch_stride) return -1;
      fb = k;
    } else if (in.sig.base == sum.out.base || in.sig.base == rd.out.base || ((uintptr_t)in.sig.base & 15) || (in.sig.ch_stride & 3) ||
               (in.sig.inst_stride & 3)) {
      return -1;
    }
  }
  if ((uint64_t)sum.n_tiles * TILE >= (1ull << 31)) return -1;
  if (fb < 0 || ((uintptr_t)sum.out.base & 15) || (sum.out.ch_stride & 3) || (sum.out.inst_stride & 3)) return -1;
  int ch = 0;
  for (int cand : {ECHO_BQ_WAVES * 256, 1024, 512, 256, 128})
    if (!ch && (float)(cand + 8) <= delay_min_max_frames[0] && delay_min_max_frames[1] <= (float)(ECHO_RING - cand - 8)) ch = cand;
  if (!ch) return -1;
  *chunk_subtiles = ch;
  EchoBq q{};
  q.coefs = f.coefs;
  q.coef_stride = f.coef_stride;
  q.y = f.out;
  q.store_y = 1;
  q.store_line = 1;
  q.delay = D.offset;
  q.sample_rate = D.sample_rate;
  *bq = q;
  return fb;
}

int echo_feed_forward(const ChainDesc& st, ChainDesc* line, EchoTail* tail, const char** why) {
  auto no = [&](const char* reason) {
    *why = reason;
    return 0;
  };
  int dk = -1;
  for (int k = 0; k < st.n_inputs; k++)
    if (st.in[k].kind == IN_DELAYED) {
      if (dk >= 0) return no("two delayed inputs");
      dk = k;
    }
  if (dk < 0) return no("no delayed input");
  const InputRef& D = st.in[dk];
  if (D.feedback) return no("the delay line is written inside a loop");
  if (!(D.offset.mode == 0 || D.offset.mode == 3) || D.delay_hi < D.delay_lo) return no("the delay is not one host-known value per instance");
  const int chunk = echo_chunk_for(D.delay_lo, D.delay_hi);
  if (chunk < 1024) return no("a delay outside the ring's window");  // (chunks below 1024 frames: the tile-parallel launch is the better fit)
  if ((uint64_t)st.n_tiles * TILE >= (1ull << 31)) return no("more than 2^31 frames");
  if (D.nch < 1 || D.nch > 2 || ((uintptr_t)D.sig.base & 15) || (D.sig.ch_stride & 3) || (D.sig.inst_stride & 3))
    return no("the delayed signal is wider than stereo or not 16-byte aligned");
  // the stand-in loop stage: line = X, nothing fed back, never stored
  ChainDesc l{};
  l.n_inputs = 1;
  l.in_nch = D.nch;
  l.in[0].kind = IN_SIGNAL;
  l.in[0].nch = D.nch;
  l.in[0].sig = D.sig;
  l.in[0].valid = D.valid;
  l.out = D.sig;
  l.out.nch = D.nch;
  l.n_inst = st.n_inst;
  l.n_tiles = st.n_tiles;
  l.n_quanta = st.n_quanta;
  l.tile0 = 0;
  l.tile1 = st.n_tiles;
  EchoTail t{};
  if (!echo_tail_applicable(l, l.n_inputs, st, &t, why)) return 0;
  t.store_line = 0;
  t.delay = D.offset;
  t.sample_rate = D.sample_rate;
  *line = l;
  *tail = t;
  return chunk;
}

void launch_echo_ring(const ChainDesc& d, int fb, int chunk_frames, int ring_frames, const EchoTail* tail, void* stream, const EchoBq* bq) {
  const int chunk_subtiles = std::max(1, chunk_frames / 256);
  if (ring_frames < 1024 || ring_frames > ECHO_RING || (ring_frames & (ring_frames - 1))) ring_frames = ECHO_RING;
  const size_t lds = (size_t)d.in_nch * (size_t)ring_frames * sizeof(float);
  const int ct = tail ? tail->in_nch : 0;
  EchoTail t{};
  if (tail) t = *tail;
  t.ring_frames = ring_frames;
  t.sub_frames = chunk_frames >= 256 ? 256 : 128;
  if (bq) {
    t.delay = bq->delay;
    t.sample_rate = bq->sample_rate;
  } else if (fb < d.n_inputs) {
    t.delay = d.in[fb].offset;
    t.sample_rate = d.in[fb].sample_rate;
  }
  const EchoBq q = bq ? *bq : EchoBq{};
  const dim3 block((unsigned)chunk_subtiles * 64), grid(d.n_inst);
  const bool store = !tail || tail->store_line;
  auto go = [&](auto kern) {
    raise_lds_limit(reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, d, fb, chunk_subtiles, t, q);
  };
  if (bq) {
    const bool sl = bq->store_line != 0;
    if (d.in_nch == 1) {
      if (ct == 0) sl ? go(echo_ring_kernel<1, 0, true, true>) : go(echo_ring_kernel<1, 0, false, true>);
      else if (ct == 1) sl ? go(echo_ring_kernel<1, 1, true, true>) : go(echo_ring_kernel<1, 1, false, true>);
      else sl ? go(echo_ring_kernel<1, 2, true, true>) : go(echo_ring_kernel<1, 2, false, true>);
    } else {
      if (ct == 0) sl ? go(echo_ring_kernel<2, 0, true, true>) : go(echo_ring_kernel<2, 0, false, true>);
      else sl ? go(echo_ring_kernel<2, 2, true, true>) : go(echo_ring_kernel<2, 2, false, true>);
    }
    return;
  }
  if (d.in_nch == 1) {
    if (ct == 0) go(echo_ring_kernel<1, 0, true>);
    else if (ct == 1) store ? go(echo_ring_kernel<1, 1, true>) : go(echo_ring_kernel<1, 1, false>);
    else store ? go(echo_ring_kernel<1, 2, true>) : go(echo_ring_kernel<1, 2, false>);
  } else {
    if (ct == 0) go(echo_ring_kernel<2, 0, true>);
    else store ? go(echo_ring_kernel<2, 2, true>) : go(echo_ring_kernel<2, 2, false>);
  }
}

}  // namespace waa
