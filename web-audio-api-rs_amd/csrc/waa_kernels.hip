// waa_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the batched offline render engine.
//
// chain_kernel: one 64-lane wavefront renders ONE instance (all its channels) for the whole
// duration, tile by tile (2048 frames = 16 render quanta), carrying recurrence state in
// registers.  A chain is  input(s) -> [mix] -> op* -> output  where the ops are the fused node
// kernels of a single-consumer path of the graph (source fetch, gain, biquad, waveshaper,
// stereo panner, equal-power panner, channel mixing).  Global loads/stores are 16 B per lane,
// fully coalesced (1 KiB per wave instruction); the biquad recurrence runs on an LDS-transposed
// layout (32 consecutive frames per lane) as zero-state pass + wavefront scan of 2x2 affine maps
// + exact-order final pass in f64.  HBM-bound by design: no MFMA anywhere.
//
// Compiled with -ffp-contract=off: the reference (Rust) never fuses a*b+c unless it says
// mul_add; every fma below is explicit.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"
#include "waa_mix.hpp"

namespace waa {

constexpr int LDS_ROW = TILE_K + 4;  // padded row (floats): 144 B, conflict-free b128 access
constexpr int CARRY_BYTES = MAX_OPS * 8 * 4 * 8;  // [MAX_OPS][C<=8][4] doubles

struct f4 {
  float x, y, z, w;
};

__device__ __forceinline__ float param_at(const ParamRef& p, uint32_t inst, uint32_t q, uint64_t frame) {
  if (p.mode == 0) return p.base[inst];
  if (p.mode == 1) return p.base[(uint64_t)inst * p.stride + q];
  return p.base[(uint64_t)inst * p.stride + frame];
}

// (channel mixing on register tiles: waa_mix.hpp, shared with waa_dyn.hip)

// ---- input fetch (layout A: lane holds float4 j at frame tile*TILE + j*256 + lane*4) ------
// K = frames per lane: a tile is 64*K frames = K/2 render quanta (K = 32: the serial 2048-frame tiles of chains
// with a recurrence; K = 4: the 256-frame sub-tiles of tile-parallel element-wise chains)
template <int C, int K>
__device__ __forceinline__ void load_input(const InputRef& in, uint32_t inst, uint32_t tile, int lane, uint32_t n_quanta,
                                           float (&v)[C][K]) {
  constexpr int NV4 = K / 4;
  constexpr int TILE_FR = 64 * K;
  constexpr int QPT = K / 2;
  const uint64_t f_tile = (uint64_t)tile * TILE_FR;
  if (in.kind == IN_SIGNAL) {
    // (valid != 0: a BufferSource's AudioBuffer read in place — frames past its end are silence; whole float4 groups,
    // the buffer length is a multiple of the render quantum)
#pragma unroll
    for (int c = 0; c < C; c++) {
      if (c < in.nch) {
        const float* p = in.sig.base + (uint64_t)inst * in.sig.inst_stride + (uint64_t)c * in.sig.ch_stride + f_tile;
#pragma unroll
        for (int j = 0; j < NV4; j++) {
          const bool inside = in.valid == 0 || f_tile + (uint64_t)(j * 256 + lane * 4) + 3 < in.valid;
          const float4 t = inside ? *reinterpret_cast<const float4*>(p + j * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[c][j * 4 + 0] = t.x;
          v[c][j * 4 + 1] = t.y;
          v[c][j * 4 + 2] = t.z;
          v[c][j * 4 + 3] = t.w;
        }
      }
    }
    return;
  }
  if (in.kind == IN_DELAYED) {
    // the gather of waa_delay.hip in the input stage (delay.rs:560-590, 622-642); 4 | 128: a float4 group sits in one quantum
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const uint64_t f = f_tile + (uint64_t)(j * 256 + lane * 4);
      const uint32_t q = (uint32_t)(f / RQ);
      const bool live = q < n_quanta;
      const float dv = in.offset.mode == 3   ? __uint_as_float((uint32_t)in.offset.stride)  // one value for the batch
                       : in.offset.mode == 0 ? load_global(in.offset.base + inst)
                                             : load_global(in.offset.base + (uint64_t)inst * in.offset.stride + (live ? q : 0));
#pragma unroll
      for (int c = 0; c < C; c++) {
        if (c < in.nch) {
          const float* p = in.sig.base + (uint64_t)inst * in.sig.inst_stride + (uint64_t)c * in.sig.ch_stride;
          float r[4] = {0.f, 0.f, 0.f, 0.f};
          if (live) delay_read4(p, in.valid, dv, in.sample_rate, in.num_quanta, false, 0., q, (int)(f % RQ), r);
#pragma unroll
          for (int e = 0; e < 4; e++) v[c][j * 4 + e] = r[e];
        }
      }
    }
    return;
  }
  if (in.kind == IN_SOURCE) {
    const SrcInst si = in.src[inst];
    const SrcSchedule sc = si.sc;
    // (inside the linear prefix of the schedule the tile's start is arithmetic: no dependent table loads in front of
    // the samples — the streaming kernels' shortcut, SrcInst::fast_prefix)
    const bool in_prefix = si.aligned && f_tile / TILE < si.fast_prefix;
    if (in_prefix || (si.aligned && load_global(sc.tile_fast + f_tile / TILE))) {
      // the enclosing 2048-frame tile is one contiguous, in-range, 16B-aligned run of the AudioBuffer
      const int64_t start = in_prefix ? si.linear_start + (int64_t)f_tile : load_global(&sc.qrec[(uint64_t)tile * QPT].start);
#pragma unroll
      for (int c = 0; c < C; c++) {
        if (c < in.nch) {
          const float* p = si.base + (uint64_t)c * si.ch_stride + start;
#pragma unroll
          for (int j = 0; j < NV4; j++) {
            const f4v t = load_global_f4(p + j * 256 + lane * 4);
            v[c][j * 4 + 0] = t.x;
            v[c][j * 4 + 1] = t.y;
            v[c][j * 4 + 2] = t.z;
            v[c][j * 4 + 3] = t.w;
          }
        }
      }
      return;
    }
    // generic path: per-quantum records (silent / fast copy with end-of-buffer or loop wrap / slow track)
    if constexpr (K == 4) {
      // tile-parallel chains run one 256-frame sub-tile per wave, so a wave's time is the latency of its chain of
      // dependent loads (instance record -> quantum record -> playback records -> samples) and not bandwidth:
      // the playback records are requested together with the quantum record, before the mode is known, and the
      // interpolating track is straight-line code so that all its gathers are in flight at once
      const uint32_t q = tile * 2 + (lane >> 5);
      const bool valid_q = q < n_quanta;
      const uint32_t qc = valid_q ? q : 0;
      const uint32_t i0 = (lane & 31) * 4;  // index of this lane's first frame within the quantum
      const QRec r = load_global(sc.qrec + qc);
      SlowRec s[4];
#pragma unroll
      for (int e = 0; e < 4; e++) s[e] = SlowRec{-1, -1, 0.};
      if (sc.slow) {
#pragma unroll
        for (int e = 0; e < 4; e++) s[e] = load_global(sc.slow + (uint64_t)qc * RQ + i0 + e);
      }
      const uint32_t mode = valid_q ? r.mode : (uint32_t)Q_SILENT;
      if (mode == Q_SLOW) {
        // audio_buffer_source.rs:754-822
        float g0[C][4], g1[C][4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int32_t ip = s[e].prev >= 0 ? s[e].prev : 0;
          // next >= 0: the next frame; -1: silence after the end; -2: extrapolate from the frame BEFORE prev
          const int32_t in2 = s[e].prev < 0 ? 0 : (s[e].next >= 0 ? s[e].next : (s[e].next == -1 ? 0 : s[e].prev - 1));
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) {
              const float* ch = si.base + (uint64_t)c * si.ch_stride;
              g0[c][e] = load_global(ch + ip);
              g1[c][e] = load_global(ch + in2);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) {
              const double prev_sample = (double)g0[c][e];
              const double other = (double)g1[c][e];
              const double next_sample = s[e].next >= 0 ? other : (s[e].next == -1 ? 0. : 2. * prev_sample - other);
              const float o = (float)__builtin_fma(1. - s[e].k, prev_sample, s[e].k * next_sample);
              v[c][e] = s[e].prev >= 0 ? o : 0.f;
            }
        }
      } else if (mode == Q_FAST || mode == Q_FAST_LOOP) {
        // audio_buffer_source.rs:562-607: index start+i, zero past the end, or wrap when looping
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
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) v[c][e] = ok ? load_global(si.base + (uint64_t)c * si.ch_stride + bi) : 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < in.nch) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[c][e] = 0.f;
          }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const uint32_t fq = j * 256 + lane * 4;  // frame within tile
      uint32_t q = tile * QPT + fq / RQ;
      const bool valid_q = q < n_quanta;
      const QRec r = load_global(sc.qrec + (valid_q ? q : 0));
      const uint32_t mode = valid_q ? r.mode : (uint32_t)Q_SILENT;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t i = (fq % RQ) + e;  // index within quantum
        if (mode == Q_FAST || mode == Q_FAST_LOOP) {
          // audio_buffer_source.rs:562-607: index start+i, zero past the end, or wrap when looping
          uint64_t bi = (uint64_t)r.start + i;
          bool ok = true;
          if (bi >= si.frames) {
            if (mode == Q_FAST_LOOP)
              bi = bi % si.frames;
            else
              ok = false;
          }
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) v[c][j * 4 + e] = ok ? load_global(si.base + (uint64_t)c * si.ch_stride + bi) : 0.f;
        } else if (mode == Q_SLOW) {
          const SlowRec s = load_global(sc.slow + (uint64_t)q * RQ + i);
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) {
              float o = 0.f;
              if (s.prev >= 0) {
                // audio_buffer_source.rs:754-822
                const float* ch = si.base + (uint64_t)c * si.ch_stride;
                const double prev_sample = (double)load_global(ch + s.prev);
                double next_sample;
                if (s.next >= 0)
                  next_sample = (double)load_global(ch + s.next);
                else if (s.next == -1)
                  next_sample = 0.;
                else
                  next_sample = 2. * prev_sample - (double)load_global(ch + s.prev - 1);
                o = (float)__builtin_fma(1. - s.k, prev_sample, s.k * next_sample);
              }
              v[c][j * 4 + e] = o;
            }
        } else {
#pragma unroll
          for (int c = 0; c < C; c++)
            if (c < in.nch) v[c][j * 4 + e] = 0.f;
        }
      }
    }
    return;
  }
  if (in.kind == IN_CONSTANT) {
    // constant_source.rs:190-275; the active frame range was resolved on the host
    const int64_t a0 = in.active[(uint64_t)inst * 2], a1 = in.active[(uint64_t)inst * 2 + 1];
#pragma unroll
    for (int j = 0; j < NV4; j++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint64_t f = f_tile + j * 256 + lane * 4 + e;
        uint32_t q = (uint32_t)(f / RQ);
        if (q >= n_quanta) q = n_quanta - 1;
        const uint64_t fc = f < (uint64_t)n_quanta * RQ ? f : (uint64_t)n_quanta * RQ - 1;
        const float val = param_at(in.offset, inst, q, fc);
        v[0][j * 4 + e] = ((int64_t)f >= a0 && (int64_t)f < a1) ? val : 0.f;
      }
    }
    return;
  }
  // IN_SILENT
#pragma unroll
  for (int i = 0; i < K; i++) v[0][i] = 0.f;
}

// ---- GainNode (gain.rs:143-199) on a register tile; also the gain folded into an input edge ----------------
template <int C, int K>
__device__ __forceinline__ void gain_regs(float (&v)[C][K], int nch, const ParamRef& p0, uint32_t inst, uint32_t tile, int lane,
                                          uint32_t n_quanta) {
  constexpr int NV4 = K / 4;
  constexpr int TILE_FR = 64 * K;
  constexpr int QPT = K / 2;
#pragma unroll
  for (int j = 0; j < NV4; j++) {
    const uint32_t q = tile * QPT + j * 2 + (lane >> 5);
    const uint32_t qc = q < n_quanta ? q : n_quanta - 1;
    const uint64_t f = (uint64_t)tile * TILE_FR + j * 256 + lane * 4;
    if (p0.mode == 2) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint64_t fc = f + e < (uint64_t)n_quanta * RQ ? f + e : (uint64_t)n_quanta * RQ - 1;
        const float g = p0.base[(uint64_t)inst * p0.stride + fc];
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < nch) v[c][j * 4 + e] *= g;
      }
    } else {
      // gain.rs:163-179: |g| <= 1e-6 -> silent, |1-g| <= 1e-6 -> passthrough
      const float g = param_at(p0, inst, qc, 0);
      const bool mute = fabsf(g) <= 1e-6f;
      const bool pass = fabsf(1.f - g) <= 1e-6f;
#pragma unroll
      for (int c = 0; c < C; c++)
        if (c < nch) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[c][j * 4 + e] = mute ? 0.f : (pass ? v[c][j * 4 + e] : v[c][j * 4 + e] * g);
        }
    }
  }
}

// ---- biquad (biquad_filter.rs:764-899) on the transposed layout --------------------------
struct Mat2 {
  double a, b, c, d;  // [[a b],[c d]]
};
__device__ __forceinline__ Mat2 matmul(const Mat2& x, const Mat2& y) {
  Mat2 r;
  r.a = __builtin_fma(x.a, y.a, x.b * y.c);
  r.b = __builtin_fma(x.a, y.b, x.b * y.d);
  r.c = __builtin_fma(x.c, y.a, x.d * y.c);
  r.d = __builtin_fma(x.c, y.b, x.d * y.d);
  return r;
}
__device__ __forceinline__ double shfl_up_d(double v, int delta) { return __shfl_up(v, delta, 64); }

template <int C>
__device__ __forceinline__ void biquad_op(const OpDesc& op, uint32_t inst, uint32_t tile, int lane, uint32_t n_quanta,
                                          float* lds, float (&v)[C][TILE_K], double* carry, const double* coef_inst) {
  // carry: LDS, [C][4] = (x1, x2, y1, y2) per channel at the start of this tile
  constexpr int NV4 = TILE_K / 4;
  const int nch = op.nch_in;
  // A -> LDS
#pragma unroll
  for (int c = 0; c < C; c++) {
    if (c < nch) {
      float* base = lds + c * (64 * LDS_ROW);
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const int row = j * 8 + (lane >> 3), col = (lane & 7) * 4;
        *reinterpret_cast<float4*>(base + row * LDS_ROW + col) =
            make_float4(v[c][j * 4 + 0], v[c][j * 4 + 1], v[c][j * 4 + 2], v[c][j * 4 + 3]);
      }
    }
  }
  __syncthreads();
  // coefficients: per instance (mode 0), per quantum (mode 1: 4 lanes per quantum) or per frame (mode 2:
  // a-rate params, biquad_filter.rs:837-855 — streamed from the table written by biquad_coef_kernel)
  const bool per_frame = op.i0 == 2;
  const uint64_t f_lane = (uint64_t)tile * TILE + (uint64_t)lane * TILE_K;  // first frame of this lane's chunk
  const uint64_t f_max = (uint64_t)n_quanta * RQ - 1;
  double b0 = 0., b1 = 0., b2 = 0., a1 = 0., a2 = 0.;
  Mat2 A;
  if (!per_frame) {
    const double* cp = coef_inst;
    if (op.i0 == 1) {
      uint32_t q = tile * QUANTA_PER_TILE + (lane >> 2);
      if (q >= n_quanta) q = n_quanta - 1;
      cp += (uint64_t)q * 5;
    }
    b0 = cp[0];
    b1 = cp[1];
    b2 = cp[2];
    a1 = cp[3];
    a2 = cp[4];
    // A_l = M^32 with M = [[-a1, -a2], [1, 0]] acting on (y[n-1], y[n-2])
    Mat2 m = {-a1, -a2, 1., 0.};
#pragma unroll
    for (int s = 0; s < 5; s++) m = matmul(m, m);
    A = m;
  } else {
    // A_l = M_31 * ... * M_0 ; left-multiplying by M_i = [[-a1_i, -a2_i], [1, 0]] costs 4 flops
    Mat2 pm = {1., 0., 0., 1.};
#pragma unroll
    for (int i = 0; i < TILE_K; i++) {
      const uint64_t f = f_lane + i < f_max ? f_lane + i : f_max;
      const double ca1 = coef_inst[f * 5 + 3], ca2 = coef_inst[f * 5 + 4];
      const double na = __builtin_fma(-ca1, pm.a, -(ca2 * pm.c)), nb = __builtin_fma(-ca1, pm.b, -(ca2 * pm.d));
      pm.c = pm.a;
      pm.d = pm.b;
      pm.a = na;
      pm.b = nb;
    }
    A = pm;
  }
#pragma unroll
  for (int c = 0; c < C; c++) {
    if (c < nch) {
      float* base = lds + c * (64 * LDS_ROW);
      float x[TILE_K];
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const float4 t = *reinterpret_cast<const float4*>(base + lane * LDS_ROW + j * 4);
        x[j * 4 + 0] = t.x;
        x[j * 4 + 1] = t.y;
        x[j * 4 + 2] = t.z;
        x[j * 4 + 3] = t.w;
      }
      // x history at the chunk boundary: previous lane's last two samples (lane 0: carried state)
      float xm1 = __shfl_up(x[TILE_K - 1], 1, 64), xm2 = __shfl_up(x[TILE_K - 2], 1, 64);
      const double c_x1 = carry[c * 4 + 0], c_x2 = carry[c * 4 + 1], c_y1 = carry[c * 4 + 2], c_y2 = carry[c * 4 + 3];
      double x1 = lane == 0 ? c_x1 : (double)xm1;
      double x2 = lane == 0 ? c_x2 : (double)xm2;
      // FIR part in the reference's evaluation order: (b0*x + b1*x1) + b2*x2, then zero-state recurrence
      double w[TILE_K];
      double z1 = 0., z2 = 0.;
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        if (per_frame) {
          const uint64_t f = f_lane + i < f_max ? f_lane + i : f_max;
          const double* cp = coef_inst + f * 5;
          b0 = cp[0];
          b1 = cp[1];
          b2 = cp[2];
          a1 = cp[3];
          a2 = cp[4];
        }
        const double xd = (double)x[i];
        w[i] = (b0 * xd + b1 * x1) + b2 * x2;
        x2 = x1;
        x1 = xd;
        const double t = __builtin_fma(-a2, z2, w[i]);
        const double y = __builtin_fma(-a1, z1, t);
        z2 = z1;
        z1 = y;
      }
      // inclusive wavefront scan of the affine maps s -> A s + z  (s = (y1, y2) at chunk end)
      Mat2 P = A;
      double r1 = z1, r2 = z2;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        Mat2 Q;
        Q.a = shfl_up_d(P.a, d);
        Q.b = shfl_up_d(P.b, d);
        Q.c = shfl_up_d(P.c, d);
        Q.d = shfl_up_d(P.d, d);
        const double q1 = shfl_up_d(r1, d), q2 = shfl_up_d(r2, d);
        if (lane >= d) {
          const double n1 = __builtin_fma(P.a, q1, __builtin_fma(P.b, q2, r1));
          const double n2 = __builtin_fma(P.c, q1, __builtin_fma(P.d, q2, r2));
          r1 = n1;
          r2 = n2;
          P = matmul(P, Q);
        }
      }
      // incoming state of this lane = map of lane-1 applied to the tile's incoming state
      double y1, y2;
      {
        const double pa = shfl_up_d(P.a, 1), pb = shfl_up_d(P.b, 1), pc = shfl_up_d(P.c, 1), pd = shfl_up_d(P.d, 1);
        const double q1 = shfl_up_d(r1, 1), q2 = shfl_up_d(r2, 1);
        const double s1 = c_y1, s2 = c_y2;
        y1 = lane == 0 ? s1 : __builtin_fma(pa, s1, __builtin_fma(pb, s2, q1));
        y2 = lane == 0 ? s2 : __builtin_fma(pc, s1, __builtin_fma(pd, s2, q2));
      }
      // final pass, exact reference order: y = ((w) - a1*y1) - a2*y2 ; flush !is_normal to 0
      float yo[TILE_K];
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        if (per_frame) {
          const uint64_t f = f_lane + i < f_max ? f_lane + i : f_max;
          a1 = coef_inst[f * 5 + 3];
          a2 = coef_inst[f * 5 + 4];
        }
        double y = (w[i] - a1 * y1) - a2 * y2;
        if (!__builtin_isnormal(y)) y = 0.;
        y2 = y1;
        y1 = y;
        yo[i] = (float)y;
      }
      // new carried state = lane 63's end state
      if (lane == 63) {
        carry[c * 4 + 0] = (double)x[TILE_K - 1];
        carry[c * 4 + 1] = (double)x[TILE_K - 2];
        carry[c * 4 + 2] = y1;
        carry[c * 4 + 3] = y2;
      }
      // T -> LDS (own row)
#pragma unroll
      for (int j = 0; j < NV4; j++)
        *reinterpret_cast<float4*>(base + lane * LDS_ROW + j * 4) =
            make_float4(yo[j * 4 + 0], yo[j * 4 + 1], yo[j * 4 + 2], yo[j * 4 + 3]);
    }
  }
  __syncthreads();
  // LDS -> A
#pragma unroll
  for (int c = 0; c < C; c++) {
    if (c < nch) {
      const float* base = lds + c * (64 * LDS_ROW);
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const int row = j * 8 + (lane >> 3), col = (lane & 7) * 4;
        const float4 t = *reinterpret_cast<const float4*>(base + row * LDS_ROW + col);
        v[c][j * 4 + 0] = t.x;
        v[c][j * 4 + 1] = t.y;
        v[c][j * 4 + 2] = t.z;
        v[c][j * 4 + 3] = t.w;
      }
    }
  }
  __syncthreads();
}

// waveshaper.rs:555-573
__device__ __forceinline__ float apply_curve(const float* curve, int nn, float input) {
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

// the same lookup on a curve staged in LDS
__device__ __forceinline__ float apply_curve_lds(const __attribute__((address_space(3))) float* curve, int nn, float input) {
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

__device__ __forceinline__ void stereo_gains_dev(float x, float& gl, float& gr) {
  const float PI_F = 3.14159265358979323846f;
  gl = sinf((1.f - x) * PI_F / 2.f);
  gr = sinf(x * PI_F / 2.f);
}

// SERIAL = true : chains with a recurrence (OP_BIQUAD): one wave per instance walks the 2048-frame tiles in order.
// SERIAL = false: element-wise chains: one wave per (instance, 256-frame sub-tile), 4 waves per workgroup.
// FANIN = false: single-input chains skip the summing loop; the second inlined copy of the input fetch is what
// doubles the kernel's VGPR count (57 -> 110 for C = 2), i.e. halves the waves per SIMD of kernels that are bound
// by the latency of their dependent loads.
// PERSIST = true: the persistent form of a block-scheduled loop (ChainDesc::persist_block) — its own instantiation: the
// block bookkeeping cost the plain single-input kernels 12-17 registers, i.e. waves per SIMD, when it lived in them.
// NOSPILL = true (stereo summing form only): five wavefronts per SIMD and 96 registers instead of six and 80 — the six-wave
// form pays for its last wave with four registers spilled to scratch memory.
template <int C, int K, bool SERIAL, bool FANIN = true, bool PERSIST = false, bool NOSPILL = false>
__global__ __launch_bounds__(SERIAL ? 64 : 256)
    __attribute__((amdgpu_waves_per_eu((!SERIAL && FANIN && C == 2) ? (NOSPILL ? 5 : 6) : 1)))
void chain_kernel(const ChainDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NV4 = K / 4;
  constexpr int TILE_FR = 64 * K;
  constexpr int QPT = K / 2;
  const int lane = threadIdx.x & 63;
  const uint32_t n_tiles_k = (d.tile1 - d.tile0) * (TILE / TILE_FR);  // sub-tiles of this launch
  uint32_t inst, tile_first, tile_last;
  uint32_t tile_step = 1, per_block = 0;  // (persistent form: sub-tiles a wavefront renders between two workgroup barriers)
  if (SERIAL) {
    inst = blockIdx.x;
    tile_first = d.tile0 * (TILE / TILE_FR);
    tile_last = d.tile1 * (TILE / TILE_FR);
  } else if (PERSIST) {
    // one workgroup per instance, blocks in order: wavefront w renders sub-tiles w, w + 4, ... of every block (the block
    // and the tile range are multiples of four sub-tiles, so the four wavefronts meet at every barrier).  What a block
    // reads of the loop's history was written by THIS workgroup before an earlier barrier: same CU, same L1.
    inst = blockIdx.x;
    tile_first = d.tile0 * (TILE / TILE_FR) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    tile_last = d.tile1 * (TILE / TILE_FR);
    tile_step = 4;
    per_block = d.persist_block / 4;
  } else {
    // (readfirstlane: the wave index is uniform; telling the compiler keeps instance / tile addressing in SGPRs)
    // Workgroup b runs on XCD b % 8.  In (instance, sub-tile) order the XCDs are dealt CONTIGUOUS ranges of it: what a
    // workgroup re-reads of its predecessors' input — a folded delay line reads the same signal a few sub-tiles back —
    // was then fetched by the same XCD a moment ago and is an L2 hit instead of a second trip to memory.
    uint32_t blk = blockIdx.x;
    if (d.xcd_remap) {
      const uint32_t q = gridDim.x / 8, r = gridDim.x % 8, x = blk % 8, j = blk / 8;
      blk = x * q + (x < r ? x : r) + j;
    }
    const uint64_t wid = (uint64_t)blk * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (d.tile_major) {
      // neighbouring waves render the SAME sub-tile of different instances: the per-frame playback table of a
      // resampling source (16 B per frame, shared by all instances of a schedule) is then reused out of L2 instead
      // of being re-fetched once per instance (C5: 15.7 GB of the 35 GB the kernel moved)
      inst = (uint32_t)(wid % d.n_inst);
      tile_first = d.tile0 * (TILE / TILE_FR) + (uint32_t)(wid / d.n_inst);
      if (wid / d.n_inst >= n_tiles_k) inst = d.n_inst;  // past the end: retire below
    } else {
      inst = (uint32_t)(wid / n_tiles_k);
      tile_first = d.tile0 * (TILE / TILE_FR) + (uint32_t)(wid % n_tiles_k);
    }
    tile_last = tile_first + 1;
  }
  // tile-parallel variant: the WaveShaper curve is staged in LDS once per workgroup; per-sample lookups are then
  // LDS gathers instead of global gathers that cost one L1 line access per distinct line (C5: the curve lookups
  // were 16 of the 36 vector-memory instructions per lane and sub-tile)
  if constexpr (!SERIAL) {
    if (d.lds_curve_op >= 0) {
      const int nn = d.ops[d.lds_curve_op].i0;
      const float* src = reinterpret_cast<const float*>(d.ops[d.lds_curve_op].ptr0);
      for (int i = threadIdx.x; i < nn; i += 256) lds[i] = src[i];
      __syncthreads();
    }
  }
  if (inst >= d.n_inst) return;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (FTZ/DAZ render scope)

  // recurrence state of every biquad op of the chain lives in LDS: [MAX_OPS][C][4] doubles
  double* carry_all = reinterpret_cast<double*>(lds + C * 64 * LDS_ROW);
  if constexpr (SERIAL) {
    for (int o = 0; o < d.n_ops; o++) {
      if (d.ops[o].kind == OP_BIQUAD) {
        const double* st = reinterpret_cast<const double*>(d.ops[o].ptr1) + (uint64_t)inst * STATE_STRIDE;
        if (lane < C * 4) carry_all[o * C * 4 + lane] = (lane >> 2) < d.ops[o].nch_in ? st[lane] : 0.;
      }
    }
    __syncthreads();
  }

  uint32_t done_in_block = 0;
  for (uint32_t tile = tile_first; tile < tile_last; tile += (PERSIST ? tile_step : 1u)) {
    if constexpr (PERSIST) {
      if (per_block) {
        if (done_in_block == per_block) {
          __syncthreads();  // (waits for this wavefront's stores, then for the other three: the block is in L2 / L1)
          done_in_block = 0;
        }
        done_in_block++;
      }
    }
    float v[C][K];
    // ---- inputs: mix every incoming edge to the node's computed channel count and sum in edge order
    if constexpr (!FANIN) {
      load_input<C, K>(d.in[0], inst, tile, lane, d.n_quanta, v);
      if (d.in[0].has_gain) gain_regs<C, K>(v, d.in[0].nch, d.in[0].gain, inst, tile, lane, d.n_quanta);
      mix_regs<C, K>(v, d.in[0].nch, d.in_nch, d.in_interp);
    } else {
      // one copy of the input fetch for all edges (not an unrolled first edge + loop): half the registers
#pragma unroll
      for (int c = 0; c < C; c++) {
#pragma unroll
        for (int i = 0; i < K; i++) v[c][i] = 0.f;
      }
#pragma nounroll
      for (int k = 0; k < d.n_inputs; k++) {
        float u[C][K];
        int lane_k = lane;  // opaque per iteration: per-lane address terms are recomputed, not kept live (LICM)
        asm volatile("" : "+v"(lane_k));
        load_input<C, K>(d.in[k], inst, tile, lane_k, d.n_quanta, u);
        if (d.in[k].has_gain) gain_regs<C, K>(u, d.in[k].nch, d.in[k].gain, inst, tile, lane_k, d.n_quanta);
        mix_regs<C, K>(u, d.in[k].nch, d.in_nch, d.in_interp);
#pragma unroll
        for (int c = 0; c < C; c++)
          if (c < d.in_nch) {
#pragma unroll
            for (int i = 0; i < K; i++) v[c][i] = k == 0 ? u[c][i] : v[c][i] + u[c][i];
          }
      }
    }
    // ---- fused node ops
    for (int o = 0; o < d.n_ops; o++) {
      const OpDesc& op = d.ops[o];
      switch (op.kind) {
        case OP_GAIN:
          gain_regs<C, K>(v, op.nch_in, op.p0, inst, tile, lane, d.n_quanta);
          break;
        case OP_PARAM_ADD: {
          // AudioParamProcessor::mix_to_output (param.rs:737-795): input (already mixed to one channel) + intrinsic
          // value, NaN -> default, clamp with max/min (not `clamp`: no NaN branch)
          const float pmin = __int_as_float(op.i0), pmax = __int_as_float(op.i1), pdef = __int_as_float(op.i2);
#pragma unroll
          for (int j = 0; j < NV4; j++) {
            const uint32_t q = tile * QPT + j * 2 + (lane >> 5);
            const uint32_t qc = q < d.n_quanta ? q : d.n_quanta - 1;
            const uint64_t f = (uint64_t)tile * TILE_FR + j * 256 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const uint64_t fc = f + e < (uint64_t)d.n_quanta * RQ ? f + e : (uint64_t)d.n_quanta * RQ - 1;
              float o1 = v[0][j * 4 + e] + param_at(op.p0, inst, qc, fc);
              o1 = o1 != o1 ? pdef : fminf(fmaxf(o1, pmin), pmax);
              v[0][j * 4 + e] = o1;
            }
          }
          break;
        }
        case OP_BIQUAD: {
          if constexpr (SERIAL) {
            const double* coef = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0;
            biquad_op<C>(op, inst, tile, lane, d.n_quanta, lds, v, carry_all + o * C * 4, coef);
          }
          break;
        }
        case OP_WAVESHAPER: {
          if (!SERIAL && o == d.lds_curve_op) {
            // (indexing `lds` directly keeps the LDS address space: ds_read, not flat loads)
#pragma unroll
            for (int c = 0; c < C; c++)
              if (c < op.nch_in) {
#pragma unroll
                for (int i = 0; i < K; i++)
                  v[c][i] = apply_curve_lds((const __attribute__((address_space(3))) float*)lds, op.i0, v[c][i]);
              }
          } else {
            const float* curve = reinterpret_cast<const float*>(op.ptr0);
#pragma unroll
            for (int c = 0; c < C; c++)
              if (c < op.nch_in) {
#pragma unroll
                for (int i = 0; i < K; i++) v[c][i] = apply_curve(curve, op.i0, v[c][i]);
              }
          }
          break;
        }
        case OP_STEREO_PAN: {
          if constexpr (C >= 2) {
#pragma unroll
            for (int j = 0; j < NV4; j++) {
              const uint32_t q = tile * QPT + j * 2 + (lane >> 5);
              const uint32_t qc = q < d.n_quanta ? q : d.n_quanta - 1;
              const uint64_t f = (uint64_t)tile * TILE_FR + j * 256 + lane * 4;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const uint64_t fc = f + e < (uint64_t)d.n_quanta * RQ ? f + e : (uint64_t)d.n_quanta * RQ - 1;
                const float pan = param_at(op.p0, inst, qc, fc);
                float gl, gr;
                if (op.p0.mode == 2) {
                  const float x = op.nch_in == 1 ? (pan + 1.f) * 0.5f : (pan <= 0.f ? pan + 1.f : pan);
                  stereo_gains_dev(x, gl, gr);
                } else {
                  gl = param_at(op.p1, inst, qc, fc);
                  gr = param_at(op.p2, inst, qc, fc);
                }
                if (op.nch_in == 1) {
                  const float in = v[0][j * 4 + e];
                  v[0][j * 4 + e] = in * gl;
                  v[1][j * 4 + e] = in * gr;
                } else {
                  const float il = v[0][j * 4 + e], ir = v[1][j * 4 + e];
                  if (pan <= 0.f) {
                    v[0][j * 4 + e] = __builtin_fmaf(ir, gl, il);
                    v[1][j * 4 + e] = ir * gr;
                  } else {
                    v[0][j * 4 + e] = il * gl;
                    v[1][j * 4 + e] = __builtin_fmaf(il, gr, ir);
                  }
                }
              }
            }
          }
          break;
        }
        case OP_PANNER: {
          if constexpr (C >= 2) {
#pragma unroll
            for (int j = 0; j < NV4; j++) {
              const uint32_t q = tile * QPT + j * 2 + (lane >> 5);
              const uint32_t qc = q < d.n_quanta ? q : d.n_quanta - 1;
              float az = param_at(op.p0, inst, qc, 0);
              float gl = param_at(op.p1, inst, qc, 0), gr = param_at(op.p2, inst, qc, 0);
              float dg = param_at(op.p3, inst, qc, 0), cg = param_at(op.p4, inst, qc, 0);
              const bool per_frame = op.p0.mode == 2;  // audio-rate AudioListener: tables of waa_panner.hip
              const uint64_t f = (uint64_t)tile * TILE_FR + j * 256 + lane * 4;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                if (per_frame) {
                  const uint64_t fc = f + e < (uint64_t)d.n_quanta * RQ ? f + e : (uint64_t)d.n_quanta * RQ - 1;
                  az = param_at(op.p0, inst, qc, fc);
                  gl = param_at(op.p1, inst, qc, fc);
                  gr = param_at(op.p2, inst, qc, fc);
                  dg = param_at(op.p3, inst, qc, fc);
                  cg = param_at(op.p4, inst, qc, fc);
                }
                if (op.nch_in == 1) {
                  // panner.rs:988-1014 (after the mono -> stereo up-mix l = r = in)
                  const float in = v[0][j * 4 + e];
                  v[0][j * 4 + e] = in * (gl * dg * cg);
                  v[1][j * 4 + e] = in * (gr * dg * cg);
                } else {
                  // panner.rs:1016-1057
                  const float il = v[0][j * 4 + e], ir = v[1][j * 4 + e];
                  if (az <= 0.f) {
                    v[0][j * 4 + e] = (il + ir * gl) * dg * cg;
                    v[1][j * 4 + e] = ir * gr * dg * cg;
                  } else {
                    v[0][j * 4 + e] = il * gl * dg * cg;
                    v[1][j * 4 + e] = (ir + il * gr) * dg * cg;
                  }
                }
              }
            }
          }
          break;
        }
        case OP_MIX:
          mix_regs<C, K>(v, op.nch_in, op.nch_out, op.i0);
          break;
        default:
          break;
      }
    }
    // ---- store
#pragma unroll
    for (int c = 0; c < C; c++) {
      if (c < d.out.nch) {
        float* p = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride + (uint64_t)tile * TILE_FR;
#pragma unroll
        for (int j = 0; j < NV4; j++)
          *reinterpret_cast<float4*>(p + j * 256 + lane * 4) =
              make_float4(v[c][j * 4 + 0], v[c][j * 4 + 1], v[c][j * 4 + 2], v[c][j * 4 + 3]);
      }
    }
  }
  // persist recurrence state (lets a later render range continue; also what tail logic would inspect)
  if constexpr (!SERIAL) return;
  __syncthreads();
  for (int o = 0; o < d.n_ops; o++) {
    if (d.ops[o].kind == OP_BIQUAD) {
      double* st = reinterpret_cast<double*>(d.ops[o].ptr1) + (uint64_t)inst * STATE_STRIDE;
      if (lane < C * 4 && (lane >> 2) < d.ops[o].nch_in) st[lane] = carry_all[o * C * 4 + lane];
    }
  }
}

// ---- per-frame biquad coefficients (biquad_filter.rs:28-373 calculate_coefs, get_computed_freq) in f64 ----
__device__ __forceinline__ void norm_coefs(double b0, double b1, double b2, double a0, double a1, double a2, double* o) {
  const double s = 1. / a0;
  o[0] = b0 * s;
  o[1] = b1 * s;
  o[2] = b2 * s;
  o[3] = a1 * s;
  o[4] = a2 * s;
}
__device__ __forceinline__ void raw_coefs(double b0, double* o) {
  o[0] = b0;
  o[1] = o[2] = o[3] = o[4] = 0.;
}
__global__ __launch_bounds__(256) void biquad_coef_kernel(const BiquadCoefDesc d) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (uint64_t)d.rows * d.frames_padded) return;
  const uint32_t inst = (uint32_t)(idx / d.frames_padded);
  const uint64_t r = idx % d.frames_padded;
  uint64_t frame = r, slot = r * 5, cstep = 1;
  if (d.lane_major) {
    // a wave covers the 64 lanes of one (tile, k): the streaming kernel's lane l owns frames tile*2048 + l*32 + k, and
    // its loads of coefficient c for step k are 64 consecutive doubles
    const uint64_t tile = r / TILE, in = r % TILE, k = in / 64, lane = in % 64;
    frame = tile * TILE + lane * TILE_K + k;
    slot = ((tile * TILE_K + k) * 5) * 64 + lane;
    cstep = 64;
  }
  if (frame >= d.n_frames) frame = d.n_frames - 1;  // padded tail of the last tile: any finite set will do
  const uint32_t q = (uint32_t)(frame / RQ);
  const float freq = param_at(d.frequency, inst, q, frame), det = param_at(d.detune, inst, q, frame);
  const double Q = (double)param_at(d.q, inst, q, frame), gain = (double)param_at(d.gain, inst, q, frame);
  const float cf = det != 0.f ? freq * exp2f(det / 1200.f) : freq;  // get_computed_freq: f32
  const double PI = 3.14159265358979323846;
  const double nyq = (double)d.sample_rate / 2.;
  double f = (double)cf / nyq;
  f = f < 0. ? 0. : f > 1. ? 1. : f;
  double o[5];
  const double A = pow(10., gain / 40.);
  const double w0 = PI * f, sw = sin(w0), cw = cos(w0);
  switch (d.type) {
    case 0: {  // lowpass
      if (f == 1.) { raw_coefs(1., o); break; }
      const double al = sw / (2. * pow(10., Q / 20.)), be = (1. - cw) / 2.;
      norm_coefs(be, 2. * be, be, 1. + al, -2. * cw, 1. - al, o);
      break;
    }
    case 1: {  // highpass
      if (f == 1.) { raw_coefs(0., o); break; }
      if (f == 0.) { raw_coefs(1., o); break; }
      const double al = sw / (2. * pow(10., Q / 20.)), be = (1. + cw) / 2.;
      norm_coefs(be, -2. * be, be, 1. + al, -2. * cw, 1. - al, o);
      break;
    }
    case 2: {  // bandpass
      if (!(f > 0. && f < 1.)) { raw_coefs(0., o); break; }
      if (!(Q > 0.)) { raw_coefs(1., o); break; }
      const double al = sw / (2. * Q);
      norm_coefs(al, 0., -al, 1. + al, -2. * cw, 1. - al, o);
      break;
    }
    case 3: {  // notch
      if (!(f > 0. && f < 1.)) { raw_coefs(1., o); break; }
      if (!(Q > 0.)) { raw_coefs(0., o); break; }
      const double al = sw / (2. * Q);
      norm_coefs(1., -2. * cw, 1., 1. + al, -2. * cw, 1. - al, o);
      break;
    }
    case 4: {  // allpass
      if (!(f > 0. && f < 1.)) { raw_coefs(1., o); break; }
      if (!(Q > 0.)) { raw_coefs(-1., o); break; }
      const double al = sw / (2. * Q);
      norm_coefs(1. - al, -2. * cw, 1. + al, 1. + al, -2. * cw, 1. - al, o);
      break;
    }
    case 5: {  // peaking
      if (!(f > 0. && f < 1.)) { raw_coefs(1., o); break; }
      if (!(Q > 0.)) { raw_coefs(A * A, o); break; }
      const double al = sw / (2. * Q);
      norm_coefs(1. + al * A, -2. * cw, 1. - al * A, 1. + al / A, -2. * cw, 1. - al / A, o);
      break;
    }
    case 6: {  // lowshelf
      if (f == 1.) { raw_coefs(A * A, o); break; }
      if (f == 0.) { raw_coefs(1., o); break; }
      const double as = sw / 2. * 1.41421356237309504880, k = 2. * as * sqrt(A), ap = A + 1., am = A - 1.;
      norm_coefs(A * (ap - am * cw + k), 2. * A * (am - ap * cw), A * (ap - am * cw - k), ap + am * cw + k,
                 -2. * (am + ap * cw), ap + am * cw - k, o);
      break;
    }
    default: {  // highshelf
      if (f == 1.) { raw_coefs(1., o); break; }
      if (!(f > 0.)) { raw_coefs(A * A, o); break; }
      const double as = sw / 2. * 1.41421356237309504880, k = 2. * as * sqrt(A), ap = A + 1., am = A - 1.;
      norm_coefs(A * (ap + am * cw + k), -2. * A * (am + ap * cw), A * (ap + am * cw - k), ap - am * cw + k,
                 2. * (am - ap * cw), ap - am * cw - k, o);
      break;
    }
  }
  double* out = d.coefs + (uint64_t)inst * d.frames_padded * 5 + slot;
#pragma unroll
  for (int c = 0; c < 5; c++) out[(uint64_t)c * cstep] = o[c];
}
void launch_biquad_coefs(const BiquadCoefDesc& d, void* stream) {
  const uint64_t total = (uint64_t)d.rows * d.frames_padded;
  hipLaunchKernelGGL(biquad_coef_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d);
}

// digest of a shared per-frame coefficient table (see BiquadHpDesc): one thread per (tile, lane), 32 frames backwards
__global__ __launch_bounds__(64) void biquad_hp_kernel(const BiquadHpDesc d) {
  const uint32_t tile = blockIdx.x;
  const int lane = threadIdx.x;
  const double* ct = d.coefs + (uint64_t)tile * (TILE_K * 5 * 64) + lane;  // element (k, coef) at ct[(k * 5 + coef) * 64]
  double* out = d.hp + (uint64_t)tile * (HP_WORDS * 64) + lane;
  // Phi_i = M_31 ... M_{i+1}; G_i = its first column; Phi_{i-1} = Phi_i M_i with M_i = [[-a1, -a2], [1, 0]]
  double pa = 1., pb = 0., pc = 0., pd = 1.;
  double g1a = 0., g1c = 0., g2a = 0., g2c = 0.;  // G_{i+1}, G_{i+2}
  double b1n = 0., b2n = 0., b2nn = 0.;           // b1_{i+1}, b2_{i+1}, b2_{i+2}
  for (int i = TILE_K - 1; i >= 0; i--) {
    const double b0 = ct[(i * 5 + 0) * 64], b1 = ct[(i * 5 + 1) * 64], b2 = ct[(i * 5 + 2) * 64], a1 = ct[(i * 5 + 3) * 64],
                 a2 = ct[(i * 5 + 4) * 64];
    // H_i = G_i b0_i + G_{i+1} b1_{i+1} + G_{i+2} b2_{i+2}
    out[(2 * i) * 64] = pa * b0 + g1a * b1n + g2a * b2nn;
    out[(2 * i + 1) * 64] = pc * b0 + g1c * b1n + g2c * b2nn;
    if (i == 0) {
      out[64 * 64] = pa * b1 + g1a * b2n;  // Hm1 = G_0 b1_0 + G_1 b2_1
      out[65 * 64] = pc * b1 + g1c * b2n;
      out[66 * 64] = pa * b2;              // Hm2 = G_0 b2_0
      out[67 * 64] = pc * b2;
    }
    g2a = g1a;
    g2c = g1c;
    g1a = pa;
    g1c = pc;
    b2nn = b2n;
    b1n = b1;
    b2n = b2;
    // Phi <- Phi M_i: columns (a, c | b, d): new first column = -a1 * col1 + col2, new second column = -a2 * col1
    const double na = __builtin_fma(-a1, pa, pb), nc = __builtin_fma(-a1, pc, pd);
    pb = -a2 * pa;
    pd = -a2 * pc;
    pa = na;
    pc = nc;
  }
  out[68 * 64] = pa;
  out[69 * 64] = pb;
  out[70 * 64] = pc;
  out[71 * 64] = pd;
}
void launch_biquad_hp(const BiquadHpDesc& d, void* stream) {
  hipLaunchKernelGGL(biquad_hp_kernel, dim3(d.n_tiles), dim3(64), 0, (hipStream_t)stream, d);
}

__global__ void quantum_heads_kernel(const float* src, uint64_t inst_stride, uint32_t n_inst, uint32_t n_quanta, float* dst) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (uint64_t)n_inst * n_quanta) return;
  const uint64_t inst = idx / n_quanta, q = idx % n_quanta;
  dst[idx] = src[inst * inst_stride + q * RQ];
}
void launch_quantum_heads(const float* src, uint64_t inst_stride, uint32_t n_inst, uint32_t n_quanta, float* dst, void* stream) {
  const uint64_t count = (uint64_t)n_inst * n_quanta;
  hipLaunchKernelGGL(quantum_heads_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, inst_stride,
                     n_inst, n_quanta, dst);
}

void launch_chain(const ChainDesc& d, int cmax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  bool serial = false;
  for (int o = 0; o < d.n_ops; o++) serial |= d.ops[o].kind == OP_BIQUAD;
  if (serial) {
    dim3 grid(d.n_inst), block(64);
    if (cmax <= 1)
      hipLaunchKernelGGL((chain_kernel<1, TILE_K, true>), grid, block, 1 * 64 * LDS_ROW * sizeof(float) + CARRY_BYTES, s, d);
    else
      hipLaunchKernelGGL((chain_kernel<2, TILE_K, true>), grid, block, 2 * 64 * LDS_ROW * sizeof(float) + CARRY_BYTES, s, d);
  } else {
    int curve_op = -1;
    if (resample_shape(d, &curve_op)) {  // source [-> WaveShaper] -> signal: the specialised kernel of waa_resample.hip
      launch_resample(d, curve_op, stream);
      return;
    }
    ChainDesc dd = d;
    dd.lds_curve_op = -1;
    dd.tile_major = 0;
    for (int k = 0; k < d.n_inputs; k++) dd.tile_major |= d.in[k].kind == IN_SOURCE;
    if (measure_switch("WAA_NO_TILE_MAJOR")) dd.tile_major = 0;  // measurement aid
    // (same-batch A/B, tools/placement_probe.py with ALT=WAA_NO_XCD_REMAP=1: echo 2.01 against 2.06 ms, the pan stage of C4
    // 1.39-1.44 against 1.44-1.46 ms)
    dd.xcd_remap = !dd.tile_major && !measure_switch("WAA_NO_XCD_REMAP");
    for (int o = 0; o < d.n_ops; o++)
      if (d.ops[o].kind == OP_WAVESHAPER && d.ops[o].i0 > 0 && d.ops[o].i0 <= 8192) {
        dd.lds_curve_op = o;
        break;
      }
    const size_t lds = dd.lds_curve_op >= 0 ? (size_t)d.ops[dd.lds_curve_op].i0 * sizeof(float) : 0;
    // (8 frames per lane instead of 4 was measured: fewer waves fit per SIMD and every workload got slower.  So were 2 / 4 / 8
    // consecutive sub-tiles per wavefront, round 3: echo 2.06-2.14 -> 2.11-2.29 ms, C4's pan stage 1.40 -> 1.75 ms — it is
    // not the number of waves launched; a wavefront that walks its stream needs the next sub-tile in flight, see waa_echo.hip)
    const uint64_t waves = (uint64_t)d.n_inst * (d.tile1 - d.tile0) * (TILE / 256);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (d.persist_block) {  // (mono / stereo only: the caller checks)
      grid = dim3(d.n_inst);
      dd.tile_major = 0;
      dd.xcd_remap = 0;
      if (d.n_inputs <= 1) {
        if (cmax <= 1)
          hipLaunchKernelGGL((chain_kernel<1, 4, false, false, true>), grid, block, lds, s, dd);
        else
          hipLaunchKernelGGL((chain_kernel<2, 4, false, false, true>), grid, block, lds, s, dd);
      } else if (cmax <= 1) {
        hipLaunchKernelGGL((chain_kernel<1, 4, false, true, true>), grid, block, lds, s, dd);
      } else {
        hipLaunchKernelGGL((chain_kernel<2, 4, false, true, true>), grid, block, lds, s, dd);
      }
      return;
    }
    if (d.n_inputs <= 1) {
      if (cmax <= 1)
        hipLaunchKernelGGL((chain_kernel<1, 4, false, false>), grid, block, lds, s, dd);
      else if (cmax <= 2)
        hipLaunchKernelGGL((chain_kernel<2, 4, false, false>), grid, block, lds, s, dd);
      else if (cmax <= 4)
        hipLaunchKernelGGL((chain_kernel<4, 4, false, false>), grid, block, lds, s, dd);
      else
        hipLaunchKernelGGL((chain_kernel<6, 4, false, false>), grid, block, lds, s, dd);
    } else if (cmax <= 1)
      hipLaunchKernelGGL((chain_kernel<1, 4, false>), grid, block, lds, s, dd);
    else if (cmax <= 2) {
      // The six-wave form keeps four registers in scratch memory, on the per-frame panning path.  Twice in 120 000 random
      // graphs rendered by eight processes on one device, 16 samples of the right channel behind an a-rate-automated
      // StereoPanner — lanes 16..31, element 3: ONE 64-byte piece of one spilled register's scratch row — came out wrong in
      // a first render and right in the next.  Unproven, but cheap to rule out: chains that pan with
      // per-frame values take the five-wave instantiation, which has no scratch at all (plain sums keep six waves: the
      // five-wave form costs them 9-10 %, same-box A/B on echo / fb).
      bool frame_pan = false;
      for (int o = 0; o < d.n_ops; o++)
        frame_pan |= (d.ops[o].kind == OP_STEREO_PAN || d.ops[o].kind == OP_PANNER) && d.ops[o].p0.mode == 2;
      if (frame_pan || measure_switch("WAA_CHAIN_NOSPILL"))
        hipLaunchKernelGGL((chain_kernel<2, 4, false, true, false, true>), grid, block, lds, s, dd);
      else
        hipLaunchKernelGGL((chain_kernel<2, 4, false>), grid, block, lds, s, dd);
    } else if (cmax <= 4)
      hipLaunchKernelGGL((chain_kernel<4, 4, false>), grid, block, lds, s, dd);
    else
      hipLaunchKernelGGL((chain_kernel<6, 4, false>), grid, block, lds, s, dd);
  }
}

}  // namespace waa
