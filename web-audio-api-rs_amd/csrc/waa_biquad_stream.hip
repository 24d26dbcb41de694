// waa_biquad_stream.hip — the streaming biquad kernel (BASELINE config C2 and the Biquad stage of T1/C4).
//
// One 64-lane wavefront renders one (instance, channel) stream for the whole duration, 2048 frames
// (16 render quanta) per tile:
//   global (16 B/lane, coalesced, next tile prefetched in registers)
//     -> LDS transpose -> 32 consecutive frames per lane
//     -> FIR part w = (b0*x + b1*x1) + b2*x2 in the reference's evaluation order (f64, unfused)
//     -> zero-state recurrence per lane, then the true incoming state of every lane from a wavefront
//        scan: 4 DPP row_shr steps with the uniform matrix powers A, A^2, A^4, A^8 (A = M^32), row
//        carries through A^16, per-lane A^(lane%16)
//     -> final pass y = (w - a1*y1) - a2*y2 in the reference's order (biquad_filter.rs:877)
//     -> LDS transpose back -> gains -> global store (16 B/lane).
// f64 denormals are flushed by hardware mode (the reference renders under FTZ/DAZ, thread.rs:374-382);
// the explicit `!y.is_normal() -> 0` of biquad_filter.rs:881-883 then only matters for inf/NaN, which is
// detected off the critical path and handled by re-running the lane's pass with the flush.
// HBM-bound by design (8 B of traffic per frame-channel); no MFMA.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"
#include "waa_stream_common.hpp"

namespace waa {

namespace {
struct M2 {
  double a, b, c, d;
};
__device__ __forceinline__ M2 mm(const M2& x, const M2& y) {
  M2 r;
  r.a = __builtin_fma(x.a, y.a, x.b * y.c);
  r.b = __builtin_fma(x.a, y.b, x.b * y.d);
  r.c = __builtin_fma(x.c, y.a, x.d * y.c);
  r.d = __builtin_fma(x.c, y.b, x.d * y.d);
  return r;
}

}  // namespace

// (WAA_STREAM_DEBUG=7, tools/dispatch_probe.py: where and when every wavefront of the last launch ran)
__device__ unsigned long long g_stream_trace[8192 * 4];

// DBG is a measurement aid (WAA_STREAM_DEBUG): 0 = product kernel, 1 = same memory pattern without the recurrence,
// 2 = recurrence without the stores (results are wrong by construction in modes 1 and 2)
// VARY = 1: coefficients change per render quantum (k-rate automation; d.coefs holds n_quanta sets per instance):
// per-lane coefficients (4 lanes share a quantum), per-lane A = M^32, general scan of the affine maps.
// VARY = 2: coefficients change per FRAME (a-rate params, biquad_filter.rs:837-855).  The table of biquad_coef_kernel is
// lane-major — element (tile, k, coef, lane) — so step k of all 64 lanes reads 64 consecutive doubles.  A lane streams
// its 32 coefficient sets twice (zero-state pass with the transition product P = M_31 ... M_0, then the exact-order
// pass from the true incoming state) instead of holding 160 doubles in registers; the table is one per batch when the
// automation is the same for every instance (then it lives in L2 / Infinity Cache and HBM only carries the samples).
// DUP: a mono stream whose result goes to both channels of a stereo signal (the speakers up-mix 1 -> 2 of the node behind it,
// quantum.rs:330-340): the second store instead of a chain launch that reads the mono signal back and writes two copies.
template <int DBG, int VARY, int NBUF = 2, bool DUP = false>
__global__ __launch_bounds__(64, 2) void biquad_stream_kernel_t(const BiquadStreamDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const uint32_t wid = blockIdx.x;
  const uint32_t inst = wid / (uint32_t)d.nch;
  const int ch = (int)(wid % (uint32_t)d.nch);
  const int lane = threadIdx.x;
  if (inst >= d.n_inst) return;

  // f64 (and f16) denormals: flush inputs and outputs, like the reference's FTZ/DAZ render scope.
  // hwreg(HW_REG_MODE = 1, offset 6, width 2)
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  unsigned long long trace_t0 = 0;
  if constexpr (DBG == 7) trace_t0 = wall_clock64();

  // coefficients: uniform per wave (constant params) or reloaded per tile and lane (VARY)
  const double* cp = d.coefs + (uint64_t)inst * d.coef_stride;
  double b0 = cp[0], b1 = cp[1], b2 = cp[2], a1 = cp[3], a2 = cp[4];
  // matrix powers of the 32-step state transition: A = M^32, M = [[-a1, -a2], [1, 0]] on (y[n-1], y[n-2])
  M2 A1;
  {
    M2 m = {-a1, -a2, 1., 0.};
#pragma unroll
    for (int s = 0; s < 5; s++) m = mm(m, m);
    A1 = m;
  }
  const M2 A2 = mm(A1, A1), A4 = mm(A2, A2), A8 = mm(A4, A4), A16 = mm(A8, A8);
  // per-lane A^(lane % 16)
  M2 Aj = {1., 0., 0., 1.};
  if constexpr (VARY == 0) {
    const int j = lane & 15;
    if (j & 1) Aj = mm(Aj, A1);
    if (j & 2) Aj = mm(Aj, A2);
    if (j & 4) Aj = mm(Aj, A4);
    if (j & 8) Aj = mm(Aj, A8);
  }
  const int row = lane >> 4;

  // carried state (uniform): x[n-1], x[n-2], y[n-1], y[n-2]
  double* st = d.state + (uint64_t)inst * STATE_STRIDE + ch * 4;
  double cx1 = st[0], cx2 = st[1], cy1 = st[2], cy2 = st[3];

  // gains (gain.rs:163-179 fast paths are per render quantum with a constant gain => per kernel here)
  float g[2] = {1.f, 1.f};
  bool g_mute[2] = {false, false}, g_pass[2] = {true, true};
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (k < d.n_gain) {
      g[k] = d.gain[k].base[inst];
      g_mute[k] = fabsf(g[k]) <= 1e-6f;
      g_pass[k] = fabsf(1.f - g[k]) <= 1e-6f;
    }

  // input addressing
  const bool is_src = d.in.kind == IN_SOURCE;
  SrcInst si{};
  SrcSchedule sc{};
  const float* sig_base = nullptr;
  if (is_src) {
    si = d.in.src[inst];
    sc = si.sc;
  } else {
    sig_base = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
  }
  float* out_base = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride;

  // One workgroup = one wavefront: LDS hand-offs between lanes only need program order (DS operations of a
  // wave execute in order), so a compiler-level wave barrier replaces __syncthreads() — whose fences would
  // drain every outstanding global load/store (s_waitcnt vmcnt(0)) and serialise the software pipeline.
  auto lds_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };

  auto fetch_fast = [&](uint32_t tile, float (&dst)[TILE_K]) __attribute__((always_inline)) {
    // (inside the linear prefix the start of a tile is arithmetic; behind it, it comes from the schedule table)
    const float* p = is_src ? si.base + (uint64_t)ch * si.ch_stride +
                                  (tile < si.fast_prefix ? si.linear_start + (int64_t)tile * TILE
                                                         : load_global(&sc.qrec[(uint64_t)tile * QUANTA_PER_TILE].start))
                            : sig_base + (uint64_t)tile * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      // streamed once: non-temporal (same-box A/B on C2, tools/ab_env.py WAA_STREAM_DEBUG=5: 1.588 against 1.610 ms, ten
      // alternations; stores alone make no difference)
      const f4v t = DBG == 5 ? load_global_f4(p + j * 256 + lane * 4)
                             : __builtin_nontemporal_load((const WAA_GLOBAL_AS f4v*)(p + j * 256 + lane * 4));
      dst[j * 4 + 0] = t.x;
      dst[j * 4 + 1] = t.y;
      dst[j * 4 + 2] = t.z;
      dst[j * 4 + 3] = t.w;
    }
  };
  auto tile_is_fast = [&](uint32_t tile) __attribute__((always_inline)) -> bool {
    return !is_src || tile < si.fast_prefix || (si.aligned && load_global(sc.tile_fast + tile));
  };

  // everything after the input fetch: transposes, recurrence, gains, store
  // stage the tile (A layout registers) into LDS
  auto stage = [&](const float (&cur)[TILE_K]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      *reinterpret_cast<float4*>(lds + r * LDS_ROW + c) =
          make_float4(cur[j * 4 + 0], cur[j * 4 + 1], cur[j * 4 + 2], cur[j * 4 + 3]);
    }
  };
  float* lds_out = lds + 64 * LDS_ROW;  // second buffer: results of the previous tile, stored one iteration later
  // store the tile whose results sit in lds_out (T layout rows) — gains applied on the way out
  auto flush = [&](uint32_t tile) __attribute__((always_inline)) {
    lds_sync();
    float* op = out_base + (uint64_t)tile * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      float4 t = *reinterpret_cast<const float4*>(lds_out + r * LDS_ROW + c);
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (k < d.n_gain) {
          if (g_mute[k]) {
            t = make_float4(0.f, 0.f, 0.f, 0.f);
          } else if (!g_pass[k]) {
            t.x *= g[k];
            t.y *= g[k];
            t.z *= g[k];
            t.w *= g[k];
          }
        }
      if constexpr (DBG == 2) {
        asm volatile("" ::"v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
      } else if constexpr (DBG == 5) {  // (A/B: the plain loads and stores)
        *reinterpret_cast<float4*>(op + j * 256 + lane * 4) = t;
      } else {
        __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, (WAA_GLOBAL_AS f4v*)(op + j * 256 + lane * 4));
        if constexpr (DUP)
          __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, (WAA_GLOBAL_AS f4v*)(op + d.out.ch_stride + j * 256 + lane * 4));
      }
    }
    lds_sync();
  };
  // ---- a-rate params: per-frame coefficient sets streamed from the lane-major table, two sweeps per tile
  auto process_arate = [&](uint32_t tile) __attribute__((always_inline)) {
    lds_sync();
    const float* xrow = lds + lane * LDS_ROW;     // this lane's 32 frames (T layout), stable during the tile
    float* yrow = lds_out + lane * LDS_ROW;
    const double* ct = cp + (uint64_t)tile * (TILE_K * 5 * 64) + lane;  // element (k, coef) at ct[(k * 5 + coef) * 64]
    auto load_chunk = [&](double (&c)[20], int chunk) __attribute__((always_inline)) {
      if (DBG >= 3 && tile != d.tile0) return;  // (measurement aid: the coefficient sets of the first tile for all tiles)
#pragma unroll
      for (int j = 0; j < 20; j++) c[j] = load_global(ct + (chunk * 20 + j) * 64);
    };
    // x history at the chunk boundary
    const float xl1 = xrow[TILE_K - 1], xl2 = xrow[TILE_K - 2];
    const float xm1 = __shfl_up(xl1, 1, 64), xm2 = __shfl_up(xl2, 1, 64);
    const double xs1 = lane == 0 ? cx1 : (double)xm1, xs2 = lane == 0 ? cx2 : (double)xm2;
    // sweep 1: zero-state response of the chunk and its transition P = M_31 ... M_0, M_i = [[-a1, -a2], [1, 0]]
    double x1 = xs1, x2 = xs2, z1 = 0., z2 = 0.;
    M2 P = {1., 0., 0., 1.};
    double ca[20], cb[20], cc[NBUF == 3 ? 20 : 1];
    if constexpr (VARY == 3) {
      // shared table: the end state is a 34-tap dot product with the digest of biquad_hp_kernel (BiquadHpDesc)
      const double* hp = d.hp + (uint64_t)(DBG == 4 ? d.tile0 : tile) * (HP_WORDS * 64) + lane;
      load_chunk(ca, 0);  // sweep 2's first coefficient sets are requested before the dot product starts
      load_chunk(cb, 1);
      double za = load_global(hp + 64 * 64) * xs1, zb = load_global(hp + 65 * 64) * xs1;
      double zc = load_global(hp + 66 * 64) * xs2, zd = load_global(hp + 67 * 64) * xs2;
#pragma unroll 2
      for (int chunk = 0; chunk < TILE_K / 4; chunk++) {
        const float4 xv = *reinterpret_cast<const float4*>(xrow + chunk * 4);
        const double* hc = hp + chunk * 8 * 64;
        const double h0 = load_global(hc), h1 = load_global(hc + 64), h2 = load_global(hc + 128), h3 = load_global(hc + 192),
                     h4 = load_global(hc + 256), h5 = load_global(hc + 320), h6 = load_global(hc + 384), h7 = load_global(hc + 448);
        za = __builtin_fma(h0, (double)xv.x, za);
        zb = __builtin_fma(h1, (double)xv.x, zb);
        zc = __builtin_fma(h2, (double)xv.y, zc);
        zd = __builtin_fma(h3, (double)xv.y, zd);
        za = __builtin_fma(h4, (double)xv.z, za);
        zb = __builtin_fma(h5, (double)xv.z, zb);
        zc = __builtin_fma(h6, (double)xv.w, zc);
        zd = __builtin_fma(h7, (double)xv.w, zd);
      }
      z1 = za + zc;
      z2 = zb + zd;
      P.a = load_global(hp + 68 * 64);
      P.b = load_global(hp + 69 * 64);
      P.c = load_global(hp + 70 * 64);
      P.d = load_global(hp + 71 * 64);
    }
    auto step_a = [&](const double (&c)[20], int chunk) __attribute__((always_inline)) {
      const float4 xv = *reinterpret_cast<const float4*>(xrow + chunk * 4);
      const float xf[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const double cb0 = c[j * 5], cb1 = c[j * 5 + 1], cb2 = c[j * 5 + 2], ca1 = c[j * 5 + 3], ca2 = c[j * 5 + 4];
        const double xd = (double)xf[j];
        const double wi = (cb0 * xd + cb1 * x1) + cb2 * x2;
        x2 = x1;
        x1 = xd;
        const double t = __builtin_fma(-ca2, z2, wi);
        const double y = __builtin_fma(-ca1, z1, t);
        z2 = z1;
        z1 = y;
        const double na = __builtin_fma(-ca1, P.a, -(ca2 * P.c)), nb = __builtin_fma(-ca1, P.b, -(ca2 * P.d));
        P.c = P.a;
        P.d = P.b;
        P.a = na;
        P.b = nb;
      }
    };
    {
      if constexpr (VARY == 2) {
        load_chunk(ca, 0);
#pragma unroll 1
        for (int chunk = 0; chunk < TILE_K / 4; chunk += 2) {
          load_chunk(cb, chunk + 1);
          step_a(ca, chunk);
          load_chunk(ca, chunk + 2 < TILE_K / 4 ? chunk + 2 : 0);  // (the last one is the prefetch of sweep 2)
          step_a(cb, chunk + 1);
        }
        load_chunk(cb, 1);
      }
      // inclusive in-row scan of the composed maps (P, r), row carries, exclusive composite: as in the k-rate path
      double r1 = z1, r2 = z2;
      auto step = [&](auto shr) __attribute__((always_inline)) {
        M2 Q;
        Q.a = shr(P.a, 1.);
        Q.b = shr(P.b, 0.);
        Q.c = shr(P.c, 0.);
        Q.d = shr(P.d, 1.);
        const double q1 = shr(r1, 0.), q2 = shr(r2, 0.);
        r1 = __builtin_fma(P.a, q1, __builtin_fma(P.b, q2, r1));
        r2 = __builtin_fma(P.c, q1, __builtin_fma(P.d, q2, r2));
        P = mm(P, Q);
      };
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<1>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<2>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<4>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<8>(v, idv); });
      double t1[4], t2[4];
      t1[0] = cy1;
      t2[0] = cy2;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int e = 16 * k + 15;
        const double pa = read_lane(P.a, e), pb = read_lane(P.b, e), pc = read_lane(P.c, e), pd = read_lane(P.d, e);
        const double e1 = read_lane(r1, e), e2 = read_lane(r2, e);
        t1[k + 1] = __builtin_fma(pa, t1[k], __builtin_fma(pb, t2[k], e1));
        t2[k + 1] = __builtin_fma(pc, t1[k], __builtin_fma(pd, t2[k], e2));
      }
      const double T1 = row == 0 ? t1[0] : row == 1 ? t1[1] : row == 2 ? t1[2] : t1[3];
      const double T2 = row == 0 ? t2[0] : row == 1 ? t2[1] : row == 2 ? t2[2] : t2[3];
      const double xa = row_shr_keep<1>(P.a, 1.), xb = row_shr_keep<1>(P.b, 0.), xc = row_shr_keep<1>(P.c, 0.),
                   xd = row_shr_keep<1>(P.d, 1.);
      const double x1r = row_shr_keep<1>(r1, 0.), x2r = row_shr_keep<1>(r2, 0.);
      const double s1 = __builtin_fma(xa, T1, __builtin_fma(xb, T2, x1r));
      const double s2 = __builtin_fma(xc, T1, __builtin_fma(xd, T2, x2r));
      // sweep 2: the reference's evaluation order from the true incoming state (biquad_filter.rs:877-883)
      double y1 = s1, y2 = s2;
      float badacc = 0.f;
      auto sweep = [&](bool flush_bad) __attribute__((always_inline)) {
        double p1 = xs1, p2 = xs2;
        auto step_b = [&](const double (&c)[20], int chunk) __attribute__((always_inline)) {
          const float4 xv = *reinterpret_cast<const float4*>(xrow + chunk * 4);
          const float xf[4] = {xv.x, xv.y, xv.z, xv.w};
          float yo[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const double xd2 = (double)xf[j];
            const double wi = (c[j * 5] * xd2 + c[j * 5 + 1] * p1) + c[j * 5 + 2] * p2;
            p2 = p1;
            p1 = xd2;
            double y = (wi - c[j * 5 + 3] * y1) - c[j * 5 + 4] * y2;
            if (flush_bad && !__builtin_isnormal(y)) y = 0.;
            y2 = y1;
            y1 = y;
            yo[j] = (float)y;
            badacc = __builtin_fmaf(yo[j], 0.f, badacc);
          }
          *reinterpret_cast<float4*>(yrow + chunk * 4) = make_float4(yo[0], yo[1], yo[2], yo[3]);
        };
        if constexpr (NBUF == 3) {
          // three rotating buffers: a coefficient set is requested two chunks (8 frames of recurrence) before its use
#pragma unroll 1
          for (int chunk = 0; chunk < 6; chunk += 3) {
            load_chunk(cc, chunk + 2);
            step_b(ca, chunk);
            load_chunk(ca, chunk + 3);
            step_b(cb, chunk + 1);
            load_chunk(cb, chunk + 4);
            step_b(cc, chunk + 2);
          }
          step_b(ca, 6);
          step_b(cb, 7);
        } else {
#pragma unroll 1
          for (int chunk = 0; chunk < TILE_K / 4; chunk += 2) {
            step_b(ca, chunk);
            if (chunk + 2 < TILE_K / 4) load_chunk(ca, chunk + 2);
            step_b(cb, chunk + 1);
            if (chunk + 3 < TILE_K / 4) load_chunk(cb, chunk + 3);
          }
        }
      };
      sweep(false);
      if (__any(badacc != badacc)) {  // inf / NaN somewhere: redo with the explicit flush
        y1 = __builtin_isfinite(s1) ? s1 : 0.;
        y2 = __builtin_isfinite(s2) ? s2 : 0.;
        load_chunk(ca, 0);
        load_chunk(cb, 1);
        sweep(true);
      }
      cx1 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl1), 63));
      cx2 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl2), 63));
      cy1 = read_lane(y1, 63);
      cy2 = read_lane(y2, 63);
    }
    lds_sync();
  };
  auto process = [&](uint32_t tile) __attribute__((always_inline)) {
    lds_sync();
    float x[TILE_K];
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const float4 t = *reinterpret_cast<const float4*>(lds + lane * LDS_ROW + j * 4);
      x[j * 4 + 0] = t.x;
      x[j * 4 + 1] = t.y;
      x[j * 4 + 2] = t.z;
      x[j * 4 + 3] = t.w;
    }
    if constexpr (DBG == 1) {
      lds_sync();
#pragma unroll
      for (int j = 0; j < NV4; j++)
        *reinterpret_cast<float4*>(lds_out + lane * LDS_ROW + j * 4) =
            make_float4(x[j * 4 + 0], x[j * 4 + 1], x[j * 4 + 2], x[j * 4 + 3]);
      lds_sync();
      return;
    }
    M2 Al = A1;  // this lane's 32-step transition
    if constexpr (VARY == 1) {
      uint32_t q = tile * QUANTA_PER_TILE + (lane >> 2);
      if (q >= d.n_quanta) q = d.n_quanta - 1;
      const double* cq = cp + (uint64_t)q * 5;
      b0 = cq[0];
      b1 = cq[1];
      b2 = cq[2];
      a1 = cq[3];
      a2 = cq[4];
      M2 m = {-a1, -a2, 1., 0.};
#pragma unroll
      for (int s = 0; s < 5; s++) m = mm(m, m);
      Al = m;
    }
    // x history at the chunk boundary
    const float xm1 = __shfl_up(x[TILE_K - 1], 1, 64), xm2 = __shfl_up(x[TILE_K - 2], 1, 64);
    double x1 = lane == 0 ? cx1 : (double)xm1;
    double x2 = lane == 0 ? cx2 : (double)xm2;
    // FIR part + zero-state recurrence
    double w[TILE_K];
    double z1 = 0., z2 = 0.;
#pragma unroll
    for (int i = 0; i < TILE_K; i++) {
      const double xd = (double)x[i];
      w[i] = (b0 * xd + b1 * x1) + b2 * x2;
      x2 = x1;
      x1 = xd;
      const double t = __builtin_fma(-a2, z2, w[i]);
      const double y = __builtin_fma(-a1, z1, t);
      z2 = z1;
      z1 = y;
    }
    double s1, s2;  // state entering this lane's chunk
    if constexpr (VARY == 0) {
      // in-row inclusive scan (rows of 16 lanes): R_l = sum_{i in row, i<=l} A^(l-i) z_i
      double r1 = z1, r2 = z2;
      {
        double q1 = row_shr<1>(r1), q2 = row_shr<1>(r2);
        r1 = __builtin_fma(A1.a, q1, __builtin_fma(A1.b, q2, r1));
        r2 = __builtin_fma(A1.c, q1, __builtin_fma(A1.d, q2, r2));
        q1 = row_shr<2>(r1);
        q2 = row_shr<2>(r2);
        r1 = __builtin_fma(A2.a, q1, __builtin_fma(A2.b, q2, r1));
        r2 = __builtin_fma(A2.c, q1, __builtin_fma(A2.d, q2, r2));
        q1 = row_shr<4>(r1);
        q2 = row_shr<4>(r2);
        r1 = __builtin_fma(A4.a, q1, __builtin_fma(A4.b, q2, r1));
        r2 = __builtin_fma(A4.c, q1, __builtin_fma(A4.d, q2, r2));
        q1 = row_shr<8>(r1);
        q2 = row_shr<8>(r2);
        r1 = __builtin_fma(A8.a, q1, __builtin_fma(A8.b, q2, r1));
        r2 = __builtin_fma(A8.c, q1, __builtin_fma(A8.d, q2, r2));
      }
      // state entering each row: T0 = carried, T_{k+1} = A^16 T_k + R(end of row k)
      const double e01 = read_lane(r1, 15), e02 = read_lane(r2, 15);
      const double e11 = read_lane(r1, 31), e12 = read_lane(r2, 31);
      const double e21 = read_lane(r1, 47), e22 = read_lane(r2, 47);
      const double t01 = cy1, t02 = cy2;
      const double t11 = __builtin_fma(A16.a, t01, __builtin_fma(A16.b, t02, e01));
      const double t12 = __builtin_fma(A16.c, t01, __builtin_fma(A16.d, t02, e02));
      const double t21 = __builtin_fma(A16.a, t11, __builtin_fma(A16.b, t12, e11));
      const double t22 = __builtin_fma(A16.c, t11, __builtin_fma(A16.d, t12, e12));
      const double t31 = __builtin_fma(A16.a, t21, __builtin_fma(A16.b, t22, e21));
      const double t32 = __builtin_fma(A16.c, t21, __builtin_fma(A16.d, t22, e22));
      const double T1 = row == 0 ? t01 : row == 1 ? t11 : row == 2 ? t21 : t31;
      const double T2 = row == 0 ? t02 : row == 1 ? t12 : row == 2 ? t22 : t32;
      // state entering this lane = A^(lane%16) * T_row + exclusive in-row scan
      const double ex1 = row_shr<1>(r1), ex2 = row_shr<1>(r2);
      s1 = __builtin_fma(Aj.a, T1, __builtin_fma(Aj.b, T2, ex1));
      s2 = __builtin_fma(Aj.c, T1, __builtin_fma(Aj.d, T2, ex2));
    } else {
      // general case: every lane has its own map s -> Al s + z.  Inclusive in-row scan of the composed maps
      // (P, r) with DPP row shifts (a lane without a source composes with the identity), then the row carries.
      M2 P = Al;
      double r1 = z1, r2 = z2;
      auto step = [&](auto shr) __attribute__((always_inline)) {
        M2 Q;
        Q.a = shr(P.a, 1.);
        Q.b = shr(P.b, 0.);
        Q.c = shr(P.c, 0.);
        Q.d = shr(P.d, 1.);
        const double q1 = shr(r1, 0.), q2 = shr(r2, 0.);
        r1 = __builtin_fma(P.a, q1, __builtin_fma(P.b, q2, r1));
        r2 = __builtin_fma(P.c, q1, __builtin_fma(P.d, q2, r2));
        P = mm(P, Q);
      };
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<1>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<2>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<4>(v, idv); });
      step([](double v, double idv) __attribute__((always_inline)) { return row_shr_keep<8>(v, idv); });
      // state entering each row: T0 = carried, T_{k+1} = P(end of row k) T_k + r(end of row k)
      double t1[4], t2[4];
      t1[0] = cy1;
      t2[0] = cy2;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int e = 16 * k + 15;
        const double pa = read_lane(P.a, e), pb = read_lane(P.b, e), pc = read_lane(P.c, e), pd = read_lane(P.d, e);
        const double e1 = read_lane(r1, e), e2 = read_lane(r2, e);
        t1[k + 1] = __builtin_fma(pa, t1[k], __builtin_fma(pb, t2[k], e1));
        t2[k + 1] = __builtin_fma(pc, t1[k], __builtin_fma(pd, t2[k], e2));
      }
      const double T1 = row == 0 ? t1[0] : row == 1 ? t1[1] : row == 2 ? t1[2] : t1[3];
      const double T2 = row == 0 ? t2[0] : row == 1 ? t2[1] : row == 2 ? t2[2] : t2[3];
      // exclusive composite of this lane within its row (identity for the first lane of a row)
      const double xa = row_shr_keep<1>(P.a, 1.), xb = row_shr_keep<1>(P.b, 0.), xc = row_shr_keep<1>(P.c, 0.),
                   xd = row_shr_keep<1>(P.d, 1.);
      const double x1r = row_shr_keep<1>(r1, 0.), x2r = row_shr_keep<1>(r2, 0.);
      s1 = __builtin_fma(xa, T1, __builtin_fma(xb, T2, x1r));
      s2 = __builtin_fma(xc, T1, __builtin_fma(xd, T2, x2r));
    }
    // final pass in the reference's order
    double y1 = s1, y2 = s2;
    float badacc = 0.f;  // becomes NaN as soon as one output is inf/NaN (x * 0 + acc), off the critical path
    float yo[TILE_K];
#pragma unroll
    for (int i = 0; i < TILE_K; i++) {
      const double y = (w[i] - a1 * y1) - a2 * y2;
      y2 = y1;
      y1 = y;
      yo[i] = (float)y;
      badacc = __builtin_fmaf(yo[i], 0.f, badacc);
    }
    if (__any(badacc != badacc)) {
      // inf / NaN appeared somewhere: redo with the explicit flush of biquad_filter.rs:881-883
      y1 = s1;
      y2 = s2;
      if (!__builtin_isfinite(y1)) y1 = 0.;
      if (!__builtin_isfinite(y2)) y2 = 0.;
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        double y = (w[i] - a1 * y1) - a2 * y2;
        if (!__builtin_isnormal(y)) y = 0.;
        y2 = y1;
        y1 = y;
        yo[i] = (float)y;
      }
    }
    // carried state for the next tile: lane 63's end state
    cx1 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[TILE_K - 1]), 63));
    cx2 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[TILE_K - 2]), 63));
    cy1 = read_lane(y1, 63);
    cy2 = read_lane(y2, 63);
    // T layout -> second LDS buffer (own row); stored to HBM by flush() during the next iteration
    lds_sync();
#pragma unroll
    for (int j = 0; j < NV4; j++)
      *reinterpret_cast<float4*>(lds_out + lane * LDS_ROW + j * 4) =
          make_float4(yo[j * 4 + 0], yo[j * 4 + 1], yo[j * 4 + 2], yo[j * 4 + 3]);
    lds_sync();
  };

  // Tiles are visited in order (the recurrence is serial).  Software pipeline per iteration:
  //   wait for tile t's input (requested one iteration ago) -> stage it into LDS -> request tile t+1
  //   -> store tile t-1's results (kept in the second LDS buffer) -> recurrence of tile t.
  // The compiler drains the whole VMEM queue (vmcnt(0)) before the staged registers are read because loads and
  // stores share one counter on gfx9; with this order everything still outstanding at that point was issued a
  // full iteration earlier, so the wait is (almost) free and stores never sit on the critical path.
  bool pending = false;   // a finished tile waits in lds_out
  uint32_t pending_tile = 0;
  uint32_t tile = d.tile0;
  while (tile < d.tile1) {
    if (!tile_is_fast(tile)) {
      float tmp[TILE_K];
      load_channel_generic(d.in, si, sc, ch, tile, lane, d.n_quanta, tmp);
      float cur[TILE_K];
#pragma unroll
      for (int i = 0; i < TILE_K; i++) cur[i] = tmp[i];
      stage(cur);
      if (pending) flush(pending_tile);
      if constexpr (VARY >= 2) process_arate(tile); else process(tile);
      pending = true;
      pending_tile = tile;
      tile++;
      continue;
    }
    uint32_t end = tile + 1;  // [tile, end) = maximal run of fast tiles
    if (is_src && end < si.fast_prefix) end = si.fast_prefix < d.tile1 ? si.fast_prefix : d.tile1;  // no table walk
    while (end < d.tile1 && tile_is_fast(end)) end++;
    if constexpr (VARY == 0 && NBUF == 4) {
      // prefetch distance 2: a wave's loads are in flight for two iterations instead of one — the recurrence keeps a
      // wave busy for ~6 us per tile while its next tile arrives in ~2 us, so with distance 1 most waves have nothing
      // in flight most of the time and HBM runs below what the same access pattern reaches as a plain copy
      float na[TILE_K], nb[TILE_K];
      fetch_fast(tile, na);
      fetch_fast(tile + 1 < end ? tile + 1 : tile, nb);
      for (; tile + 1 < end; tile += 2) {
        stage(na);
        fetch_fast(tile + 2 < end ? tile + 2 : tile, na);
        if (pending) flush(pending_tile);
        process(tile);
        pending = true;
        pending_tile = tile;
        stage(nb);
        fetch_fast(tile + 3 < end ? tile + 3 : tile + 1, nb);
        flush(pending_tile);
        process(tile + 1);
        pending_tile = tile + 1;
      }
      if (tile < end) {
        stage(na);
        if (pending) flush(pending_tile);
        process(tile);
        pending = true;
        pending_tile = tile;
        tile++;
      }
      continue;
    }
    float nx[TILE_K];
    fetch_fast(tile, nx);
    for (; tile < end; tile++) {
      stage(nx);
      fetch_fast(tile + 1 < end ? tile + 1 : tile, nx);  // clamped: the last iteration re-reads its own tile (L2 hit)
      if (pending) flush(pending_tile);
      if constexpr (VARY >= 2) process_arate(tile); else process(tile);
      pending = true;
      pending_tile = tile;
    }
  }
  if (pending) flush(pending_tile);
  if (lane == 0) {
    st[0] = cx1;
    st[1] = cx2;
    st[2] = cy1;
    st[3] = cy2;
  }
  if constexpr (DBG == 7) {
    if (lane == 0 && wid < 8192) {
      g_stream_trace[wid * 4 + 0] = trace_t0;
      g_stream_trace[wid * 4 + 1] = wall_clock64();
      g_stream_trace[wid * 4 + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
      g_stream_trace[wid * 4 + 3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Constant coefficients, second form ("digest"): the same arithmetic seen by the samples, restructured for occupancy.
// biquad_stream_kernel_t<0, 0> keeps a tile's FIR part w[32] (64 registers) between its two sweeps and owns two LDS
// buffers: 230 registers and 18 KB per wave = 2 waves per SIMD, and a wave that is inside its ~6 us recurrence has
// nothing but one prefetched tile in flight.  Here
//   * sweep 1 does not run the recurrence at all: with constant coefficients the zero-state end state of a lane's 32
//     frames is LINEAR in them, (y31, y30) = sum_i H_i x_i + Hm1 x[-1] + Hm2 x[-2] with
//     H_i = b0 g(31-i) + b1 g(30-i) + b2 g(29-i), g(n) = first column of M^n (0 for n < 0) — 68 uniform doubles per
//     wave, computed once and kept in LDS (broadcast reads); two independent 34-tap dot products instead of a serial
//     chain of 64 FMAs, and no w[] to keep;
//   * sweep 2 re-reads x from the lane's LDS row, evaluates the reference's expression in the reference's order
//     (biquad_filter.rs:877, exactly as the first form) and writes y back IN PLACE;
//   * one LDS buffer: at the top of the next iteration every lane swaps, chunk by chunk, the finished tile out of LDS
//     (-> gains -> global store) and the prefetched tile in — same addresses in both directions.
// 9.2 KB of LDS and < 128 registers per wave: 4 waves per SIMD.  Outputs: the incoming state of a lane is the only
// thing that is computed differently from the serial reference (as in the first form), f64-accurate either way.
__global__ __launch_bounds__(64, 4) void biquad_stream_digest_kernel(const BiquadStreamDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  double* htab = reinterpret_cast<double*>(lds + 64 * LDS_ROW);  // [34][2]: H_0..H_31, Hm1, Hm2
  const uint32_t wid = blockIdx.x;
  const uint32_t inst = wid / (uint32_t)d.nch;
  const int ch = (int)(wid % (uint32_t)d.nch);
  const int lane = threadIdx.x;
  if (inst >= d.n_inst) return;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (thread.rs:374-382)

  // wave-uniform values live in SGPRs (one scalar operand per f64 FMA is free): the compiler cannot prove that a value
  // COMPUTED on the vector unit is uniform and would keep the 25 doubles below in 50 VGPRs
  auto uni = [](double v) __attribute__((always_inline)) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  auto uni_m = [&](const M2& m) __attribute__((always_inline)) { return M2{uni(m.a), uni(m.b), uni(m.c), uni(m.d)}; };
  const double* cp = d.coefs + (uint64_t)inst * d.coef_stride;
  const double b0 = uni(cp[0]), b1 = uni(cp[1]), b2 = uni(cp[2]), a1 = uni(cp[3]), a2 = uni(cp[4]);
  M2 A1;
  {
    M2 m = {-a1, -a2, 1., 0.};
#pragma unroll
    for (int s = 0; s < 5; s++) m = mm(m, m);
    A1 = uni_m(m);
  }
  const M2 A2 = uni_m(mm(A1, A1)), A4 = uni_m(mm(A2, A2)), A8 = uni_m(mm(A4, A4)), A16 = uni_m(mm(A8, A8));
  M2 Aj = {1., 0., 0., 1.};
  {
    const int j = lane & 15;
    if (j & 1) Aj = mm(Aj, A1);
    if (j & 2) Aj = mm(Aj, A2);
    if (j & 4) Aj = mm(Aj, A4);
    if (j & 8) Aj = mm(Aj, A8);
  }
  const int row = lane >> 4;
  {
    // g(n) = M^n e1, n = 0..31; lane i < 32 keeps H_i, lanes 32 / 33 keep Hm1 / Hm2
    double g1 = 1., g2 = 0.;          // g(n)
    double p1 = 0., p2 = 0.;          // g(n - 1)
    double q1 = 0., q2 = 0.;          // g(n - 2)
    double h1 = 0., h2 = 0.;
    for (int n = 0; n < TILE_K; n++) {
      // H_{31-n} = b0 g(n) + b1 g(n-1) + b2 g(n-2)
      if (lane == TILE_K - 1 - n) {
        h1 = __builtin_fma(b0, g1, __builtin_fma(b1, p1, b2 * q1));
        h2 = __builtin_fma(b0, g2, __builtin_fma(b1, p2, b2 * q2));
      }
      if (n == TILE_K - 1) {
        if (lane == 32) {  // Hm1 = b1 g(31) + b2 g(30)
          h1 = __builtin_fma(b1, g1, b2 * p1);
          h2 = __builtin_fma(b1, g2, b2 * p2);
        }
        if (lane == 33) {  // Hm2 = b2 g(31)
          h1 = b2 * g1;
          h2 = b2 * g2;
        }
      }
      const double n1 = __builtin_fma(-a1, g1, -(a2 * g2)), n2 = g1;
      q1 = p1;
      q2 = p2;
      p1 = g1;
      p2 = g2;
      g1 = n1;
      g2 = n2;
    }
    if (lane < 34) {
      htab[lane * 2] = h1;
      htab[lane * 2 + 1] = h2;
    }
  }
  double* st = d.state + (uint64_t)inst * STATE_STRIDE + ch * 4;
  double cx1 = st[0], cx2 = st[1], cy1 = st[2], cy2 = st[3];
  float g[2] = {1.f, 1.f};
  bool g_mute[2] = {false, false}, g_pass[2] = {true, true};
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (k < d.n_gain) {
      g[k] = d.gain[k].base[inst];
      g_mute[k] = fabsf(g[k]) <= 1e-6f;
      g_pass[k] = fabsf(1.f - g[k]) <= 1e-6f;
    }
  const bool is_src = d.in.kind == IN_SOURCE;
  SrcInst si{};
  SrcSchedule sc{};
  const float* sig_base = nullptr;
  if (is_src) {
    si = d.in.src[inst];
    sc = si.sc;
  } else {
    sig_base = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
  }
  float* out_base = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride;
  auto lds_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };
  auto fetch_fast = [&](uint32_t tile, float (&dst)[TILE_K]) __attribute__((always_inline)) {
    const float* p = is_src ? si.base + (uint64_t)ch * si.ch_stride +
                                  (tile < si.fast_prefix ? si.linear_start + (int64_t)tile * TILE
                                                         : load_global(&sc.qrec[(uint64_t)tile * QUANTA_PER_TILE].start))
                            : sig_base + (uint64_t)tile * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const f4v t = load_global_f4(p + j * 256 + lane * 4);
      dst[j * 4 + 0] = t.x;
      dst[j * 4 + 1] = t.y;
      dst[j * 4 + 2] = t.z;
      dst[j * 4 + 3] = t.w;
    }
  };
  auto tile_is_fast = [&](uint32_t tile) __attribute__((always_inline)) -> bool {
    return !is_src || tile < si.fast_prefix || (si.aligned && load_global(sc.tile_fast + tile));
  };
  // swap: the finished tile `done` (if any) leaves LDS for HBM, the staged registers take its place
  auto swap_in = [&](const float (&cur)[TILE_K], bool have_done, uint32_t done) __attribute__((always_inline)) {
    lds_sync();
    float* op = out_base + (uint64_t)done * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      float4* cell = reinterpret_cast<float4*>(lds + r * LDS_ROW + c);
      float4 t = *cell;
      *cell = make_float4(cur[j * 4 + 0], cur[j * 4 + 1], cur[j * 4 + 2], cur[j * 4 + 3]);
      if (have_done) {
#pragma unroll
        for (int k = 0; k < 2; k++)
          if (k < d.n_gain) {
            if (g_mute[k]) {
              t = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (!g_pass[k]) {
              t.x *= g[k];
              t.y *= g[k];
              t.z *= g[k];
              t.w *= g[k];
            }
          }
        *reinterpret_cast<float4*>(op + j * 256 + lane * 4) = t;
      }
    }
    lds_sync();
  };
  auto flush_last = [&](uint32_t done) __attribute__((always_inline)) {
    lds_sync();
    float* op = out_base + (uint64_t)done * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      float4 t = *reinterpret_cast<const float4*>(lds + r * LDS_ROW + c);
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (k < d.n_gain) {
          if (g_mute[k]) {
            t = make_float4(0.f, 0.f, 0.f, 0.f);
          } else if (!g_pass[k]) {
            t.x *= g[k];
            t.y *= g[k];
            t.z *= g[k];
            t.w *= g[k];
          }
        }
      *reinterpret_cast<float4*>(op + j * 256 + lane * 4) = t;
    }
  };
  auto process = [&]() __attribute__((always_inline)) {
    float* xrow = lds + lane * LDS_ROW;  // this lane's 32 frames; y replaces x chunk by chunk
    const float4 xlast = *reinterpret_cast<const float4*>(xrow + TILE_K - 4);
    const float xl1 = xlast.w, xl2 = xlast.z;
    const float xm1 = __shfl_up(xl1, 1, 64), xm2 = __shfl_up(xl2, 1, 64);
    const double xs1 = lane == 0 ? cx1 : (double)xm1, xs2 = lane == 0 ? cx2 : (double)xm2;
    // sweep 1: zero-state end state as two dot products
    double za, zb, zc, zd;
    {
      const double2 hm1 = *reinterpret_cast<const double2*>(htab + 64), hm2 = *reinterpret_cast<const double2*>(htab + 66);
      za = hm1.x * xs1;
      zb = hm1.y * xs1;
      zc = hm2.x * xs2;
      zd = hm2.y * xs2;
    }
#pragma unroll 2
    for (int c = 0; c < NV4; c++) {
      const float4 xv = *reinterpret_cast<const float4*>(xrow + c * 4);
      const double2 h0 = *reinterpret_cast<const double2*>(htab + (c * 4 + 0) * 2);
      const double2 h1 = *reinterpret_cast<const double2*>(htab + (c * 4 + 1) * 2);
      const double2 h2 = *reinterpret_cast<const double2*>(htab + (c * 4 + 2) * 2);
      const double2 h3 = *reinterpret_cast<const double2*>(htab + (c * 4 + 3) * 2);
      za = __builtin_fma(h0.x, (double)xv.x, za);
      zb = __builtin_fma(h0.y, (double)xv.x, zb);
      zc = __builtin_fma(h1.x, (double)xv.y, zc);
      zd = __builtin_fma(h1.y, (double)xv.y, zd);
      za = __builtin_fma(h2.x, (double)xv.z, za);
      zb = __builtin_fma(h2.y, (double)xv.z, zb);
      zc = __builtin_fma(h3.x, (double)xv.w, zc);
      zd = __builtin_fma(h3.y, (double)xv.w, zd);
    }
    const double z1 = za + zc, z2 = zb + zd;
    // wavefront scan of the affine maps s -> A s + z (as in the first form)
    double s1, s2;
    {
      double r1 = z1, r2 = z2;
      double q1 = row_shr<1>(r1), q2 = row_shr<1>(r2);
      r1 = __builtin_fma(A1.a, q1, __builtin_fma(A1.b, q2, r1));
      r2 = __builtin_fma(A1.c, q1, __builtin_fma(A1.d, q2, r2));
      q1 = row_shr<2>(r1);
      q2 = row_shr<2>(r2);
      r1 = __builtin_fma(A2.a, q1, __builtin_fma(A2.b, q2, r1));
      r2 = __builtin_fma(A2.c, q1, __builtin_fma(A2.d, q2, r2));
      q1 = row_shr<4>(r1);
      q2 = row_shr<4>(r2);
      r1 = __builtin_fma(A4.a, q1, __builtin_fma(A4.b, q2, r1));
      r2 = __builtin_fma(A4.c, q1, __builtin_fma(A4.d, q2, r2));
      q1 = row_shr<8>(r1);
      q2 = row_shr<8>(r2);
      r1 = __builtin_fma(A8.a, q1, __builtin_fma(A8.b, q2, r1));
      r2 = __builtin_fma(A8.c, q1, __builtin_fma(A8.d, q2, r2));
      const double e01 = read_lane(r1, 15), e02 = read_lane(r2, 15);
      const double e11 = read_lane(r1, 31), e12 = read_lane(r2, 31);
      const double e21 = read_lane(r1, 47), e22 = read_lane(r2, 47);
      const double t01 = cy1, t02 = cy2;
      const double t11 = __builtin_fma(A16.a, t01, __builtin_fma(A16.b, t02, e01));
      const double t12 = __builtin_fma(A16.c, t01, __builtin_fma(A16.d, t02, e02));
      const double t21 = __builtin_fma(A16.a, t11, __builtin_fma(A16.b, t12, e11));
      const double t22 = __builtin_fma(A16.c, t11, __builtin_fma(A16.d, t12, e12));
      const double t31 = __builtin_fma(A16.a, t21, __builtin_fma(A16.b, t22, e21));
      const double t32 = __builtin_fma(A16.c, t21, __builtin_fma(A16.d, t22, e22));
      const double T1 = row == 0 ? t01 : row == 1 ? t11 : row == 2 ? t21 : t31;
      const double T2 = row == 0 ? t02 : row == 1 ? t12 : row == 2 ? t22 : t32;
      const double ex1 = row_shr<1>(r1), ex2 = row_shr<1>(r2);
      s1 = __builtin_fma(Aj.a, T1, __builtin_fma(Aj.b, T2, ex1));
      s2 = __builtin_fma(Aj.c, T1, __builtin_fma(Aj.d, T2, ex2));
    }
    // sweep 2: the reference's expression in the reference's order (biquad_filter.rs:877-883), from the true incoming
    // state; y replaces x in place, 4 frames at a time.  `!y.is_normal() -> 0` only matters for inf / NaN (denormals are
    // flushed by the hardware mode): a chunk in which one shows up anywhere in the wave is redone with the explicit test
    double y1 = __builtin_isfinite(s1) ? s1 : 0., y2 = __builtin_isfinite(s2) ? s2 : 0.;
    double p1 = xs1, p2 = xs2;
#pragma unroll 1
    for (int c = 0; c < NV4; c++) {
      const float4 xv = *reinterpret_cast<const float4*>(xrow + c * 4);
      const float xf[4] = {xv.x, xv.y, xv.z, xv.w};
      const double sy1 = y1, sy2 = y2, sp1 = p1, sp2 = p2;
      float yo[4];
      float badacc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const double xd = (double)xf[j];
        const double wi = (b0 * xd + b1 * p1) + b2 * p2;
        p2 = p1;
        p1 = xd;
        const double y = (wi - a1 * y1) - a2 * y2;
        y2 = y1;
        y1 = y;
        yo[j] = (float)y;
        badacc = __builtin_fmaf(yo[j], 0.f, badacc);  // NaN as soon as one output is inf / NaN
      }
      if (__any(badacc != badacc)) {
        y1 = sy1;
        y2 = sy2;
        p1 = sp1;
        p2 = sp2;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double xd = (double)xf[j];
          const double wi = (b0 * xd + b1 * p1) + b2 * p2;
          p2 = p1;
          p1 = xd;
          double y = (wi - a1 * y1) - a2 * y2;
          if (!__builtin_isnormal(y)) y = 0.;
          y2 = y1;
          y1 = y;
          yo[j] = (float)y;
        }
      }
      *reinterpret_cast<float4*>(xrow + c * 4) = make_float4(yo[0], yo[1], yo[2], yo[3]);
    }
    cx1 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl1), 63));
    cx2 = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl2), 63));
    cy1 = read_lane(y1, 63);
    cy2 = read_lane(y2, 63);
  };
  // Software pipeline per iteration: wait for tile t (requested one iteration ago) -> swap it into LDS while tile t-1
  // leaves for HBM -> request tile t+1 -> both sweeps of tile t.  Stores and loads are issued back to back, one whole
  // process() before the next wait: neither sits on the critical path.
  bool have = false;
  uint32_t done = 0;
  uint32_t tile = d.tile0;
  while (tile < d.tile1) {
    if (!tile_is_fast(tile)) {
      float tmp[TILE_K];
      load_channel_generic(d.in, si, sc, ch, tile, lane, d.n_quanta, tmp);
      float cur[TILE_K];
#pragma unroll
      for (int i = 0; i < TILE_K; i++) cur[i] = tmp[i];
      swap_in(cur, have, done);
      process();
      have = true;
      done = tile;
      tile++;
      continue;
    }
    uint32_t end = tile + 1;
    if (is_src && end < si.fast_prefix) end = si.fast_prefix < d.tile1 ? si.fast_prefix : d.tile1;
    while (end < d.tile1 && tile_is_fast(end)) end++;
    float nx[TILE_K];
    fetch_fast(tile, nx);
    for (; tile < end; tile++) {
      swap_in(nx, have, done);
      fetch_fast(tile + 1 < end ? tile + 1 : tile, nx);
      process();
      have = true;
      done = tile;
    }
  }
  if (have) flush_last(done);
  if (lane == 0) {
    st[0] = cx1;
    st[1] = cx2;
    st[2] = cy1;
    st[3] = cy2;
  }
}

void launch_biquad_stream(const BiquadStreamDesc& d, void* stream) {
  const dim3 grid(d.n_inst * (uint32_t)d.nch), block(64);
  const size_t lds = 2 * 64 * LDS_ROW * sizeof(float);
  const char* dbg = measure_switch("WAA_STREAM_DEBUG");  // measurement aid only, see profiles/r01_c2_memory_pattern.txt
  if (d.dup_out && d.vary == 0) {
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 0, 4, true>), grid, block, lds, (hipStream_t)stream, d);
    return;
  }
  if (d.vary == 3 && dbg && dbg[0] == '3')
    hipLaunchKernelGGL((biquad_stream_kernel_t<3, 3>), grid, block, lds, (hipStream_t)stream, d);
  else if (d.vary == 3 && dbg && dbg[0] == '4')
    hipLaunchKernelGGL((biquad_stream_kernel_t<4, 3>), grid, block, lds, (hipStream_t)stream, d);
  else if (d.vary == 3 && measure_switch("WAA_ARATE_BUFS3"))  // experiment: deeper coefficient prefetch (spills a few registers)
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 3, 3>), grid, block, lds, (hipStream_t)stream, d);
  else if (d.vary == 3)
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 3>), grid, block, lds, (hipStream_t)stream, d);
  else if (d.vary == 2)
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 2>), grid, block, lds, (hipStream_t)stream, d);
  else if (d.vary)
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 1>), grid, block, lds, (hipStream_t)stream, d);
  else if (dbg && dbg[0] == '1')
    hipLaunchKernelGGL((biquad_stream_kernel_t<1, 0>), grid, block, lds, (hipStream_t)stream, d);
  else if (dbg && dbg[0] == '2')
    hipLaunchKernelGGL((biquad_stream_kernel_t<2, 0>), grid, block, lds, (hipStream_t)stream, d);
  else if (dbg && dbg[0] == '5')
    hipLaunchKernelGGL((biquad_stream_kernel_t<5, 0>), grid, block, lds, (hipStream_t)stream, d);
  else if (dbg && dbg[0] == '7')
    hipLaunchKernelGGL((biquad_stream_kernel_t<7, 0>), grid, block, lds, (hipStream_t)stream, d);
  else if (measure_switch("WAA_STREAM_PREFETCH1"))  // the form of rounds 1-5: one tile in flight (A/B with tools/ab_env.py)
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 0, 2>), grid, block, lds, (hipStream_t)stream, d);
  else if (measure_switch("WAA_BIQUAD_DIGEST"))  // experiment, bit-identical output; same-box A/B (tools/ab_env.py): no gain — with 4
                                         // instead of 2 waves per SIMD the kernel runs at the same 1.5-1.65 ms, i.e. what bounds
                                         // C2 is the memory side of 2048 concurrent streams, not the wave's latency hiding
    hipLaunchKernelGGL(biquad_stream_digest_kernel, grid, block, 64 * LDS_ROW * sizeof(float) + 34 * 2 * sizeof(double),
                       (hipStream_t)stream, d);
  else
    // Two tiles in flight per wavefront (round 6).  Rounds 2-5 measured this form "identical" — on batches whose OUTPUT lay in
    // slow-to-write memory (1.58 against 1.58 ms: the memory system was the bound either way).  In fast-to-write memory
    // (waa_device_arena_reserve_graded) the kernel sits 3.5 % above its own copy-only form with one tile in flight and 1.5 % with
    // two: 1.345 -> 1.320 ms on six batches out of six (profiles/r06k_c2_variants.txt).  256 registers, no spill.
    hipLaunchKernelGGL((biquad_stream_kernel_t<0, 0, 4>), grid, block, lds, (hipStream_t)stream, d);
}

}  // namespace waa

// (measurement aid, not part of include/waa_hip.h: copies the trace of the last WAA_STREAM_DEBUG=7 launch)
extern "C" int waa_debug_stream_trace(unsigned long long* dst, unsigned n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(waa::g_stream_trace), (size_t)n_words * 8);
}
