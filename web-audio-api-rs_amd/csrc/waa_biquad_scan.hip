// waa_biquad_scan.hip — the streaming Biquad, parallel in TIME as well as over streams.
//
// waa_biquad_stream.hip gives every (instance, channel) stream ONE wavefront that walks the stream's tiles in order: at
// BASELINE config 2 that is 2048 wavefronts — two per SIMD — however long the render is, and the per-frame-coefficient form
// (C1 a-rate) is bound by the latency of that one wavefront's dependent chain (DESIGN.md section 3, 0.25 of the HBM peak).
// Here the unit of work is one TILE of one stream (2048 frames, a lane owns 32), units are handed out by an atomic counter,
// and the state that enters a tile comes from a chained scan over the tiles of its stream ("decoupled look-back"):
//
//   1. zero-state response of the tile (sweep 1 + the in-wavefront scan of waa_biquad_stream.hip)  -> Z_t
//   2. publish the AGGREGATE (Z_t; the tile's transition P_t is A^64 for constant coefficients)     status 1
//   3. look back: walk the predecessors of the same stream, newest first, folding aggregates
//        s = P_{t-1} (... ) + Z_{t-1}   until one has published its INCLUSIVE state (status 2) or the launch's first tile
//   4. the reference's evaluation order from every lane's true incoming state (biquad_filter.rs:877-883)
//   5. publish the inclusive state (y[n-1], y[n-2] after the tile's last frame)                     status 2
//
// Units are numbered tile-major, stream-minor, in EIGHT independent shards (stream s belongs to shard s mod 8, wavefront w pulls
// from shard w mod 8 — about one per XCD; one device-scope counter saturates near 88 dequeues per microsecond, the render has
// 481 000 units), and a wavefront takes the next number of its shard when it is free: every predecessor a unit can wait for has
// a SMALLER number in the same shard, i.e. was taken by a wavefront that is running or done — progress does not depend on
// dispatch order or placement (MI355X_MICROARCH.md, correctness boundaries).  Spins are bounded: a wavefront that gives up sets
// `error` (waa_sync then fails the render; nothing hangs).
// Publication (MI355X_MICROARCH.md, hand-off price list): payload and flag are write-through device-scope stores (`sc1`: relaxed
// agent atomics), the payload is drained (s_waitcnt vmcnt(0)) before the flag is stored; readers poll the flag with a relaxed
// `sc1` load and read the payload with `sc1` loads afterwards.  NO acquire / release fences: a release writes back the XCD's
// whole L2 and an acquire invalidates the CU's L1 — with two of each per 16 KB unit the first version of this kernel took
// 36 ms for C2 instead of 1.5.
//
// The arithmetic a sample sees is that of the serial kernel: the only thing computed differently is a lane's INCOMING state
// (f64-accurate either way).  WAA_BIQUAD_SERIAL=1 selects the one-wavefront-per-stream kernel (same-box A/B, cross-check).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"
#include "waa_stream_common.hpp"

namespace waa {
#ifdef WAA_MEASURE

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
__device__ __forceinline__ void mv(const M2& m, double x1, double x2, double e1, double e2, double& o1, double& o2) {
  const double t1 = __builtin_fma(m.a, x1, __builtin_fma(m.b, x2, e1));
  const double t2 = __builtin_fma(m.c, x1, __builtin_fma(m.d, x2, e2));
  o1 = t1;
  o2 = t2;
}
__device__ __forceinline__ double ald(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ast(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr int PAY = 8;         // doubles per unit: Z1 Z2 | incl y1 y2 | x[last] x[last - 1] | spare
constexpr uint32_t SPIN_MAX = 4000000u;
}  // namespace

// Per-instance matrices of the constant-coefficient scan, computed once per plan: A = M^32, A^2, A^4, A^8, A^16, A^64 (6 x 4
// doubles), then A^j for j = 0..15 (16 x 4) — one thread per instance.
__global__ void biquad_scan_powers_kernel(const double* coefs, uint64_t coef_stride, double* pw, uint32_t n_inst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inst) return;
  const double* cp = coefs + (uint64_t)i * coef_stride;
  M2 m = {-cp[3], -cp[4], 1., 0.};
  for (int s = 0; s < 5; s++) m = mm(m, m);
  const M2 A1 = m, A2 = mm(A1, A1), A4 = mm(A2, A2), A8 = mm(A4, A4), A16 = mm(A8, A8), A32 = mm(A16, A16), A64 = mm(A32, A32);
  double* o = pw + (uint64_t)i * BIQUAD_SCAN_PW;
  const M2 ms[6] = {A1, A2, A4, A8, A16, A64};
  for (int q = 0; q < 6; q++) {
    o[q * 4 + 0] = ms[q].a;
    o[q * 4 + 1] = ms[q].b;
    o[q * 4 + 2] = ms[q].c;
    o[q * 4 + 3] = ms[q].d;
  }
  for (int j = 0; j < 16; j++) {
    M2 Aj = {1., 0., 0., 1.};
    if (j & 1) Aj = mm(Aj, A1);
    if (j & 2) Aj = mm(Aj, A2);
    if (j & 4) Aj = mm(Aj, A4);
    if (j & 8) Aj = mm(Aj, A8);
    o[24 + j * 4 + 0] = Aj.a;
    o[24 + j * 4 + 1] = Aj.b;
    o[24 + j * 4 + 2] = Aj.c;
    o[24 + j * 4 + 3] = Aj.d;
  }
}
void launch_biquad_scan_powers(const double* coefs, uint64_t coef_stride, double* pw, uint32_t n_inst, void* stream) {
  hipLaunchKernelGGL(biquad_scan_powers_kernel, dim3((n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, coefs, coef_stride, pw, n_inst);
}

// Payload words (8 doubles per unit): [0,1] the tile's zero-state end state (aggregate), [2,3] the inclusive state
// (y[n-1], y[n-2] after the tile's last frame), [4,5] the tile's last two input samples.  Every word is written at most ONCE per
// render and starts out as SENTINEL (the host fills the table with 0xFF bytes before every render): a reader polls a word until
// it is something else — no flags, no fences, no draining of the writer's other memory traffic.  (All-ones is a NaN no
// arithmetic produces: hardware NaNs are canonical, a propagated input NaN keeps its own payload, and an f32 all-ones NaN
// widens to 0xFFFFFFFFE0000000.)
namespace {
constexpr unsigned long long SENTINEL = ~0ull;
__device__ __forceinline__ unsigned long long aldu(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double as_d(unsigned long long v) { return __longlong_as_double((long long)v); }
// polls one word until it is valid (bounded); returns false when it gave up
__device__ __forceinline__ bool wait_word(const double* p, unsigned long long& v, uint32_t* error, int lane) {
  uint32_t spins = 0;
  while ((v = aldu(p)) == SENTINEL) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_MAX) {
      if (lane == 0) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = 0ull;
      return false;
    }
  }
  return true;
}
}  // namespace

template <bool DUP, int DBG = 0>
__global__ __launch_bounds__(64, 3) void biquad_scan_kernel(const BiquadStreamDesc d, const BiquadScanCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int row = lane >> 4;
  // f64 (and f16) denormals: flush inputs and outputs, like the reference's FTZ/DAZ render scope
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  auto lds_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch;
  const uint32_t ntl = d.tile1 - d.tile0;
  const uint32_t shard = blockIdx.x & 7u;
  const uint32_t ns_sh = (n_streams + 7u - shard) / 8u;   // streams shard, shard + 8, ...
  const uint32_t n_units = ntl * ns_sh;
  uint32_t* head = ctl.counter + shard * 16u;             // (one cache line per shard)
  const uint32_t base = ctl.counter_base[shard];
  const bool is_src = d.in.kind == IN_SOURCE;
  uint32_t static_next = blockIdx.x >> 3;  // (DBG & 2, measurement aid: numbers without the atomic counter)
  auto dequeue = [&]() __attribute__((always_inline)) -> uint32_t {
    if (DBG & 2) {
      const uint32_t r = static_next;
      static_next += gridDim.x >> 3;
      return r;
    }
    uint32_t u = 0;
    if (lane == 0) u = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
  };
  // where a unit's tile starts in a LINEAR input (null: the generic loader renders it), and whether the two frames in front
  // of the tile can be read from the input as well
  auto locate = [&](uint32_t inst, int ch, uint32_t tile, bool& prev_direct) __attribute__((always_inline)) -> const float* {
    prev_direct = false;
    if (!is_src) {
      prev_direct = tile > d.tile0;
      return d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride + (uint64_t)tile * TILE;
    }
    const SrcInst* sp = d.in.src + inst;
    const uint32_t fast_prefix = load_global(&sp->fast_prefix);
    if (tile < fast_prefix) {
      prev_direct = tile > d.tile0;
      return load_global(&sp->base) + (uint64_t)ch * load_global(&sp->ch_stride) + load_global(&sp->linear_start) + (int64_t)tile * TILE;
    }
    if (load_global(&sp->aligned) && load_global(load_global(&sp->sc.tile_fast) + tile))
      return load_global(&sp->base) + (uint64_t)ch * load_global(&sp->ch_stride) +
             load_global(&load_global(&sp->sc.qrec)[(uint64_t)tile * QUANTA_PER_TILE].start);
    return nullptr;
  };
  f4v raw[NV4];  // the prefetched tile of the unit that is processed next
  auto fetch = [&](const float* p) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NV4; j++) raw[j] = __builtin_nontemporal_load((const WAA_GLOBAL_AS f4v*)(p + j * 256 + lane * 4));
  };
  // Unit numbers are pulled TWO units ahead and tiles ONE unit ahead: a device-scope dequeue takes 1-3 us under load and the
  // tile's address depends on it — waited for on the spot (the first version) the wave spent more time there than rendering.
  uint32_t u = dequeue();
  if (u >= n_units) return;
  uint32_t un = dequeue();
  {
    const uint32_t sid = shard + 8u * (u % ns_sh);
    bool pd;
    const float* p = locate(sid / (uint32_t)d.nch, (int)(sid % (uint32_t)d.nch), d.tile0 + u / ns_sh, pd);
    if (p) fetch(p);
  }
  for (;;) {
    // the number of the unit after next: requested now, read at the bottom of the loop (none once the shard has run dry)
    uint32_t unn_raw = 0xffffffffu;
    if (DBG & 2) {
      if (un < n_units) unn_raw = dequeue();
    } else if (un < n_units && lane == 0) {
      unn_raw = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    }
    const uint32_t sid = shard + 8u * (u % ns_sh), tile = d.tile0 + u / ns_sh;
    const uint32_t inst = sid / (uint32_t)d.nch;
    const int ch = (int)(sid % (uint32_t)d.nch);
    const double* cp = d.coefs + (uint64_t)inst * d.coef_stride;
    const double* pw = ctl.pw + (uint64_t)inst * BIQUAD_SCAN_PW;
    double* st = d.state + (uint64_t)inst * STATE_STRIDE + ch * 4;
    double* pay = ctl.payload + ((uint64_t)sid * d.n_tiles + tile) * PAY;
    bool prev_direct;
    const float* in_fast = locate(inst, ch, tile, prev_direct);
    // ---- early requests: the predecessor's inclusive state and the two samples in front of the tile
    unsigned long long pin1 = 0, pin2 = 0;
    float hx1 = 0.f, hx2 = 0.f;
    if (DBG & 1) {
      pin1 = pin2 = 0;
    } else if (tile > d.tile0) {
      pin1 = aldu(pay - PAY + 2);
      pin2 = aldu(pay - PAY + 3);
      if (prev_direct) {
        hx1 = load_global(in_fast - 1);
        hx2 = load_global(in_fast - 2);
      }
    }
    // ---- this unit's tile -> LDS rows
    if (in_fast) {
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
        *reinterpret_cast<float4*>(lds + r * LDS_ROW + c) = make_float4(raw[j].x, raw[j].y, raw[j].z, raw[j].w);
      }
    } else {
      float tmp[TILE_K];  // (its address escapes into the out-of-line loader)
      const SrcInst si = d.in.src[inst];
      load_channel_generic(d.in, si, si.sc, ch, tile, lane, d.n_quanta, tmp);
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
        *reinterpret_cast<float4*>(lds + r * LDS_ROW + c) = make_float4(tmp[j * 4 + 0], tmp[j * 4 + 1], tmp[j * 4 + 2], tmp[j * 4 + 3]);
      }
    }
    // ---- the next unit's tile: in flight while this one is rendered
    if (un < n_units) {
      const uint32_t nsid = shard + 8u * (un % ns_sh);
      bool pd;
      const float* p = locate(nsid / (uint32_t)d.nch, (int)(nsid % (uint32_t)d.nch), d.tile0 + un / ns_sh, pd);
      if (p) fetch(p);
    }
    lds_sync();
    float* myrow = lds + lane * LDS_ROW;
    float x[TILE_K];
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const float4 t = *reinterpret_cast<const float4*>(myrow + j * 4);
      x[j * 4 + 0] = t.x;
      x[j * 4 + 1] = t.y;
      x[j * 4 + 2] = t.z;
      x[j * 4 + 3] = t.w;
    }
    const double b0 = cp[0], b1 = cp[1], b2 = cp[2], a1 = cp[3], a2 = cp[4];
    // ---- x[n-1], x[n-2] in front of the tile: the carried state, the input itself, or what the predecessor published
    double tx1, tx2;
    if (tile == d.tile0) {
      tx1 = st[0];
      tx2 = st[1];
    } else if (prev_direct) {
      tx1 = (double)hx1;
      tx2 = (double)hx2;
    } else {
      unsigned long long v1, v2;
      wait_word(pay - PAY + 4, v1, ctl.error, lane);
      wait_word(pay - PAY + 5, v2, ctl.error, lane);
      tx1 = as_d(v1);
      tx2 = as_d(v2);
    }
    const float xm1 = __shfl_up(x[TILE_K - 1], 1, 64), xm2 = __shfl_up(x[TILE_K - 2], 1, 64);
    const double xs1 = lane == 0 ? tx1 : (double)xm1, xs2 = lane == 0 ? tx2 : (double)xm2;
    // ---- sweep 1: zero-state response of the lane's chunk
    double z1 = 0., z2 = 0.;
    {
      double x1 = xs1, x2 = xs2;
#pragma unroll
      for (int i = 0; i < TILE_K; i++) {
        double xd = (double)x[i];
        asm volatile("" : "+v"(xd) : "v"(z2));  // (bounds the compiler's look-ahead: see waa_conv3.hip)
        const double w = __builtin_fma(b2, x2, __builtin_fma(b1, x1, b0 * xd));
        x2 = x1;
        x1 = xd;
        const double t = __builtin_fma(-a2, z2, w);
        const double y = __builtin_fma(-a1, z1, t);
        z2 = z1;
        z1 = y;
      }
    }
    // in-row inclusive scan (rows of 16 lanes), then the rows' ends
    double r1 = z1, r2 = z2;
    {
      M2 P = {pw[0], pw[1], pw[2], pw[3]};
      double q1 = row_shr<1>(r1), q2 = row_shr<1>(r2);
      mv(P, q1, q2, r1, r2, r1, r2);
      P = M2{pw[4], pw[5], pw[6], pw[7]};
      q1 = row_shr<2>(r1);
      q2 = row_shr<2>(r2);
      mv(P, q1, q2, r1, r2, r1, r2);
      P = M2{pw[8], pw[9], pw[10], pw[11]};
      q1 = row_shr<4>(r1);
      q2 = row_shr<4>(r2);
      mv(P, q1, q2, r1, r2, r1, r2);
      P = M2{pw[12], pw[13], pw[14], pw[15]};
      q1 = row_shr<8>(r1);
      q2 = row_shr<8>(r2);
      mv(P, q1, q2, r1, r2, r1, r2);
    }
    const M2 A16 = {pw[16], pw[17], pw[18], pw[19]};
    const double e01 = read_lane(r1, 15), e02 = read_lane(r2, 15), e11 = read_lane(r1, 31), e12 = read_lane(r2, 31);
    const double e21 = read_lane(r1, 47), e22 = read_lane(r2, 47), e31 = read_lane(r1, 63), e32 = read_lane(r2, 63);
    const float xl1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[TILE_K - 1]), 63));
    const float xl2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[TILE_K - 2]), 63));
    {
      // the tile's zero-state end state = its aggregate
      double u1 = e01, u2 = e02;
      mv(A16, u1, u2, e11, e12, u1, u2);
      mv(A16, u1, u2, e21, e22, u1, u2);
      mv(A16, u1, u2, e31, e32, u1, u2);
      if (lane == 0 && !(DBG & 1)) {
        ast(pay + 0, u1);
        ast(pay + 1, u2);
        ast(pay + 4, (double)xl1);
        ast(pay + 5, (double)xl2);
      }
    }
    // ---- look-back: the state that enters this tile
    double t1, t2;
    if (tile == d.tile0) {
      t1 = st[2];
      t2 = st[3];
    } else if (pin1 != SENTINEL && pin2 != SENTINEL) {  // the predecessor had finished when this unit started: the usual case
      t1 = as_d(pin1);
      t2 = as_d(pin2);
    } else {
      const M2 A64 = {pw[20], pw[21], pw[22], pw[23]};
      double acc1 = 0., acc2 = 0.;
      M2 Mx = {1., 0., 0., 1.};
      t1 = t2 = 0.;
      bool done = false;
      for (uint32_t j = tile; j-- > d.tile0 && !done;) {
        const double* p = ctl.payload + ((uint64_t)sid * d.n_tiles + j) * PAY;
        // whichever is there first: the inclusive state (the walk ends) or the aggregate (fold it, one tile further back)
        uint32_t spins = 0;
        for (;;) {
          const unsigned long long i1 = aldu(p + 2), i2 = aldu(p + 3);
          if (i1 != SENTINEL && i2 != SENTINEL) {
            mv(Mx, as_d(i1), as_d(i2), acc1, acc2, t1, t2);
            done = true;
            break;
          }
          const unsigned long long g1 = aldu(p + 0), g2 = aldu(p + 1);
          if (g1 != SENTINEL && g2 != SENTINEL) {
            mv(Mx, as_d(g1), as_d(g2), acc1, acc2, acc1, acc2);
            Mx = mm(Mx, A64);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) {
            if (lane == 0) __hip_atomic_store(ctl.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = true;
            break;
          }
        }
      }
      if (!done) mv(Mx, st[2], st[3], acc1, acc2, t1, t2);  // every predecessor back to the launch's first tile was an aggregate
    }
    double T1 = t1, T2 = t2;
    {
      double v1 = t1, v2 = t2;
      mv(A16, v1, v2, e01, e02, v1, v2);
      if (row == 1) {
        T1 = v1;
        T2 = v2;
      }
      mv(A16, v1, v2, e11, e12, v1, v2);
      if (row == 2) {
        T1 = v1;
        T2 = v2;
      }
      mv(A16, v1, v2, e21, e22, v1, v2);
      if (row == 3) {
        T1 = v1;
        T2 = v2;
      }
    }
    double s1, s2;
    {
      const double* aj = pw + 24 + (lane & 15) * 4;
      const M2 Aj = {load_global(aj), load_global(aj + 1), load_global(aj + 2), load_global(aj + 3)};
      const double ex1 = row_shr<1>(r1), ex2 = row_shr<1>(r2);
      mv(Aj, T1, T2, ex1, ex2, s1, s2);
    }
    // ---- sweep 2: the reference's evaluation order from the true incoming state; results into the lane's own row
    double y1 = s1, y2 = s2;
    {
      float badacc = 0.f;
      double x1 = xs1, x2 = xs2;
#pragma unroll
      for (int q = 0; q < NV4; q++) {
        float yo[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double xd = (double)x[q * 4 + e];
          const double w = (b0 * xd + b1 * x1) + b2 * x2;
          x2 = x1;
          x1 = xd;
          const double y = (w - a1 * y1) - a2 * y2;
          y2 = y1;
          y1 = y;
          yo[e] = (float)y;
          badacc = __builtin_fmaf(yo[e], 0.f, badacc);
        }
        *reinterpret_cast<float4*>(myrow + q * 4) = make_float4(yo[0], yo[1], yo[2], yo[3]);
      }
      if (__any(badacc != badacc)) {  // inf / NaN somewhere: redo with the explicit flush of biquad_filter.rs:881-883
        y1 = __builtin_isfinite(s1) ? s1 : 0.;
        y2 = __builtin_isfinite(s2) ? s2 : 0.;
        x1 = xs1;
        x2 = xs2;
#pragma unroll
        for (int q = 0; q < NV4; q++) {
          float yo[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const double xd = (double)x[q * 4 + e];
            const double w = (b0 * xd + b1 * x1) + b2 * x2;
            x2 = x1;
            x1 = xd;
            double y = (w - a1 * y1) - a2 * y2;
            if (!__builtin_isnormal(y)) y = 0.;
            y2 = y1;
            y1 = y;
            yo[e] = (float)y;
          }
          *reinterpret_cast<float4*>(myrow + q * 4) = make_float4(yo[0], yo[1], yo[2], yo[3]);
        }
      }
    }
    {
      // the inclusive state: lane 63's end state (a non-finite state is published as 0 — what the successor's flush rule
      // would make of it, and never the sentinel)
      double i1 = read_lane(y1, 63), i2 = read_lane(y2, 63);
      if (!__builtin_isfinite(i1)) i1 = 0.;
      if (!__builtin_isfinite(i2)) i2 = 0.;
      if (lane == 0 && !(DBG & 1)) {
        ast(pay + 2, i1);
        ast(pay + 3, i2);
        if (tile + 1 == d.tile1) {  // carried state for a later launch of a block-scheduled plan
          st[0] = (double)xl1;
          st[1] = (double)xl2;
          st[2] = i1;
          st[3] = i2;
        }
      }
    }
    lds_sync();
    // ---- rows -> coalesced store (gains on the way out, gain.rs:163-179 fast paths per launch)
    {
      float* op = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride + (uint64_t)tile * TILE;
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
        float4 t = *reinterpret_cast<const float4*>(lds + r * LDS_ROW + c);
#pragma unroll
        for (int k = 0; k < 2; k++)
          if (k < d.n_gain) {
            const float g = d.gain[k].base[inst];
            if (fabsf(g) <= 1e-6f) {
              t = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (!(fabsf(1.f - g) <= 1e-6f)) {
              t.x *= g;
              t.y *= g;
              t.z *= g;
              t.w *= g;
            }
          }
        if (DBG & 4)
          asm volatile("" ::"v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
        else
          __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, (WAA_GLOBAL_AS f4v*)(op + j * 256 + lane * 4));
        if constexpr (DUP)
          __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, (WAA_GLOBAL_AS f4v*)(op + d.out.ch_stride + j * 256 + lane * 4));
      }
    }
    lds_sync();  // the rows are rewritten by the next unit
    if (un >= n_units) break;
    u = un;
    un = (uint32_t)__builtin_amdgcn_readfirstlane((int)unn_raw);
  }
}

// `issued[8]`: the host's mirror of the eight shard counters; updated for this launch (every unit takes one number, every
// wavefront one more when it finds its shard empty)
void launch_biquad_scan(const BiquadStreamDesc& d, const BiquadScanCtl& ctl_in, uint32_t* issued, void* stream) {
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch, ntl = d.tile1 - d.tile0;
  const uint64_t n_units = (uint64_t)ntl * n_streams;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const char* wpc = measure_switch("WAA_BIQUAD_SCAN_WAVES");  // resident wavefronts per CU (LDS: 9.2 KB each, registers: 4 per SIMD)
  const uint64_t resident = (uint64_t)cus * (wpc ? (unsigned)atoi(wpc) : 16u);
  uint32_t waves = (uint32_t)(n_units < resident ? n_units : resident);
  waves = (waves + 7u) & ~7u;
  BiquadScanCtl ctl = ctl_in;
  for (uint32_t sh = 0; sh < 8; sh++) {
    ctl.counter_base[sh] = issued[sh];
    issued[sh] += ntl * ((n_streams + 7u - sh) / 8u) + waves / 8u;
  }
  const size_t lds = 64 * LDS_ROW * sizeof(float);
  const char* dbg = measure_switch("WAA_SCAN_DEBUG");  // measurement aid (results wrong by construction): 1 no state hand-off,
                                               // 2 static unit numbers, 3 both, 4 no output stores, 7 all three
  const int dm = dbg ? atoi(dbg) : 0;
  if (d.dup_out)
    hipLaunchKernelGGL((biquad_scan_kernel<true>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else if (dm == 1)
    hipLaunchKernelGGL((biquad_scan_kernel<false, 1>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else if (dm == 2)
    hipLaunchKernelGGL((biquad_scan_kernel<false, 2>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else if (dm == 3)
    hipLaunchKernelGGL((biquad_scan_kernel<false, 3>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else if (dm == 4)
    hipLaunchKernelGGL((biquad_scan_kernel<false, 4>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else if (dm == 7)
    hipLaunchKernelGGL((biquad_scan_kernel<false, 7>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
  else
    hipLaunchKernelGGL((biquad_scan_kernel<false>), dim3(waves), dim3(64), lds, (hipStream_t)stream, d, ctl);
}

#else
// (product build: the chained scan is a measured negative result, opt-in through WAA_BIQUAD_SCAN in the measurement build only —
// DESIGN.md 3.1g; its kernels are not part of libwaa_hip.so)
void launch_biquad_scan_powers(const double*, uint64_t, double*, uint32_t, void*) {}
void launch_biquad_scan(const BiquadStreamDesc&, const BiquadScanCtl&, uint32_t*, void*) {}
#endif  // WAA_MEASURE
}  // namespace waa
