// waa_biquad_lanes.hip — the a-rate Biquad with ONE coefficient table for all instances (biquad_filter.rs:837-855: per-frame
// coefficients; the usual automation: the same ramp scheduled on every context — BASELINE config 1's a-rate variant x 1024),
// parallel over streams AND time, one LANE per stream.
//
// The per-stream wavefront of waa_biquad_stream.hip streams 116 KB of coefficient sets and digests per tile and is bound by
// one wavefront's dependent f64 chain: 2048 wavefronts however long the render is, 0.25 of the HBM peak.
// Here the coefficient set of a frame is the same for every stream, so a wavefront takes 64 STREAMS — one per lane — and one
// TILE of 2048 frames: the five coefficients of a frame are scalar operands (one scalar load for 64 streams instead of 64
// vector loads), every lane runs the reference's recurrence in the reference's order with its explicit flush
// (biquad_filter.rs:877-883) — no scan, no zero-state pass, nothing computed twice — and (tile, stream group) units fill the
// chip: 235 x 32 = 7520 wavefronts for C1a.  What a tile needs from the past is two numbers per stream, its incoming y state:
//
//   digest (once per plan)  per tile T and frame i: H_i with  y_end(T; zero incoming state) = sum_i H_i x_i + Hm1 x[-1] +
//                           Hm2 x[-2], and the tile's transition P_T (products of the per-frame [[-a1, -a2], [1, 0]])
//   pass A                  Z[T][s] = that sum over the tile's samples: two FMAs per frame, H_i scalar
//   chain                   S[T+1][s] = P_T S[T][s] + Z[T][s]: one thread per stream, 235 steps
//   pass B                  the exact recurrence of every (tile, stream) from S[T][s]
//
// The samples are read twice (pass A, pass B): 11.8 GB per C1a render instead of 7.9 — the price of tiles that do not wait
// for each other.  Samples travel as full 128-byte lines: a wavefront's load instruction fetches 8 rows (streams) x 128 B, an
// LDS transpose hands every lane its stream's 32 frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "waa_internal.hpp"
#include "waa_stream_common.hpp"

namespace waa {

namespace {
struct V2 {
  double x, y;
};
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
__device__ __forceinline__ V2 mvv(const M2& m, const V2& v) { return V2{__builtin_fma(m.a, v.x, m.b * v.y), __builtin_fma(m.c, v.x, m.d * v.y)}; }
constexpr int CHUNK = 32;                 // frames per lane and LDS round trip
constexpr int NCHUNK = TILE / CHUNK;      // 64 chunks per tile
constexpr int ROWF = CHUNK + 4;           // LDS row: 32 frames + 4 (conflict-free 16-byte row accesses)
// Uniform tables (digests, coefficient sets) are read through the CONSTANT address space: scalar loads into scalar
// registers, one per wavefront instead of one per lane.  (Through the kernel argument's generic pointer the compiler may
// not assume the tables are unchanged by the kernel's own stores and emits per-lane vector loads — 128 registers of hoisted
// loads and 207 spilled in the first build.)  The tables are written by earlier launches only.
typedef const __attribute__((address_space(4))) double* cdp;
}  // namespace

// ---- tile digests from the frame-major coefficient table ---------------------------------------------------------------------
// ht[tile]: 2048 x (Hx, Hy), then Hm1 (2), Hm2 (2), P (4).  One wavefront per tile; lane l digests frames 32 l .. 32 l + 31
// backwards (biquad_hp_kernel's recurrence), the lanes' pieces are chained with the suffix products S_l = P_63 ... P_{l+1}.
__global__ __launch_bounds__(64) void biquad_tile_digest_kernel(const BiquadLanesDesc d) {
  __shared__ double sh[64][12];  // per lane: P (4), Hm1 (2), Hm2 (2), S (4)
  const uint32_t tile = blockIdx.x;
  const int lane = threadIdx.x;
  const double* ct = d.coefs + ((uint64_t)tile * TILE + (uint64_t)lane * CHUNK) * 5;
  double* out = d.ht + (uint64_t)tile * BIQUAD_HT_WORDS;
  // Phi_i = M_31 ... M_{i+1}; G_i = its first column; Phi_{i-1} = Phi_i M_i with M_i = [[-a1, -a2], [1, 0]]
  double pa = 1., pb = 0., pc = 0., pd = 1.;
  double g1a = 0., g1c = 0., g2a = 0., g2c = 0.;  // G_{i+1}, G_{i+2}
  double b1n = 0., b2n = 0., b2nn = 0.;           // b1_{i+1}, b2_{i+1}, b2_{i+2}
  double hx[CHUNK], hy[CHUNK];
  double m1x = 0., m1y = 0., m2x = 0., m2y = 0.;
  for (int i = CHUNK - 1; i >= 0; i--) {
    const double b0 = ct[i * 5 + 0], b1 = ct[i * 5 + 1], b2 = ct[i * 5 + 2], a1 = ct[i * 5 + 3], a2 = ct[i * 5 + 4];
    hx[i] = pa * b0 + g1a * b1n + g2a * b2nn;  // H_i = G_i b0_i + G_{i+1} b1_{i+1} + G_{i+2} b2_{i+2}
    hy[i] = pc * b0 + g1c * b1n + g2c * b2nn;
    if (i == 0) {
      m1x = pa * b1 + g1a * b2n;  // Hm1 = G_0 b1_0 + G_1 b2_1
      m1y = pc * b1 + g1c * b2n;
      m2x = pa * b2;              // Hm2 = G_0 b2_0
      m2y = pc * b2;
    }
    g2a = g1a;
    g2c = g1c;
    g1a = pa;
    g1c = pc;
    b2nn = b2n;
    b1n = b1;
    b2n = b2;
    const double na = __builtin_fma(-a1, pa, pb), nc = __builtin_fma(-a1, pc, pd);
    pb = -a2 * pa;
    pd = -a2 * pc;
    pa = na;
    pc = nc;
  }
  sh[lane][0] = pa;
  sh[lane][1] = pb;
  sh[lane][2] = pc;
  sh[lane][3] = pd;
  sh[lane][4] = m1x;
  sh[lane][5] = m1y;
  sh[lane][6] = m2x;
  sh[lane][7] = m2y;
  __syncthreads();
  if (lane == 0) {  // suffix products (plan time: 63 small products, serially)
    M2 S = {1., 0., 0., 1.};
    for (int l = 63; l >= 0; l--) {
      sh[l][8] = S.a;
      sh[l][9] = S.b;
      sh[l][10] = S.c;
      sh[l][11] = S.d;
      S = mm(S, M2{sh[l][0], sh[l][1], sh[l][2], sh[l][3]});
    }
    // S is now P_63 ... P_0: the tile's transition
    out[2 * TILE + 4] = S.a;
    out[2 * TILE + 5] = S.b;
    out[2 * TILE + 6] = S.c;
    out[2 * TILE + 7] = S.d;
  }
  __syncthreads();
  const M2 S = {sh[lane][8], sh[lane][9], sh[lane][10], sh[lane][11]};
  for (int i = 0; i < CHUNK; i++) {
    V2 h = mvv(S, V2{hx[i], hy[i]});
    if (lane < 63 && i >= CHUNK - 2) {
      // the next lane's history terms belong to this lane's last two frames: x[l+1][-1] = x[l][31], x[l+1][-2] = x[l][30]
      const M2 Sn = {sh[lane + 1][8], sh[lane + 1][9], sh[lane + 1][10], sh[lane + 1][11]};
      const V2 t = i == CHUNK - 1 ? mvv(Sn, V2{sh[lane + 1][4], sh[lane + 1][5]}) : mvv(Sn, V2{sh[lane + 1][6], sh[lane + 1][7]});
      h.x += t.x;
      h.y += t.y;
    }
    out[2 * (lane * CHUNK + i)] = h.x;
    out[2 * (lane * CHUNK + i) + 1] = h.y;
  }
  if (lane == 0) {
    const V2 a = mvv(S, V2{m1x, m1y}), bb = mvv(S, V2{m2x, m2y});
    out[2 * TILE + 0] = a.x;
    out[2 * TILE + 1] = a.y;
    out[2 * TILE + 2] = bb.x;
    out[2 * TILE + 3] = bb.y;
  }
}

namespace {
// one frame of one (instance, channel) stream the slow way (load_channel_generic's rules, one element): frames in front of a
// tile whose predecessor the fast track does not cover
__device__ float source_frame(const SrcInst& si, int ch, uint64_t frame, uint32_t n_quanta) {
  const uint32_t q = (uint32_t)(frame / RQ);
  if (q >= n_quanta) return 0.f;
  const QRec r = load_global(si.sc.qrec + q);
  const uint32_t i = (uint32_t)(frame % RQ);
  const float* chp = si.base + (uint64_t)ch * si.ch_stride;
  if (r.mode == Q_FAST || r.mode == Q_FAST_LOOP) {
    uint64_t bi = (uint64_t)r.start + i;
    if (bi >= si.frames) {
      if (r.mode != Q_FAST_LOOP) return 0.f;
      bi = bi % si.frames;
    }
    return load_global(chp + bi);
  }
  if (r.mode == Q_SLOW) {
    const SlowRec s = load_global(si.sc.slow + (uint64_t)q * RQ + i);
    if (s.prev < 0) return 0.f;
    const double ps = (double)load_global(chp + s.prev);
    const double ns = s.next >= 0 ? (double)load_global(chp + s.next) : s.next == -1 ? 0. : 2. * ps - (double)load_global(chp + s.prev - 1);
    return (float)__builtin_fma(1. - s.k, ps, s.k * ns);
  }
  return 0.f;
}
}  // namespace

// PASS 0: the tiles' zero-state end states (pass A).  PASS 1: the exact render (pass B).
// FAST: every stream is one linear run over all tiles of the launch (the host knows: BiquadLanesDesc::fast_tiles) — that
// instantiation carries no schedule record, no limits and no per-load pointer shuffles; the general one renders the rest
// (tiles behind a source's fast prefix, bounded signals).
template <int PASS, bool FAST>
__global__ __launch_bounds__(64, 4) void biquad_lanes_kernel(const BiquadLanesDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [64][ROWF] chunk rows, then [2][TW] doubles of table
  const int lane = threadIdx.x;
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch;
  const uint32_t n_groups = (n_streams + 63u) / 64u;
  const uint32_t tile = d.lt0 + blockIdx.x / n_groups, g = blockIdx.x % n_groups;
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (FTZ/DAZ render scope, thread.rs:374-382)
  auto lds_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };
  // ---- this lane's stream: where its tile starts (null: not linear there), how many frames may be read
  const uint32_t sid = g * 64u + lane;
  const bool live = sid < n_streams;
  const uint32_t inst = live ? sid / (uint32_t)d.nch : 0u;
  const int ch = live ? (int)(sid % (uint32_t)d.nch) : 0;
  const bool is_src = d.in.kind == IN_SOURCE;
  const float* base = nullptr;    // frame 0 of the render in a linear input
  uint64_t lin_frames = 0;        // frames [0, lin_frames) of the render are base[frame]; zeros beyond for a bounded signal
  bool generic_tail = false;      // (source) frames >= lin_frames follow the schedule tables
  SrcInst si{};
  if (FAST) {
    if (is_src) {
      const SrcInst* sp = d.in.src + inst;
      base = load_global(&sp->base) + (uint64_t)ch * load_global(&sp->ch_stride) + load_global(&sp->linear_start);
    } else {
      base = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
    }
    lin_frames = (uint64_t)d.n_tiles * TILE;
  } else if (live) {
    if (is_src) {
      si = d.in.src[inst];
      base = si.base + (uint64_t)ch * si.ch_stride + si.linear_start;
      lin_frames = (uint64_t)si.fast_prefix * TILE;
      generic_tail = true;
      if (si.linear_all) {  // the render's partial last tile continues the run: a bounded signal, nothing behind it is rendered
        lin_frames = (uint64_t)d.n_quanta * RQ;
        generic_tail = false;
      }
    } else {
      base = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
      lin_frames = d.in.valid ? d.in.valid : (uint64_t)d.n_tiles * TILE;
    }
  }
  const uint64_t f_tile = (uint64_t)tile * TILE;
  // does any stream of the group need the slow loader in this tile?  (uniform decision: the slow path is wave-cooperative)
  const bool slow_tile = !FAST && __any(live && generic_tail && f_tile + TILE > lin_frames);
  // rows are fetched by OTHER lanes: lane l loads 16 B of row (l >> 3) + 8 j — pointers and limits by shuffle.
  // The common case — every stream of the group linear over the whole tile — keeps the eight row pointers in registers
  // (advanced by one chunk per iteration) instead of shuffling pointer and limit for every load: the shuffles were 32
  // LDS-pipe operations and eight dependent waits per chunk.  (A group's dead lanes alias its first stream.)
  const float* prow[8];
  if (FAST) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int r = j * 8 + (lane >> 3);
      const bool r_live = g * 64u + (uint32_t)r < n_streams;
      const uint64_t pb = (uint64_t)__shfl((unsigned long long)(uintptr_t)base, r_live ? r : 0, 64);
      prow[j] = reinterpret_cast<const float*>(pb) + f_tile + (uint64_t)(lane & 7) * 4;
      if (d.debug == 1)  // measurement aid: the 64 rows of a chunk as ONE contiguous 8 KB piece (what a transposed layout would give)
        prow[j] = (is_src ? load_global(&d.in.src->base) : d.in.sig.base) + ((uint64_t)tile * n_groups + g) * 64 * TILE + r * CHUNK + (lane & 7) * 4;
    }
  }
  const uint32_t row_step = FAST && d.debug == 1 ? 64u * CHUNK : (uint32_t)CHUNK;
  // measurement aid (debug 4, pass A only — its sum does not care about the order): every workgroup starts at a different
  // chunk of its tile, so that the wavefronts of the device do not walk addresses that are congruent modulo the stream stride
  const int rot = (PASS == 0 && FAST && d.debug == 4) ? (int)((g * 5u + tile * 3u) & (NCHUNK - 1)) : 0;
  int fetch_c = rot;
  if (rot) {
#pragma unroll
    for (int j = 0; j < 8; j++) prow[j] += (uint32_t)rot * row_step;
  }
  auto fetch_chunk_fast = [&](f4v (&raw)[8]) __attribute__((always_inline)) {
    if (d.debug == 2) return;  // measurement aid: no sample loads (everything but the memory side)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      raw[j] = __builtin_nontemporal_load((const WAA_GLOBAL_AS f4v*)prow[j]);
      prow[j] += row_step;
    }
    if (PASS == 0 && FAST && d.debug == 4) {
      fetch_c++;
      if (fetch_c == NCHUNK) {
        fetch_c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) prow[j] -= (uint32_t)NCHUNK * row_step;
      }
    }
  };
  auto fetch_chunk = [&](int c, f4v (&raw)[8]) __attribute__((always_inline)) {
    const uint64_t f0 = f_tile + (uint64_t)c * CHUNK + (uint64_t)(lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int r = j * 8 + (lane >> 3);
      const uint64_t pb = (uint64_t)__shfl((unsigned long long)(uintptr_t)base, r, 64);
      const uint64_t lim = __shfl((unsigned long long)lin_frames, r, 64);
      const float* p = reinterpret_cast<const float*>(pb);
      const bool ok = p && f0 + 3 < lim;
      // (unconditional load, the zero selected afterwards: a load behind a branch is waited for on the spot)
      const float* q = ok ? p + f0 : reinterpret_cast<const float*>(d.coefs);
      raw[j] = __builtin_nontemporal_load((const WAA_GLOBAL_AS f4v*)q);
      if (!ok) raw[j] = f4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage_chunk = [&](int buf, const f4v (&raw)[8]) __attribute__((always_inline)) {
    float* dst = lds + buf * (64 * ROWF);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int r = j * 8 + (lane >> 3);
      *reinterpret_cast<f4v*>(dst + r * ROWF + (lane & 7) * 4) = raw[j];
    }
  };
  // slow path: the chunk's frames one by one (source tiles behind the fast prefix: the buffer's end, loops, the slow track)
  auto stage_chunk_slow = [&](int buf, int c) __attribute__((always_inline)) {
    float* dst = lds + buf * (64 * ROWF) + lane * ROWF;
    for (int k = 0; k < CHUNK; k++) {
      const uint64_t f = f_tile + (uint64_t)c * CHUNK + k;
      float v = 0.f;
      if (live) {
        if (f < lin_frames)
          v = load_global(base + f);
        else if (generic_tail)
          v = source_frame(si, ch, f, d.n_quanta);
      }
      dst[k] = v;
    }
  };
  // ---- state in front of the tile
  double x1 = 0., x2 = 0., y1 = 0., y2 = 0.;
  double* st = d.state + (uint64_t)inst * STATE_STRIDE + ch * 4;
  if (live) {
    if (tile == d.tile0) {
      if (PASS == 0) {  // (pass B: from the chain kernel's copy, below — `state` is being written by this launch)
        x1 = st[0];
        x2 = st[1];
      }
    } else if (f_tile <= lin_frames) {
      x1 = (double)load_global(base + f_tile - 1);
      x2 = (double)load_global(base + f_tile - 2);
    } else if (generic_tail) {
      x1 = (double)source_frame(si, ch, f_tile - 1, d.n_quanta);
      x2 = (double)source_frame(si, ch, f_tile - 2, d.n_quanta);
    }
    if (PASS == 1) {
      const double* sp = d.sin + ((uint64_t)tile * n_streams + sid) * 2;  // (also for the first tile: the chain kernel's copy)
      y1 = load_global(sp);
      y2 = load_global(sp + 1);
      if (tile == d.tile0) {
        const double* xs = d.z + ((uint64_t)tile * n_streams + sid) * 2;
        x1 = load_global(xs);
        x2 = load_global(xs + 1);
      }
    }
  }
  const cdp ht = (cdp)(d.ht + (uint64_t)tile * BIQUAD_HT_WORDS);
  double z1 = 0., z2 = 0.;
  if (PASS == 0) {
    // history terms of the zero-state response
    z1 = __builtin_fma(ht[2 * TILE + 0], x1, ht[2 * TILE + 2] * x2);
    z2 = __builtin_fma(ht[2 * TILE + 1], x1, ht[2 * TILE + 3] * x2);
  }
  float g0 = 1.f, g1 = 1.f;  // constant gains behind the filter (gain.rs:163-179 fast paths)
  bool mute = false;
  if (PASS == 1 && live) {
    if (d.n_gain > 0) g0 = d.gain[0].base[inst];
    if (d.n_gain > 1) g1 = d.gain[1].base[inst];
    mute = (d.n_gain > 0 && fabsf(g0) <= 1e-6f) || (d.n_gain > 1 && fabsf(g1) <= 1e-6f);
    if (fabsf(1.f - g0) <= 1e-6f) g0 = 1.f;
    if (fabsf(1.f - g1) <= 1e-6f) g1 = 1.f;
  }
  // The per-frame table of a chunk — pass A: 32 x (Hx, Hy) = 512 B, pass B: 32 x 5 coefficients = 1280 B — is the same for
  // all 64 lanes.  It travels like the samples: one coalesced vector load a chunk ahead (registers), staged into LDS, read
  // back as BROADCAST reads (every lane the same address: one bank access).  As scalar loads (first build: s_load_dwordx16,
  // two or three in flight for lack of scalar registers, each an L2 round trip) the table was what every wave waited for:
  // 71 % of the wave-cycles parked, 620 cycles per frame.
  constexpr int TW = PASS == 0 ? CHUNK * 2 : CHUNK * 5;   // doubles per chunk
  double* tabs = reinterpret_cast<double*>(lds + 64 * ROWF);   // [TW]: staged at the top of an iteration, when the previous chunk's sums are done
  const double* tsrc = PASS == 0 ? d.ht + (uint64_t)tile * BIQUAD_HT_WORDS : d.coefs + f_tile * 5;
  typedef double d2v __attribute__((ext_vector_type(2)));
  d2v traw[PASS == 0 ? 1 : 2];
  auto fetch_tab = [&](int c) __attribute__((always_inline)) {
    const double* p = tsrc + (uint64_t)c * TW;
    // lane l carries doubles 2 l, 2 l + 1 (pass B: lanes 0..15 also 128 + 2 l, 129 + 2 l); lanes past the table re-read its start
    traw[0] = *(const WAA_GLOBAL_AS d2v*)(p + (2 * lane < TW ? 2 * lane : 0));
    if constexpr (PASS == 1) traw[1] = *(const WAA_GLOBAL_AS d2v*)(p + 128 + (lane < 16 ? 2 * lane : 0));
  };
  auto stage_tab = [&](int buf) __attribute__((always_inline)) {
    double* t = tabs + buf * TW;
    if (2 * lane < TW) *reinterpret_cast<d2v*>(t + 2 * lane) = traw[0];
    if constexpr (PASS == 1) {
      if (lane < 16) *reinterpret_cast<d2v*>(t + 128 + 2 * lane) = traw[1];
    }
  };
  // ---- 64 chunks of 32 frames, software-pipelined: chunk c + 1 (samples and table) is in flight in registers while chunk c
  // is rendered; ONE row buffer in LDS (9 KB per wavefront: four wavefronts per SIMD)
  f4v raw[8];
  if (FAST)
    fetch_chunk_fast(raw);
  else if (!slow_tile)
    fetch_chunk(0, raw);
  fetch_tab(rot);
  // output rows as 32-bit element offsets from the signal's base (the planner keeps signals below 2^32 elements on this path)
  uint32_t orow[8];
  if (PASS == 1) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int r = j * 8 + (lane >> 3);
      const uint32_t sr = g * 64u + (uint32_t)r;
      const bool r_live = sr < n_streams;
      const uint32_t ri = sr / (uint32_t)d.nch, rc = sr % (uint32_t)d.nch;
      orow[j] = r_live ? (uint32_t)((uint64_t)ri * d.out.inst_stride + (uint64_t)rc * d.out.ch_stride + f_tile + (uint64_t)(lane & 7) * 4)
                       : 0xFFFFFFFFu;
      if (FAST && d.debug == 1) orow[j] = (uint32_t)(((uint64_t)tile * n_groups + g) * 64 * TILE + r * CHUNK + (lane & 7) * 4);
    }
  }
  // (the render's last tile may be partial: chunks behind the render's end are not read and written as zeros — a source whose
  // whole render is one linear run, SrcInst::linear_all, then needs no general launch for that tile: a lone wavefront per CU
  // there paid a full memory round trip per chunk, 0.5 ms per pass behind the other 234 tiles before it was bounded)
  const uint64_t f_end = (uint64_t)d.n_quanta * RQ;
  const int n_chunk = f_tile + TILE <= f_end ? NCHUNK : (int)((f_end - f_tile + CHUNK - 1) / CHUNK);  // (FAST: f_end is a multiple of CHUNK)
  for (int c = 0; c < n_chunk; c++) {
    if (FAST || !slow_tile)
      stage_chunk(0, raw);
    else
      stage_chunk_slow(0, c);
    stage_tab(0);
    if (c + 1 < n_chunk) {
      if (FAST)
        fetch_chunk_fast(raw);
      else if (!slow_tile)
        fetch_chunk(c + 1, raw);
      fetch_tab((c + 1 + rot) & (NCHUNK - 1));
    }
    lds_sync();
    float* row = lds + lane * ROWF;
    const double* tb = tabs;
    if (d.debug == 3) {
      // measurement aid: no arithmetic (loads, staging and stores only)
    } else if (PASS == 0) {
      // (eight frames at a time: with all 32 table reads of the chunk hoisted in front of the sums the kernel spilled)
#pragma unroll
      for (int k8 = 0; k8 < CHUNK / 8; k8++) {
        const f4v va = *reinterpret_cast<const f4v*>(row + k8 * 8), vb = *reinterpret_cast<const f4v*>(row + k8 * 8 + 4);
        const float xs[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const d2v h = *reinterpret_cast<const d2v*>(tb + 2 * (k8 * 8 + e));
          const double xd = (double)xs[e];
          z1 = __builtin_fma(h.x, xd, z1);
          z2 = __builtin_fma(h.y, xd, z2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int k4 = 0; k4 < CHUNK / 4; k4++) {
        float yo[4];
        const f4v xv = *reinterpret_cast<const f4v*>(row + k4 * 4);  // (read per group: 32 samples held in registers cost a wave per SIMD)
        const float x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          double cs[10];  // two frames' coefficient sets: five broadcast 16-byte reads (four frames' worth cost 20 more registers)
#pragma unroll
          for (int e = 0; e < 5; e++) {
            const d2v v = *reinterpret_cast<const d2v*>(tb + k4 * 20 + h * 10 + 2 * e);
            cs[2 * e] = v.x;
            cs[2 * e + 1] = v.y;
          }
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const double b0 = cs[e * 5 + 0], b1 = cs[e * 5 + 1], b2 = cs[e * 5 + 2], a1 = cs[e * 5 + 3], a2 = cs[e * 5 + 4];
            const double xd = (double)x4[h * 2 + e];
            // biquad_filter.rs:877-883, the reference's order, unfused, with its flush (denormals: hardware mode)
            double y = (b0 * xd + b1 * x1) + b2 * x2;
            y = (y - a1 * y1) - a2 * y2;
            if (!__builtin_isfinite(y)) y = 0.;
            x2 = x1;
            x1 = xd;
            y2 = y1;
            y1 = y;
            float o = (float)y;
            if (d.n_gain > 0 && g0 != 1.f) o *= g0;
            if (d.n_gain > 1 && g1 != 1.f) o *= g1;
            yo[h * 2 + e] = mute ? 0.f : o;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        *reinterpret_cast<f4v*>(row + k4 * 4) = f4v{yo[0], yo[1], yo[2], yo[3]};
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_sync();
    if (PASS == 1) {
      // rows -> full 128-byte lines of the streams' outputs
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int r = j * 8 + (lane >> 3);
        const f4v v = *reinterpret_cast<const f4v*>(lds + r * ROWF + (lane & 7) * 4);
        if (orow[j] != 0xFFFFFFFFu) {  // (offsets are multiples of 4: never the marker)
          __builtin_nontemporal_store(v, (WAA_GLOBAL_AS f4v*)(d.out.base + orow[j]));
          orow[j] += FAST ? row_step : (uint32_t)CHUNK;
        }
      }
      lds_sync();  // (the rows are restaged at the top of the next iteration)
    }
  }
  if (PASS == 1) {
    // the padding behind the render's end stays defined (whole-tile consumers read it): zeros, no round trips
    for (int c = n_chunk; c < NCHUNK; c++) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (orow[j] != 0xFFFFFFFFu) {
          __builtin_nontemporal_store(f4v{0.f, 0.f, 0.f, 0.f}, (WAA_GLOBAL_AS f4v*)(d.out.base + orow[j]));
          orow[j] += FAST ? row_step : (uint32_t)CHUNK;
        }
    }
  }
  if (!live) return;
  if (PASS == 0) {
    double* zp = d.z + ((uint64_t)tile * n_streams + sid) * 2;
    zp[0] = z1;
    zp[1] = z2;
  } else if (tile + 1 == d.tile1) {  // carried state for a later launch of a block-scheduled plan
    st[0] = x1;
    st[1] = x2;
    st[2] = y1;
    st[3] = y2;
  }
}

// S[T+1] = P_T S[T] + Z[T] per stream: one thread per stream, the tiles in order; Z loads do not depend on the chain
__global__ __launch_bounds__(256) void biquad_lanes_chain_kernel(const BiquadLanesDesc d) {
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch;
  const uint32_t sid = blockIdx.x * 256 + threadIdx.x;
  if (sid >= n_streams) return;
  const uint32_t inst = sid / (uint32_t)d.nch;
  const int ch = (int)(sid % (uint32_t)d.nch);
  const double* st = d.state + (uint64_t)inst * STATE_STRIDE + ch * 4;
  double s1 = st[2], s2 = st[3];
  for (uint32_t t0 = d.tile0; t0 < d.tile1; t0 += 8) {
    double zz[8][2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t t = t0 + k < d.tile1 ? t0 + k : d.tile1 - 1;
      const double* zp = d.z + ((uint64_t)t * n_streams + sid) * 2;
      zz[k][0] = load_global(zp);
      zz[k][1] = load_global(zp + 1);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t t = t0 + k;
      if (t >= d.tile1) break;
      double* sp = d.sin + ((uint64_t)t * n_streams + sid) * 2;
      sp[0] = s1;
      sp[1] = s2;
      const double* pt = d.ht + (uint64_t)t * BIQUAD_HT_WORDS + 2 * TILE + 4;
      const double n1 = __builtin_fma(pt[0], s1, __builtin_fma(pt[1], s2, zz[k][0]));
      const double n2 = __builtin_fma(pt[2], s1, __builtin_fma(pt[3], s2, zz[k][1]));
      s1 = n1;
      s2 = n2;
    }
  }
  // Pass B takes the state in front of the launch's FIRST tile from here (y: sin[tile0], above; x: the slot z[tile0], consumed
  // by now) and not from `state`: the workgroup of the launch's LAST tile writes the carried state there, and nothing orders
  // it behind the first tile's read — with the last tile bounded by the render's end it is the shortest workgroup of the
  // launch and, under load, finished before the first one started (found by the fuzz campaign with eight processes on one
  // device: ~1 graph in 1000 rendered its first tile from the END state).
  double* xs = d.z + ((uint64_t)d.tile0 * n_streams + sid) * 2;
  xs[0] = st[0];
  xs[1] = st[1];
}

void launch_biquad_tile_digest(const BiquadLanesDesc& d, void* stream) {
  hipLaunchKernelGGL(biquad_tile_digest_kernel, dim3(d.n_tiles), dim3(64), 0, (hipStream_t)stream, d);
}
void launch_biquad_lanes(const BiquadLanesDesc& d0, void* stream) {
  BiquadLanesDesc d = d0;
  d.debug = measure_switch("WAA_LANES_DEBUG") ? (uint32_t)atoi(measure_switch("WAA_LANES_DEBUG")) : 0u;
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch, n_groups = (n_streams + 63u) / 64u;
  // LDS per wavefront: the row buffer + ONE table buffer — 9.5 KB (pass A) / 10.25 KB (pass B): 16 / 15 wavefronts per CU
  // (a second table buffer is not needed: the table is staged at the top of an iteration, after the previous chunk's sums).
  const size_t lds_a = 64 * ROWF * sizeof(float) + CHUNK * 2 * sizeof(double);
  const size_t lds_b = 64 * ROWF * sizeof(float) + CHUNK * 5 * sizeof(double);
  // tiles [tile0, tf): every stream linear (the instantiation without the general loader); [tf, tile1): the general one
  const uint32_t tf = measure_switch("WAA_LANES_GENERAL") ? d.tile0 : std::min(std::max(d.fast_tiles, d.tile0), d.tile1);
  auto pass = [&](auto pass_c) {
    constexpr int PASS = decltype(pass_c)::value;
    const size_t lds = (PASS == 0 ? lds_a : lds_b) + (measure_switch("WAA_LANES_LDS_PAD") ? (size_t)atoi(measure_switch("WAA_LANES_LDS_PAD")) : 0);  // (measurement aid: fewer wavefronts per CU)
    if (tf > d.tile0) {
      d.lt0 = d.tile0;
      d.lt1 = tf;
      hipLaunchKernelGGL((biquad_lanes_kernel<PASS, true>), dim3((tf - d.tile0) * n_groups), dim3(64), lds, (hipStream_t)stream, d);
    }
    if (tf < d.tile1) {
      d.lt0 = tf;
      d.lt1 = d.tile1;
      hipLaunchKernelGGL((biquad_lanes_kernel<PASS, false>), dim3((d.tile1 - tf) * n_groups), dim3(64), lds, (hipStream_t)stream, d);
    }
  };
  pass(std::integral_constant<int, 0>{});
  hipLaunchKernelGGL(biquad_lanes_chain_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, (hipStream_t)stream, d);
  pass(std::integral_constant<int, 1>{});
}

}  // namespace waa
