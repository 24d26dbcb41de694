// waa_osfft.hip — WaveShaperNode 2x / 4x oversampling in ONE launch: up-sample, curve, down-sample as 256-point
// transforms in registers with one LDS exchange each (waa_osfft.hpp has the algebra), replacing the two dense matrix
// products over render quanta of waa_frozen.hip (14 ms / 28.7 ms for 1024 stereo contexts x 10 s, compute-bound on six bf16
// MFMA products per f32 product; the up-sampled signal crossed HBM twice).  Reference: waveshaper.rs:290-347 (the
// resamplers), :395-481 (process), :555-573 (the curve); rubato's FftFixedInOut restated in DESIGN.md 3.5.
//
// Work decomposition.  A GROUP of 16 lanes renders a run of `seg_len` render quanta of one instance, both channels packed
// as z = L + i R, quantum after quantum with the two stages' overlaps in registers (the only state the node has).  The
// overlap a run starts with is recomputed, not communicated: the two PROCESSED quanta in front of the run are rendered first
// without being stored (the up-sampling overlap of the first feeds the second, whose two overlaps are what the run needs;
// found through the node's `prev` table, which is also how skipped quanta — silent input, curve(0) = 0 — and re-created
// resamplers — channel count changes — are followed: LINK_SKIP leaves the overlaps alone and writes silence, LINK_FRESH
// clears them).  Four groups per wavefront run the same instruction stream on their own quanta; what differs between them
// (skip / fresh / store) is a per-lane predicate.  A workgroup is four wavefronts that share the spectral tables and the
// curve in LDS; the exchange buffers are per group.  seg_len is chosen on the host so that the launch has ~16 k groups.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "waa_internal.hpp"
#include "waa_osfft.hpp"

namespace waa {
namespace {
using namespace osfft;
constexpr int WAVES = 4;
constexpr int CURVE_LDS_MAX = 4100;

// (the scheduling barriers keep the phases apart: without them the scheduler starts the next branch's transform early
// and the live values of two phases add up)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <int R, bool CLDS, bool STEREO>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void osfft_kernel(const OsFftDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  // LDS map (8-byte slots): 2 R tables | WAVES * 4 exchange buffers | curve (floats)
  const ldsp tab = (ldsp)lds_raw;
  const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, t = lane & 15;
  const ldsp ex = tab + 2 * R * TAB_SLOTS + (wv * 4 + g) * XSLOTS;
  const __attribute__((address_space(3))) float* cv =
      (const __attribute__((address_space(3))) float*)(tab + 2 * R * TAB_SLOTS + WAVES * 4 * XSLOTS);
  {
    const f4v* src = reinterpret_cast<const f4v*>(d.tables);
    __attribute__((address_space(3))) f4v* dst = (__attribute__((address_space(3))) f4v*)tab;
    for (int i = threadIdx.x; i < 2 * R * TAB_SLOTS / 2; i += WAVES * 64) dst[i] = load_global_f4(reinterpret_cast<const float*>(src + i));
    if (CLDS) {
      __attribute__((address_space(3))) float* cw = (__attribute__((address_space(3))) float*)cv;
      for (int i = threadIdx.x; i <= d.curve_n; i += WAVES * 64) cw[i] = load_global(d.curve + (i < d.curve_n ? i : d.curve_n - 1));  // (+ the pad)
    }
  }
  __syncthreads();
  const uint64_t gid = ((uint64_t)blockIdx.x * WAVES + wv) * 4 + g;
  const uint32_t inst = (uint32_t)(gid / d.n_seg), seg = (uint32_t)(gid % d.n_seg);
  const bool alive = inst < d.n_inst;
  const int32_t* prev = d.prev + (uint64_t)(alive ? inst : 0) * d.prev_stride;
  const int q_lo = (int)(d.q0 + seg * d.seg_len);
  const int q_hi = (int)((uint64_t)q_lo + d.seg_len < d.q1 ? q_lo + d.seg_len : d.q1);
  // the two processed quanta in front of the run (-1: none).  The 16 lanes of the group look at 16 table entries per step.
  int p1 = -1, p2 = -1;
  if (q_lo > 0) {  // (q_lo depends on the group only through `seg`; groups past the end idle through the loop)
    int base = q_lo - 1;
    bool searching = alive && q_lo < (int)d.q1;
    while (__builtin_amdgcn_ballot_w64(searching) != 0) {
      const int qq = base - t;
      const int32_t l = (searching && qq >= 0) ? load_global(prev + qq) : LINK_SKIP;
      const uint64_t hit = __builtin_amdgcn_ballot_w64(l != LINK_SKIP) >> (g * 16) & 0xffffull;
      if (searching) {
        if (hit) {
          p1 = base - __builtin_ctzll(hit);
          searching = false;
        } else {
          base -= 16;
          if (base < 0) searching = false;
        }
      }
    }
    if (p1 >= 0) {
      const int32_t l1 = load_global(prev + p1);
      p2 = l1 >= 0 ? l1 : -1;
    }
  }
  Lane<R> L;
  load_tw(reinterpret_cast<const c2v*>(d.tw256), t, L.tws);
  lane_reset(L);
  const float* src = d.src + (uint64_t)(alive ? inst : 0) * d.src_inst;
  float* dst = d.dst + (uint64_t)(alive ? inst : 0) * d.dst_inst;
  constexpr bool stereo = STEREO;
  const float c_first = d.curve_n > 0 ? load_global(d.curve) : 0.f, c_last = d.curve_n > 0 ? load_global(d.curve + d.curve_n - 1) : 0.f;
  const int n_it = (int)d.seg_len + 2;
  // step `it` of a group: its quantum (the two run heads first), whether the node processes it, whether it is stored
  auto quantum_of = [&](int it) { return it == 0 ? p2 : it == 1 ? p1 : q_lo + it - 2; };
  auto valid_at = [&](int it, int q) { return alive && it < n_it && q >= 0 && (it < 2 || q < q_hi); };
  const float* safe = d.tw256;  // (any 128 readable floats: what a lane that does not process loads and throws away)
  c2v xn[8];                    // the next step's input frames, requested one step ahead
  int32_t link_n;
  auto request = [&](int q, bool procn) __attribute__((always_inline)) {
    const float* p0 = procn ? src + (uint64_t)q * RQ + t : safe + t;
    const float* p1c = procn ? p0 + d.src_ch : safe + t;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      xn[j].x = load_global(p0 + 16 * j);
      xn[j].y = stereo ? load_global(p1c + 16 * j) : 0.f;
    }
  };
  {
    const int q0 = quantum_of(0);
    link_n = valid_at(0, q0) ? load_global(prev + q0) : LINK_SKIP;
    request(q0, link_n != LINK_SKIP);
  }
#pragma unroll 1
  for (int it = 0; it < n_it; it++) {
    const int q = quantum_of(it);
    const int32_t link = link_n;
    const bool proc = link != LINK_SKIP;           // (LINK_SKIP also stands for "no quantum in this step")
    const bool store = alive && it >= 2 && q < q_hi;
    const int qn = quantum_of(it + 1);
    link_n = valid_at(it + 1, qn) ? load_global(prev + qn) : LINK_SKIP;
    // (no early exit for a step in which no group processes anything: a second loop latch costs ~50 registers of copies —
    // measured — and with them the second wavefront per SIMD; such a step computes on zeros and stores zeros)
    lane_reset_if(L, proc && link == LINK_FRESH);  // re-created resamplers (or the first processed quantum)
    // (the tables are loop-invariant LDS reads: the address is made opaque so that they are not hoisted out of the loop)
    int tofs = 0;
    asm volatile("" : "+s"(tofs));
    const ldsp tabq = tab + tofs;
    {
      c2v x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x[j].x = proc ? xn[j].x : 0.f;
        x[j].y = proc ? xn[j].y : 0.f;
      }
      ph_in(L, x);
    }
    xwrite(L.a, ex, t);
    wsync();
    xread(L.a, ex, t);
    wsync();
    ph_spec(L);
#pragma unroll
    for (int r = 0; r < R; r++) {
      ph_up(L, tabq + r * TAB_SLOTS, t);
      xwrite(L.a, ex, t);
      wsync();
      xread(L.a, ex, t);
      wsync();
      c2v u[8];
      ph_up_out(L, r, proc, u);
#pragma unroll
      for (int j = 0; j < 8; j++)
        u[j] = CLDS ? shape2<true>(cv, d.curve_n, c_first, c_last, u[j])
                    : shape2<false>((const WAA_GLOBAL_AS float*)d.curve, d.curve_n, c_first, c_last, u[j]);
      ph_dn(L, u);
      xwrite(L.a, ex, t);
      wsync();
      xread(L.a, ex, t);
      wsync();
      ph_dn_acc(L, r, tabq + (R + r) * TAB_SLOTS, t);
    }
    ph_out(L);
    // the next step's frames: requested here, where the spectra of this step are dead, one exchange and a transform pass
    // (and the other wavefront of the SIMD) ahead of their use
    request(qn, link_n != LINK_SKIP);
    xwrite(L.a, ex, t);
    wsync();
    xread(L.a, ex, t);
    wsync();
    c2v o[8];
    ph_out_end(L, proc, o);
    {
      // (lanes that do not store write their values to a scratch row instead of branching around sixteen stores)
      float* p0 = store ? dst + (uint64_t)q * RQ + t : d.trash + lane;
      float* p1c = store && stereo ? p0 + d.dst_ch : d.trash + lane;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        store_global(p0 + (store ? 16 * j : 0), proc ? o[j].x : 0.f);
        if (stereo) store_global(p1c + (store ? 16 * j : 0), proc ? o[j].y : 0.f);
      }
    }
  }
}
}  // namespace

size_t osfft_lds_bytes(int R, int curve_n) {
  const bool clds = curve_n <= CURVE_LDS_MAX;
  return ((size_t)2 * R * TAB_SLOTS + (size_t)WAVES * 4 * XSLOTS) * 8 + (clds ? (size_t)((curve_n + 4) & ~3) * 4 : 0);
}
void launch_osfft(const OsFftDesc& d0, void* stream) {
  OsFftDesc d = d0;
  if (d.q1 == 0) d.q1 = d.n_quanta;  // (the whole render)
  if (d.q0 != 0 || d.q1 != d.n_quanta) {  // a range: as many runs as it needs
    const uint32_t nq = d.q1 > d.q0 ? d.q1 - d.q0 : 0;
    if (nq == 0) return;
    // (the plan's run length fills the device with the WHOLE render's quanta; a block of a quantum-blocked loop is a few dozen quanta
    // per instance: runs as short as it takes to have ~16 k groups again, at least 8 quanta — two unstored heads per run)
    const uint32_t fill = (uint32_t)(((uint64_t)d.n_inst * nq + 16383) / 16384);
    d.seg_len = std::min(nq, std::max<uint32_t>(8, fill));
    d.n_seg = (nq + d.seg_len - 1) / d.seg_len;
  }
  const bool clds = d.curve_n <= CURVE_LDS_MAX;
  const size_t lds = osfft_lds_bytes(d.R, d.curve_n);
  const uint64_t groups = (uint64_t)d.n_inst * d.n_seg;
  const dim3 grid((unsigned)((groups + WAVES * 4 - 1) / (WAVES * 4))), block(WAVES * 64);
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) raise_lds_limit(reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, d);
  };
  const bool st = d.nch == 2;
  if (d.R == 2) {
    if (st)
      clds ? go(osfft_kernel<2, true, true>) : go(osfft_kernel<2, false, true>);
    else
      clds ? go(osfft_kernel<2, true, false>) : go(osfft_kernel<2, false, false>);
  } else {
    if (st)
      clds ? go(osfft_kernel<4, true, true>) : go(osfft_kernel<4, false, true>);
    else
      clds ? go(osfft_kernel<4, true, false>) : go(osfft_kernel<4, false, false>);
  }
}

}  // namespace waa
