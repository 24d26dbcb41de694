// waa_dyn.hip — exact dynamic channel counts.  The reference's AudioRenderQuantum (quantum.rs:178-586) carries a
// channel count and a "silent" flag that change from render quantum to render quantum: a silent quantum is mono,
// `add` mixes both operands to the count computed from the receiver's channel config and the CURRENT counts
// (quantum.rs:532-569), and count-sensitive processors follow that count — the Biquad / IIR filters keep state per
// channel and start a new channel from zero (biquad_filter.rs:778-815, iir_filter.rs:323-360), the StereoPanner and
// the Panner have a mono and a stereo law (stereo_panner.rs:218-317, panner.rs:988-1057), the DelayNode re-mixes its
// whole line to the count of its current input (delay.rs:469-489) and reports silence from the DATA it read
// (delay.rs:660-668), the filters' tails end when their state leaves the normal range.  The node-major kernels render
// static counts; that is exact as long as the planner's replay finds no count change at a count-sensitive node
// (DESIGN.md section 5).  When it does, the graph (everything but sources and FFT convolvers) is rendered here:
// one wavefront per instance, render quantum by render quantum, items in the reference's processing order, every
// signal accompanied by a per-quantum code = count | CODE_SILENT.  Feedback loops need nothing extra in this form
// (the loop kernel of waa_loop.hip is the static-count special case).  Latency-bound by construction; throughput
// comes from the batch (1024 instances = 1024 wavefronts).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

#include "waa_internal.hpp"
#include "waa_mix.hpp"

namespace waa {

namespace {
__device__ __forceinline__ float pval(const ParamRef& p, uint32_t inst, uint32_t q, uint64_t frame) {
  if (p.mode == 0) return load_global(p.base + inst);
  if (p.mode == 1) return load_global(p.base + (uint64_t)inst * p.stride + q);
  return load_global(p.base + (uint64_t)inst * p.stride + frame);
}
__device__ __forceinline__ float coherent_f(const float* p) {
  return __int_as_float(__hip_atomic_load((const WAA_GLOBAL_AS int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ uint32_t coherent_u(const uint32_t* p) {
  return (uint32_t)__hip_atomic_load((const WAA_GLOBAL_AS int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waveshaper.rs:555-573
__device__ __forceinline__ float shape(const float* curve, int nn, float input) {
  if (nn == 0) return 0.f;
  const float n = (float)nn;
  const float v = (n - 1.f) / 2.0f * (input + 1.f);
  if (v <= 0.f) return load_global(curve);
  if (v >= n - 1.f) return load_global(curve + nn - 1);
  const float k = floorf(v);
  const float f = v - k;
  const int ki = (int)k;
  return (1.f - f) * load_global(curve + ki) + f * load_global(curve + ki + 1);
}
// quantum.rs:285-505 mix_inner; `sil` = every channel is the allocator's zero block.  A computed down-mix goes through
// make_mut, so its result is no longer silent by pointer identity (quantum.rs:96-104); up-mixes copy or pad.
template <int CM>
__device__ __forceinline__ void mixn(float (&u)[CM][2], int from, int to, int interp, bool& sil) {
  if (from == to) return;
  mix_regs<CM, 2>(u, from, to, interp);
  if (mix_is_computed(from, to, interp)) sil = false;
}
struct M2d {
  double a, b, c, d;
};
__device__ __forceinline__ M2d mm2(const M2d& x, const M2d& y) {
  M2d r;
  r.a = __builtin_fma(x.a, y.a, x.b * y.c);
  r.b = __builtin_fma(x.a, y.b, x.b * y.d);
  r.c = __builtin_fma(x.c, y.a, x.d * y.c);
  r.d = __builtin_fma(x.c, y.b, x.d * y.d);
  return r;
}
__device__ __forceinline__ void stereo_gains(float x, float& gl, float& gr) {  // stereo_panner.rs:74-79
  const float PI_F = 3.14159265358979323846f;
  gl = sinf((1.f - x) * PI_F / 2.f);
  gr = sinf(x * PI_F / 2.f);
}
}  // namespace

// One workgroup = one wavefront: LDS hand-offs between lanes need program order only (the DS operations of a wave execute
// in order).  __syncthreads() would also drain every outstanding global store (its release fence waits for vmcnt(0)) —
// with four to five of them per filter item and quantum the wave waited for its own output stores a dozen times per quantum.
__device__ __forceinline__ void lds_sync() { __builtin_amdgcn_wave_barrier(); }

// CM = 2: mono / stereo graphs (the round-2 kernel); CM = 6: layouts up to 5.1 (DelayNodes stay mono / stereo: the planner checks)
static_assert(offsetof(DynItem, kind) == 0 && offsetof(DynItem, cc) == 8 && offsetof(DynItem, interp) == 16 && offsetof(DynItem, nch_pub) == 24 &&
                  offsetof(DynItem, compact_ch1) == 32 && offsetof(DynItem, writer_item) == 40 && offsetof(DynItem, num_quanta) == 48 &&
                  offsetof(DynItem, in) == 56 && sizeof(DynItem) % 8 == 0,
              "dyn_kernel reads the item's scalar fields as seven 8-byte words");
// W = 1: one wavefront per instance walks items and quanta (the form of rounds 2-4).  W > 1 (round 5): the items — in THIRDS: the
// gather + mix of an item's inputs, its node (result into LDS), the result's publication to memory — are cut into W
// contiguous STAGES (DynDesc::stage_begin, the planner's choice: a DelayNode's writer and reader and everything between them
// share a stage, so do the members of a feedback loop) and stage w renders quantum t - w in step t: a software pipeline over
// the render quanta, one wavefront per stage, the items' outputs and codes of the last W quanta in an LDS ring, one workgroup
// barrier per step.  Every item runs exactly the code of the W = 1 form on exactly the same values: bit-identical
// (tests/test_dynamic_counts.py, WAA_DYN_NO_PIPE), 1 / W of the dependent instruction chain per quantum and W wavefronts per
// instance instead of one (the kernel is latency-bound: DESIGN.md section 8).
template <int CM, int W>
__device__ __forceinline__ void dyn_body(const DynDesc& d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cur_ring = lds;                                                      // [W][n_items][CM][128] outputs of the last W quanta
  // (W > 1) the MIXED INPUTS of the last W quanta: an item's front half (gather + mix) and back half (node + hand-over) may
  // belong to different stages
  // (two slots: an item's gather and node are consecutive units, so their stages are the same or neighbours)
  constexpr int MIXD = W > 1 ? 2 : 0;
  float* mix_ring = cur_ring + (size_t)W * d.n_items * CM * RQ;               // [2][n_items][CM][128] (W > 1 only)
  float* scratch_all = mix_ring + (size_t)MIXD * d.n_items * CM * RQ;         // [W][CM][128]
  double* fst = reinterpret_cast<double*>(scratch_all + (size_t)W * CM * RQ); // [n_items][CM][DYN_STATE] filter state
  int* ist = reinterpret_cast<int*>(fst + (size_t)d.n_items * CM * DYN_STATE);  // [n_items][4] integer state
  int* codes_ring = ist + (size_t)d.n_items * 4;                              // [W][n_items] codes of the last W quanta
  __shared__ double coef_all[W][2 * (DYN_STATE + 1)];                        // IIR coefficient block of the item at hand, per stage
  // The item descriptors once into LDS: read through the pointer in the kernel argument they were ~80 dependent
  // vector loads per quantum (uniform addresses, but not provably read-only: no scalar loads), each one an exposed L2
  // round trip — with one wave per instance that WAS the kernel's time (23 k cycles per quantum for ~1100 instructions).
  // per-item caches of what never changes from quantum to quantum: the params with ONE value per instance (ParamRef mode 0: a global
  // load per use, ~700 cycles each, three in a row in a panner item) and a Biquad's constant coefficient set (five doubles)
  int* meta_ring = codes_ring + (size_t)W * d.n_items;                                // [2][n_items] count | silent << 8 of the mixed inputs (W > 1 only)
  int* pmask_s = meta_ring + (size_t)MIXD * d.n_items;                                // [n_items] bit s: slot s is cached; bit 8: the coefficients
  float* pcs = reinterpret_cast<float*>(pmask_s + d.n_items);                         // [n_items][8]: op.p0 .. op.p4, alt1, alt2
  // [n_items][5], 8-byte aligned: (4 + W (+ W) + 1) n_items ints + 8 n_items floats in front, one pad word when that count is odd
  double* cfs = reinterpret_cast<double*>(pcs + (size_t)d.n_items * 8 + (((13 + W + MIXD) * d.n_items) & 1));
  DynItem* items_s = reinterpret_cast<DynItem*>(cfs + (size_t)d.n_items * 5);
  // (round 6) q_split > 1: the quanta of this ranged launch spread over q_split workgroups per instance (stateless items only)
  const uint32_t split = W == 1 && d.q_split > 1 ? d.q_split : 1u;
  const uint32_t inst = blockIdx.x / split;
  const uint32_t bq0 = d.q0, bq1 = d.q1 ? d.q1 : d.n_quanta;  // the launch's range (a block of a quantum-blocked loop, or the whole render)
  uint32_t sub_q0 = bq0, sub_q1 = bq1;
  if (split > 1) {
    const uint32_t per = (bq1 - bq0 + split - 1) / split;
    sub_q0 = bq0 + (blockIdx.x % split) * per;
    sub_q1 = sub_q0 + per < bq1 ? sub_q0 + per : bq1;
    if (sub_q0 >= sub_q1) return;  // (the whole workgroup: nothing has been synchronised yet)
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = W > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;   // this wavefront's stage
  float* scratch = scratch_all + (size_t)wv * CM * RQ;
  double* coef_s = coef_all[wv];
  auto all_sync = []() __attribute__((always_inline)) {
    if constexpr (W > 1)
      __syncthreads();
    else
      lds_sync();
  };
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.items);
    uint32_t* dst = reinterpret_cast<uint32_t*>(items_s);
    const int words = d.n_items * (int)(sizeof(DynItem) / 4);
    for (int i = tid; i < words; i += 64 * W) dst[i] = load_global(src + i);
  }
  all_sync();
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (FTZ/DAZ render scope, thread.rs:374-382)
  for (int i = tid; i < d.n_items * CM * DYN_STATE; i += 64 * W) fst[i] = 0.;
  for (int i = tid; i < W * d.n_items; i += 64 * W) codes_ring[i] = (int)(1u | CODE_SILENT);
  for (int i = tid; i < d.n_items; i += 64 * W) {
    const DynItem& li = items_s[i];
    // ist[0]: channels of the filter state (xy_len = 0, iir_filter.rs:303-306 "eagerly assume stereo" = 2) /
    //         delay writer: the line's channel count (ring of silent = mono quanta, delay.rs:386-397)
    ist[i * 4 + 0] = (li.kind == DI_NODE && li.dk == DK_IIR) ? 2 : (li.kind == DI_DELAY_W ? 1 : 0);
    ist[i * 4 + 1] = -1;  // delay writer: last quantum in which the line was (re-)mixed to mono
    ist[i * 4 + 2] = 0;   // DK_CONV_IN: compacted quantum slots used by channel 1
    ist[i * 4 + 3] = 0;
    int mask = 0;
    if (li.kind == DI_NODE) {
      const ParamRef* slots[7] = {&li.op.p0, &li.op.p1, &li.op.p2, &li.op.p3, &li.op.p4, &li.alt1, &li.alt2};
      const bool has_params = li.dk == DK_GAIN || li.dk == DK_STEREO_PAN || li.dk == DK_PANNER;
      for (int sl = 0; sl < 7; sl++)
        if (has_params && slots[sl]->mode == 0 && slots[sl]->base) {
          pcs[i * 8 + sl] = load_global(slots[sl]->base + inst);
          mask |= 1 << sl;
        }
      if (li.dk == DK_BIQUAD && li.op.i0 == 0 && li.op.ptr0) {
        const double* cf = reinterpret_cast<const double*>(li.op.ptr0) + (uint64_t)inst * li.op.u0;
        for (int j = 0; j < 5; j++) cfs[i * 5 + j] = load_global(cf + j);
        mask |= 1 << 8;
      }
    }
    pmask_s[i] = mask;
  }
  all_sync();
  // a launch over a range of quanta that continues an earlier one (a loop rendered block by block): the items' state from memory
  const uint32_t rq0 = sub_q0, rq1 = sub_q1;
  if (d.save_f && bq0 > 0) {
    const double* sf = d.save_f + (uint64_t)inst * (uint64_t)(d.n_items * CM * DYN_STATE);
    const int32_t* si = d.save_i + (uint64_t)inst * (uint64_t)(d.n_items * 4);
    for (int i = tid; i < d.n_items * CM * DYN_STATE; i += 64 * W) fst[i] = load_global(sf + i);
    for (int i = tid; i < d.n_items * 4; i += 64 * W) ist[i] = load_global(si + i);
    all_sync();
  }

#ifdef WAA_MEASURE
  unsigned long long cyc[8][3] = {};
#define DYN_STAMP(ph)                                                    \
  if (W == 1 && d.cycles) {                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime();         \
    if (it < 8) cyc[it][ph] += now - t_last;                             \
    t_last = now;                                                        \
  }
  unsigned long long t_last = d.cycles ? __builtin_amdgcn_s_memtime() : 0ull;
#else
#define DYN_STAMP(ph)
#endif
  // a delay reader sees its writer's stores of this quantum (same wavefront: they share a stage) once they have reached L2
  // (Writer and reader share a wavefront, its stores go to the XCD's L2 and the reader's loads are L2-served — coherent_f /
  // coherent_u, `sc1` — so all that is needed is that the stores have been acknowledged: workgroup scope = s_waitcnt vmcnt(0).
  // The agent-scope pair used until the end of round 5 added `buffer_wbl2 sc1` + `buffer_inv sc1` — an L2 write-back per
  // quantum and workgroup: 29 us per quantum on a 1024-context delay group, tools/conv_noise_probe.py, profiles/r05v_*.)
  auto vm_sync = []() __attribute__((always_inline)) {
    if constexpr (W > 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // (whether a workgroup-scope fence lowers to a wait for the outstanding stores is the compiler's memory-model
      // choice for the target, not a promise; the wait the hand-over NEEDS is stated)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  // stage_begin counts THIRDS of items: unit 3 i = item i's inputs gathered and mixed, 3 i + 1 = its node (result and code into the
  // LDS ring), 3 i + 2 = the result published to memory
  const int u0 = W > 1 ? d.stage_begin[wv] : 0, u1 = W > 1 ? d.stage_begin[wv + 1] : 3 * d.n_items;
  const int it0 = u0 / 3, it1 = (u1 + 2) / 3;
  for (uint32_t t = rq0; t < rq1 + (uint32_t)(W - 1); t++) {
    const uint32_t q = t - (uint32_t)wv;  // (wraps to a huge value / falls below the range while this stage has nothing to do yet)
    if (q >= rq0 && q < rq1) {
    const uint64_t f0 = (uint64_t)q * RQ;
    const int slot = W > 1 ? (int)(q % (uint32_t)W) : 0;
    float* cur = cur_ring + (size_t)slot * d.n_items * CM * RQ;
    int* codes = codes_ring + (size_t)slot * d.n_items;
    float* mixb = mix_ring + (size_t)(q & 1u) * d.n_items * CM * RQ;
    int* metab = meta_ring + (size_t)(q & 1u) * d.n_items;
    for (int it = it0; it < it1; it++) {
      const DynItem& li = items_s[it];
      // The item's scalar fields (the first 14 words of the descriptor) in ONE batch of LDS reads into scalar registers: read where
      // they are used, every decision below (kind -> n_in -> input item -> ...) was a dependent LDS read -> readfirstlane -> branch,
      // ~130 cycles each, eight to twelve in a row per phase (WAA_DYN_CYCLES: 1200-2000 cycles for a phase that computes nothing).
      const int2* hp = reinterpret_cast<const int2*>(&li);
      const int2 hw0 = hp[0], hw1 = hp[1], hw2 = hp[2], hw3 = hp[3], hw4 = hp[4], hw5 = hp[5], hw6 = hp[6];
      const int h_kind = __builtin_amdgcn_readfirstlane(hw0.x), h_dk = __builtin_amdgcn_readfirstlane(hw0.y);
      const int h_cc = __builtin_amdgcn_readfirstlane(hw1.x), h_mode = __builtin_amdgcn_readfirstlane(hw1.y);
      const int h_interp = __builtin_amdgcn_readfirstlane(hw2.x), h_n_in = __builtin_amdgcn_readfirstlane(hw2.y);
      const int h_nch_pub = __builtin_amdgcn_readfirstlane(hw3.x), h_publish_upmix = __builtin_amdgcn_readfirstlane(hw3.y);
      const int h_compact_ch1 = __builtin_amdgcn_readfirstlane(hw4.x), h_flags = __builtin_amdgcn_readfirstlane(hw4.y);
      const int h_writer_item = __builtin_amdgcn_readfirstlane(hw5.x), h_in_cycle = __builtin_amdgcn_readfirstlane(hw5.y);
      const int h_num_quanta = __builtin_amdgcn_readfirstlane(hw6.x);
      const int h_pmask = __builtin_amdgcn_readfirstlane(pmask_s[it]);
      auto pvc = [&](int slot, const ParamRef& p, uint64_t frame) __attribute__((always_inline)) {
        return (h_pmask >> slot) & 1 ? pcs[it * 8 + slot] : pval(p, inst, q, frame);
      };
      const bool do_front = W == 1 || (3 * it >= u0 && 3 * it < u1), do_node = W == 1 || (3 * it + 1 >= u0 && 3 * it + 1 < u1),
                 do_pub = W == 1 || (3 * it + 2 >= u0 && 3 * it + 2 < u1);  // (uniform; a stage's units are contiguous)
      float v[CM][2];
#pragma unroll
      for (int c = 0; c < CM; c++) v[c][0] = v[c][1] = 0.f;
      int sn = 1;       // number_of_channels of the mixed input
      bool ss = true;   // is_silent
      if (!do_front && !do_node) {
        // (only the publishing third of this item belongs to this stage: its result is recovered from the ring below)
      } else if (!do_front) {
        // the gather ran in an earlier stage: the mixed input of this quantum waits in the ring
        const int m = __builtin_amdgcn_readfirstlane(metab[it]);
        sn = m & 0xff;
        ss = (m >> 8) != 0;
#pragma unroll
        for (int c = 0; c < CM; c++) {
          v[c][0] = mixb[((size_t)it * CM + c) * RQ + lane];
          v[c][1] = mixb[((size_t)it * CM + c) * RQ + 64 + lane];
        }
      } else if (h_kind != DI_DELAY_R) {
        // ---- graph.rs:524-535: the input starts silent (mono); every incoming edge is `add`ed in order
        for (int k = 0; k < h_n_in; k++) {
          const DynInput& in = li.in[k];
          float u[CM][2];
#pragma unroll
          for (int c = 0; c < CM; c++) u[c][0] = u[c][1] = 0.f;
          uint32_t oc;
          if (in.item >= 0) {
            oc = (uint32_t)codes[in.item];
            const float* src = cur + (size_t)in.item * CM * RQ;
#pragma unroll
            for (int c = 0; c < CM; c++)
              if (c < (int)(oc & 63u)) {
                u[c][0] = src[c * RQ + lane];
                u[c][1] = src[c * RQ + 64 + lane];
              }
          } else {
            oc = in.code ? (uint32_t)load_global(in.code + (uint64_t)inst * in.code_stride + q) : (uint32_t)in.nch;
            const float* src = in.sig.base + (uint64_t)inst * in.sig.inst_stride;
            const int on = (int)(oc & 63u);
            if (!(oc & CODE_SILENT)) {
              u[0][0] = load_global(src + f0 + lane);
              u[0][1] = load_global(src + f0 + 64 + lane);
              if (on >= 2) {
                const uint64_t f1 = in.remap ? (uint64_t)load_global(in.remap + (uint64_t)inst * in.code_stride + q) * RQ : f0;
                u[1][0] = load_global(src + in.sig.ch_stride + f1 + lane);
                u[1][1] = load_global(src + in.sig.ch_stride + f1 + 64 + lane);
              }
#pragma unroll
              for (int c = 2; c < CM; c++)
                if (c < on) {
                  u[c][0] = load_global(src + (uint64_t)c * in.sig.ch_stride + f0 + lane);
                  u[c][1] = load_global(src + (uint64_t)c * in.sig.ch_stride + f0 + 64 + lane);
                }
            }
          }
          int on = (int)(oc & 63u);
          bool os = (oc & CODE_SILENT) != 0;
          if (on > CM) on = CM;  // (the planner picks the instantiation by the widest signal)
          // quantum.rs:532-569
          const int maxc = sn > on ? sn : on;
          int newc = h_mode == 0 ? maxc : (h_mode == 2 ? h_cc : (maxc < h_cc ? maxc : h_cc));
          if (newc > CM) newc = CM;
          if (newc < 1) newc = 1;
          mixn<CM>(v, sn, newc, h_interp, ss);
          mixn<CM>(u, on, newc, h_interp, os);
          if (ss) {  // quantum.rs:114-120: a silent channel takes the other operand (and its flag)
#pragma unroll
            for (int c = 0; c < CM; c++)
#pragma unroll
              for (int e = 0; e < 2; e++) v[c][e] = u[c][e];
            ss = os;
          } else if (!os) {
#pragma unroll
            for (int c = 0; c < CM; c++)
#pragma unroll
              for (int e = 0; e < 2; e++) v[c][e] = v[c][e] + u[c][e];
          }
          sn = newc;
        }
        if (ss) {  // silent data is the zero block
#pragma unroll
          for (int c = 0; c < CM; c++) v[c][0] = v[c][1] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < CM; c++)
          if (c >= sn) v[c][0] = v[c][1] = 0.f;
      }
      if (!do_node && !do_pub) {  // this stage ends behind the item's gather: hand the mixed input over
#pragma unroll
        for (int c = 0; c < CM; c++) {
          mixb[((size_t)it * CM + c) * RQ + lane] = v[c][0];
          mixb[((size_t)it * CM + c) * RQ + 64 + lane] = v[c][1];
        }
        if (lane == 0) metab[it] = sn | ((ss ? 1 : 0) << 8);
        lds_sync();
        continue;
      }
      int outn = sn;
      bool outs = ss;
      DYN_STAMP(0)
      if (!do_node) {
        // the node ran in an earlier stage: its result and code of this quantum wait in the ring
        const uint32_t oc = (uint32_t)__builtin_amdgcn_readfirstlane(codes[it]);
        outn = (int)(oc & 63u);
        outs = (oc & CODE_SILENT) != 0;
        const float* rs = cur + (size_t)it * CM * RQ;
#pragma unroll
        for (int c = 0; c < CM; c++) {
          v[c][0] = rs[c * RQ + lane];
          v[c][1] = rs[c * RQ + 64 + lane];
        }
      } else if (h_kind == DI_NODE) {
        const OpDesc& op = li.op;
        switch (h_dk) {
          case DK_GAIN: {  // gain.rs:143-199
            if (ss) {
              outn = 1;
              break;
            }
            const int pm = __builtin_amdgcn_readfirstlane(li.pmod_n);
            if (pm > 0) {
              // the gain param's inputs are members of this loop (round 6): their outputs of THIS quantum, out of the ring
              float m[2] = {0.f, 0.f};
              for (int j = 0; j < pm; j++) {
                const int it2 = __builtin_amdgcn_readfirstlane(li.pmod_item[j]);
                const uint32_t c2 = (uint32_t)__builtin_amdgcn_readfirstlane(codes[it2]);
                if (c2 & CODE_SILENT) continue;
                const float* rs = cur + (size_t)it2 * CM * RQ;
                m[0] = j == 0 ? rs[lane] : m[0] + rs[lane];
                m[1] = j == 0 ? rs[64 + lane] : m[1] + rs[64 + lane];
              }
#pragma unroll
              for (int e = 0; e < 2; e++) {
                float g = m[e] + pval(op.p0, inst, q, f0 + e * 64 + lane);
                g = g != g ? li.pmod_def : fminf(fmaxf(g, li.pmod_min), li.pmod_max);
#pragma unroll
                for (int c = 0; c < CM; c++) v[c][e] *= g;
              }
            } else if (op.p0.mode == 2) {
#pragma unroll
              for (int e = 0; e < 2; e++) {
                const float g = load_global(op.p0.base + (uint64_t)inst * op.p0.stride + f0 + e * 64 + lane);
#pragma unroll
                for (int c = 0; c < CM; c++) v[c][e] *= g;
              }
            } else {
              const float g = pvc(0, op.p0, 0);
              if (fabsf(g) <= 1e-6f) {  // :163-171 silent output
                outs = true;
                outn = 1;
#pragma unroll
                for (int c = 0; c < CM; c++) v[c][0] = v[c][1] = 0.f;
              } else if (!(fabsf(1.f - g) <= 1e-6f)) {
#pragma unroll
                for (int c = 0; c < CM; c++)
#pragma unroll
                  for (int e = 0; e < 2; e++) v[c][e] *= g;
              }
            }
            break;
          }
          case DK_BIQUAD:
          case DK_IIR: {
            // biquad_filter.rs:764-899 / iir_filter.rs:323-405: per-channel state, tail until the state is denormal
            const bool iir = h_dk == DK_IIR;
            const int ns = iir ? (op.i0 < 0 ? -op.i0 : op.i0) : 4;  // state doubles that decide the tail
            double* st = fst + (size_t)it * CM * DYN_STATE;
            int nst = ist[it * 4 + 0];
            if (ss) {
              bool normal = false;
              for (int j = lane; j < nst * DYN_STATE; j += 64) {
                const int c = j / DYN_STATE, jj = j % DYN_STATE;
                if (jj < (iir ? ns + 1 : 4)) normal |= __builtin_isnormal(st[c * DYN_STATE + jj]);
              }
              if (!__any(normal)) {
                outs = true;
                outn = 1;
                break;
              }
              outn = nst;
            } else {
              if (sn != nst) {
                lds_sync();
                for (int j = lane; j < CM * DYN_STATE; j += 64)
                  if (j / DYN_STATE >= nst && j / DYN_STATE < sn) st[j] = 0.;
                if (lane == 0) ist[it * 4 + 0] = sn;
                nst = sn;
              }
              outn = sn;
            }
            outs = false;
            lds_sync();
#pragma unroll
            for (int c = 0; c < CM; c++) {
              // (a silent input reads its single zero channel for every state channel, :858-862)
              scratch[c * RQ + lane] = ss ? 0.f : v[c][0];
              scratch[c * RQ + 64 + lane] = ss ? 0.f : v[c][1];
            }
            lds_sync();
            if (iir) {  // the coefficient block (shared by all instances) once per quantum into LDS: b then a, zero padded
              const double* cb = reinterpret_cast<const double*>(op.ptr0);
              for (int j = lane; j < 2 * (ns + 1); j += 64) coef_s[j] = load_global(cb + j);
              lds_sync();
            }
            // One coefficient set for the quantum (constant / k-rate params): the 128-frame recurrence spread over the
            // wavefront like the streaming kernel does it (waa_biquad_stream.hip) — 32 lanes per channel, 4 frames per lane:
            // zero-state response of the lane's frames, the lanes' true incoming states from a 5-step scan with the uniform
            // powers of A = M^4 (M the one-frame transition), then the reference's evaluation order from that state.
            // On two lanes the recurrence was a chain of ~640 dependent f64 operations per quantum (85 % of the kernel).
            // Denormals: flushed by hardware mode like in the reference's render scope; inf / NaN: the quantum is redone
            // serially below (the scan assumes linearity).
            bool scan_done = false;
            if (!iir && op.i0 != 2 && !d.no_scan) {
              const int l = lane & 31;
              const double* cf = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0 + (op.i0 == 1 ? (uint64_t)q * 5 : 0);
              const bool cfc = ((h_pmask >> 8) & 1) != 0;  // (constant set: out of the LDS cache)
              const double b0 = cfc ? cfs[it * 5] : load_global(cf), b1 = cfc ? cfs[it * 5 + 1] : load_global(cf + 1),
                           b2 = cfc ? cfs[it * 5 + 2] : load_global(cf + 2), a1 = cfc ? cfs[it * 5 + 3] : load_global(cf + 3),
                           a2 = cfc ? cfs[it * 5 + 4] : load_global(cf + 4);
              // two channels per pass (32 lanes each); wider layouts take CM / 2 passes, and nothing is written back before
              // every pass has found its quantum free of inf / NaN / near-flush values (the serial form below redoes ALL channels)
              float yo_all[CM / 2][4];
              double fin[CM / 2][4];
              bool bad_any = false;
              auto tiny = [](double v) { return v != 0. && __builtin_fabs(v) < 1e-280; };
#pragma unroll
              for (int pi = 0; pi < CM / 2; pi++) {
                if (pi * 2 >= outn) continue;  // (uniform)
                const int ch = pi * 2 + (lane >> 5);
                const bool act = ch < outn;
                const double* s = st + ch * DYN_STATE;
                const float* row = scratch + ch * RQ;
                const f4v xv = *reinterpret_cast<const f4v*>(row + 4 * l);
                const double x0 = (double)xv.x, x1 = (double)xv.y, x2 = (double)xv.z, x3 = (double)xv.w;
                double xm1 = __shfl_up(x3, 1, 32), xm2 = __shfl_up(x2, 1, 32);
                const double cx1 = s[0], cx2 = s[1], cy1 = s[2], cy2 = s[3];
                if (l == 0) {
                  xm1 = cx1;
                  xm2 = cx2;
                }
                // zero-state response of the lane's four frames (incoming y state 0, true x history)
                const double w0 = b0 * x0 + b1 * xm1 + b2 * xm2, w1 = b0 * x1 + b1 * x0 + b2 * xm1, w2 = b0 * x2 + b1 * x1 + b2 * x0,
                             w3 = b0 * x3 + b1 * x2 + b2 * x1;
                const double z0 = w0, z1 = w1 - a1 * z0, z2 = w2 - a1 * z1 - a2 * z0, z3 = w3 - a1 * z2 - a2 * z1;
                double r1 = z3, r2 = z2;
                M2d P = {-a1, -a2, 1., 0.};
                P = mm2(P, P);
                P = mm2(P, P);  // A = M^4
                if (l == 0) {
                  r1 = __builtin_fma(P.a, cy1, __builtin_fma(P.b, cy2, r1));
                  r2 = __builtin_fma(P.c, cy1, __builtin_fma(P.d, cy2, r2));
                }
#pragma unroll
                for (int dd = 1; dd < 32; dd <<= 1) {
                  const double q1 = __shfl_up(r1, dd, 32), q2 = __shfl_up(r2, dd, 32);
                  if (l >= dd) {
                    r1 = __builtin_fma(P.a, q1, __builtin_fma(P.b, q2, r1));
                    r2 = __builtin_fma(P.c, q1, __builtin_fma(P.d, q2, r2));
                  }
                  P = mm2(P, P);
                }
                double y1 = __shfl_up(r1, 1, 32), y2 = __shfl_up(r2, 1, 32);
                if (l == 0) {
                  y1 = cy1;
                  y2 = cy2;
                }
                // The scan is linear algebra; the reference flushes y to zero FRAME BY FRAME once it leaves the normal
                // range (:881-883).  While a tail decays through the last decades above 2.2e-308 the two differ in WHEN
                // the state reaches zero — by a quantum, and the quantum in which the tail ends is this item's silence
                // flag (fuzz seed 6278: a DelayNode behind it re-mixed its line one quantum early).  Any state or output
                // of this quantum that is non-zero but below 1e-280: the quantum is rendered serially.
                bool bad = tiny(y1) || tiny(y2);
                // the reference's evaluation order from the true incoming state (biquad_filter.rs:877-883)
                double p1 = xm1, p2 = xm2;
                const double xs[4] = {x0, x1, x2, x3};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  const double x = xs[e];
                  double y = b0 * x + b1 * p1 + b2 * p2 - a1 * y1 - a2 * y2;
                  bad |= !(__builtin_fabs(y) <= 1.7976931348623157e308) || tiny(y);  // inf / NaN, or close to the flush
                  if (!__builtin_isnormal(y)) y = 0.;
                  p2 = p1;
                  p1 = x;
                  y2 = y1;
                  y1 = y;
                  yo_all[pi][e] = (float)y;
                }
                fin[pi][0] = p1;
                fin[pi][1] = p2;
                fin[pi][2] = y1;
                fin[pi][3] = y2;
                bad_any |= act && bad;
              }
              if (!__any(bad_any)) {
                scan_done = true;
#pragma unroll
                for (int pi = 0; pi < CM / 2; pi++) {
                  if (pi * 2 >= outn) continue;
                  const int ch = pi * 2 + (lane >> 5);
                  if (ch < outn) {
                    *reinterpret_cast<f4v*>(scratch + ch * RQ + 4 * l) = f4v{yo_all[pi][0], yo_all[pi][1], yo_all[pi][2], yo_all[pi][3]};
                    if (l == 31) {
                      double* sw = st + ch * DYN_STATE;
                      sw[0] = fin[pi][0];
                      sw[1] = fin[pi][1];
                      sw[2] = fin[pi][2];
                      sw[3] = fin[pi][3];
                    }
                  }
                }
              }
            }
            if (lane < outn) {
              double* s = st + lane * DYN_STATE;
              float* row = scratch + lane * RQ;
              if (!iir && op.i0 == 2) {
                // per-frame coefficient sets (a-rate params)
                double x1 = s[0], x2 = s[1], y1 = s[2], y2 = s[3];
                const double* cbase = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0;
                for (int i = 0; i < RQ; i++) {
                  const double* cf = cbase + (f0 + i) * 5;
                  const double x = (double)row[i];
                  double y = load_global(cf) * x + load_global(cf + 1) * x1 + load_global(cf + 2) * x2 - load_global(cf + 3) * y1 -
                             load_global(cf + 4) * y2;
                  if (!__builtin_isnormal(y)) y = 0.;
                  x2 = x1;
                  x1 = x;
                  y2 = y1;
                  y1 = y;
                  row[i] = (float)y;
                }
                s[0] = x1;
                s[1] = x2;
                s[2] = y1;
                s[3] = y2;
              } else if (!iir && scan_done) {
                // (rendered by all 64 lanes above)
              } else if (!iir) {
                // one coefficient set for the quantum: in registers, eight frames per LDS round trip (the loop used to load
                // the five doubles from global memory in every frame: 128 exposed latencies per channel and quantum)
                double x1 = s[0], x2 = s[1], y1 = s[2], y2 = s[3];
                const double* cf = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0 + (op.i0 == 1 ? (uint64_t)q * 5 : 0);
                const double b0 = load_global(cf), b1 = load_global(cf + 1), b2 = load_global(cf + 2), a1 = load_global(cf + 3),
                             a2 = load_global(cf + 4);
                for (int i = 0; i < RQ; i += 8) {
                  const f4v p = *reinterpret_cast<const f4v*>(row + i), r = *reinterpret_cast<const f4v*>(row + i + 4);
                  const float xin[8] = {p.x, p.y, p.z, p.w, r.x, r.y, r.z, r.w};
                  float yo[8];
#pragma unroll
                  for (int e = 0; e < 8; e++) {
                    const double x = (double)xin[e];
                    double y = b0 * x + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
                    if (!__builtin_isnormal(y)) y = 0.;
                    x2 = x1;
                    x1 = x;
                    y2 = y1;
                    y1 = y;
                    yo[e] = (float)y;
                  }
                  *reinterpret_cast<f4v*>(row + i) = f4v{yo[0], yo[1], yo[2], yo[3]};
                  *reinterpret_cast<f4v*>(row + i + 4) = f4v{yo[4], yo[5], yo[6], yo[7]};
                }
                s[0] = x1;
                s[1] = x2;
                s[2] = y1;
                s[3] = y2;
              } else {
                const double* cb = coef_s;  // [2][ns + 1]: b then a, zero padded
                const double* ca = cb + (ns + 1);
                const double b0 = cb[0];
                for (int i = 0; i < RQ; i++) {
                  const double x = (double)row[i];
                  double y = __builtin_fma(b0, x, s[0]);
                  if (!__builtin_isnormal(y)) y = 0.;
                  for (int k = 1; k <= ns; k++) {
                    const double next = k < DYN_STATE ? s[k] : 0.;
                    s[k - 1] = cb[k] * x - ca[k] * y + next;
                  }
                  row[i] = (float)y;
                }
              }
            }
            lds_sync();
#pragma unroll
            for (int c = 0; c < CM; c++) {
              v[c][0] = c < outn ? scratch[c * RQ + lane] : 0.f;
              v[c][1] = c < outn ? scratch[c * RQ + 64 + lane] : 0.f;
            }
            break;
          }
          case DK_WAVESHAPER: {  // waveshaper.rs:383-487 (oversample none)
            if (ss && (h_flags & 1)) {
              outn = 1;
              break;
            }
            if (op.ptr0) {
              const float* curve = reinterpret_cast<const float*>(op.ptr0);
#pragma unroll
              for (int c = 0; c < CM; c++)
                if (c < sn) {
#pragma unroll
                  for (int e = 0; e < 2; e++) v[c][e] = shape(curve, op.i0, v[c][e]);
                }
              outs = false;
            }
            break;
          }
          case DK_STEREO_PAN: {  // stereo_panner.rs:218-317
            if (ss) {
              outn = 1;
              break;
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const uint64_t f = f0 + e * 64 + lane;
              const float pan = pvc(0, op.p0, f);
              float gl, gr;
              if (sn == 1) {
                if (op.p0.mode == 2) {
                  stereo_gains((pan + 1.f) * 0.5f, gl, gr);
                } else {
                  gl = pvc(5, li.alt1, 0);
                  gr = pvc(6, li.alt2, 0);
                }
                const float x = v[0][e];
                v[0][e] = x * gl;
                v[1][e] = x * gr;
              } else {
                if (op.p0.mode == 2) {
                  stereo_gains(pan <= 0.f ? pan + 1.f : pan, gl, gr);
                } else {
                  gl = pvc(1, op.p1, 0);
                  gr = pvc(2, op.p2, 0);
                }
                const float il = v[0][e], ir = v[1][e];
                if (pan <= 0.f) {
                  v[0][e] = __builtin_fmaf(ir, gl, il);
                  v[1][e] = ir * gr;
                } else {
                  v[0][e] = il * gl;
                  v[1][e] = __builtin_fmaf(il, gr, ir);
                }
              }
            }
            outn = 2;
            outs = false;
            break;
          }
          case DK_PANNER: {  // panner.rs:830-897, 988-1057 (equal power)
            if (ss) {
              outn = 1;
              break;
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const uint64_t f = f0 + e * 64 + lane;  // (per-frame tables with an audio-rate AudioListener)
              const float az = pvc(0, op.p0, f);
              const float dg = pvc(3, op.p3, f), cg = pvc(4, op.p4, f);
              if (sn == 1) {
                const float gl = pvc(5, li.alt1, f), gr = pvc(6, li.alt2, f);
                const float x = v[0][e];
                v[0][e] = x * (gl * dg * cg);
                v[1][e] = x * (gr * dg * cg);
              } else {
                const float gl = pvc(1, op.p1, f), gr = pvc(2, op.p2, f);
                const float il = v[0][e], ir = v[1][e];
                if (az <= 0.f) {
                  v[0][e] = (il + ir * gl) * dg * cg;
                  v[1][e] = ir * gr * dg * cg;
                } else {
                  v[0][e] = il * gl * dg * cg;
                  v[1][e] = (ir + il * gr) * dg * cg;
                }
              }
            }
            outn = 2;
            outs = false;
            break;
          }
          case DK_PASS:
            if (h_flags & 2) {  // ConvolverNode without a buffer (convolver.rs:357-374): no tail, then passthrough
              if (ss) outn = 1;
            }
            break;
          default: break;  // DK_CONV_IN: the mixed input as it is
        }
      } else if (h_kind == DI_DELAY_W) {
        // delay.rs:428-489: the ring is re-mixed to the count of the current input, then the input is stored
        if constexpr (CM > 2) {
          // layouts above stereo: the ring entries are REALLY re-mixed, like DelayWriter::check_ring_buffer_up_down_mix does
          // it (every stored quantum through `mix(new count, Speakers)`) — a chain of changes such as 4 -> 2 -> 4 has no
          // closed form per entry the way mono <-> stereo has (below).  Count changes are rare; the ring is <= 376 quanta.
          const int old = ist[it * 4 + 0];
          if (sn != old && q > 0) {
            vm_sync();  // (this wave's earlier stores to the line have reached L2)
            const uint32_t cap = (uint32_t)h_num_quanta + 1u;
            uint32_t* wc = li.aux32 + (uint64_t)inst * li.code_stride;
            float* hbw = li.out.base + (uint64_t)inst * li.out.inst_stride;
            for (uint32_t p = q > cap ? q - cap : 0u; p < q; p++) {
              const uint32_t pc = coherent_u(wc + p);
              const int np = (int)(pc & 63u);
              if (np == sn) continue;
              float w[CM][2];
#pragma unroll
              for (int c = 0; c < CM; c++) {
                w[c][0] = c < np ? coherent_f(hbw + (uint64_t)c * li.out.ch_stride + (uint64_t)p * RQ + lane) : 0.f;
                w[c][1] = c < np ? coherent_f(hbw + (uint64_t)c * li.out.ch_stride + (uint64_t)p * RQ + 64 + lane) : 0.f;
              }
              mix_regs<CM, 2>(w, np, sn, 0);
#pragma unroll
              for (int c = 0; c < CM; c++)
                if (c < sn && c < h_nch_pub) {
                  store_global(hbw + (uint64_t)c * li.out.ch_stride + (uint64_t)p * RQ + lane, w[c][0]);
                  store_global(hbw + (uint64_t)c * li.out.ch_stride + (uint64_t)p * RQ + 64 + lane, w[c][1]);
                }
              if (lane == 0) store_global(wc + p, (uint32_t)sn | (pc & CODE_SILENT));
            }
            vm_sync();
          }
        }
        if (lane == 0) {
          // (round 6) blocks of several quanta: the reader in the EARLIER launch of this block has already rendered the block's
          // later quanta with the ring count it found at the block's start.  The count it should have seen is this one, one
          // quantum late (delay.rs:515-530: ring[0].number_of_channels() right now) — a change anywhere but in the block's last
          // quantum invalidates the block: flagged, and the host renders the loop again one quantum per block (waa_abi.cpp).
          if (li.xstate && sn != ist[it * 4 + 0] && q + 1 < bq1 && bq1 - bq0 > 1) store_global(li.xstate + (uint64_t)d.n_inst * 2, 1);
          ist[it * 4 + 0] = sn;
          if (sn == 1) ist[it * 4 + 1] = (int)q;
          // (a split launch: the state the next block's reader finds is the one behind the block's LAST quantum; the workgroups of
          // the earlier quanta keep out of it — they run concurrently)
          if (li.xstate && (split == 1 || q + 1 == bq1)) {  // a reader in another launch follows the ring's state (round 5)
            store_global(li.xstate + (uint64_t)inst * 2, sn - 1);
            store_global(li.xstate + (uint64_t)inst * 2 + 1, (sn == 1 ? (int)q : ist[it * 4 + 1]) + 1);
          }
        }
      } else {
        // ---- DelayReader::process, delay.rs:515-745, on the writer's line in absolute time
        vm_sync();  // the writer's stores of this quantum (if it rendered first) have reached L2
        // (the writer: an item of this launch, or — a loop cut at a frozen-state node — of another one, seen through memory)
        const bool xw = h_writer_item < 0;
        const DynItem& wi = items_s[xw ? it : h_writer_item];
        const SignalRef& hs = xw ? li.xline : wi.out;
        const int nch = xw ? (int)coherent_u(reinterpret_cast<const uint32_t*>(li.xstate + (uint64_t)inst * 2)) + 1
                           : ist[h_writer_item * 4 + 0];         // ring[0].number_of_channels() right now
        const int last_mono = xw ? (int)coherent_u(reinterpret_cast<const uint32_t*>(li.xstate + (uint64_t)inst * 2 + 1)) - 1
                                 : ist[h_writer_item * 4 + 1];   // entries written before it were collapsed to mono
        const uint32_t* wcode = xw ? li.xaux32 + (uint64_t)inst * li.code_stride : wi.aux32 + (uint64_t)inst * wi.code_stride;
        const OpDesc& op = li.op;
        int64_t pf0 = 0;
        float k0 = 0.f;
        if (op.p0.mode != 2) {
          double dv = (double)pval(op.p0, inst, q, 0);
          if (h_in_cycle) dv = fmax(dv, d.quantum_duration);
          const double position = 0. - dv * d.sample_rate;
          const double fl = floor(position);
          pf0 = (int64_t)fl;
          k0 = (float)(position - fl);
        }
        bool active = false;
        const float* hb = hs.base + (uint64_t)inst * hs.inst_stride;
        // one sample of the line as the reader sees it NOW: entries written with two channels were averaged when the
        // line was re-mixed to mono after they were written, and duplicated again if it went back to stereo
        auto sample = [&](int c, int64_t idx) -> float {
          if (idx < 0) return 0.f;
          const int p = (int)(idx >> 7);
          const uint32_t pc = coherent_u(wcode + p);
          if (pc & CODE_SILENT) return 0.f;
          const int np = (int)(pc & 63u);
          if constexpr (CM > 2)  // (the writer re-mixed the ring in place: every entry carries the ring's count)
            return c < np ? coherent_f(hb + (uint64_t)c * hs.ch_stride + idx) : 0.f;
          if (np <= 1) return coherent_f(hb + idx);
          if (last_mono > p) return 0.5f * (coherent_f(hb + idx) + coherent_f(hb + hs.ch_stride + idx));
          return coherent_f(hb + (uint64_t)c * hs.ch_stride + idx);
        };
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int i = e * 64 + lane;
          int64_t pf;
          float k;
          if (op.p0.mode != 2) {
            pf = pf0 + i;
            k = k0;
          } else {
            double dv = (double)load_global(op.p0.base + (uint64_t)inst * op.p0.stride + f0 + i);
            if (h_in_cycle) dv = fmax(dv, d.quantum_duration);
            const double position = (double)i - dv * d.sample_rate;
            const double fl = floor(position);
            pf = (int64_t)fl;
            k = (float)(position - fl);
          }
          const int64_t prev = (int64_t)f0 + pf;
          // frame 128 of the newest block wraps to the OLDEST ring block (delay.rs:622-626); a reader that renders
          // before its writer sees, in the slot of the current quantum, the block written ring-capacity quanta ago
          int64_t next = prev + 1;
          if (!h_in_cycle && pf == RQ - 1) next = ((int64_t)q - h_num_quanta) * RQ;
          if (h_in_cycle && next >= (int64_t)f0) next -= ((int64_t)h_num_quanta + 1) * RQ;
#pragma unroll
          for (int c = 0; c < CM; c++)
            if (c < nch) {
              const float ps = sample(c, prev), nsv = sample(c, next);
              const float val = __builtin_fmaf(1.f - k, ps, k * nsv);
              active |= __builtin_isnormal(val);
              v[c][e] = val;
            }
        }
        if (__any(active)) {
          outn = nch;
          outs = false;
        } else {  // delay.rs:660-668: nothing but zeros / denormals came out: the output is silent
          outn = 1;
          outs = true;
#pragma unroll
          for (int c = 0; c < CM; c++) v[c][0] = v[c][1] = 0.f;
        }
      }
      if (outs) {
#pragma unroll
        for (int c = 0; c < CM; c++) v[c][0] = v[c][1] = 0.f;
      }
      DYN_STAMP(1)
      // ---- hand over (LDS) and publish (HBM)
      const uint32_t code = (uint32_t)outn | (outs ? CODE_SILENT : 0u);
      if (do_node) {
        float* dst = cur + (size_t)it * CM * RQ;
#pragma unroll
        for (int c = 0; c < CM; c++) {
          dst[c * RQ + lane] = c < outn ? v[c][0] : 0.f;
          dst[c * RQ + 64 + lane] = c < outn ? v[c][1] : 0.f;
        }
        if (lane == 0) codes[it] = (int)code;
      }
      if (!do_pub) {  // (published by a later stage, out of the ring)
        lds_sync();
        continue;
      }
      if (li.out.base) {
        float* gout = li.out.base + (uint64_t)inst * li.out.inst_stride;
        uint64_t f1 = f0;
        if (h_compact_ch1) {  // mono impulse response: convolver 1 only runs (= its time only advances) on stereo quanta
          const int slot = ist[it * 4 + 2];
          f1 = (uint64_t)slot * RQ;
          if (lane == 0) {
            store_global(li.aux32 + (uint64_t)inst * li.code_stride + q, (uint32_t)slot);
            if (outn >= 2) ist[it * 4 + 2] = slot + 1;
          }
        }
#pragma unroll
        for (int c = 0; c < CM; c++)
          if (c < h_nch_pub) {
            const bool have = c < outn;
            const bool dup = !have && h_publish_upmix && !outs;
            if (c == 1 && h_compact_ch1 && !have) continue;  // nothing enters convolver 1 in this quantum
            const uint64_t fc = c == 1 ? f1 : f0;
            store_global(gout + (uint64_t)c * li.out.ch_stride + fc + lane, have ? v[c][0] : (dup ? v[0][0] : 0.f));
            store_global(gout + (uint64_t)c * li.out.ch_stride + fc + 64 + lane, have ? v[c][1] : (dup ? v[0][1] : 0.f));
          }
        if (lane == 0) {
          if (li.out_code) li.out_code[(uint64_t)inst * li.code_stride + q] = (uint8_t)code;
          if (h_kind == DI_DELAY_W) store_global(li.aux32 + (uint64_t)inst * li.code_stride + q, code);
        }
      }
      lds_sync();
      DYN_STAMP(2)
    }
    }
    if constexpr (W > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the step's hand-over (LDS only)
  }
  if (d.save_f && (split == 1 || rq1 == bq1)) {  // ... and back, for the launch that renders the next block (split: the last quanta's workgroup)
    all_sync();
    double* sf = d.save_f + (uint64_t)inst * (uint64_t)(d.n_items * CM * DYN_STATE);
    int32_t* si = d.save_i + (uint64_t)inst * (uint64_t)(d.n_items * 4);
    for (int i = tid; i < d.n_items * CM * DYN_STATE; i += 64 * W) store_global(sf + i, fst[i]);
    for (int i = tid; i < d.n_items * 4; i += 64 * W) store_global(si + i, ist[i]);
  }
#ifdef WAA_MEASURE
  if (W == 1 && d.cycles && inst == 0 && lane == 0)
    for (int i = 0; i < 8; i++)
      for (int p = 0; p < 3; p++) d.cycles[i * 3 + p] = cyc[i][p];
#endif
}

template <int CM, int W>
__global__ __launch_bounds__(64 * W) void dyn_kernel(const DynDesc d) {
  dyn_body<CM, W>(d);
}
void launch_dyn(const DynDesc& d, void* stream) {
  const int cm = dyn_planes(d.cmax);
  // the pipelined form: mono / stereo groups the planner cut into stages (WAA_DYN_NO_PIPE=1: the one-wavefront form, A/B and cross-check)
  const bool split = d.split_ok && cm == 2 && !measure_switch("WAA_DYN_NO_SPLIT") && !measure_switch("WAA_DYN_CYCLES");
  const int stages = cm == 2 && d.n_stages > 1 && !split && !measure_switch("WAA_DYN_NO_PIPE") && !measure_switch("WAA_DYN_CYCLES") ? d.n_stages : 1;
  const size_t lds = dyn_lds_bytes(d.n_items, d.cmax, stages);
  DynDesc dd = d;
  dd.no_scan = measure_switch("WAA_DYN_NO_SCAN") ? 1u : 0u;
  dd.cycles = nullptr;
#ifdef WAA_MEASURE
  static unsigned long long* d_cycles = nullptr;
  if (measure_switch("WAA_DYN_CYCLES")) {
    if (!d_cycles) (void)hipMalloc(&d_cycles, 24 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_cycles, 0, 24 * sizeof(unsigned long long), (hipStream_t)stream);
    dd.cycles = d_cycles;
  }
  struct Report {
    unsigned long long* p;
    hipStream_t st;
    uint32_t nq;
    int n_items;
    ~Report() {
      if (!p) return;
      unsigned long long h[24];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, p, sizeof h, hipMemcpyDeviceToHost);
      for (int i = 0; i < n_items && i < 8; i++)
        fprintf(stderr, "[dyn cycles] item %d: gather %.0f  node %.0f  hand-over %.0f  (shader-clock ticks per quantum, instance 0)\n", i,
                (double)h[i * 3] / nq, (double)h[i * 3 + 1] / nq, (double)h[i * 3 + 2] / nq);
    }
  } report{dd.cycles, (hipStream_t)stream, d.n_quanta, d.n_items};
#endif
  dd.q_split = 1;
  if (split) {
    // (at least six quanta per workgroup: each one loads the descriptors and the state; ~16 k workgroups fill the device)
    const uint32_t nq = (dd.q1 ? dd.q1 : dd.n_quanta) - dd.q0;
    const uint32_t want = (uint32_t)std::max<uint64_t>(1, 16384 / std::max<uint32_t>(d.n_inst, 1));
    dd.q_split = std::max<uint32_t>(1, std::min<uint32_t>(std::max<uint32_t>(want, 16), nq / 6));
  }
  auto go = [&](auto kernel, int w) {
    if (lds > 64 * 1024) raise_lds_limit(reinterpret_cast<const void*>(kernel));
    hipLaunchKernelGGL(kernel, dim3(d.n_inst * (w == 1 ? dd.q_split : 1u)), dim3(64 * w), lds, (hipStream_t)stream, dd);
  };
  if (cm == 32) {
    go(dyn_kernel<32, 1>, 1);  // (round 6) signals of 7 ... 32 channels: every mix is discrete there, the count rules are the same
  } else if (cm == 16) {
    go(dyn_kernel<16, 1>, 1);
  } else if (cm == 8) {
    go(dyn_kernel<8, 1>, 1);
  } else if (cm != 2) {
    go(dyn_kernel<6, 1>, 1);
  } else {
    switch (stages) {
      case 8: go(dyn_kernel<2, 8>, 8); break;
      case 7: go(dyn_kernel<2, 7>, 7); break;
      case 6: go(dyn_kernel<2, 6>, 6); break;
      case 5: go(dyn_kernel<2, 5>, 5); break;
      case 4: go(dyn_kernel<2, 4>, 4); break;
      case 3: go(dyn_kernel<2, 3>, 3); break;
      case 2: go(dyn_kernel<2, 2>, 2); break;
      default: go(dyn_kernel<2, 1>, 1); break;
    }
  }
}
size_t dyn_lds_bytes(int n_items, int cmax, int stages) {
  const int cm = dyn_planes(cmax);
  const int w = cm == 2 && stages > 1 ? stages : 1;
  // signals (a ring of w quanta; w > 1: a second ring, the mixed inputs) + scratch (per stage), filter state, ist (4) + codes (w)
  // (+ meta (w)) + pmask (1) ints (+ the pad word in front of the doubles), the param cache (8 floats), the coefficient cache
  // (5 doubles), the item descriptors
  const size_t mixd = w > 1 ? 2 : 0;  // (the mixed inputs: two slots, neighbours hand over)
  return (((size_t)w + mixd) * n_items * cm * RQ + (size_t)w * cm * RQ) * sizeof(float) + (size_t)n_items * cm * DYN_STATE * sizeof(double) +
         (size_t)(n_items * (5 + w + mixd) + 2) * sizeof(int) + (size_t)n_items * 8 * sizeof(float) + (size_t)n_items * 5 * sizeof(double) +
         (size_t)n_items * sizeof(DynItem);
}

// ConvolverRenderer::process on codes (convolver.rs:343-392): the tail counter cuts the output off once a silent input
// has lasted for the length of the impulse response; the output count follows the routing table (:384-466).
struct ConvCodeState {
  ConvNoiseNode node;  // the tail counter and the node's FFTConvolvers (waa_conv_noise.hpp)
  bool ever_active;    // (an input that has never been active: the convolver's output is exact zeros, see ConvCodeDesc)
};
// quanta [qa, qb) of one instance.  On entry clean[q] holds the non-zero flags of the input quantum (conv_nz_kernel; noise form).
__device__ inline void conv_code_quanta(const ConvCodeDesc& d, uint32_t inst, uint32_t qa, uint32_t qb, ConvCodeState& s) {
  const uint8_t* in = d.in_code + (uint64_t)inst * d.code_stride;
  uint8_t* out = d.out_code + (uint64_t)inst * d.code_stride;
  uint8_t* clean = d.clean + (uint64_t)inst * d.code_stride;
  const uint32_t chmask = d.cout >= 2 ? 3u : 1u;  // (channels of `out` in absolute time)
  for (uint32_t q = qa; q < qb; q++) {
    const uint32_t c = in[q];
    const uint32_t nz = d.noise ? clean[q] : 0u;
    const bool silent = (c & CODE_SILENT) != 0;
    const uint32_t r = conv_noise_node_step(d.nir, d.ir_nch, d.impulse_length, s.node, silent, (int)(c & 63u), nz);
    clean[q] = 0;
    if (r == CONV_NOISE_CUT) {
      out[q] = (uint8_t)(1u | CODE_SILENT);
      continue;
    }
    const uint32_t outn = r >> 4, noisy = r & 3u;
    out[q] = (uint8_t)outn;
    if (d.noise) {
      const uint32_t live = (outn == 2 ? 3u : 1u) & chmask;
      clean[q] = (uint8_t)((~noisy & live) | ((noisy & live) << 2));
    } else if (silent) {
      clean[q] = s.ever_active ? 0 : 1;
    }
    if (!silent) s.ever_active = true;
  }
}
__global__ void conv_code_kernel(const ConvCodeDesc d) {
  const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= d.n_inst) return;
  ConvCodeState s;
  conv_noise_node_reset(s.node);
  s.ever_active = false;
  conv_code_quanta(d, inst, 0, d.n_quanta, s);
}
// the same automaton over a range of quanta, its state in memory between launches
__global__ void conv_code_range_kernel(const ConvCodeDesc d) {
  const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= d.n_inst) return;
  int32_t* stw = d.state + (uint64_t)inst * CONV_CODE_STATE_INTS;
  ConvCodeState s;
  s.node.tail = (uint64_t)(uint32_t)stw[0] | ((uint64_t)(uint32_t)stw[1] << 32);
  s.ever_active = stw[2] != 0;
  for (int k = 0; k < 4; k++) {
    if (stw[3]) {
      s.node.cv[k].hist = (uint64_t)(uint32_t)stw[4 + 4 * k] | ((uint64_t)(uint32_t)stw[5 + 4 * k] << 32);
      s.node.cv[k].age = (uint32_t)stw[6 + 4 * k];
      s.node.cv[k].flags = (uint32_t)stw[7 + 4 * k];
    } else {
      conv_noise_reset(s.node.cv[k]);  // (the state arrives zero-filled)
    }
  }
  conv_code_quanta(d, inst, d.q0, d.q1, s);
  stw[0] = (int32_t)(uint32_t)s.node.tail;
  stw[1] = (int32_t)(uint32_t)(s.node.tail >> 32);
  stw[2] = s.ever_active ? 1 : 0;
  stw[3] = 1;
  for (int k = 0; k < 4; k++) {
    stw[4 + 4 * k] = (int32_t)(uint32_t)s.node.cv[k].hist;
    stw[5 + 4 * k] = (int32_t)(uint32_t)(s.node.cv[k].hist >> 32);
    stw[6 + 4 * k] = (int32_t)s.node.cv[k].age;
    stw[7 + 4 * k] = (int32_t)s.node.cv[k].flags;
  }
}
// does channel c of input quantum q hold a non-zero sample?  (bit c of clean[q], read back by the code kernel; a quantum coded
// silent is zeros in the reference whatever the buffer holds)
// (one workgroup per (quantum, instance) pair, walked with a grid stride: neither count is bounded by a grid dimension)
__global__ __launch_bounds__(128) void conv_nz_kernel(const ConvCodeDesc d) {
  const uint64_t nq = d.q1 - d.q0, total = nq * d.n_inst;
  for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {
    const uint32_t q = d.q0 + (uint32_t)(w % nq), inst = (uint32_t)(w / nq);
    const uint32_t code = d.in_code[(uint64_t)inst * d.code_stride + q];
    uint32_t bits = 0;
    if (!(code & CODE_SILENT)) {
      const int ic = (int)(code & 63u);
      for (int c = 0; c < ic && c < d.in_test_nch; c++) {
        const float v = d.in.base[(uint64_t)inst * d.in.inst_stride + (uint64_t)c * d.in.ch_stride + (uint64_t)q * RQ + threadIdx.x];
        if (__syncthreads_or(v != 0.f)) bits |= 1u << c;
      }
    }
    if (threadIdx.x == 0) d.clean[(uint64_t)inst * d.code_stride + q] = (uint8_t)bits;
  }
}
__global__ __launch_bounds__(128) void conv_floor_kernel(const ConvCodeDesc d) {
  const uint64_t nq = d.q1 - d.q0, total = nq * d.n_inst;
  for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {
    const uint32_t q = d.q0 + (uint32_t)(w % nq), inst = (uint32_t)(w / nq);
    const uint32_t m = d.clean[(uint64_t)inst * d.code_stride + q];
    if (!m) continue;
    if (!d.noise) {  // the round-3 form: the whole quantum is exact zeros
      for (int c = 0; c < d.cout; c++)
        d.out.base[(uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride + (uint64_t)q * RQ + threadIdx.x] = 0.f;
      continue;
    }
    for (int c = 0; c < 2 && c < d.cout; c++) {
      float* p = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)c * d.out.ch_stride + (uint64_t)q * RQ + threadIdx.x;
      if (m & (1u << c)) {
        *p = 0.f;
      } else if (m & (4u << c)) {
        const float v = *p;
        if (__builtin_fabsf(v) < CONV_NOISE_FLOOR) *p = __builtin_copysignf(CONV_NOISE_FLOOR, v);
      }
    }
  }
}
void launch_conv_codes(const ConvCodeDesc& d0, void* stream) {
  ConvCodeDesc d = d0;
  const bool ranged = d.state != nullptr;
  if (!ranged) {
    d.q0 = 0;
    d.q1 = d.n_quanta;
  }
  if (d.q1 <= d.q0 || d.n_inst == 0) return;
  const uint64_t pairs = (uint64_t)(d.q1 - d.q0) * d.n_inst;
  const dim3 grid((uint32_t)std::min<uint64_t>(pairs, 1u << 22));
  if (d.noise) hipLaunchKernelGGL(conv_nz_kernel, grid, dim3(128), 0, (hipStream_t)stream, d);
  if (ranged)
    hipLaunchKernelGGL(conv_code_range_kernel, dim3((d.n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL(conv_code_kernel, dim3((d.n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
  hipLaunchKernelGGL(conv_floor_kernel, grid, dim3(128), 0, (hipStream_t)stream, d);
}

}  // namespace waa
