// waa_hrtf_fft.hip — the HRTF panner for directions that do not change during the render (one for the batch or one per context), as uniform partitioned overlap-add on
// 256-point transforms in registers (waa_hrtf_fft.hpp has the algebra; the oversampled WaveShaper's kernel, waa_osfft.hip, the
// machinery).  Replaces hrtf8_kernel's direct form (415 taps x 128 frames x 2 ears of fused multiply-adds per quantum: 8.5-9 ms for
// 1024 contexts x 10 s at 0.6 of the packed-f32 peak) for batches in which PannerNode and AudioListener are at rest; moving
// sources keep the direct kernels (their HRIR pair changes per quantum, panner.rs:781-829).
//
// Work decomposition as in waa_osfft.hip: a GROUP of 16 lanes renders a run of `seg_len` render quanta of one instance, quantum
// after quantum, with the node's state — the spectra of the last three processed quanta and the overlap-add carry — in registers;
// the state a run starts with is recomputed from the FOUR processed quanta in front of it (found through the node's `prev` table,
// which is also how skipped quanta and the frozen history of panner.rs:697-711 are followed).  Four groups per wavefront, four
// wavefronts per workgroup share the spectral tables in LDS.
#include <hip/hip_runtime.h>

#include "waa_hrtf_fft.hpp"
#include "waa_internal.hpp"

namespace waa {
namespace {
using namespace hrtffft;
constexpr int WAVES = 4;

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// EXACTZ (dynamic plans in which a node behind the panner may DECIDE on exact zeros — a DelayNode that read nothing normal, a Biquad whose
// state is no longer normal): the direct form's sum is exactly zero where no non-zero input frame lies within the response's reach
// (every product is +-0), the transforms leave 1e-10 of roundoff there.  The kernel follows the position of the last non-zero input frame
// through the processed quanta (per row of 16 frames: a ballot of the group's lanes, the highest set bit at or below the lane's frame)
// and puts out 0 where that frame is further back than the ear's last non-zero tap (jmax).  The quanta in front of a run are 4 x 128 =
// 512 frames >= taps: enough history.  (Leading zero taps are not modelled: the few frames right at an onset keep their roundoff — the
// quantum is not silent there anyway.)
template <bool EXACTZ>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void hrtf_fft_kernel(const HrtfDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  // LDS map (8-byte slots): PARTS tables | WAVES * 4 exchange buffers
  const ldsp tab = (ldsp)lds_raw;
  const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, t = lane & 15;
  const ldsp ex = tab + PARTS * TAB_SLOTS + (wv * 4 + g) * XSLOTS;
  // One direction for the whole batch (rows == 1): any group renders any (instance, run).  One direction PER instance (rows == n_inst):
  // the sixteen groups of a workgroup render sixteen runs of ONE instance and share its table — the runs of an instance are padded to
  // a multiple of sixteen (n_seg_pad, idle groups at the end of an instance's last workgroup).
  const uint64_t gid = ((uint64_t)blockIdx.x * WAVES + wv) * 4 + g;
  const uint32_t inst = (uint32_t)(gid / d.n_seg_pad), seg = (uint32_t)(gid % d.n_seg_pad);
  const bool alive = inst < d.n_inst && seg < d.n_seg;
  const uint32_t row = d.rows > 1 && inst < d.n_inst ? inst : 0u;  // (uniform over the workgroup when rows > 1)
  {
    const f4v* src = reinterpret_cast<const f4v*>(d.fft_tables + (uint64_t)row * (PARTS * TAB_SLOTS * 2));
    __attribute__((address_space(3))) f4v* dst = (__attribute__((address_space(3))) f4v*)tab;
    for (int i = threadIdx.x; i < PARTS * TAB_SLOTS / 2; i += WAVES * 64) dst[i] = load_global_f4(reinterpret_cast<const float*>(src + i));
  }
  __syncthreads();
  const int32_t* prev = d.prev + (uint64_t)(alive ? inst : 0) * d.prev_stride;
  const uint8_t* code = d.in_code + (uint64_t)(alive ? inst : 0) * d.code_stride;
  const int q_lo = (int)(d.q0 + seg * d.seg_len);
  const int q_hi = (int)((uint64_t)q_lo + d.seg_len < d.q1 ? q_lo + d.seg_len : d.q1);
  // the four processed quanta in front of the run, ph[3] the nearest (-1: none)
  int ph[HEADS];
#pragma unroll
  for (int k = 0; k < HEADS; k++) ph[k] = -1;
  if (q_lo > 0) {
    int base = q_lo - 1, p1 = -1;
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
    int p = p1;
#pragma unroll
    for (int k = HEADS - 1; k >= 0; k--) {
      ph[k] = p;
      if (p >= 0) {
        const int32_t l = load_global(prev + p);
        p = l >= 0 ? l : -1;
      }
    }
  }
  HLane L;
  load_tw(reinterpret_cast<const c2v*>(d.tw256), t, L.tws);
  lane_reset_if(L, true);
  const float* src = d.in.base + (uint64_t)(alive ? inst : 0) * d.in.inst_stride;
  float* dst = d.out.base + (uint64_t)(alive ? inst : 0) * d.out.inst_stride;
  const float gain = load_global(&d.table[row].gain);  // (per_row == 1: one geometry row per instance, or one for the batch)
  int jmax_l = 0, jmax_r = 0, lastnz = -8192;  // EXACTZ: last non-zero tap per ear; last non-zero input frame relative to the quantum's start
  if constexpr (EXACTZ) {
    jmax_l = load_global(d.jmax + 2 * row);
    jmax_r = load_global(d.jmax + 2 * row + 1);
  }
  const int n_it = (int)d.seg_len + HEADS;
  auto quantum_of = [&](int it) { return it < HEADS ? (it == 0 ? ph[0] : it == 1 ? ph[1] : it == 2 ? ph[2] : ph[3]) : q_lo + it - HEADS; };
  auto valid_at = [&](int it, int q) { return alive && it < n_it && q >= 0 && (it < HEADS || q < q_hi); };
  const float* safe = d.tw256;  // (any 128 readable floats: what a lane that does not process loads and throws away)
  float xn0[8], xn1[8];         // the next step's input frames (channel 0 / channel 1), requested one step ahead
  int zpos[8];                  // EXACTZ: per frame t + 16 j of the quantum at hand, the last non-zero input frame at or in front of it
  int32_t link_n;
  uint32_t code_n;
  auto request = [&](int q, bool procn, uint32_t c) __attribute__((always_inline)) {
    const bool live = procn && !(c & CODE_SILENT);
    const float* p0 = live ? src + (uint64_t)q * RQ + t : safe + t;
    const float* p1c = live && (c & 63u) >= 2 ? p0 + d.in.ch_stride : safe + t;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      xn0[j] = load_global(p0 + 16 * j);
      xn1[j] = load_global(p1c + 16 * j);
    }
  };
  {
    const int q0 = quantum_of(0);
    const bool v = valid_at(0, q0);
    link_n = v ? load_global(prev + q0) : LINK_SKIP;
    code_n = v ? (uint32_t)load_global(code + q0) : (uint32_t)CODE_SILENT;
    request(q0, link_n != LINK_SKIP, code_n);
  }
#pragma unroll 1
  for (int it = 0; it < n_it; it++) {
    const int q = quantum_of(it);
    const int32_t link = link_n;
    const uint32_t c = code_n;
    const bool proc = link != LINK_SKIP;  // (LINK_SKIP also stands for "no quantum in this step")
    const bool store = alive && it >= HEADS && q < q_hi;
    const int qn = quantum_of(it + 1);
    {
      const bool v = valid_at(it + 1, qn);
      link_n = v ? load_global(prev + qn) : LINK_SKIP;
      code_n = v ? (uint32_t)load_global(code + qn) : (uint32_t)CODE_SILENT;
    }
    lane_reset_if(L, proc && link == LINK_FRESH);
    if constexpr (EXACTZ) lastnz = proc && link == LINK_FRESH ? -8192 : lastnz;
    int tofs = 0;
    asm volatile("" : "+s"(tofs));
    const ldsp tabq = tab + tofs;  // (loop-invariant LDS reads: the address is opaque so that they are not hoisted out of the loop)
    {
      // the quantum's mono mix (panner.rs:800-810; quantum.rs:387-397): a silent input is zeros, a stereo one 0.5 (L + R)
      const bool live = proc && !(c & CODE_SILENT), st2 = (c & 63u) >= 2;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float m = st2 ? 0.5f * (xn0[j] + xn1[j]) : xn0[j];
        x[j] = live ? m : 0.f;
      }
      if constexpr (EXACTZ) {
        int run_last = lastnz;  // last non-zero frame in front of row j (relative to this quantum's first frame; < 0: an earlier quantum)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint64_t bal = __builtin_amdgcn_ballot_w64(x[j] != 0.f);
          const uint32_t rm = (uint32_t)(bal >> (g * 16)) & 0xffffu;   // the group's sixteen frames 16 j + t
          const uint32_t below = rm & ((2u << t) - 1u);                 // ... at or below this lane's
          zpos[j] = below ? 16 * j + (31 - __builtin_clz(below)) : run_last;
          run_last = rm ? 16 * j + (31 - __builtin_clz(rm)) : run_last;
        }
        const int nxt = run_last - 128 < -8192 ? -8192 : run_last - 128;
        lastnz = proc ? nxt : lastnz;
      }
      ph_in(L, x);
    }
    xwrite(L.a, ex, t);
    wsync();
    xread(L.a, ex, t);
    wsync();
    ph_spec(L, tabq, t, proc);
    request(qn, link_n != LINK_SKIP, code_n);
    xwrite(L.Y, ex, t);
    wsync();
    xread(L.Y, ex, t);
    wsync();
    c2v o[8];
    ph_out(L, proc, o);
    {
      // acc * gain * corr in hrtf8_kernel's order; corr = 2 behind a stereo input (by the quantum's count, silent or not)
      const float corr = (c & 63u) >= 2 ? 2.f : 1.f;
      float* p0 = store ? dst + (uint64_t)q * RQ + t : d.trash + lane;
      float* p1c = store ? p0 + d.out.ch_stride : d.trash + lane;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float ol = proc ? o[j].x * gain * corr : 0.f, orr = proc ? o[j].y * gain * corr : 0.f;
        if constexpr (EXACTZ) {
          const int back = 16 * j + t - zpos[j];  // frames between this one and the last non-zero input frame
          ol = back > jmax_l ? 0.f : ol;
          orr = back > jmax_r ? 0.f : orr;
        }
        store_global(p0 + (store ? 16 * j : 0), ol);
        store_global(p1c + (store ? 16 * j : 0), orr);
      }
    }
  }
}
// The partition spectra of MANY HRIR pairs (one direction per context), hrtffft::make_tables on the device: one thread per (row,
// partition, bin), the 128-tap sum in f64 in the host function's order (no contraction), cos / sin from the host's table.
__global__ __launch_bounds__(256) void hrtf_fft_tables_kernel(const float* pairs, uint32_t pair_stride, int taps, const double* cs_sn, float* out) {
  const uint32_t row = blockIdx.x / PARTS, p = blockIdx.x % PARTS;
  const int kp = (int)threadIdx.x;
  const float* pair = pairs + (uint64_t)row * pair_stride;
  double re = 0., im = 0.;
  for (int n = 0; n < 128; n++) {
    const int tap = 128 * (int)p + n;
    if (tap >= taps) break;
    const double hl = (double)pair[(size_t)tap * 2], hr = (double)pair[(size_t)tap * 2 + 1];
    const int e = (kp * n) & 255;
    const double c = cs_sn[e], sn = cs_sn[256 + e];
    re += hl * c - hr * sn;
    im += hl * sn + hr * c;
  }
  const int t = kp & 15, j = kp >> 4;
  const size_t slot = (size_t)t * ROW + (size_t)K16(j);
  float* o = out + (((uint64_t)row * PARTS + p) * TAB_SLOTS + slot) * 2;
  o[0] = (float)(re / 256.);
  o[1] = (float)(im / 256.);
}
}  // namespace

void launch_hrtf_fft_tables(const float* pairs, uint32_t pair_stride, int taps, uint32_t rows, const double* cs_sn, float* out, void* stream) {
  (void)hipMemsetAsync(out, 0, (size_t)rows * PARTS * TAB_SLOTS * 2 * sizeof(float), (hipStream_t)stream);  // (the rows' padding slots)
  hipLaunchKernelGGL(hrtf_fft_tables_kernel, dim3(rows * PARTS), dim3(256), 0, (hipStream_t)stream, pairs, pair_stride, taps, cs_sn, out);
}

void launch_hrtf_fft(const HrtfDesc& d0, void* stream) {
  HrtfDesc d = d0;
  if (d.q1 == 0) d.q1 = d.n_quanta;
  if (d.q1 <= d.q0) return;
  const uint32_t nq = d.q1 - d.q0;
  if (d.q0 != 0 || d.q1 != d.n_quanta || d.seg_len == 0) {  // a range: as many runs as it needs
    // (a block of a quantum-blocked loop: runs short enough for ~16 k groups, at least 8 quanta — four unstored heads per run)
    const uint32_t fill = (uint32_t)(((uint64_t)d.n_inst * nq + 16383) / 16384);
    d.seg_len = nq < 8 ? nq : (fill > 8 ? (fill < nq ? fill : nq) : 8);
  }
  d.n_seg = (nq + d.seg_len - 1) / d.seg_len;
  d.n_seg_pad = d.rows > 1 ? (d.n_seg + WAVES * 4 - 1) / (WAVES * 4) * (WAVES * 4) : d.n_seg;
  const size_t lds = ((size_t)PARTS * TAB_SLOTS + (size_t)WAVES * 4 * XSLOTS) * 8;
  const uint64_t groups = (uint64_t)d.n_inst * d.n_seg_pad;
  const dim3 grid((unsigned)((groups + WAVES * 4 - 1) / (WAVES * 4))), block(WAVES * 64);
  if (d.jmax)
    hipLaunchKernelGGL(hrtf_fft_kernel<true>, grid, block, lds, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL(hrtf_fft_kernel<false>, grid, block, lds, (hipStream_t)stream, d);
}

}  // namespace waa
