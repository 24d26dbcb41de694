// waa_frozen.hip — the two node kinds whose render state FREEZES while they do not process, rendered node-major:
//
//   * WaveShaperNode with 2x / 4x oversampling (waveshaper.rs:395-481): up-sample the quantum (rubato FftFixedInOut,
//     third party), apply the curve, down-sample.  A silent input (with a curve that maps 0 to 0) skips the whole
//     block — the resamplers' overlap is NOT flushed, it is added to the next block that is processed; a change of the
//     quantum's channel count re-creates the resamplers (overlap lost).
//   * PannerNode with the HRTF panning model (panner.rs:697-711,781-829): an FIR with a per-quantum impulse response
//     whose history is the input of the quanta the node PROCESSED (silent quanta after the tail counter ran out are
//     skipped and leave no trace in the history).
//
// link_kernel replays that control flow once per instance over the per-quantum codes of the node's input and leaves
// a `prev` table; everything else is parallel over (instance, quantum):
//   qgemm_*      — both resampling stages are LINEAR maps of (this block, previous processed block), i.e. matrix
//                  products over render quanta.  Product path: qgemm_bf16x6_w8_kernel, the bf16 matrix
//                  cores at f32 accuracy (exact three-way bf16 split of both operands, six MFMA products per f32
//                  product); cross-checks: qgemm_mfma_kernel (f32 MFMA) and qgemm_kernel (f32 vector FMA).
//   hrtf_kernel  — direct-form FIR, one wavefront per (instance, quantum), sliding register window.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"

namespace waa {

// ---------------------------------------------------------------------------------------------------------------
// One thread per instance walks its row of codes in order; the rows of the 64 instances of a workgroup are moved
// through LDS in chunks of 128 quanta, so that global loads and stores are 16-byte pieces of contiguous rows instead of
// one byte / one word per lane at a stride of a whole row.
namespace {
constexpr int LCH = 128;               // quanta per chunk
constexpr int LCP = LCH + 4;           // row pitch of the byte tiles
constexpr int LPP = LCH + 1;           // row pitch (words) of the prev tile
}  // namespace
__global__ __launch_bounds__(64) void link_kernel(const LinkDesc d) {
  __shared__ __attribute__((aligned(16))) uint8_t cin[64 * LCP];
  __shared__ __attribute__((aligned(16))) uint8_t cout[64 * LCP];
  __shared__ int32_t pv[64 * LPP];
  const int t = threadIdx.x;
  const uint32_t inst0 = blockIdx.x * 64, inst = inst0 + t;
  const bool live = inst < d.n_inst;
  int32_t last = LINK_FRESH;
  int cur_ch = 1;               // kind 0: channels_x2 / channels_x4 start at 1 (waveshaper.rs:526-527)
  uint64_t tail_counter = 0;    // kind 1: only ever grows (panner.rs:697-711)
  for (uint32_t q0 = 0; q0 < d.n_quanta; q0 += LCH) {
    // rows in: 64 x 128 B as 512 pieces of 16 B (code_stride is a multiple of 16, rows are 16-byte aligned)
    for (int p = t; p < 64 * (LCH / 16); p += 64) {
      const int r = p / (LCH / 16), off = (p % (LCH / 16)) * 16;
      uint4 v = make_uint4(0x81818181u, 0x81818181u, 0x81818181u, 0x81818181u);
      if (inst0 + r < d.n_inst && q0 + off < d.code_stride)
        v = *reinterpret_cast<const uint4*>(d.in_code + (uint64_t)(inst0 + r) * d.code_stride + q0 + off);
      *reinterpret_cast<uint32_t*>(cin + r * LCP + off + 0) = v.x;
      *reinterpret_cast<uint32_t*>(cin + r * LCP + off + 4) = v.y;
      *reinterpret_cast<uint32_t*>(cin + r * LCP + off + 8) = v.z;
      *reinterpret_cast<uint32_t*>(cin + r * LCP + off + 12) = v.w;
    }
    __syncthreads();
    if (live) {
      const uint32_t n = d.n_quanta - q0 < (uint32_t)LCH ? d.n_quanta - q0 : (uint32_t)LCH;
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = cin[t * LCP + i];
        const bool silent = (c & CODE_SILENT) != 0;
        int32_t link;
        uint8_t oc;
        if (d.kind == 0) {
          // WaveShaperRenderer::process, X2 / X4 (waveshaper.rs:395-400, 409-425)
          if (silent && d.can_propagate_silence) {
            link = LINK_SKIP;
            oc = (uint8_t)(1u | CODE_SILENT);
          } else {
            // (a silent quantum keeps the count it was MIXED to — explicit mode: the node's channelCount, quantum.rs:532-569 —
            // not 1: the code carries it)
            const int nch = (int)(c & 63u);
            if (nch != cur_ch) {  // the resamplers are re-created for the new channel count: their overlap is gone
              cur_ch = nch;
              last = LINK_FRESH;
            }
            link = last;
            last = (int32_t)(q0 + i);
            oc = (uint8_t)nch;
          }
        } else {
          // PannerRenderer::process, HRTF (panner.rs:697-711)
          bool skip = false;
          if (silent) {
            if (!((uint64_t)d.tail_frames > tail_counter))
              skip = true;
            else
              tail_counter += RQ;
          }
          if (skip) {
            link = LINK_SKIP;
            oc = (uint8_t)(1u | CODE_SILENT);
          } else {
            link = last;
            last = (int32_t)(q0 + i);
            oc = (uint8_t)2u;
          }
        }
        pv[t * LPP + i] = link;
        cout[t * LCP + i] = oc;
      }
    }
    __syncthreads();
    // rows out: prev as 16-byte pieces (4 words), codes as 16-byte pieces
    for (int p = t; p < 64 * (LCH / 4); p += 64) {
      const int r = p / (LCH / 4), off = (p % (LCH / 4)) * 4;
      if (inst0 + r >= d.n_inst) continue;
      int32_t* dst = d.prev + (uint64_t)(inst0 + r) * d.prev_stride + q0 + off;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (q0 + off + e < d.n_quanta) dst[e] = pv[r * LPP + off + e];
    }
    if (d.out_code)
      for (int p = t; p < 64 * (LCH / 16); p += 64) {
        const int r = p / (LCH / 16), off = (p % (LCH / 16)) * 16;
        if (inst0 + r >= d.n_inst || q0 + off >= d.code_stride) continue;
        uint4 v;
        v.x = *reinterpret_cast<const uint32_t*>(cout + r * LCP + off + 0);
        v.y = *reinterpret_cast<const uint32_t*>(cout + r * LCP + off + 4);
        v.z = *reinterpret_cast<const uint32_t*>(cout + r * LCP + off + 8);
        v.w = *reinterpret_cast<const uint32_t*>(cout + r * LCP + off + 12);
        *reinterpret_cast<uint4*>(d.out_code + (uint64_t)(inst0 + r) * d.code_stride + q0 + off) = v;
      }
    __syncthreads();
  }
}
// The same automaton over a RANGE of quanta with its state in memory between launches: the node sits inside a feedback loop that
// is rendered block by block (round 5), so its input codes of quantum q only exist once the loop's items have rendered q.  One
// thread per instance, a handful of quanta per launch.  State words: last + 2 (0 = LINK_SKIP never stored: the initial LINK_FRESH
// = -1 is 1 ... so 0 means "never written" = LINK_FRESH), cur_ch - 1, the tail counter in two halves.
__global__ __launch_bounds__(64) void link_range_kernel(const LinkDesc d) {
  const uint32_t inst = blockIdx.x * 64 + threadIdx.x;
  if (inst >= d.n_inst) return;
  int32_t* stw = d.state + (uint64_t)inst * 4;
  int32_t last = stw[0] == 0 ? LINK_FRESH : stw[0] - 2;
  int cur_ch = stw[1] + 1;
  uint64_t tail_counter = (uint64_t)(uint32_t)stw[2] | ((uint64_t)(uint32_t)stw[3] << 32);
  const uint8_t* cin = d.in_code + (uint64_t)inst * d.code_stride;
  uint8_t* cout = d.out_code ? d.out_code + (uint64_t)inst * d.code_stride : nullptr;
  int32_t* pv = d.prev + (uint64_t)inst * d.prev_stride;
  for (uint32_t q = d.q0; q < d.q1; q++) {
    const uint32_t c = cin[q];
    const bool silent = (c & CODE_SILENT) != 0;
    int32_t link;
    uint8_t oc;
    if (d.kind == 0) {  // (as in link_kernel: waveshaper.rs:395-400, 409-425)
      if (silent && d.can_propagate_silence) {
        link = LINK_SKIP;
        oc = (uint8_t)(1u | CODE_SILENT);
      } else {
        const int nch = (int)(c & 63u);
        if (nch != cur_ch) {
          cur_ch = nch;
          last = LINK_FRESH;
        }
        link = last;
        last = (int32_t)q;
        oc = (uint8_t)nch;
      }
    } else {  // (panner.rs:697-711)
      bool skip = false;
      if (silent) {
        if (!((uint64_t)d.tail_frames > tail_counter))
          skip = true;
        else
          tail_counter += RQ;
      }
      if (skip) {
        link = LINK_SKIP;
        oc = (uint8_t)(1u | CODE_SILENT);
      } else {
        link = last;
        last = (int32_t)q;
        oc = (uint8_t)2u;
      }
    }
    pv[q] = link;
    if (cout) cout[q] = oc;
  }
  stw[0] = last + 2;
  stw[1] = cur_ch - 1;
  stw[2] = (int32_t)(uint32_t)tail_counter;
  stw[3] = (int32_t)(uint32_t)(tail_counter >> 32);
}
void launch_link(const LinkDesc& d, void* stream) {
  if (d.state) {  // the ranged form
    if (d.q1 > d.q0) hipLaunchKernelGGL(link_range_kernel, dim3((d.n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
    return;
  }
  hipLaunchKernelGGL(link_kernel, dim3((d.n_inst + 63) / 64), dim3(64), 0, (hipStream_t)stream, d);
}

// ---------------------------------------------------------------------------------------------------------------
#ifdef WAA_MEASURE  // the matrix (qgemm) form of the oversampling stages: measurement build only
namespace {
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA = BM + 4, LDB = BN + 4;  // padded LDS rows (bank spread of the transposing stores)

__device__ __forceinline__ float shape_curve(const float* curve, int nn, float input) {  // waveshaper.rs:555-573
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
constexpr int CURVE_LDS = 4100;
__device__ __forceinline__ float shape_curve_lds(const float* curve, int nn, float input) {  // waveshaper.rs:555-573
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
}  // namespace

__global__ __launch_bounds__(256) void qgemm_kernel(const QGemmDesc d) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  __shared__ int32_t prev_s[BN];
  const int tid = threadIdx.x;
  const int ty = tid & 15;   // row group: rows ty * 4 + {0..3} and 64 + ty * 4 + {0..3} of the tile
  const int tx = tid >> 4;   // column group: columns tx * 4 + {0..3} and 64 + tx * 4 + {0..3}
  const uint32_t q0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const uint32_t inst = blockIdx.z / (uint32_t)d.nch, ch = blockIdx.z % (uint32_t)d.nch;
  const float* src = d.src + (uint64_t)inst * d.src_inst + (uint64_t)ch * d.src_ch;
  if (tid < BN) {
    const uint32_t q = q0 + tid;
    prev_s[tid] = q < d.n_quanta ? load_global(d.prev + (uint64_t)inst * d.prev_stride + q) : LINK_SKIP;
  }
  __syncthreads();
  const int K = 2 * d.Kh;
  // global -> register staging: A tile BK x BM = 512 float4 (2 per thread), B tile BN columns x 4 float4 (2 per thread)
  f4v ra[2], rb[2];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + i * 256;
      const int ak = idx >> 5, am = (idx & 31) * 4;  // 32 float4 per A row
      ra[i] = load_global_f4(d.A + (uint64_t)(kk + ak) * d.M + m0 + am);
      const int col = idx >> 2, part = idx & 3;
      const uint32_t q = q0 + col;
      const int32_t p = prev_s[col];
      const bool second = kk >= d.Kh;
      const int32_t sq = second ? p : (int32_t)q;
      f4v v = {0.f, 0.f, 0.f, 0.f};
      if (p != LINK_SKIP && sq >= 0)
        v = load_global_f4(src + (uint64_t)sq * d.src_q + (uint64_t)((second ? kk - d.Kh : kk) + part * 4));
      rb[i] = v;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + i * 256;
      const int ak = idx >> 5, am = (idx & 31) * 4;
      *reinterpret_cast<f4v*>(&As[buf][ak][am]) = ra[i];
      const int col = idx >> 2, part = idx & 3;
      Bs[buf][part * 4 + 0][col] = rb[i].x;
      Bs[buf][part * 4 + 1][col] = rb[i].y;
      Bs[buf][part * 4 + 2][col] = rb[i].z;
      Bs[buf][part * 4 + 3][col] = rb[i].w;
    }
  };
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = 0.f;
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int kk = 0; kk < K; kk += BK) {
    const bool more = kk + BK < K;
    if (more) fetch(kk + BK);
#pragma unroll
    for (int k = 0; k < BK; k++) {
      const f4v a0 = *reinterpret_cast<const f4v*>(&As[buf][k][ty * 4]);
      const f4v a1 = *reinterpret_cast<const f4v*>(&As[buf][k][64 + ty * 4]);
      const f4v b0 = *reinterpret_cast<const f4v*>(&Bs[buf][k][tx * 4]);
      const f4v b1 = *reinterpret_cast<const f4v*>(&Bs[buf][k][64 + tx * 4]);
      const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_fmaf(a[i], bb[j], acc[i][j]);
    }
    if (more) {
      stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  // epilogue: column j of the thread = quantum q0 + (j < 4 ? tx * 4 + j : 64 + tx * 4 + j - 4); rows in two float4
  float* dst = d.dst + (uint64_t)inst * d.dst_inst + (uint64_t)ch * d.dst_ch;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int col = (j < 4 ? tx * 4 + j : 64 + tx * 4 + (j - 4));
    const uint32_t q = q0 + col;
    if (q >= d.n_quanta) continue;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      f4v v = {acc[h * 4 + 0][j], acc[h * 4 + 1][j], acc[h * 4 + 2][j], acc[h * 4 + 3][j]};
      if (d.curve) {
        v.x = shape_curve(d.curve, d.curve_n, v.x);
        v.y = shape_curve(d.curve, d.curve_n, v.y);
        v.z = shape_curve(d.curve, d.curve_n, v.z);
        v.w = shape_curve(d.curve, d.curve_n, v.w);
      }
      *(WAA_GLOBAL_AS f4v*)(dst + (uint64_t)q * d.dst_q + m0 + h * 64 + ty * 4) = v;
    }
  }
}
// The same product on the matrix cores: v_mfma_f32_32x32x2_f32 multiplies f32 by f32 into f32 (exact products, the
// same peak as the vector FMA pipe — 64 flop / clk / SIMD — but reachable: the FMA form above spends a third of its
// issue slots on LDS reads and register shuffles).  Operand roles are swapped with respect to the maths: the SOURCE
// tile (render quanta x k) is the MFMA's A operand and the matrix tile (k x output frames) its B operand, so a lane of
// the 32 x 32 result block holds 16 quanta of ONE output frame and the 32 lanes of a half-wave hold 32 consecutive
// frames of one quantum — the epilogue stores are 128-byte runs.  Workgroup tile 128 x 128, one 64 x 64 quadrant (2 x 2
// result blocks, 64 accumulator registers) per wavefront; staging through LDS exactly as in qgemm_kernel.
// DBG (WAA_QGEMM_DEBUG, measurement aid; results are wrong by construction): 1 = no stores, 2 = no source loads,
// 3 = no matrix loads, 4 = no MFMAs
typedef float f16v __attribute__((ext_vector_type(16)));
template <int DBG>
__global__ __launch_bounds__(256) void qgemm_mfma_kernel(const QGemmDesc d) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  __shared__ int32_t prev_s[BN];
  __shared__ float curve_s[CURVE_LDS];  // the WaveShaper curve (two gathers per output element in the epilogue)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const bool curve_in_lds = d.curve && d.curve_n <= CURVE_LDS;
  if (curve_in_lds)
    for (int i = tid; i < d.curve_n; i += 256) curve_s[i] = load_global(d.curve + i);
  const int wq = (wave >> 1) * 64, wm = (wave & 1) * 64;  // this wave's quadrant: quanta [wq, wq + 64), frames [wm, wm + 64)
  const int l32 = lane & 31, kh = lane >> 5;
  const uint32_t q0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const uint32_t inst = blockIdx.z / (uint32_t)d.nch, ch = blockIdx.z % (uint32_t)d.nch;
  const float* src = d.src + (uint64_t)inst * d.src_inst + (uint64_t)ch * d.src_ch;
  if (tid < BN) {
    const uint32_t q = q0 + tid;
    prev_s[tid] = q < d.n_quanta ? load_global(d.prev + (uint64_t)inst * d.prev_stride + q) : LINK_SKIP;
  }
  __syncthreads();
  const int K = 2 * d.Kh;
  // two register sets: a tile is requested TWO k-tiles (two MFMA phases, ~1.7 us) before it is staged into LDS — one
  // phase (0.85 us) is shorter than the HBM latency under load, and every k-tile then ended in a wait for its loads
  f4v ra0[2], rb0[2], ra1[2], rb1[2];
  auto fetch = [&](int kk, f4v (&ra)[2], f4v (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + i * 256;
      const int ak = idx >> 5, am = (idx & 31) * 4;
      if constexpr (DBG == 3)
        ra[i] = f4v{1.f, 1.f, 1.f, 1.f};
      else
        ra[i] = load_global_f4(d.A + (uint64_t)(kk + ak) * d.M + m0 + am);
      const int col = idx >> 2, part = idx & 3;
      const uint32_t q = q0 + col;
      const int32_t p = prev_s[col];
      const bool second = kk >= d.Kh;
      const int32_t sq = second ? p : (int32_t)q;
      f4v v = {0.f, 0.f, 0.f, 0.f};
      if (DBG != 2 && p != LINK_SKIP && sq >= 0)
        v = load_global_f4(src + (uint64_t)sq * d.src_q + (uint64_t)((second ? kk - d.Kh : kk) + part * 4));
      rb[i] = v;
    }
  };
  auto stage = [&](int buf, const f4v (&ra)[2], const f4v (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + i * 256;
      const int ak = idx >> 5, am = (idx & 31) * 4;
      *reinterpret_cast<f4v*>(&As[buf][ak][am]) = ra[i];
      const int col = idx >> 2, part = idx & 3;
      Bs[buf][part * 4 + 0][col] = rb[i].x;
      Bs[buf][part * 4 + 1][col] = rb[i].y;
      Bs[buf][part * 4 + 2][col] = rb[i].z;
      Bs[buf][part * 4 + 3][col] = rb[i].w;
    }
  };
  f16v acc[2][2];  // [quanta block][frame block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
    // all operands of the k-tile first (32 registers), then 32 MFMAs back to back: with the operands read step by step
    // into the same four registers every group of four MFMAs waited for its LDS reads
    // A operand (32 x 2): row = quantum l32, k = kh;  B operand (2 x 32): k = kh, column = output frame l32
    float sv[BK / 2][2], cv[BK / 2][2];
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
      sv[k / 2][0] = Bs[buf][k + kh][wq + l32];
      sv[k / 2][1] = Bs[buf][k + kh][wq + 32 + l32];
      cv[k / 2][0] = As[buf][k + kh][wm + l32];
      cv[k / 2][1] = As[buf][k + kh][wm + 32 + l32];
    }
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads back between the MFMA groups)
#pragma unroll
    for (int k = 0; k < BK / 2; k++) {
      if constexpr (DBG == 4) {
        acc[0][0][k] += sv[k][0] * cv[k][0];
        acc[1][1][k] += sv[k][1] * cv[k][1];
        continue;
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[k][0], cv[k][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[k][0], cv[k][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[k][1], cv[k][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[k][1], cv[k][1], acc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  fetch(0, ra0, rb0);
  stage(0, ra0, rb0);
  __syncthreads();
  fetch(BK, ra0, rb0);  // (K >= 2 * BK always: K = 2 * Kh, Kh a multiple of 128)
  int buf = 0;
  for (int kk = 0; kk < K; kk += 2 * BK) {
    // tile kk is in LDS[buf], tile kk + BK in flight in set 0
    if (kk + 2 * BK < K) fetch(kk + 2 * BK, ra1, rb1);
    mfma_tile(buf);
    stage(buf ^ 1, ra0, rb0);
    __syncthreads();
    buf ^= 1;
    // tile kk + BK is in LDS[buf], tile kk + 2 BK in flight in set 1
    if (kk + 3 * BK < K) fetch(kk + 3 * BK, ra0, rb0);
    mfma_tile(buf);
    if (kk + 2 * BK < K) {
      stage(buf ^ 1, ra1, rb1);
      __syncthreads();
      buf ^= 1;
    }
  }
  // result block layout: element e of a lane = row (e / 4) * 8 + kh * 4 + e % 4 (quantum), column l32 (output frame)
  float* dst = d.dst + (uint64_t)inst * d.dst_inst + (uint64_t)ch * d.dst_ch;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const uint32_t q = q0 + (uint32_t)(wq + i * 32 + (e >> 2) * 8 + kh * 4 + (e & 3));
      if (q >= d.n_quanta) continue;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        float v = acc[i][j][e];
        if (curve_in_lds)
          v = shape_curve_lds(curve_s, d.curve_n, v);
        else if (d.curve)
          v = shape_curve(d.curve, d.curve_n, v);
        if constexpr (DBG == 1) {
          asm volatile("" ::"v"(v));
        } else {
          store_global(dst + (uint64_t)q * d.dst_q + m0 + wm + j * 32 + l32, v);
        }
      }
    }
}
// The same product on the bf16 matrix cores, at f32 accuracy.  Every f32 value is the EXACT sum of three bf16 values
// (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid: 8 + 8 + 8 significant bits), so
//     s * a = (sh + sm + sl) * (ah + am + al)
// and the six products of weight >= 2^-16 — sh ah, sh am, sm ah, sh al, sl ah, sm am — carry it to 2^-24 relative (the
// three dropped ones are below the rounding of an f32 product); each bf16 x bf16 product is exact in the f32 accumulator.
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32 MFMA / vector FMA, six of them per f32 product leave 2.7x.
// Measured against the f64 product on resampler-shaped data: 2.0e-7 relative RMS (a chain of f32 FMAs: 2.9e-7).
// Tile: 128 quanta x 128 output frames per workgroup, one 64 x 64 quadrant per wavefront, k in steps of 16.  The matrix
// comes pre-split and pre-tiled from the host (QGemmDesc::A16); the source tile is split on the fly while it is staged.
// LDS image of a plane of a tile: fragment (block of 32 rows, k half h, row) = 16 B at slot (block * 2 + h) * 32 + row, so
// the 64 lanes of a fragment read (lane & 31 = row, lane >> 5 = h) touch 64 consecutive slots: conflict-free ds_read_b128.
namespace {
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
constexpr int XK = 16;
struct XSet {          // one k-tile in flight: 2 x 4 source floats, 3 x 8 matrix bf16 per thread
  f4v s[2];
  u4v m[3];
};
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
// (x0, x1) = hi + mid + lo, each a pair of bf16 in one word: v_cvt_pk_bf16_f32 rounds to nearest even, the residuals
// x - hi and (x - hi) - mid are exact in f32 and the last one has at most 8 significant bits.  inf / NaN: hi keeps the
// class, the residuals are NaN — the products are NaN either way.
__device__ __forceinline__ void split_bf16x3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const bf2v h = {(__bf16)x0, (__bf16)x1};
  hi = __builtin_bit_cast(uint32_t, h);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  const bf2v m = {(__bf16)r0, (__bf16)r1};
  mid = __builtin_bit_cast(uint32_t, m);
  const float l0 = r0 - __uint_as_float(mid << 16), l1 = r1 - __uint_as_float(mid & 0xffff0000u);
  const bf2v l = {(__bf16)l0, (__bf16)l1};
  lo = __builtin_bit_cast(uint32_t, l);
}
// LDS slot (16 B) of fragment (block of 32 rows, k half h, row): the two halves of a block 40 slots apart, so that the
// staging writes (8 rows x both halves per 32 lanes) spread over all banks like the fragment reads do
__device__ __forceinline__ int xslot(int blk, int h, int row) { return blk * 72 + h * 40 + row; }
constexpr int XSLOTS = 4 * 72;
}  // namespace
// DBG (WAA_QGEMM_DEBUG=a|b|c, measurement aid, results wrong by construction): 1 = no global loads after the first four
// tiles, 2 = no MFMAs, 3 = nothing staged after the first tile (no split, no LDS writes)
template <int DBG>
__global__ __launch_bounds__(256, 2) void qgemm_bf16x6_kernel(const QGemmDesc d) {
  __shared__ u4v Ss[2][3][XSLOTS];  // [buffer][plane][slot]: source tile, 128 quanta x 16 k
  __shared__ u4v Ms[2][3][XSLOTS];  // matrix tile, 128 output frames x 16 k
  __shared__ int32_t prev_s[BN];
  __shared__ float curve_s[CURVE_LDS];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const bool curve_in_lds = d.curve && d.curve_n <= CURVE_LDS;
  if (curve_in_lds)
    for (int i = tid; i < d.curve_n; i += 256) curve_s[i] = load_global(d.curve + i);
  const int wq = (wave >> 1) * 64, wm = (wave & 1) * 64;
  const int l32 = lane & 31, kh = lane >> 5;
  const uint32_t q0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const uint32_t inst = blockIdx.z / (uint32_t)d.nch, ch = blockIdx.z % (uint32_t)d.nch;
  const float* src = d.src + (uint64_t)inst * d.src_inst + (uint64_t)ch * d.src_ch;
  if (tid < BN) {
    const uint32_t q = q0 + tid;
    prev_s[tid] = q < d.n_quanta ? load_global(d.prev + (uint64_t)inst * d.prev_stride + q) : LINK_SKIP;
  }
  __syncthreads();
  const int KT = 2 * d.Kh / XK;  // (a multiple of 16: Kh is a multiple of 128)
  // this thread's two source pieces: quantum row `col`, 4 consecutive k; its three matrix pieces: 16 B of plane p
  int32_t srow[2][2];  // [piece][first / second half of k]: source quantum or -1
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int col = (tid + i * 256) >> 2;
    const int32_t p = prev_s[col];
    srow[i][0] = p != LINK_SKIP ? (int32_t)(q0 + col) : -1;
    srow[i][1] = p != LINK_SKIP ? p : -1;  // (LINK_FRESH = -1: no previous block)
  }
  const u4v* a16 = reinterpret_cast<const u4v*>(d.A16);  // tile kt, plane p, column m, half h: ((kt * 3 + p) * M + m) * 2 + h
  bool warm = false;  // (DBG 1 / 3)
  auto fetch = [&](int kt, XSet& x) __attribute__((always_inline)) {
    if (DBG == 1 && warm) return;
    // tile order: (k-tile p of this block, k-tile p of the previous block), p = 0, 1, ...: the two read the same source
    // columns of neighbouring quanta (the previous processed block is almost always quantum q - 1), i.e. the same cache
    // lines one phase apart.  In plain k order the second pass over the rows came half a workgroup lifetime later, after
    // the 128 KB panel had left the L2 (64 workgroups per XCD): the source was fetched from HBM several times.
    const bool second = kt & 1;
    const int koff = (kt >> 1) * XK;
    const int mt = (kt >> 1) + (second ? d.Kh / XK : 0);  // matrix tile (rows mt * 16 ... of the k-major matrix)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int part = (tid + i * 256) & 3;
      const int32_t sq = srow[i][second ? 1 : 0];
      // (unconditional load, row 0 instead of a skipped row, zeroed when staged: a load inside a branch is waited for at once)
      x.s[i] = load_global_f4(src + (uint64_t)(sq < 0 ? 0 : sq) * d.src_q + (uint64_t)(koff + part * 4));
    }
#pragma unroll
    for (int p = 0; p < 3; p++)
      x.m[p] = *(const WAA_GLOBAL_AS u4v*)(a16 + ((uint64_t)(mt * 3 + p) * d.M + m0) * 2 + tid);
  };
  auto stage = [&](int kt, int buf, const XSet& x) __attribute__((always_inline)) {
    if (DBG == 3 && warm) {
      asm volatile("" ::"v"(x.s[0]), "v"(x.s[1]), "v"(x.m[0]), "v"(x.m[1]), "v"(x.m[2]));
      return;
    }
    const bool second = kt & 1;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + i * 256;
      const int col = idx >> 2, part = idx & 3;
      const bool live = srow[i][second ? 1 : 0] >= 0;
      uint32_t h[2], m[2], l[2];
      split_bf16x3(x.s[i].x, x.s[i].y, h[0], m[0], l[0]);
      split_bf16x3(x.s[i].z, x.s[i].w, h[1], m[1], l[1]);
      const int slot = xslot(col >> 5, part >> 1, col & 31);
      const u2v z = {0u, 0u};
      *(reinterpret_cast<u2v*>(&Ss[buf][0][slot]) + (part & 1)) = live ? u2v{h[0], h[1]} : z;
      *(reinterpret_cast<u2v*>(&Ss[buf][1][slot]) + (part & 1)) = live ? u2v{m[0], m[1]} : z;
      *(reinterpret_cast<u2v*>(&Ss[buf][2][slot]) + (part & 1)) = live ? u2v{l[0], l[1]} : z;
    }
    // matrix piece tid of a plane = column tid >> 1, half tid & 1
    const int mslot = xslot((tid >> 1) >> 5, tid & 1, (tid >> 1) & 31);
#pragma unroll
    for (int p = 0; p < 3; p++) Ms[buf][p][mslot] = x.m[p];
  };
  f16v acc[2][2];  // [quanta block][frame block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    bf8v sf[2][3], mf[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        sf[i][p] = __builtin_bit_cast(bf8v, Ss[buf][p][xslot((wq >> 5) + i, kh, l32)]);
        mf[i][p] = __builtin_bit_cast(bf8v, Ms[buf][p][xslot((wm >> 5) + i, kh, l32)]);
      }
    __builtin_amdgcn_sched_barrier(0);
    // smallest products first; the four result blocks interleaved, so that consecutive MFMAs are independent
    constexpr int PS[6] = {2, 0, 1, 1, 0, 0}, PM[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (DBG == 2)
            asm volatile("" : "+v"(acc[i][j]) : "v"(sf[i][PS[t]]), "v"(mf[j][PM[t]]));
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sf[i][PS[t]], mf[j][PM[t]], acc[i][j], 0, 0, 0);
        }
  };
  // four register sets by tile index mod 4: a tile is requested four compute phases (~1.3 us) before it is staged
  XSet x0, x1, x2, x3;
  fetch(0, x0);
  fetch(1, x1);
  fetch(2, x2);
  fetch(3, x3);
  stage(0, 0, x0);
  __syncthreads();
  warm = true;
  int buf = 0;
  // Every fetch and every stage is UNCONDITIONAL (past the last tile: the last tile again, staged into the buffer nobody
  // reads): the wait counts of the loads are static, and a fetch inside an `if` makes the compiler assume it did not
  // happen — the stage of the oldest set then waited for the fetch issued a few instructions earlier (measured: 3x).
  const int KL = KT - 1;
  for (int kt = 0; kt < KT; kt += 4) {
    // tile kt in LDS[buf]; kt + 1, kt + 2, kt + 3 in flight in x1, x2, x3; x0 free
    fetch(kt + 4 < KL ? kt + 4 : KL, x0);
    compute(buf);
    stage(kt + 1, buf ^ 1, x1);
    __syncthreads();
    buf ^= 1;
    fetch(kt + 5 < KL ? kt + 5 : KL, x1);
    compute(buf);
    stage(kt + 2, buf ^ 1, x2);
    __syncthreads();
    buf ^= 1;
    fetch(kt + 6 < KL ? kt + 6 : KL, x2);
    compute(buf);
    stage(kt + 3, buf ^ 1, x3);
    __syncthreads();
    buf ^= 1;
    fetch(kt + 7 < KL ? kt + 7 : KL, x3);
    compute(buf);
    stage(kt + 4 < KL ? kt + 4 : KL, buf ^ 1, x0);
    __syncthreads();
    buf ^= 1;
  }
  // result block layout: element e of a lane = row (e / 4) * 8 + kh * 4 + e % 4 (quantum), column l32 (output frame)
  float* dst = d.dst + (uint64_t)inst * d.dst_inst + (uint64_t)ch * d.dst_ch;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const uint32_t q = q0 + (uint32_t)(wq + i * 32 + (e >> 2) * 8 + kh * 4 + (e & 3));
      if (q >= d.n_quanta) continue;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        float v = acc[i][j][e];
        if (curve_in_lds)
          v = shape_curve_lds(curve_s, d.curve_n, v);
        else if (d.curve)
          v = shape_curve(d.curve, d.curve_n, v);
        store_global(dst + (uint64_t)q * d.dst_q + m0 + wm + j * 32 + l32, v);
      }
    }
}
// The same tile with EIGHT wavefronts (64 quanta x 32 frames each, 32 accumulator registers): four waves per SIMD
// with two workgroups per CU.  The four-wave form ran as the plain SUM of its parts (MFMA time + staging + loads + LDS
// hand-offs: removing any one of them removed exactly its own time) — two waves per SIMD do not overlap them.
struct XSet8 {  // one k-tile in flight: 4 source floats, 2 x 8 matrix bf16 per thread
  f4v s[1];
  u4v m[2];
};
template <int DBG>
__global__ __launch_bounds__(512, 4) void qgemm_bf16x6_w8_kernel(const QGemmDesc d) {
  __shared__ u4v Ss[2][3][XSLOTS];  // [buffer][plane][slot]: source tile, 128 quanta x 16 k
  __shared__ u4v Ms[2][3][XSLOTS];  // matrix tile, 128 output frames x 16 k
  __shared__ int32_t prev_s[BN];
  __shared__ float curve_s[CURVE_LDS];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const bool curve_in_lds = d.curve && d.curve_n <= CURVE_LDS;
  if (curve_in_lds)
    for (int i = tid; i < d.curve_n; i += 512) curve_s[i] = load_global(d.curve + i);
  const int wq = (wave >> 2) * 64, wm = (wave & 3) * 32;
  const int l32 = lane & 31, kh = lane >> 5;
  const uint32_t q0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const uint32_t inst = blockIdx.z / (uint32_t)d.nch, ch = blockIdx.z % (uint32_t)d.nch;
  const float* src = d.src + (uint64_t)inst * d.src_inst + (uint64_t)ch * d.src_ch;
  if (tid < BN) {
    const uint32_t q = q0 + tid;
    prev_s[tid] = q < d.n_quanta ? load_global(d.prev + (uint64_t)inst * d.prev_stride + q) : LINK_SKIP;
  }
  __syncthreads();
  const int KT = 2 * d.Kh / XK;  // (a multiple of 16: Kh is a multiple of 128)
  // this thread's two source pieces: quantum row `col`, 4 consecutive k; its three matrix pieces: 16 B of plane p
  int32_t srow[1][2];  // [piece][first / second half of k]: source quantum or -1
#pragma unroll
  for (int i = 0; i < 1; i++) {
    const int col = tid >> 2;
    const int32_t p = prev_s[col];
    srow[i][0] = p != LINK_SKIP ? (int32_t)(q0 + col) : -1;
    srow[i][1] = p != LINK_SKIP ? p : -1;  // (LINK_FRESH = -1: no previous block)
  }
  const u4v* a16 = reinterpret_cast<const u4v*>(d.A16);  // tile kt, plane p, column m, half h: ((kt * 3 + p) * M + m) * 2 + h
  bool warm = false;  // (DBG 1 / 3)
  auto fetch = [&](int kt, XSet8& x) __attribute__((always_inline)) {
    if (DBG == 1 && warm) return;
    // tile order: (k-tile p of this block, k-tile p of the previous block), p = 0, 1, ...: the two read the same source
    // columns of neighbouring quanta (the previous processed block is almost always quantum q - 1), i.e. the same cache
    // lines one phase apart.  In plain k order the second pass over the rows came half a workgroup lifetime later, after
    // the 128 KB panel had left the L2 (64 workgroups per XCD): the source was fetched from HBM several times.
    const bool second = kt & 1;
    const int koff = (kt >> 1) * XK;
    const int mt = (kt >> 1) + (second ? d.Kh / XK : 0);  // matrix tile (rows mt * 16 ... of the k-major matrix)
#pragma unroll
    for (int i = 0; i < 1; i++) {
      const int part = tid & 3;
      const int32_t sq = srow[i][second ? 1 : 0];
      // (unconditional load, row 0 instead of a skipped row, zeroed when staged: a load inside a branch is waited for at once)
      x.s[i] = load_global_f4(src + (uint64_t)(sq < 0 || DBG == 4 ? 0 : sq) * d.src_q + (uint64_t)(koff + part * 4));  // (DBG 4: one hot row)
    }
    // 768 matrix pieces for 512 threads: piece tid & 255 of plane tid >> 8, and of plane 2 (both halves of the workgroup:
    // the same load and the same LDS store twice, instead of a branch around a load)
    x.m[0] = *(const WAA_GLOBAL_AS u4v*)(a16 + ((uint64_t)(mt * 3 + (tid >> 8)) * d.M + m0) * 2 + (tid & 255));
    x.m[1] = *(const WAA_GLOBAL_AS u4v*)(a16 + ((uint64_t)(mt * 3 + 2) * d.M + m0) * 2 + (tid & 255));
  };
  auto stage = [&](int kt, int buf, const XSet8& x) __attribute__((always_inline)) {
    if (DBG == 3 && warm) {
      asm volatile("" ::"v"(x.s[0]), "v"(x.m[0]), "v"(x.m[1]));
      return;
    }
    const bool second = kt & 1;
#pragma unroll
    for (int i = 0; i < 1; i++) {
      const int idx = tid;
      const int col = idx >> 2, part = idx & 3;
      const bool live = srow[i][second ? 1 : 0] >= 0;
      uint32_t h[2], m[2], l[2];
      split_bf16x3(x.s[i].x, x.s[i].y, h[0], m[0], l[0]);
      split_bf16x3(x.s[i].z, x.s[i].w, h[1], m[1], l[1]);
      const int slot = xslot(col >> 5, part >> 1, col & 31);
      const u2v z = {0u, 0u};
      *(reinterpret_cast<u2v*>(&Ss[buf][0][slot]) + (part & 1)) = live ? u2v{h[0], h[1]} : z;
      *(reinterpret_cast<u2v*>(&Ss[buf][1][slot]) + (part & 1)) = live ? u2v{m[0], m[1]} : z;
      *(reinterpret_cast<u2v*>(&Ss[buf][2][slot]) + (part & 1)) = live ? u2v{l[0], l[1]} : z;
    }
    // matrix piece tid of a plane = column tid >> 1, half tid & 1
    const int mslot = xslot(((tid & 255) >> 1) >> 5, tid & 1, ((tid & 255) >> 1) & 31);
    Ms[buf][tid >> 8][mslot] = x.m[0];
    Ms[buf][2][mslot] = x.m[1];
  };
  f16v acc[2][1];  // [quanta block][frame block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][0][e] = 0.f;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    bf8v sf[2][3], mf[1][3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
      sf[0][p] = __builtin_bit_cast(bf8v, Ss[buf][p][xslot((wq >> 5), kh, l32)]);
      sf[1][p] = __builtin_bit_cast(bf8v, Ss[buf][p][xslot((wq >> 5) + 1, kh, l32)]);
      mf[0][p] = __builtin_bit_cast(bf8v, Ms[buf][p][xslot(wm >> 5, kh, l32)]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // smallest products first; the four result blocks interleaved, so that consecutive MFMAs are independent
    constexpr int PS[6] = {2, 0, 1, 1, 0, 0}, PM[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 1; j++) {
          if (DBG == 2)
            asm volatile("" : "+v"(acc[i][j]) : "v"(sf[i][PS[t]]), "v"(mf[j][PM[t]]));
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sf[i][PS[t]], mf[j][PM[t]], acc[i][j], 0, 0, 0);
        }
  };
  // two register sets by tile parity: a tile is requested two compute phases before it is staged (four waves per SIMD hide
  // the rest; four sets as in the four-wave form spill at 128 registers)
  XSet8 x0, x1;
  fetch(0, x0);
  fetch(1, x1);
  stage(0, 0, x0);
  __syncthreads();
  warm = true;
  int buf = 0;
  // (every fetch and every stage unconditional, see the four-wave form)
  const int KL = KT - 1;
  for (int kt = 0; kt < KT; kt += 2) {
    fetch(kt + 2 < KL ? kt + 2 : KL, x0);
    compute(buf);
    stage(kt + 1, buf ^ 1, x1);
    __syncthreads();
    buf ^= 1;
    fetch(kt + 3 < KL ? kt + 3 : KL, x1);
    compute(buf);
    stage(kt + 2 < KL ? kt + 2 : KL, buf ^ 1, x0);
    __syncthreads();
    buf ^= 1;
  }
  // result block layout: element e of a lane = row (e / 4) * 8 + kh * 4 + e % 4 (quantum), column l32 (output frame)
  float* dst = d.dst + (uint64_t)inst * d.dst_inst + (uint64_t)ch * d.dst_ch;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const uint32_t q = q0 + (uint32_t)(wq + i * 32 + (e >> 2) * 8 + kh * 4 + (e & 3));
      if (q >= d.n_quanta) continue;
#pragma unroll
      for (int j = 0; j < 1; j++) {
        float v = acc[i][j][e];
        if (curve_in_lds)
          v = shape_curve_lds(curve_s, d.curve_n, v);
        else if (d.curve)
          v = shape_curve(d.curve, d.curve_n, v);
        store_global(dst + (uint64_t)q * d.dst_q + m0 + wm + j * 32 + l32, v);
      }
    }
}
void launch_qgemm(const QGemmDesc& d, void* stream) {
  dim3 grid((d.n_quanta + BN - 1) / BN, d.M / BM, d.n_inst * (uint32_t)d.nch);
  const char* xdbg = measure_switch("WAA_QGEMM_DEBUG");
  if (d.A16 && xdbg && xdbg[0] >= 'a' && xdbg[0] <= 'c') {
    if (xdbg[0] == 'a') hipLaunchKernelGGL(qgemm_bf16x6_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d);
    if (xdbg[0] == 'b') hipLaunchKernelGGL(qgemm_bf16x6_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, d);
    if (xdbg[0] == 'c') hipLaunchKernelGGL(qgemm_bf16x6_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, d);
  } else if (d.A16 && xdbg && xdbg[0] >= 'd' && xdbg[0] <= 'f') {
    if (xdbg[0] == 'd') hipLaunchKernelGGL(qgemm_bf16x6_w8_kernel<1>, grid, dim3(512), 0, (hipStream_t)stream, d);
    if (xdbg[0] == 'e') hipLaunchKernelGGL(qgemm_bf16x6_w8_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, d);
    if (xdbg[0] == 'f') hipLaunchKernelGGL(qgemm_bf16x6_w8_kernel<3>, grid, dim3(512), 0, (hipStream_t)stream, d);
  } else if (d.A16 && xdbg && xdbg[0] == 'g') {
    hipLaunchKernelGGL(qgemm_bf16x6_w8_kernel<4>, grid, dim3(512), 0, (hipStream_t)stream, d);
  } else if (d.A16 && !measure_switch("WAA_QGEMM_FMA") && !measure_switch("WAA_QGEMM_F32") && !xdbg && measure_switch("WAA_QGEMM_W4"))
    hipLaunchKernelGGL(qgemm_bf16x6_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d);
  else if (d.A16 && !measure_switch("WAA_QGEMM_FMA") && !measure_switch("WAA_QGEMM_F32") && !xdbg)
    hipLaunchKernelGGL(qgemm_bf16x6_w8_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, d);
  else if (measure_switch("WAA_QGEMM_FMA"))  // (switch: the vector-FMA form, same-box A/B with tools/ab_env.py)
    hipLaunchKernelGGL(qgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
  else if (const char* dbg = measure_switch("WAA_QGEMM_DEBUG")) {
    switch (dbg[0]) {
      case '1': hipLaunchKernelGGL(qgemm_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d); break;
      case '2': hipLaunchKernelGGL(qgemm_mfma_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, d); break;
      case '3': hipLaunchKernelGGL(qgemm_mfma_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, d); break;
      default: hipLaunchKernelGGL(qgemm_mfma_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, d); break;
    }
  } else
    hipLaunchKernelGGL(qgemm_mfma_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d);
}
#else
// (product build: the matrix form of the resampling stages is a measured alternative, WAA_OS_MATRIX — DESIGN.md 3.5 — that only the
// measurement build plans; its kernels are not part of libwaa_hip.so)
void launch_qgemm(const QGemmDesc&, void*) {}
#endif  // WAA_MEASURE

// ---------------------------------------------------------------------------------------------------------------
// HRTF FIR.  Workgroup = 4 wavefronts, each renders one (instance, quantum): lanes 0..31 the left ear, 32..63 the
// right ear, four consecutive output frames per lane.  LDS per wavefront: the mono input window xw[O + 128] (O = taps
// rounded up to 4: positions -O .. 127 relative to the quantum) and the interpolated HRIR pair h[2][O].
__global__ __launch_bounds__(256) void hrtf_kernel(const HrtfDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int O = (d.taps + 3) & ~3;
  float* xw = lds + (size_t)wave * (size_t)(3 * O + RQ);
  float* hh = xw + O + RQ;  // [2][O]
  const uint32_t inst = blockIdx.y;
  const uint32_t q = d.q0 + blockIdx.x * 4 + wave;
  const bool in_range = q < d.q1;
  const int32_t* prev = d.prev + (uint64_t)inst * d.prev_stride;
  const uint8_t* code = d.in_code + (uint64_t)inst * d.code_stride;
  const int32_t link = in_range ? load_global(prev + q) : LINK_SKIP;
  const bool process = link != LINK_SKIP;
  float gain = 0.f, corr = 1.f;
  if (process) {
    const HrtfQ* rec = d.table + (uint64_t)(d.rows == 1 ? 0 : inst) * d.per_row + (d.per_row == 1 ? 0 : q);
    const int v0 = load_global(&rec->v[0]), v1 = load_global(&rec->v[1]), v2 = load_global(&rec->v[2]);
    const float w0 = load_global(&rec->w[0]), w1 = load_global(&rec->w[1]), w2 = load_global(&rec->w[2]);
    gain = load_global(&rec->gain);
    // HrirSphere::sample_bilinear: a * u + b * v + c * w per tap, f32, in this order
    const float* pa = d.hrir + (uint64_t)v0 * 2 * d.taps;
    const float* pb = d.hrir + (uint64_t)v1 * 2 * d.taps;
    const float* pc = d.hrir + (uint64_t)v2 * 2 * d.taps;
    for (int i = lane; i < 2 * O; i += 64) {
      const int ear = i >= O, t = ear ? i - O : i;
      float v = 0.f;
      if (t < d.taps) {
        const int o = ear * d.taps + t;
        v = load_global(pa + o) * w0 + load_global(pb + o) * w1 + load_global(pc + o) * w2;
      }
      hh[i] = v;
    }
    // input window: this quantum and, through the prev links, the quanta processed before it
    const float* src = d.in.base + (uint64_t)inst * d.in.inst_stride;
    int32_t qe = (int32_t)q;
    for (int e = 0; e * RQ < O + RQ; e++) {  // element e covers positions [-e * 128, -e * 128 + 128)
      float m0 = 0.f, m1 = 0.f;
      if (qe >= 0) {
        const uint32_t c = load_global(code + qe);
        if (!(c & CODE_SILENT)) {
          const uint64_t f = (uint64_t)qe * RQ;
          m0 = load_global(src + f + lane);
          m1 = load_global(src + f + 64 + lane);
          if ((c & 63u) >= 2) {  // stereo input: mixed down (quantum.rs:387-397), doubled after the convolution
            m0 = 0.5f * (m0 + load_global(src + d.in.ch_stride + f + lane));
            m1 = 0.5f * (m1 + load_global(src + d.in.ch_stride + f + 64 + lane));
          }
        }
        if (e == 0) corr = (c & 63u) >= 2 ? 2.f : 1.f;  // (panner.rs:800-810: by the quantum's count, silent or not)
      }
      const int p0 = O - e * RQ + lane, p1 = p0 + 64;
      if (p0 >= 0) xw[p0] = m0;
      if (p1 >= 0) xw[p1] = m1;
      if (qe >= 0) qe = load_global(prev + qe);  // LINK_FRESH (-1): nothing before it -> zeros
    }
  }
  __syncthreads();
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride;
  const int ear = lane >> 5, n0 = (lane & 31) * 4;
  if (!in_range) return;
  f4v res = {0.f, 0.f, 0.f, 0.f};
  if (process) {
    const float* h = hh + ear * O;
    const int b = O + n0;  // xw index of output frame n0 at tap 0
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    f4v cur = *reinterpret_cast<const f4v*>(xw + b);  // x[b .. b + 3]
    for (int g = 0; g < O / 4; g++) {
      const f4v nxt = *reinterpret_cast<const f4v*>(xw + b - 4 * g - 4);  // x[b - 4g - 4 .. b - 4g - 1]
      const f4v hv = *reinterpret_cast<const f4v*>(h + 4 * g);
      const float win[7] = {nxt.y, nxt.z, nxt.w, cur.x, cur.y, cur.z, cur.w};  // x[b - 4g - 3 .. b - 4g + 3]
      const float ht[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = __builtin_fmaf(ht[t], win[3 + r - t], acc[r]);
      cur = nxt;
    }
    res.x = acc[0] * gain * corr;
    res.y = acc[1] * gain * corr;
    res.z = acc[2] * gain * corr;
    res.w = acc[3] * gain * corr;
  }
  *(WAA_GLOBAL_AS f4v*)(out + (uint64_t)ear * d.out.ch_stride + (uint64_t)q * RQ + n0) = res;
}
// The same FIR with EIGHT output frames per lane and both ears in one packed operation.  One wavefront renders four
// (instance, quantum) units, 16 lanes each.  acc[r] = (left, right) of frame n0 + r; a tap is one v_pk_fma_f32 per frame:
// (hL[j], hR[j]) * (x, x) + acc — the packed pipe is the only way past half of the f32 peak, and with eight frames per
// lane one 16-byte LDS read of new input feeds 64 multiply-adds (the four-frame form: 16, and it was LDS-bound).
// Summation order per output (taps in order, one fma each) is that of hrtf_kernel: the results are bit-identical
// (WAA_HRTF_V1=1 selects the old form; tests compare the two).
// STATIC: one HRIR pair per instance (or for the whole batch) — nothing automated: interpolated once on the host
// (HrtfDesc::hstatic, [row][tap][ear]) and read through the scalar cache — no per-unit interpolation, no LDS traffic for h.
// Directions that change per quantum keep the four-frame form (one wavefront per unit: its prologue interpolates the
// unit's HRIR pair, and four units per wavefront would serialise four of those).
typedef float f2v __attribute__((ext_vector_type(2)));
template <bool STATIC>
__global__ __launch_bounds__(64) void hrtf8_kernel(const HrtfDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int O = (d.taps + 3) & ~3;
  const int per_unit = (STATIC ? O : 3 * O) + RQ;  // xw[O + 128] (+ hh[O][2])
  __shared__ float meta[4][2];                     // gain * corr, process flag
  const uint32_t inst = blockIdx.y;
  const int32_t* prev = d.prev + (uint64_t)inst * d.prev_stride;
  const uint8_t* code = d.in_code + (uint64_t)inst * d.code_stride;
  const float* src = d.in.base + (uint64_t)inst * d.in.inst_stride;
  for (int u = 0; u < 4; u++) {
    float* xw = lds + (size_t)u * (size_t)per_unit;
    float* hh = xw + O + RQ;  // [O][2]
    const uint32_t q = d.q0 + blockIdx.x * 4 + u;
    const int32_t link = q < d.q1 ? load_global(prev + q) : LINK_SKIP;
    const bool process = link != LINK_SKIP;
    float gain = 0.f, corr = 1.f;
    if (process) {
      const HrtfQ* rec = d.table + (uint64_t)(d.rows == 1 ? 0 : inst) * d.per_row + (d.per_row == 1 ? 0 : q);
      gain = load_global(&rec->gain);
      if (!STATIC) {
        const int v0 = load_global(&rec->v[0]), v1 = load_global(&rec->v[1]), v2 = load_global(&rec->v[2]);
        const float w0 = load_global(&rec->w[0]), w1 = load_global(&rec->w[1]), w2 = load_global(&rec->w[2]);
        // HrirSphere::sample_bilinear: a * u + b * v + c * w per tap, f32, in this order
        const float* pa = d.hrir + (uint64_t)v0 * 2 * d.taps;
        const float* pb = d.hrir + (uint64_t)v1 * 2 * d.taps;
        const float* pc = d.hrir + (uint64_t)v2 * 2 * d.taps;
        for (int i = lane; i < 2 * O; i += 64) {
          const int ear = i >= O, t = ear ? i - O : i;
          float v = 0.f;
          if (t < d.taps) {
            const int o = ear * d.taps + t;
            v = load_global(pa + o) * w0 + load_global(pb + o) * w1 + load_global(pc + o) * w2;
          }
          hh[2 * t + ear] = v;
        }
      }
      // input window: this quantum and, through the prev links, the quanta processed before it
      int32_t qe = (int32_t)q;
      for (int e = 0; e * RQ < O + RQ; e++) {  // element e covers positions [-e * 128, -e * 128 + 128)
        float m0 = 0.f, m1 = 0.f;
        if (qe >= 0) {
          const uint32_t c = load_global(code + qe);
          if (!(c & CODE_SILENT)) {
            const uint64_t f = (uint64_t)qe * RQ;
            m0 = load_global(src + f + lane);
            m1 = load_global(src + f + 64 + lane);
            if ((c & 63u) >= 2) {  // stereo input: mixed down (quantum.rs:387-397), doubled after the convolution
              m0 = 0.5f * (m0 + load_global(src + d.in.ch_stride + f + lane));
              m1 = 0.5f * (m1 + load_global(src + d.in.ch_stride + f + 64 + lane));
            }
          }
          if (e == 0) corr = (c & 63u) >= 2 ? 2.f : 1.f;  // (panner.rs:800-810: by the quantum's count, silent or not)
        }
        const int p0 = O - e * RQ + lane, p1 = p0 + 64;
        if (p0 >= 0) xw[p0] = m0;
        if (p1 >= 0) xw[p1] = m1;
        if (qe >= 0) qe = load_global(prev + qe);  // LINK_FRESH (-1): nothing before it -> zeros
      }
    }
    if (lane == 0) {
      meta[u][0] = gain;
      meta[u][1] = process ? corr : 0.f;  // (corr is 1 or 2: 0 marks a skipped unit)
    }
  }
  __syncthreads();
  const int u = lane >> 4, n0 = (lane & 15) * 8;
  const uint32_t q = d.q0 + blockIdx.x * 4 + u;
  if (q >= d.q1) return;
  const float gain = meta[u][0], corr = meta[u][1];
  f2v acc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) acc[r] = f2v{0.f, 0.f};
  if (corr != 0.f) {
    const float* xw = lds + (size_t)u * (size_t)per_unit;
    const float* hh = xw + O + RQ;
    const int b = O + n0;  // xw index of output frame n0 at tap 0
    // window x[b - 4g - 4 .. b - 4g + 7] in three registers of four; the roles (new input, first four, next four) rotate
    // over three steps instead of the registers being copied (12 % of the loop's issue slots were moves)
    f4v r0, r1 = *reinterpret_cast<const f4v*>(xw + b), r2 = *reinterpret_cast<const f4v*>(xw + b + 4);  // x[b .. b + 7]
    auto step = [&](int g, f4v& nx, const f4v& c0, const f4v& c1) __attribute__((always_inline)) {
      nx = *reinterpret_cast<const f4v*>(xw + b - 4 * g - 4);  // x[b - 4g - 4 .. b - 4g - 1]
      const float w[12] = {nx.x, nx.y, nx.z, nx.w, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};  // x[b - 4g - 4 + i]
      f2v h[4];
      if (STATIC) {
        const f2v* hs = reinterpret_cast<const f2v*>(d.hstatic) + (size_t)(d.rows == 1 ? 0 : inst) * O + 4 * g;  // (uniform address: scalar loads)
#pragma unroll
        for (int t = 0; t < 4; t++) h[t] = hs[t];
      } else {
        const f4v h01 = *reinterpret_cast<const f4v*>(hh + 8 * g), h23 = *reinterpret_cast<const f4v*>(hh + 8 * g + 4);
        h[0] = f2v{h01.x, h01.y};
        h[1] = f2v{h01.z, h01.w};
        h[2] = f2v{h23.x, h23.y};
        h[3] = f2v{h23.z, h23.w};
      }
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float x = w[4 - t + r];
          acc[r] = __builtin_elementwise_fma(h[t], f2v{x, x}, acc[r]);
        }
    };
    const int G = O / 4;
    int g = 0;
    for (; g + 2 < G; g += 3) {
      step(g, r0, r1, r2);
      step(g + 1, r2, r0, r1);
      step(g + 2, r1, r2, r0);
    }
    if (g < G) step(g, r0, r1, r2);
    if (g + 1 < G) step(g + 1, r2, r0, r1);
  }
  float* out = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)q * RQ + n0;
  const float gc = gain * corr;
  // (acc * gain * corr in the old form: (acc * gain) * corr — keep that order)
  f4v l0, l1, r0, r1;
  l0.x = acc[0].x * gain * corr; l0.y = acc[1].x * gain * corr; l0.z = acc[2].x * gain * corr; l0.w = acc[3].x * gain * corr;
  l1.x = acc[4].x * gain * corr; l1.y = acc[5].x * gain * corr; l1.z = acc[6].x * gain * corr; l1.w = acc[7].x * gain * corr;
  r0.x = acc[0].y * gain * corr; r0.y = acc[1].y * gain * corr; r0.z = acc[2].y * gain * corr; r0.w = acc[3].y * gain * corr;
  r1.x = acc[4].y * gain * corr; r1.y = acc[5].y * gain * corr; r1.z = acc[6].y * gain * corr; r1.w = acc[7].y * gain * corr;
  (void)gc;
  if (corr == 0.f) l0 = l1 = r0 = r1 = f4v{0.f, 0.f, 0.f, 0.f};
  *(WAA_GLOBAL_AS f4v*)(out) = l0;
  *(WAA_GLOBAL_AS f4v*)(out + 4) = l1;
  *(WAA_GLOBAL_AS f4v*)(out + d.out.ch_stride) = r0;
  *(WAA_GLOBAL_AS f4v*)(out + d.out.ch_stride + 4) = r1;
}
void launch_hrtf(const HrtfDesc& d0, void* stream) {
  HrtfDesc d = d0;
  if (d.q1 == 0) d.q1 = d.n_quanta;  // (the whole render)
  if (d.q1 <= d.q0) return;
  const uint32_t nq_launch = d.q1 - d.q0;
  // one direction for the whole batch: the FIR as partitioned overlap-add on 256-point transforms (waa_hrtf_fft.hip) — unless the
  // launch is a short range of a loop rendered block by block (a run pays four unstored quanta up front)
  if (d.fft_tables && d.per_row == 1 && nq_launch >= 16 && !measure_switch("WAA_HRTF_DIRECT") && !measure_switch("WAA_HRTF_V1") &&
      !measure_switch("WAA_HRTF_V8") && !measure_switch("WAA_HRTF_DYNAMIC")) {
    launch_hrtf_fft(d, stream);
    return;
  }
  if (!measure_switch("WAA_HRTF_V1")) {
    const int O = (d.taps + 3) & ~3;
    const bool stat = d.hstatic != nullptr && !measure_switch("WAA_HRTF_DYNAMIC");
    const size_t lds = (size_t)4 * (size_t)((stat ? O : 3 * O) + RQ) * sizeof(float);
    dim3 grid((nq_launch + 3) / 4, d.n_inst);
    if (stat) {
      hipLaunchKernelGGL(hrtf8_kernel<true>, grid, dim3(64), lds, (hipStream_t)stream, d);
      return;
    }
    if (measure_switch("WAA_HRTF_V8")) {  // (experiment: the eight-frame form with per-unit HRIR pairs in LDS — slower, see above)
      if (lds > 64 * 1024)
        raise_lds_limit(reinterpret_cast<const void*>(hrtf8_kernel<false>));
      hipLaunchKernelGGL(hrtf8_kernel<false>, grid, dim3(64), lds, (hipStream_t)stream, d);
      return;
    }
  }
  const int O = (d.taps + 3) & ~3;
  const size_t lds = (size_t)4 * (size_t)(3 * O + RQ) * sizeof(float);
  if (lds > 64 * 1024)
    raise_lds_limit(reinterpret_cast<const void*>(hrtf_kernel));
  dim3 grid((nq_launch + 3) / 4, d.n_inst);
  hipLaunchKernelGGL(hrtf_kernel, grid, dim3(256), lds, (hipStream_t)stream, d);
}

}  // namespace waa
