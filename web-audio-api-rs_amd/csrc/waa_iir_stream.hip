// waa_iir_stream.hip — IIRFilterNode (src/node/iir_filter.rs:323-405): f64 transposed direct form II with up
// to 19 state variables, constant coefficients shared by all instances.
//
// Same streaming shape as the biquad kernel: one 64-lane wavefront per (instance, channel), 2048-frame tiles,
// LDS transpose so every lane owns 32 consecutive frames.  The recurrence
//     y    = fma(b0, x, s0)                     (iir_filter.rs:381, mul_add)
//     s_i  = b_{i+1} x - a_{i+1} y + s_{i+1}    (iir_filter.rs:389-392, unfused, left to right)
// is affine in the state, s' = M s + (input terms).  Per tile:
//   pass 1: every lane runs its 32 frames from a zero state (lane 0 from the carried state) -> z_l
//   scan  : inclusive Hillis-Steele scan over the 64 lanes of the maps s -> A s + z with the uniform matrices
//           A^(2^k), A = M^32 (host-computed, NS x NS each) -> the true state at the end of every lane
//   pass 2: every lane replays its 32 frames from its true incoming state in the reference's exact operation
//           order; only this pass produces output.
// Inf/NaN anywhere (unstable filters; the reference flushes such outputs to 0, iir_filter.rs:383-385, which is
// not affine) is detected off the critical path and that tile is redone lane after lane, exactly.
// NS (template) = number of state variables; every order 1..19 has its own instantiation.
// Roofline: HBM for small orders (8 B per frame-channel, like the biquad kernel); the f64 vector rate takes over
// around NS >= 8 (about 6 NS^2 + 4 NS * 32 DFMA per lane and tile).  No MFMA: the matrices are tiny and per
// lane-vector, the scan is latency-, not throughput-shaped.
#include <hip/hip_runtime.h>

#include "waa_internal.hpp"
#include "waa_stream_common.hpp"

namespace waa {

namespace {
typedef const __attribute__((address_space(4))) double* cdouble_ptr;  // constant address space: scalar loads

__device__ __forceinline__ double shfl_up_zero(double v, int delta, int lane) {
  const double t = __shfl_up(v, delta, 64);
  return lane >= delta ? t : 0.;
}
}  // namespace

template <int NS>
__global__ __launch_bounds__(64, 2) void iir_stream_kernel(const IirStreamDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const uint32_t wid = blockIdx.x;
  const uint32_t inst = wid / (uint32_t)d.nch;
  const int ch = (int)(wid % (uint32_t)d.nch);
  const int lane = threadIdx.x;
  if (inst >= d.n_inst) return;

  // f64 denormals: flush inputs and outputs (the reference renders under FTZ/DAZ, thread.rs:374-382)
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);

  // normalised coefficients, uniform: cb[0..NS], ca[0..NS] (ca[0] unused)
  // (re-derived from an opaque pointer inside every pass so the scalar loads are not hoisted out of the tile
  // loop: 2 (NS + 1) + 6 NS^2 loop-invariant doubles would not fit the SGPR file and spill)
  const cdouble_ptr coef_c = (cdouble_ptr)(d.coef);
  const cdouble_ptr pw_c = (cdouble_ptr)(d.pow);
  auto opaque_zero = []() __attribute__((always_inline)) {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
  };

  double* st = d.state + ((uint64_t)inst * d.nch + ch) * NS;
  double carry[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) carry[k] = st[k];

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

  // The input products b_k * x[i] do not depend on the recurrence; without a tie the compiler computes all
  // 32 * NS of them up front (and spills).  An empty asm makes frame i's input wait for frame i - AHEAD.
  constexpr int AHEAD = NS <= 4 ? 4 : 2;
  auto lds_sync = []() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); };
  auto fetch_fast = [&](uint32_t tile, float (&dst)[TILE_K]) __attribute__((always_inline)) {
    // (inside the linear prefix the start of a tile is arithmetic; behind it, it comes from the schedule table)
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
  auto stage = [&](const float (&cur)[TILE_K]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      *reinterpret_cast<float4*>(lds + r * LDS_ROW + c) =
          make_float4(cur[j * 4 + 0], cur[j * 4 + 1], cur[j * 4 + 2], cur[j * 4 + 3]);
    }
  };
  float* lds_out = lds + 64 * LDS_ROW;
  auto flush = [&](uint32_t tile) __attribute__((always_inline)) {
    lds_sync();
    float* op = out_base + (uint64_t)tile * TILE;
#pragma unroll
    for (int j = 0; j < NV4; j++) {
      const int r = j * 8 + (lane >> 3), c = (lane & 7) * 4;
      const float4 t = *reinterpret_cast<const float4*>(lds_out + r * LDS_ROW + c);
      *reinterpret_cast<float4*>(op + j * 256 + lane * 4) = t;
    }
    lds_sync();
  };

  // the reference's per-sample update, exact operation order; FLUSH adds the explicit `!is_normal -> 0`
  // (with hardware denormal flushing it only differs for inf/NaN)
  auto exact_pass = [&](const float (&x)[TILE_K], double (&s)[NS], float (&yo)[TILE_K], auto flush_tag)
                        __attribute__((always_inline)) {
    constexpr bool FLUSH = decltype(flush_tag)::value;
    const cdouble_ptr cb = coef_c + opaque_zero(), ca = cb + (NS + 1);
    double yh[TILE_K];
#pragma unroll
    for (int i = 0; i < TILE_K; i++) {
      double xd = (double)x[i];
      if (i >= AHEAD) asm volatile("" : "+v"(xd) : "v"(yh[i - AHEAD]));
      double y = __builtin_fma(cb[0], xd, s[0]);
      if constexpr (FLUSH)
        if (!__builtin_isnormal(y)) y = 0.;
      yh[i] = y;
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const double next = k + 1 < NS ? s[k + 1] : 0.;
        s[k] = (cb[k + 1] * xd - ca[k + 1] * y) + next;
      }
      yo[i] = (float)y;
    }
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
    // pass 1: end state of this lane's 32 frames from a zero state (lane 0: from the carried state)
    double z[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) z[k] = lane == 0 ? carry[k] : 0.;
    const int oz = opaque_zero();
    const cdouble_ptr cb = coef_c + oz, ca = cb + (NS + 1), pw = pw_c + oz;
    double zh[TILE_K];
#pragma unroll
    for (int i = 0; i < TILE_K; i++) {
      double xd = (double)x[i];
      if (i >= AHEAD) asm volatile("" : "+v"(xd) : "v"(zh[i - AHEAD]));
      const double y = __builtin_fma(cb[0], xd, z[0]);
      zh[i] = y;
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const double next = k + 1 < NS ? z[k + 1] : 0.;
        z[k] = __builtin_fma(cb[k + 1], xd, __builtin_fma(-ca[k + 1], y, next));
      }
    }
    // inclusive scan over lanes: z_l <- A^(2^lvl) z_(l - 2^lvl) + z_l
#pragma unroll
    for (int lvl = 0; lvl < 6; lvl++) {
      double q[NS];
#pragma unroll
      for (int k = 0; k < NS; k++) q[k] = shfl_up_zero(z[k], 1 << lvl, lane);
      const cdouble_ptr P = pw + lvl * NS * NS;
#pragma unroll
      for (int r = 0; r < NS; r++) {
        double acc = z[r];
#pragma unroll
        for (int c = 0; c < NS; c++) acc = __builtin_fma(P[r * NS + c], q[c], acc);
        z[r] = acc;
      }
    }
    // state entering this lane = inclusive result of the previous lane
    double s[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const double t = __shfl_up(z[k], 1, 64);
      s[k] = lane == 0 ? carry[k] : t;
    }
    // pass 2: the reference's arithmetic from the true incoming state
    float yo[TILE_K];
    exact_pass(x, s, yo, std::false_type{});
    float badacc = 0.f;  // NaN as soon as one output is inf/NaN
#pragma unroll
    for (int i = 0; i < TILE_K; i++) badacc = __builtin_fmaf(yo[i], 0.f, badacc);
#pragma unroll
    for (int k = 0; k < NS; k++) badacc = __builtin_fmaf((float)s[k], 0.f, badacc);
    lds_sync();
#pragma unroll
    for (int j = 0; j < NV4; j++)
      *reinterpret_cast<float4*>(lds_out + lane * LDS_ROW + j * 4) =
          make_float4(yo[j * 4 + 0], yo[j * 4 + 1], yo[j * 4 + 2], yo[j * 4 + 3]);
    if (__any(badacc != badacc)) {
      // inf/NaN in this tile: the flush to zero is not affine, redo the 64 chunks one after the other
      // (inputs re-read from the staging buffer, results written straight to the output buffer: rare path,
      // keeps its registers out of the common one)
      double cur[NS];
#pragma unroll
      for (int k = 0; k < NS; k++) cur[k] = carry[k];
      float xr[TILE_K];
#pragma unroll
      for (int j = 0; j < NV4; j++) {
        const float4 t4 = *reinterpret_cast<const float4*>(lds + lane * LDS_ROW + j * 4);
        xr[j * 4 + 0] = t4.x;
        xr[j * 4 + 1] = t4.y;
        xr[j * 4 + 2] = t4.z;
        xr[j * 4 + 3] = t4.w;
      }
      for (int l = 0; l < 64; l++) {
        double t[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) t[k] = cur[k];
        float yt[TILE_K];
        exact_pass(xr, t, yt, std::true_type{});
        if (lane == l) {
#pragma unroll
          for (int j = 0; j < NV4; j++)
            *reinterpret_cast<float4*>(lds_out + lane * LDS_ROW + j * 4) =
                make_float4(yt[j * 4 + 0], yt[j * 4 + 1], yt[j * 4 + 2], yt[j * 4 + 3]);
        }
#pragma unroll
        for (int k = 0; k < NS; k++) cur[k] = read_lane(t[k], l);
      }
#pragma unroll
      for (int k = 0; k < NS; k++) carry[k] = cur[k];
    } else {
#pragma unroll
      for (int k = 0; k < NS; k++) carry[k] = read_lane(s[k], 63);
    }
    lds_sync();
  };

  // same software pipeline as the biquad kernel: next tile's input in flight, previous tile's store deferred
  bool pending = false;
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
      process(tile);
      pending = true;
      pending_tile = tile;
      tile++;
      continue;
    }
    uint32_t end = tile + 1;
    if (is_src && end < si.fast_prefix) end = si.fast_prefix < d.tile1 ? si.fast_prefix : d.tile1;  // no table walk
    while (end < d.tile1 && tile_is_fast(end)) end++;
    float nx[TILE_K];
    fetch_fast(tile, nx);
    for (; tile < end; tile++) {
      stage(nx);
      fetch_fast(tile + 1 < end ? tile + 1 : tile, nx);
      if (pending) flush(pending_tile);
      process(tile);
      pending = true;
      pending_tile = tile;
    }
  }
  if (pending) flush(pending_tile);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS; k++) st[k] = carry[k];
  }
}

// Exact fallback for filters whose 32-step transition has large powers (ill-conditioned direct forms: the scan
// would amplify rounding differences beyond the parity tolerance) or overflows: one LANE per (instance, channel)
// stream, frames strictly in order, the reference's arithmetic including the explicit flush.  No cross-lane
// traffic at all; 64-byte runs per lane keep whole sectors in use.  Latency-bound (n_inst * nch / 64 waves).
template <int NS>
__global__ __launch_bounds__(64) void iir_lane_kernel(const IirStreamDesc d) {
  const uint32_t sid = blockIdx.x * 64 + threadIdx.x;
  if (sid >= d.n_inst * (uint32_t)d.nch) return;
  const uint32_t inst = sid / (uint32_t)d.nch;
  const int ch = (int)(sid % (uint32_t)d.nch);
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  const cdouble_ptr cb = (cdouble_ptr)(d.coef), ca = cb + (NS + 1);
  const float* ip = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
  float* op = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride;
  double* st = d.state + ((uint64_t)inst * d.nch + ch) * NS;
  double s[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) s[k] = st[k];
  constexpr int BLK = 16;
  const uint64_t blk0 = (uint64_t)d.tile0 * TILE / BLK, n_blocks = (uint64_t)d.tile1 * TILE / BLK;
  float4 nx[BLK / 4];
#pragma unroll
  for (int j = 0; j < BLK / 4; j++) nx[j] = reinterpret_cast<const float4*>(ip + blk0 * BLK)[j];
  for (uint64_t blk = blk0; blk < n_blocks; blk++) {
    float x[BLK];
#pragma unroll
    for (int j = 0; j < BLK / 4; j++) {
      x[j * 4 + 0] = nx[j].x;
      x[j * 4 + 1] = nx[j].y;
      x[j * 4 + 2] = nx[j].z;
      x[j * 4 + 3] = nx[j].w;
    }
    const uint64_t nb = blk + 1 < n_blocks ? blk + 1 : blk;
#pragma unroll
    for (int j = 0; j < BLK / 4; j++) nx[j] = reinterpret_cast<const float4*>(ip + nb * BLK)[j];
    float y4[BLK];
    double yh[BLK];
    // the input products b_k * x[i] do not depend on the recurrence; without a tie the compiler computes all
    // BLK * NS of them up front (and spills).  An empty asm makes frame i's input wait for frame i - AHEAD.
    constexpr int AHEAD = NS <= 4 ? 4 : 2;
#pragma unroll
    for (int i = 0; i < BLK; i++) {
      double xd = (double)x[i];
      if (i >= AHEAD) asm volatile("" : "+v"(xd) : "v"(yh[i - AHEAD]));
      double y = __builtin_fma(cb[0], xd, s[0]);
      if (!__builtin_isnormal(y)) y = 0.;
      yh[i] = y;
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const double next = k + 1 < NS ? s[k + 1] : 0.;
        s[k] = (cb[k + 1] * xd - ca[k + 1] * y) + next;
      }
      y4[i] = (float)y;
    }
#pragma unroll
    for (int j = 0; j < BLK / 4; j++)
      reinterpret_cast<float4*>(op + blk * BLK)[j] = make_float4(y4[j * 4 + 0], y4[j * 4 + 1], y4[j * 4 + 2], y4[j * 4 + 3]);
  }
#pragma unroll
  for (int k = 0; k < NS; k++) st[k] = s[k];
}

// Exact kernel for higher orders: one DPP row (16 lanes) per stream, lane j owns states j*M .. j*M+M-1, four
// streams per wave.  Per frame every lane forms y from lane 0's s_0 (row broadcast), takes s_{k+1} of its last
// state from its right neighbour (row shift) and updates its own states: the dependent chain per frame is the
// same handful of operations as in the scalar loop, but the NS-wide part runs across lanes instead of in time,
// and 16x more wavefronts are in flight than with one lane per stream.  Arithmetic identical to the reference.
template <int M>
__global__ __launch_bounds__(64) void iir_row_kernel(const IirStreamDesc d) {
  const int lane = threadIdx.x, j = lane & 15;
  const uint32_t n_streams = d.n_inst * (uint32_t)d.nch;
  uint32_t sid = blockIdx.x * 4 + (lane >> 4);
  const bool valid = sid < n_streams;
  if (!valid) sid = n_streams - 1;  // keep the row busy (whole-wave DPP), never store
  const uint32_t inst = sid / (uint32_t)d.nch;
  const int ch = (int)(sid % (uint32_t)d.nch);
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
  const int ns = d.ns;
  const double b0 = d.coef[0];
  double bk[M], ak[M], s[M];
  double* st = d.state + ((uint64_t)inst * d.nch + ch) * ns;
  bool last = false;  // this lane owns s_{ns-1}: its successor is the constant 0 (iir_filter.rs:389-392)
#pragma unroll
  for (int t = 0; t < M; t++) {
    const int k = j * M + t;
    bk[t] = k < ns ? d.coef[k + 1] : 0.;
    ak[t] = k < ns ? d.coef[ns + 1 + k + 1] : 0.;
    s[t] = k < ns ? st[k] : 0.;
    if (t == M - 1) last = k >= ns - 1;
  }
  const float* ip = d.in.sig.base + (uint64_t)inst * d.in.sig.inst_stride + (uint64_t)ch * d.in.sig.ch_stride;
  float* op = d.out.base + (uint64_t)inst * d.out.inst_stride + (uint64_t)ch * d.out.ch_stride;
  constexpr int BLK = 16;
  const uint64_t blk0 = (uint64_t)d.tile0 * TILE / BLK, n_blocks = (uint64_t)d.tile1 * TILE / BLK;
  float4 nx[BLK / 4];
#pragma unroll
  for (int q = 0; q < BLK / 4; q++) nx[q] = reinterpret_cast<const float4*>(ip + blk0 * BLK)[q];
  auto dpp64 = [](double v, auto ctrl, auto bound) __attribute__((always_inline)) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), decltype(ctrl)::value, 0xf, 0xf, decltype(bound)::value);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), decltype(ctrl)::value, 0xf, 0xf, decltype(bound)::value);
    return __hiloint2double(hi, lo);
  };
  for (uint64_t blk = blk0; blk < n_blocks; blk++) {
    float x[BLK];
#pragma unroll
    for (int q = 0; q < BLK / 4; q++) {
      x[q * 4 + 0] = nx[q].x;
      x[q * 4 + 1] = nx[q].y;
      x[q * 4 + 2] = nx[q].z;
      x[q * 4 + 3] = nx[q].w;
    }
    const uint64_t nb = blk + 1 < n_blocks ? blk + 1 : blk;
#pragma unroll
    for (int q = 0; q < BLK / 4; q++) nx[q] = reinterpret_cast<const float4*>(ip + nb * BLK)[q];
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < BLK; i++) {
      const double xd = (double)x[i];
      const double s0 = dpp64(s[0], std::integral_constant<int, 0x150>{}, std::false_type{});   // row_share:0
      double y = __builtin_fma(b0, xd, s0);
      if (!__builtin_isnormal(y)) y = 0.;
      double nxt = dpp64(s[0], std::integral_constant<int, 0x101>{}, std::true_type{});         // row_shl:1, lane 15 <- 0
      if (last) nxt = 0.;
#pragma unroll
      for (int t = 0; t < M; t++) {
        const double next = t + 1 < M ? s[t + 1] : nxt;
        s[t] = (bk[t] * xd - ak[t] * y) + next;
      }
      mine = i == j ? (float)y : mine;
    }
    if (valid) op[blk * BLK + j] = mine;
  }
  if (valid) {
#pragma unroll
    for (int t = 0; t < M; t++)
      if (j * M + t < ns) st[j * M + t] = s[t];
  }
}

int iir_padded_states(int n_states) {  // every order has its own instantiation: no zero-padded states
  if (n_states < 1) return 1;  // a pure gain b0 still runs as a one-state filter with zero coefficients
  return n_states <= 19 ? n_states : -1;
}

namespace {
template <int NS>
void launch_ns(const IirStreamDesc& d, hipStream_t s) {
  if (d.exact == 1) {
    const dim3 grid((d.n_inst * (uint32_t)d.nch + 63) / 64), block(64);
    hipLaunchKernelGGL((iir_lane_kernel<NS>), grid, block, 0, s, d);
  } else {
    const dim3 grid(d.n_inst * (uint32_t)d.nch), block(64);
    const size_t lds = 2 * 64 * LDS_ROW * sizeof(float);
    hipLaunchKernelGGL((iir_stream_kernel<NS>), grid, block, lds, s, d);
  }
}
template <int NS>
void dispatch_ns(const IirStreamDesc& d, hipStream_t s) {
  if constexpr (NS >= 1) {
    if (d.ns == NS)
      launch_ns<NS>(d, s);
    else
      dispatch_ns<NS - 1>(d, s);
  }
}
}  // namespace

void launch_iir_stream(const IirStreamDesc& d, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (d.exact == 2) {
    const dim3 grid((d.n_inst * (uint32_t)d.nch + 3) / 4), block(64);
    if (d.ns <= 16)
      hipLaunchKernelGGL((iir_row_kernel<1>), grid, block, 0, s, d);
    else
      hipLaunchKernelGGL((iir_row_kernel<2>), grid, block, 0, s, d);
    return;
  }
  dispatch_ns<19>(d, s);
}

}  // namespace waa
