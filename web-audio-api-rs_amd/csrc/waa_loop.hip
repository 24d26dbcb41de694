// waa_loop.hip — feedback loops.  A cycle through a DelayNode (graph.rs:323-487: the delay's writer half is the
// cycle breaker) makes quantum q of every member depend on quantum q-1 of the loop, so the node-major engine
// cannot render it one node at a time.  The members of the loop are instead rendered quantum by quantum, in the
// reference's processing order, by one wavefront per instance (instances are independent, so no cross-wave
// synchronisation exists): lane l renders frames l and l + 64 of the 128-frame quantum for up to two channels.
//   * outputs of members rendered earlier in the same quantum are handed over through LDS,
//   * every member also writes its output signal to HBM (consumers outside the loop, delay lines),
//   * a delay line is the writer's mixed input in absolute time; the reader gathers from it exactly like
//     waa_delay.hip, with the in-cycle clamp of delay.rs:693-701; it reads through agent-scope loads because the
//     same wave wrote those lines moments ago,
//   * a BiquadFilter inside a loop renders its 128 frames over the whole wavefront (zero-state + scan + exact-order pass,
//     as in waa_dyn.hip); per-frame coefficient sets and quanta with inf / NaN: serially on one lane per channel.
// Latency-bound by construction (about 1-2 us per quantum and member); throughput comes from the batch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "waa_internal.hpp"

namespace waa {

namespace {
__device__ __forceinline__ float param_val(const ParamRef& p, uint32_t inst, uint32_t q, uint64_t frame) {
  if (p.mode == 0) return load_global(p.base + inst);
  if (p.mode == 1) return load_global(p.base + (uint64_t)inst * p.stride + q);
  return load_global(p.base + (uint64_t)inst * p.stride + frame);
}
__device__ __forceinline__ float coherent_load(const float* p) {
  return __int_as_float(__hip_atomic_load((const WAA_GLOBAL_AS int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
// quantum.rs:285-505 for the layouts a loop may carry (mono / stereo)
__device__ __forceinline__ void mix12(float (&u)[2][2], int from, int to, int interp) {
  if (from == to) return;
  if (from == 1 && to == 2) {
#pragma unroll
    for (int e = 0; e < 2; e++) u[1][e] = interp == 1 ? 0.f : u[0][e];
  } else if (from == 2 && to == 1) {
    if (interp != 1) {
#pragma unroll
      for (int e = 0; e < 2; e++) u[0][e] = 0.5f * (u[0][e] + u[1][e]);
    }
  }
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
// One workgroup = one wavefront: LDS hand-offs between lanes need program order only (see waa_dyn.hip).
__device__ __forceinline__ void lds_sync() { __builtin_amdgcn_wave_barrier(); }
}  // namespace

__global__ __launch_bounds__(64) void loop_kernel(const LoopDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cur = lds;                                        // [n_items][2][128] outputs of this quantum
  float* scratch = lds + (size_t)d.n_items * 2 * RQ;       // [2][128]
  double* bq_state = reinterpret_cast<double*>(scratch + 2 * RQ);  // [n_items][2][4]
  // the item descriptors once into LDS (through the kernel argument they are dependent vector loads, see waa_dyn.hip)
  LoopItem* items_s = reinterpret_cast<LoopItem*>(bq_state + (size_t)d.n_items * 8);
  const uint32_t inst = blockIdx.x;
  const int lane = threadIdx.x;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.items);
    uint32_t* dst = reinterpret_cast<uint32_t*>(items_s);
    const int words = d.n_items * (int)(sizeof(LoopItem) / 4);
    for (int i = lane; i < words; i += 64) dst[i] = load_global(src + i);
  }
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);  // f64 denormals flushed (FTZ/DAZ render scope)
  for (int i = lane; i < d.n_items * 8; i += 64) bq_state[i] = 0.;
  lds_sync();

  for (uint32_t q = 0; q < d.n_quanta; q++) {
    const uint64_t f0 = (uint64_t)q * RQ;
    for (int it = 0; it < d.n_items; it++) {
      const LoopItem& li = items_s[it];
      float v[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
      if (li.kind != LI_DELAY_R) {
        // ---- inputs: every incoming edge mixed to the computed channel count, summed in edge order
        for (int k = 0; k < li.n_in; k++) {
          float u[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
          const int nc = li.in_nch[k];
          if (li.in_item[k] >= 0) {
            const float* src = cur + (size_t)li.in_item[k] * 2 * RQ;
#pragma unroll
            for (int c = 0; c < 2; c++)
              if (c < nc) {
                u[c][0] = src[c * RQ + lane];
                u[c][1] = src[c * RQ + 64 + lane];
              }
          } else {
            const SignalRef& sg = li.in_sig[k];
            const float* src = sg.base + (uint64_t)inst * sg.inst_stride + f0;
#pragma unroll
            for (int c = 0; c < 2; c++)
              if (c < nc) {
                u[c][0] = load_global(src + (uint64_t)c * sg.ch_stride + lane);
                u[c][1] = load_global(src + (uint64_t)c * sg.ch_stride + 64 + lane);
              }
          }
          mix12(u, nc, li.nch_in, li.interp);
#pragma unroll
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < 2; e++) v[c][e] = k == 0 ? u[c][e] : v[c][e] + u[c][e];
        }
      }
      if (li.kind == LI_NODE) {
        const OpDesc& op = li.op;
        switch (op.kind) {
          case OP_GAIN: {
            if (op.p0.mode == 2) {
#pragma unroll
              for (int e = 0; e < 2; e++) {
                const float g = load_global(op.p0.base + (uint64_t)inst * op.p0.stride + f0 + e * 64 + lane);
                v[0][e] *= g;
                v[1][e] *= g;
              }
            } else {  // gain.rs:163-179
              const float g = param_val(op.p0, inst, q, 0);
              const bool mute = fabsf(g) <= 1e-6f, pass = fabsf(1.f - g) <= 1e-6f;
#pragma unroll
              for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 2; e++) v[c][e] = mute ? 0.f : (pass ? v[c][e] : v[c][e] * g);
            }
            break;
          }
          case OP_WAVESHAPER: {
            const float* curve = reinterpret_cast<const float*>(op.ptr0);
#pragma unroll
            for (int c = 0; c < 2; c++)
              if (c < op.nch_in) {
#pragma unroll
                for (int e = 0; e < 2; e++) v[c][e] = shape(curve, op.i0, v[c][e]);
              }
            break;
          }
          case OP_STEREO_PAN: {  // stereo_panner.rs:218-317, k-rate pan (gains from the host tables)
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const float pan = param_val(op.p0, inst, q, 0);
              const float gl = param_val(op.p1, inst, q, 0), gr = param_val(op.p2, inst, q, 0);
              if (op.nch_in == 1) {
                const float in = v[0][e];
                v[0][e] = in * gl;
                v[1][e] = in * gr;
              } else {
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
            break;
          }
          case OP_BIQUAD: {  // biquad_filter.rs:857-897, 128 frames in order on one lane per channel
            lds_sync();
#pragma unroll
            for (int c = 0; c < 2; c++) {
              scratch[c * RQ + lane] = v[c][0];
              scratch[c * RQ + 64 + lane] = v[c][1];
            }
            lds_sync();
            // one coefficient set for the quantum: the recurrence over the whole wavefront (32 lanes per channel, 4 frames
            // per lane: zero-state response, 5-step scan with the powers of M^4, the reference's order from the true
            // incoming state) as in waa_dyn.hip / waa_biquad_stream.hip; per-frame sets and quanta with inf / NaN: serially
            bool scan_done = false;
            if (op.i0 != 2 && !d.no_scan) {
              const int ch = lane >> 5, l = lane & 31;
              const bool act = ch < op.nch_in;
              const double* s = bq_state + ((size_t)it * 2 + ch) * 4;
              float* row = scratch + ch * RQ;
              const double* cf = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0 + (op.i0 == 1 ? (uint64_t)q * 5 : 0);
              const double b0 = load_global(cf), b1 = load_global(cf + 1), b2 = load_global(cf + 2), a1 = load_global(cf + 3),
                           a2 = load_global(cf + 4);
              const f4v xv = *reinterpret_cast<const f4v*>(row + 4 * l);
              const double x0 = (double)xv.x, x1 = (double)xv.y, x2 = (double)xv.z, x3 = (double)xv.w;
              double xm1 = __shfl_up(x3, 1, 32), xm2 = __shfl_up(x2, 1, 32);
              const double cy1 = s[2], cy2 = s[3];
              if (l == 0) {
                xm1 = s[0];
                xm2 = s[1];
              }
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
              double p1 = xm1, p2 = xm2;
              const double xs[4] = {x0, x1, x2, x3};
              float yo[4];
              bool bad = false;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const double x = xs[e];
                double y = b0 * x + b1 * p1 + b2 * p2 - a1 * y1 - a2 * y2;
                bad |= !(__builtin_fabs(y) <= 1.7976931348623157e308);  // inf / NaN
                if (!__builtin_isnormal(y)) y = 0.;
                p2 = p1;
                p1 = x;
                y2 = y1;
                y1 = y;
                yo[e] = (float)y;
              }
              bad = act && bad;
              if (!__any(bad)) {
                scan_done = true;
                if (act) {
                  *reinterpret_cast<f4v*>(row + 4 * l) = f4v{yo[0], yo[1], yo[2], yo[3]};
                  if (l == 31) {
                    double* sw = bq_state + ((size_t)it * 2 + ch) * 4;
                    sw[0] = p1;
                    sw[1] = p2;
                    sw[2] = y1;
                    sw[3] = y2;
                  }
                }
              }
            }
            if (!scan_done && lane < op.nch_in) {
              double* st = bq_state + ((size_t)it * 2 + lane) * 4;
              double x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
              const double* cbase = reinterpret_cast<const double*>(op.ptr0) + (uint64_t)inst * op.u0;
              float* row = scratch + lane * RQ;
              for (int i = 0; i < RQ; i++) {
                const double* cf = op.i0 == 0 ? cbase : op.i0 == 1 ? cbase + (uint64_t)q * 5 : cbase + (f0 + i) * 5;
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
              st[0] = x1;
              st[1] = x2;
              st[2] = y1;
              st[3] = y2;
            }
            lds_sync();
#pragma unroll
            for (int c = 0; c < 2; c++) {
              v[c][0] = scratch[c * RQ + lane];
              v[c][1] = scratch[c * RQ + 64 + lane];
            }
            break;
          }
          default: break;  // pass-through (analyser, waveshaper without a curve)
        }
      } else if (li.kind == LI_DELAY_R) {
        // delay.rs:515-745 in absolute time (see waa_delay.hip)
        __syncthreads();  // the writer's stores of this quantum (if it rendered first) have reached L2
        const SignalRef& hs = items_s[li.writer_item].out;
        const OpDesc& op = li.op;
        int64_t pf0 = 0;
        float k0 = 0.f;
        if (op.p0.mode != 2) {
          double dv = (double)param_val(op.p0, inst, q, 0);
          if (li.in_cycle) dv = fmax(dv, d.quantum_duration);
          const double position = 0. - dv * d.sample_rate;
          const double fl = floor(position);
          pf0 = (int64_t)fl;
          k0 = (float)(position - fl);
        }
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
            if (li.in_cycle) dv = fmax(dv, d.quantum_duration);
            const double position = (double)i - dv * d.sample_rate;
            const double fl = floor(position);
            pf = (int64_t)fl;
            k = (float)(position - fl);
          }
          const int64_t prev = (int64_t)f0 + pf;
          // frame 128 of the newest block wraps to the OLDEST ring block (delay.rs:622-626); a reader that renders
          // before its writer sees, in the slot of the current quantum, the block written ring-capacity quanta ago
          int64_t next = prev + 1;
          if (!li.in_cycle && pf == RQ - 1) next = ((int64_t)q - li.num_quanta) * RQ;
          if (li.in_cycle && next >= (int64_t)f0) next -= ((int64_t)li.num_quanta + 1) * RQ;
          const float* hb = hs.base + (uint64_t)inst * hs.inst_stride;
#pragma unroll
          for (int c = 0; c < 2; c++)
            if (c < li.nch_out) {
              const float ps = prev >= 0 ? coherent_load(hb + (uint64_t)c * hs.ch_stride + prev) : 0.f;
              const float nsv = next >= 0 ? coherent_load(hb + (uint64_t)c * hs.ch_stride + next) : 0.f;
              v[c][e] = __builtin_fmaf(1.f - k, ps, k * nsv);
            }
        }
      }
      // ---- hand over (LDS) and publish (HBM)
      const int nco = li.kind == LI_DELAY_W ? li.nch_in : li.nch_out;
      float* dst = cur + (size_t)it * 2 * RQ;
      float* gout = li.out.base + (uint64_t)inst * li.out.inst_stride + f0;
#pragma unroll
      for (int c = 0; c < 2; c++)
        if (c < nco) {
          dst[c * RQ + lane] = v[c][0];
          dst[c * RQ + 64 + lane] = v[c][1];
          store_global(gout + (uint64_t)c * li.out.ch_stride + lane, v[c][0]);
          store_global(gout + (uint64_t)c * li.out.ch_stride + 64 + lane, v[c][1]);
        }
      lds_sync();
    }
  }
}

void launch_loop(const LoopDesc& d, void* stream) {
  const size_t lds = ((size_t)d.n_items * 2 * RQ + 2 * RQ) * sizeof(float) + (size_t)d.n_items * 8 * sizeof(double) +
                     (size_t)d.n_items * sizeof(LoopItem);
  if (lds > 64 * 1024)
    raise_lds_limit(reinterpret_cast<const void*>(loop_kernel));
  LoopDesc dd = d;
  dd.no_scan = measure_switch("WAA_DYN_NO_SCAN") ? 1u : 0u;
  hipLaunchKernelGGL(loop_kernel, dim3(d.n_inst), dim3(64), lds, (hipStream_t)stream, dd);
}

}  // namespace waa
