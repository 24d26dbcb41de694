// waa_panner.hip — per-frame geometry of the equal-power PannerNode when the AudioListener is automated at audio rate
// (panner.rs:720-779 a_rate_params, :830-897; spatial.rs:205-299).  With a single-valued listener the reference
// evaluates the geometry once per render quantum (from the FIRST value of every param, panner.rs:844-845) and the host
// does that (waa_plan.cpp, with the libm the reference resolves to).  As soon as one of the nine listener params is a
// 128-value slice, every frame has its own source / listener vectors: distance gain (f64), cone gain, azimuth in the
// listener's frame of reference, and from the wrapped azimuth the equal-power gains of the mono and the stereo law.
// This kernel evaluates them for all frames of all instances (one table row per instance, or one row for the batch when
// nothing depends on the instance) into per-frame tables the panner op of the chain kernels reads like a-rate params.
// f32 vector algebra as in the reference; acosf / sinf / cosf are the device's (<= 1-2 ulp from the host libm).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "waa_internal.hpp"

namespace waa {

namespace {
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 normalized(V3 a) { return scale(a, 1.f / sqrtf(sqlen(a))); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
constexpr float PI_F = 3.14159265358979323846f;

// spatial.rs:205-270
__device__ void azimuth_elevation(V3 sp, V3 lp, V3 lf, V3 lu, float* az) {
  *az = 0.f;
  const V3 rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return;
  const V3 sl = normalized(rel);
  const V3 right = cross(lf, lu);
  if (sqlen(right) == 0.f) return;
  const V3 rn = normalized(right), fn = normalized(lf), up = cross(rn, fn);
  const float up_proj = dot(sl, up);
  const V3 ps = sub(sl, scale(up, up_proj));
  if (sqlen(ps) == 0.f) return;
  const V3 psn = normalized(ps);
  float azimuth = 180.f * acosf(dot(psn, rn)) / PI_F;
  if (dot(psn, fn) < 0.f) azimuth = 360.f - azimuth;
  if (azimuth >= 0.f && azimuth <= 270.f)
    azimuth = 90.f - azimuth;
  else
    azimuth = 450.f - azimuth;
  *az = azimuth;
}
// spatial.rs:278-299
__device__ float spatial_angle(V3 sp, V3 so, V3 lp) {
  if (sqlen(so) == 0.f) return 0.f;
  const V3 son = normalized(so), rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return 0.f;
  const V3 sl = normalized(rel);
  return fabsf(180.f * acosf(dot(sl, son)) / PI_F);
}
}  // namespace

__global__ __launch_bounds__(256) void panner_geom_kernel(const PannerGeomDesc d) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (uint64_t)d.rows * d.n_frames) return;
  const uint32_t inst = (uint32_t)(idx / d.n_frames);
  const uint64_t frame = idx % d.n_frames;
  const uint32_t q = (uint32_t)(frame / RQ);
  // a single-valued listener: one geometry per quantum from the first value of every param (panner.rs:833-846)
  bool single = d.single[(uint64_t)inst * d.single_stride + q] != 0;
#pragma unroll
  for (int k = 0; k < 9; k++)
    if (d.dev_len[k]) single = single && d.dev_len[k][(uint64_t)inst * d.single_stride + q] == 1;
  const uint64_t f = single ? (uint64_t)q * RQ : frame;
  float v[15];
#pragma unroll
  for (int k = 0; k < 15; k++) {
    const ParamRef& p = d.p[k];
    v[k] = p.mode == 0 ? p.base[inst] : p.mode == 1 ? p.base[(uint64_t)inst * p.stride + q] : p.base[(uint64_t)inst * p.stride + f];
  }
  const V3 sp{v[0], v[1], v[2]}, so{v[3], v[4], v[5]}, lp{v[6], v[7], v[8]}, lf{v[9], v[10], v[11]}, lu{v[12], v[13], v[14]};
  // panner.rs:955-985 dist_gain (f64)
  float dg;
  {
    const double distance = (double)sqrtf(sqlen(sub(sp, lp)));
    const double ref = d.ref_distance, maxd = d.max_distance, roll = d.rolloff;
    double g;
    if (d.distance_model == 0) {  // linear
      const double rf = roll < 0. ? 0. : roll > 1. ? 1. : roll;
      const double lo = fmin(ref, maxd), hi = fmax(ref, maxd);
      const double dc = distance < lo ? lo : distance > hi ? hi : distance;
      g = 1. - rf * (dc - lo) / (hi - lo);
    } else if (d.distance_model == 1) {  // inverse
      const double rf = fmax(roll, 0.);
      g = distance > 0. ? ref / (ref + rf * (fmax(ref, distance) - ref)) : 1.;
    } else {
      const double rf = fmax(roll, 0.);
      g = pow(fmax(distance, ref) / ref, -rf);
    }
    dg = (float)g;
  }
  // panner.rs:927-953 cone_gain
  float cg;
  {
    const float in = fabsf(d.cone_inner) / 2.f, out = fabsf(d.cone_outer) / 2.f;
    if (in >= 180.f && out >= 180.f) {
      cg = 1.f;
    } else {
      const float a = spatial_angle(sp, so, lp);
      if (a < in)
        cg = 1.f;
      else if (a >= out)
        cg = d.cone_outer_gain;
      else {
        const float x = (a - in) / (out - in);
        cg = (1.f - x) + d.cone_outer_gain * x;
      }
    }
  }
  float a;
  azimuth_elevation(sp, lp, lf, lu, &a);
  // panner.rs:996-1004 wrap to [-90, 90]
  a = a < -180.f ? -180.f : a > 180.f ? 180.f : a;
  if (a < -90.f)
    a = -180.f - a;
  else if (a > 90.f)
    a = 180.f - a;
  const float xm = (a + 90.f) / 180.f;                                  // mono law, panner.rs:1006-1009
  const float xs = a <= 0.f ? (a + 90.f) / 90.f : a / 90.f;             // stereo law, panner.rs:1030-1040
  const uint64_t o = (uint64_t)inst * d.n_frames + frame;
  d.az[o] = a;
  d.gl_mono[o] = cosf(xm * PI_F / 2.f);
  d.gr_mono[o] = sinf(xm * PI_F / 2.f);
  d.gl_stereo[o] = cosf(xs * PI_F / 2.f);
  d.gr_stereo[o] = sinf(xs * PI_F / 2.f);
  d.dg[o] = dg;
  d.cg[o] = cg;
}

void launch_panner_geom(const PannerGeomDesc& d, void* stream) {
  const uint64_t total = (uint64_t)d.rows * d.n_frames;
  hipLaunchKernelGGL(panner_geom_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d);
}

}  // namespace waa
