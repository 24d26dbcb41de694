// waa_schedule.cpp — what the reference computes on the control side or once per quantum and what therefore stays
// on the host: biquad coefficient formulas, panner geometry, the AudioBufferSourceNode
// playhead state machine (its output is a per-quantum record table, the interpolation arithmetic runs on the
// device), per-quantum AudioParam values.
#include <atomic>
#include <cstring>

#include "waa_host.hpp"

namespace waa {
namespace host {

// ---- almost crate 0.2 (see oracle header for the provenance note) ----------------------
const double ALMOST_TOL = 1.4901161193847656e-8;
bool almost_zero(double a) { return std::fabs(a) < ALMOST_TOL; }
bool almost_equal(double a, double b) {
  if (a == b) return true;
  if (!std::isfinite(a) || !std::isfinite(b)) return false;
  double scale = std::fmax(std::fabs(a), std::fabs(b));
  if (scale < 1.0) scale = 1.0;
  return std::fabs(a - b) < scale * ALMOST_TOL;
}

// ---- biquad coefficients (biquad_filter.rs:28-373), f64 ---------------------------------
Coefs norm(double b0, double b1, double b2, double a0, double a1, double a2) {
  double s = 1. / a0;
  return {b0 * s, b1 * s, b2 * s, a1 * s, a2 * s};
}
Coefs biquad_coefs(int type, double sample_rate, double f0, double gain, double q) {
  const double PI = 3.14159265358979323846;
  double nyq = sample_rate / 2.;
  double f = f0 / nyq;
  f = f < 0. ? 0. : f > 1. ? 1. : f;
  const Coefs wire{1., 0., 0., 0., 0.}, zero{0., 0., 0., 0., 0.};
  double A = std::pow(10., gain / 40.);
  switch (type) {
    case WAA_BIQUAD_LOWPASS: {
      if (f == 1.) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * std::pow(10., q / 20.)), cw = std::cos(w0), be = (1. - cw) / 2.;
      return norm(be, 2. * be, be, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_HIGHPASS: {
      if (f == 1.) return zero;
      if (f == 0.) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * std::pow(10., q / 20.)), cw = std::cos(w0), be = (1. + cw) / 2.;
      return norm(be, -2. * be, be, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_BANDPASS: {
      if (!(f > 0. && f < 1.)) return zero;
      if (!(q > 0.)) return wire;
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(al, 0., -al, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_NOTCH: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return zero;
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1., -2. * cw, 1., 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_ALLPASS: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return Coefs{-1., 0., 0., 0., 0.};
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1. - al, -2. * cw, 1. + al, 1. + al, -2. * cw, 1. - al);
    }
    case WAA_BIQUAD_PEAKING: {
      if (!(f > 0. && f < 1.)) return wire;
      if (!(q > 0.)) return Coefs{A * A, 0., 0., 0., 0.};
      double w0 = PI * f, al = std::sin(w0) / (2. * q), cw = std::cos(w0);
      return norm(1. + al * A, -2. * cw, 1. - al * A, 1. + al / A, -2. * cw, 1. - al / A);
    }
    case WAA_BIQUAD_LOWSHELF: {
      if (f == 1.) return Coefs{A * A, 0., 0., 0., 0.};
      if (f == 0.) return wire;
      double w0 = PI * f, cw = std::cos(w0), as = std::sin(w0) / 2. * 1.41421356237309504880;
      double k = 2. * as * std::sqrt(A), ap = A + 1., am = A - 1.;
      return norm(A * (ap - am * cw + k), 2. * A * (am - ap * cw), A * (ap - am * cw - k), ap + am * cw + k,
                  -2. * (am + ap * cw), ap + am * cw - k);
    }
    default: {
      if (f == 1.) return wire;
      if (!(f > 0.)) return Coefs{A * A, 0., 0., 0., 0.};
      double w0 = PI * f, cw = std::cos(w0), as = std::sin(w0) / 2. * 1.41421356237309504880;
      double k = 2. * as * std::sqrt(A), ap = A + 1., am = A - 1.;
      return norm(A * (ap + am * cw + k), -2. * A * (am + ap * cw), A * (ap + am * cw - k), ap - am * cw + k,
                  2. * (am - ap * cw), ap - am * cw - k);
    }
  }
}
float computed_freq(float freq, float detune) { return detune != 0.f ? freq * exp2f(detune / 1200.f) : freq; }

// ---- spatial geometry (spatial.rs:205-299, panner.rs:927-985), f32 as the reference -----
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
V3 normalized(V3 a) { return scale(a, 1.f / std::sqrt(sqlen(a))); }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

void azimuth_elevation(V3 sp, V3 lp, V3 lf, V3 lu, float* az, float* el) {
  *az = 0.f;
  *el = 0.f;
  V3 rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return;
  V3 sl = normalized(rel);
  V3 right = cross(lf, lu);
  if (sqlen(right) == 0.f) return;
  V3 rn = normalized(right), fn = normalized(lf), up = cross(rn, fn);
  float elevation = 90.f - 180.f * acosf(dot(sl, up)) / PI_F;
  if (elevation > 90.f)
    elevation = 180.f - elevation;
  else if (elevation < -90.f)
    elevation = -180.f - elevation;
  float up_proj = dot(sl, up);
  V3 ps = sub(sl, scale(up, up_proj));
  *el = elevation;
  if (sqlen(ps) == 0.f) return;
  V3 psn = normalized(ps);
  float azimuth = 180.f * acosf(dot(psn, rn)) / PI_F;
  if (dot(psn, fn) < 0.f) azimuth = 360.f - azimuth;
  if (azimuth >= 0.f && azimuth <= 270.f)
    azimuth = 90.f - azimuth;
  else
    azimuth = 450.f - azimuth;
  *az = azimuth;
}
float spatial_angle(V3 sp, V3 so, V3 lp) {
  if (sqlen(so) == 0.f) return 0.f;
  V3 son = normalized(so), rel = sub(sp, lp);
  if (sqlen(rel) <= FLT_MIN) return 0.f;
  V3 sl = normalized(rel);
  return std::fabs(180.f * acosf(dot(sl, son)) / PI_F);
}
float cone_gain(const waa_node_desc& d, V3 sp, V3 so, V3 lp) {
  float in = (float)std::fabs(d.d[3]) / 2.f, out = (float)std::fabs(d.d[4]) / 2.f;
  if (in >= 180.f && out >= 180.f) return 1.f;
  float cog = (float)d.d[5];
  float a = spatial_angle(sp, so, lp);
  if (a < in) return 1.f;
  if (a >= out) return cog;
  float x = (a - in) / (out - in);
  return (1.f - x) + cog * x;
}
float dist_gain(const waa_node_desc& d, V3 sp, V3 lp) {
  double distance = (double)std::sqrt(sqlen(sub(sp, lp)));
  double ref = d.d[0], maxd = d.d[1], roll = d.d[2], g;
  switch (d.i[1]) {
    case WAA_DISTANCE_LINEAR: {
      double rf = roll < 0. ? 0. : roll > 1. ? 1. : roll;
      double lo = std::fmin(ref, maxd), hi = std::fmax(ref, maxd);
      double dc = distance < lo ? lo : distance > hi ? hi : distance;
      g = 1. - rf * (dc - lo) / (hi - lo);
      break;
    }
    case WAA_DISTANCE_INVERSE: {
      double rf = std::fmax(roll, 0.);
      g = distance > 0. ? ref / (ref + rf * (std::fmax(ref, distance) - ref)) : 1.;
      break;
    }
    default: {
      double rf = std::fmax(roll, 0.);
      g = std::pow(std::fmax(distance, ref) / ref, -rf);
    }
  }
  return (float)g;
}

// ---- AudioBufferSourceNode scheduler: port of audio_buffer_source.rs:422-845 -------------

static void schedule_source_impl(const waa_batch* b, const SourceSched& cfg, uint64_t frames, float buf_sr, bool has_buffer,
                                 const std::vector<float>& rate_q, const std::vector<float>& detune_q, SchedOut* out, bool allow_runs);

// (measurement build, WAA_SCHED_VERIFY=1: every schedule is replayed twice — with the steady-state runs and frame by frame —
// and the tables compared bit for bit; tests/test_schedule_runs.py reads the mismatch count)
static std::atomic<uint64_t> g_sched_checked{0}, g_sched_mismatches{0}, g_sched_run_frames{0};
extern "C" uint64_t waa_debug_sched_run_frames(void) { return g_sched_run_frames.load(); }  // frames rendered by steady-state runs so far
extern "C" uint64_t waa_debug_sched_verify(uint64_t* checked) {
  if (checked) *checked = g_sched_checked.load();
  return g_sched_mismatches.load();
}

void schedule_source(const waa_batch* b, const SourceSched& cfg, uint64_t frames, float buf_sr, bool has_buffer,
                     const std::vector<float>& rate_q, const std::vector<float>& detune_q, SchedOut* out) {
  const bool no_runs = measure_switch("WAA_SCHED_NO_RUNS") != nullptr;
  schedule_source_impl(b, cfg, frames, buf_sr, has_buffer, rate_q, detune_q, out, !no_runs);
  if (measure_switch("WAA_SCHED_VERIFY")) {
    SchedOut ref;
    schedule_source_impl(b, cfg, frames, buf_sr, has_buffer, rate_q, detune_q, &ref, false);
    bool same = ref.qrec.size() == out->qrec.size() && ref.slow.size() == out->slow.size() && ref.tile_fast == out->tile_fast &&
                ref.any_slow == out->any_slow && ref.ended_quantum == out->ended_quantum && ref.ended_at_unload == out->ended_at_unload;
    for (size_t i = 0; same && i < ref.qrec.size(); i++) same = ref.qrec[i].start == out->qrec[i].start && ref.qrec[i].mode == out->qrec[i].mode;
    for (size_t i = 0; same && i < ref.slow.size(); i++)
      same = ref.slow[i].prev == out->slow[i].prev && ref.slow[i].next == out->slow[i].next &&
             std::memcmp(&ref.slow[i].k, &out->slow[i].k, sizeof(double)) == 0;
    g_sched_checked++;
    if (!same) g_sched_mismatches++;
  }
}

std::shared_ptr<SchedOut> schedule_source_cached(waa_batch* b, uint32_t node, const SchedKey& key, const SourceSched& cfg, uint64_t frames,
                                                 float buf_sr, bool has_buffer, const std::vector<float>& rate_q,
                                                 const std::vector<float>& detune_q) {
  auto k = std::make_pair(node, key);
  auto it = b->sched_cache.find(k);
  if (it != b->sched_cache.end()) return it->second;
  auto so = std::make_shared<SchedOut>();
  schedule_source(b, cfg, frames, buf_sr, has_buffer, rate_q, detune_q, so.get());
  b->sched_cache.emplace(k, so);
  return so;
}

static void schedule_source_impl(const waa_batch* b, const SourceSched& cfg, uint64_t frames, float buf_sr, bool has_buffer,
                                 const std::vector<float>& rate_q, const std::vector<float>& detune_q, SchedOut* out, bool allow_runs) {
  const uint32_t nq = b->n_quanta;
  out->qrec.assign(nq, QRec{0, Q_SILENT, 0});
  out->slow.clear();
  out->any_slow = false;
  double start_time = cfg.start, stop_time = cfg.stop, offset = cfg.offset, duration = cfg.duration;
  const double sample_rate = (double)b->sr;
  const double dt = 1. / sample_rate;
  const double block_duration = dt * (double)RQ;
  const double buffer_duration = has_buffer ? (double)frames / (double)buf_sr : 0.;
  // clamp_loop_boundaries (:401-417)
  double loop_start = cfg.loop_start, loop_end = cfg.loop_end;
  if (has_buffer) {
    if (loop_start < 0.)
      loop_start = 0.;
    else if (loop_start > buffer_duration)
      loop_start = buffer_duration;
    if (loop_end <= 0. || loop_end > buffer_duration) loop_end = buffer_duration;
  }
  const bool is_looping = cfg.looping != 0;
  const double sampling_ratio = has_buffer ? (double)buf_sr / sample_rate : 1.;
  double buffer_time = 0., elapsed = 0.;
  bool started = false, entered_loop = false, is_aligned = false, ended = false;
  auto ensure_slow = [&]() {
    if (!out->any_slow) {
      out->slow.assign((size_t)nq * RQ, SlowRec{-1, -1, 0.});
      out->any_slow = true;
    }
  };
  for (uint32_t q = 0; q < nq; q++) {
    if (ended) break;
    const double block_time = (double)((uint64_t)q * RQ) / sample_rate;  // thread.rs:360
    const double next_block_time = block_time + block_duration;
    if (!has_buffer && start_time != DBL_MAX) {  // ended (start with a null buffer, :446-451)
      out->ended_quantum = q;
      break;
    }
    if (start_time >= next_block_time) {
      if (stop_time <= next_block_time) {  // stopped before it started (:457-461)
        out->ended_quantum = q;
        break;
      }
      continue;
    }
    if (!has_buffer) continue;
    const double detune = (double)detune_q[detune_q.size() == 1 ? 0 : q];
    const double playback_rate = (double)rate_q[rate_q.size() == 1 ? 0 : q];
    const double cpr = playback_rate * std::exp2(detune / 1200.);
    double actual_loop_start = 0., actual_loop_end = 0.;
    if (!started && start_time < block_time) start_time = block_time;
    if (start_time == block_time && offset == 0.) is_aligned = true;
    if (sampling_ratio != 1. || cpr != 1.) is_aligned = false;
    if (loop_start != 0. || loop_end != buffer_duration) is_aligned = false;
    if (buffer_time + block_duration > duration || block_time + block_duration > stop_time) is_aligned = false;
    if (is_aligned) {
      if (start_time == block_time) started = true;
      const int64_t start_index = (int64_t)std::llround(buffer_time * sample_rate);
      out->qrec[q] = QRec{start_index, is_looping ? (uint32_t)Q_FAST_LOOP : (uint32_t)Q_FAST, 0};
      if (buffer_time + block_duration > buffer_duration) {
        // did the playhead wrap inside this block?  (:568-607)
        int loop_point_index = -1;
        if (is_looping) {
          uint64_t si = (uint64_t)start_index, off = 0;
          for (int index = 0; index < RQ; index++) {
            uint64_t bi = si + (uint64_t)index - off;
            if (bi >= frames) {
              loop_point_index = index;
              si = 0;
              off = (uint64_t)index;
            }
          }
        }
        if (loop_point_index >= 0)
          buffer_time = std::fmod((double)(RQ - loop_point_index) / sample_rate, buffer_duration);
        else
          buffer_time += block_duration;
      } else {
        buffer_time += block_duration;
      }
      elapsed += block_duration;
    } else {
      if (is_looping) {
        if (loop_start >= 0. && loop_end > 0. && loop_start < loop_end) {
          actual_loop_start = loop_start;
          actual_loop_end = loop_end;
        } else {
          actual_loop_start = 0.;
          actual_loop_end = buffer_duration;
        }
      } else {
        entered_loop = false;
      }
      ensure_slow();
      out->qrec[q] = QRec{0, Q_SLOW, 0};
      SlowRec* rec = &out->slow[(size_t)q * RQ];
      // The steady state of the playhead (round 5: one replay of a 10 s slow-track schedule was 9 ms of a plan, 20 ns per
      // frame of tolerance tests that cannot fire): once the source has started (and entered its loop), while every frame of
      // this block lies before stop_time, the played duration cannot reach `duration` inside the block, and buffer_time sits
      // strictly inside (zone_lo, zone_hi) — farther than the `almost` tolerance from the loop points and from zero — every
      // snapping test and both wrap loops below are no-ops by construction.  The frame then only computes its record and
      // advances the clock with the SAME two additions in the same order (buffer_time += time_incr; elapsed += |time_incr|):
      // the table is bit-identical to the frame-by-frame form (tests/test_reference_kat.py, C5 every-instance: bit-exact).
      const double time_incr_blk = dt * cpr;
      const bool blk_before_stop = block_time + (double)(RQ - 1) * dt < stop_time && std::isfinite(time_incr_blk);
      const bool blk_duration_safe =
          elapsed + (double)(RQ + 1) * std::fabs(time_incr_blk) < duration - 4. * ALMOST_TOL * std::fmax(std::fabs(duration), 1.) || duration == DBL_MAX;
      const double zone_margin = 4. * ALMOST_TOL * std::fmax(1., std::fmax(std::fabs(actual_loop_end), std::fabs(buffer_duration)));
      const double zone_lo = (is_looping ? actual_loop_start : 0.) + zone_margin;
      const double zone_hi = is_looping ? actual_loop_end - zone_margin : DBL_MAX;
      const bool blk_fast = allow_runs && blk_before_stop && blk_duration_safe && (!is_looping || actual_loop_start >= 0.);
      for (int i = 0; i < RQ; i++) {
        if (blk_fast && started && (!is_looping || entered_loop) && buffer_time > zone_lo && buffer_time < zone_hi && buffer_time < buffer_duration) {
          // a run of steady-state frames: first the clock (the only serial part: one dependent addition per frame), then
          // the records of the run
          double bts[RQ + 1], els[RQ + 1];
          int n_run = 0;
          {
            double bt = buffer_time, el = elapsed;
            const double ainc = std::fabs(time_incr_blk);
            while (i + n_run < RQ && bt > zone_lo && bt < zone_hi && bt < buffer_duration) {
              bts[n_run] = bt;
              els[n_run] = el;
              n_run++;
              bt += time_incr_blk;
              el += ainc;
            }
            bts[n_run] = bt;
            els[n_run] = el;
          }
          int done = 0;
          for (; done < n_run; done++) {
            const double playhead = bts[done] * sampling_ratio * sample_rate;  // position = buffer_time * ratio; playhead = position * sr
            const double pf = std::floor(playhead);
            const uint64_t prev = (uint64_t)pf;
            if (!(prev + 1 < frames)) break;  // (the buffer's last frame: the general form below knows the end rules)
            rec[i + done] = SlowRec{(int32_t)prev, (int32_t)(prev + 1), playhead - pf};
          }
          if (done > 0) {
            g_sched_run_frames.fetch_add((uint64_t)done, std::memory_order_relaxed);
            buffer_time = bts[done];
            elapsed = els[done];
            i += done - 1;
            continue;
          }
        }
        rec[i] = SlowRec{-1, -1, 0.};
        const double current_time = block_time + (double)i * dt;
        if (!started && almost_equal(current_time, start_time)) start_time = current_time;
        if (almost_equal(elapsed, duration)) elapsed = duration;
        if (current_time < start_time || current_time >= stop_time || elapsed >= duration) continue;
        if (!started) {
          const double delta = current_time - start_time;
          offset += delta * cpr;
          offset = std::fmin(std::fmax(offset, 0.), buffer_duration);
          if (is_looping && cpr >= 0. && offset > actual_loop_end) offset = actual_loop_end;
          if (is_looping && cpr < 0. && offset < actual_loop_start) offset = actual_loop_start;
          buffer_time = offset;
          elapsed = std::fabs(delta * cpr);
          started = true;
        }
        if (is_looping) {
          if (almost_equal(buffer_time, actual_loop_end)) buffer_time = actual_loop_end;
          if (almost_equal(buffer_time, actual_loop_start)) buffer_time = actual_loop_start;
          if (!entered_loop) {
            if (offset < actual_loop_end && buffer_time >= actual_loop_start) entered_loop = true;
            if (offset >= actual_loop_end && buffer_time < actual_loop_end) entered_loop = true;
          }
          if (entered_loop) {
            while (buffer_time >= actual_loop_end) buffer_time -= actual_loop_end - actual_loop_start;
            while (buffer_time < actual_loop_start) buffer_time += actual_loop_end - actual_loop_start;
          }
        }
        if (almost_zero(buffer_time)) buffer_time = 0.;
        if (buffer_time >= 0. && buffer_time < buffer_duration) {
          const double position = buffer_time * sampling_ratio;
          const double playhead = position * sample_rate;
          const double pf = std::floor(playhead);
          const uint64_t prev = (uint64_t)pf;
          const double k = playhead - pf;
          if (prev < frames) {
            SlowRec r;
            r.prev = (int32_t)prev;
            r.k = k;
            if (prev + 1 < frames) {
              r.next = (int32_t)(prev + 1);
            } else if (is_looping) {
              if (playback_rate >= 0.) {
                const double sp = actual_loop_start * sample_rate;
                const uint64_t si = (std::floor(sp) == sp) ? (uint64_t)sp : (uint64_t)sp + 1;
                r.next = si < frames ? (int32_t)si : -1;
              } else {
                // the reference reads buffer_channel[end_index] (audio_buffer_source.rs:795-797): one past the
                // end when loop_end == duration (a Rust panic there); defined as a 0 sample here
                const double ep = actual_loop_end * sample_rate;
                const uint64_t ei = (uint64_t)ep;
                r.next = ei < frames ? (int32_t)ei : -1;
              }
            } else {
              r.next = (almost_equal(k, 1.) || prev == 0) ? -1 : -2;
            }
            rec[i] = r;
          }
        }
        const double time_incr = dt * cpr;
        buffer_time += time_incr;
        elapsed += std::fabs(time_incr);
      }
    }
    if (next_block_time >= stop_time || elapsed >= duration ||
        (!is_looping && ((cpr > 0. && buffer_time >= buffer_duration) || (cpr < 0. && buffer_time < 0.)))) {
      ended = true;
      out->ended_quantum = q;
    }
  }
  if (out->ended_quantum < 0) {  // before_drop (:872-878) with the time after the last quantum (thread.rs:399-401)
    const double end_time = (double)((uint64_t)nq * RQ) / sample_rate;
    out->ended_at_unload = end_time >= start_time || end_time >= stop_time;
  }
  // per tile: can the whole tile be fetched as one aligned contiguous run?
  out->tile_fast.assign(b->n_tiles, 0);
  for (uint32_t t = 0; t < b->n_tiles; t++) {
    bool ok = true;
    int64_t s0 = 0;
    for (int k = 0; k < QUANTA_PER_TILE && ok; k++) {
      uint32_t q = t * QUANTA_PER_TILE + k;
      if (q >= nq) {
        ok = false;
        break;
      }
      const QRec& r = out->qrec[q];
      if (r.mode != Q_FAST && r.mode != Q_FAST_LOOP) ok = false;
      if (k == 0) s0 = r.start;
      if (r.start != s0 + (int64_t)k * RQ) ok = false;
    }
    if (ok && (s0 % 4 != 0 || (uint64_t)s0 + TILE > frames)) ok = false;
    out->tile_fast[t] = ok ? 1 : 0;
  }
}

// values of one param for one instance, one value per quantum (first sample of a len-128 slice)
std::vector<float> param_per_quantum(const waa_batch* b, const ParamStore& p, uint32_t inst, bool* varies) {
  std::vector<float> v(1, p.fix(p.cst[inst]));
  bool any = false;
  for (auto& blk : p.blocks)
    if (blk.inst == WAA_ALL_INSTANCES || blk.inst == inst) any = true;
  if (!any) {
    if (varies) *varies = false;
    return v;
  }
  v.assign(b->n_quanta, p.fix(p.cst[inst]));
  for (auto& blk : p.blocks) {
    if (!(blk.inst == WAA_ALL_INSTANCES || blk.inst == inst)) continue;
    for (uint32_t k = 0; k < blk.nq; k++) {
      uint64_t q = blk.q0 + k;
      if (q < b->n_quanta) v[q] = p.fix(blk.v[(size_t)k * blk.vpq]);
    }
  }
  if (varies) *varies = true;
  return v;
}

}  // namespace host
}  // namespace waa
