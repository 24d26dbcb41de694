// waa_internal.hpp — structures shared by the host planner (waa_plan.cpp) and the gfx950 kernels
// (waa_kernels.hip).  Not part of the public C ABI (include/waa_hip.h).
#pragma once
#include <cstdint>
#include <cstdlib>

#include "waa_conv_noise.hpp"

namespace waa {

// Switches read from the environment.  Four are part of the library's behaviour and documented:
// WAA_POISON_ALLOC, WAA_STRICT_CHANNEL_COUNTS, WAA_OSC_EXACT, WAA_IIR_EXACT — plain getenv.  Everything else is an A/B,
// debugging or measurement aid (several produce wrong results by construction): measure_switch() is getenv only in
// libwaa_hip_measure.so (built from the same sources with -DWAA_MEASURE; the tests and tools that flip such switches load
// it) and a constant null pointer in the product library, where the corresponding code folds away.
inline const char* measure_switch(const char* name) {
#ifdef WAA_MEASURE
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Kernels that need more than 64 KB of dynamic LDS: the limit of a kernel function is a per-DEVICE attribute, and several
// devices may be driven from threads of one process (waa_render_sharded): raised once per (device, function), under a lock.
// Defined in waa_echo.hip (host code).
void raise_lds_limit(const void* kernel);

constexpr int RQ = 128;            // render quantum (reference src/lib.rs:18)
constexpr int TILE_K = 32;         // frames per lane in the transposed (recurrence) layout
constexpr int TILE = 64 * TILE_K;  // frames per wave-tile = 2048 = 16 render quanta
constexpr int QUANTA_PER_TILE = TILE / RQ;
constexpr int MAX_OPS = 12;
constexpr int MAX_INPUTS = 4;
constexpr int MAX_CH = 32;         // channels per signal (MAX_CHANNELS, src/lib.rs:21): per-channel filter state is laid out for it
constexpr int STATE_STRIDE = MAX_CH * 4;  // doubles of biquad state per instance per op

// An AudioParam's values as seen by a kernel (AudioParamValues::get, processor.rs:186-229).
//   mode 0: one value per instance              base[inst]
//   mode 1: one value per (instance, quantum)   base[inst*stride + q]           (len-1 slices)
//   mode 2: one value per (instance, frame)     base[inst*stride + frame]       (len-128 slices)
struct ParamRef {
  const float* base;
  uint64_t stride;
  int32_t mode;
  int32_t pad;
};

// A materialised signal: [instance][channel][frames_padded] f32, frames_padded % TILE == 0.
struct SignalRef {
  float* base;
  uint64_t inst_stride;
  uint64_t ch_stride;
  int32_t nch;
  int32_t pad;
};

// ---- AudioBufferSourceNode on device ---------------------------------------------------
// Per-quantum playback record produced by the host-side scheduler (port of the state machine
// in audio_buffer_source.rs:422-845; "host-side scheduling stays on the host").
enum : uint32_t { Q_SILENT = 0, Q_FAST = 1, Q_SLOW = 2, Q_FAST_LOOP = 3 };
struct QRec {
  int64_t start;   // FAST: buffer index of the first frame of this quantum
  uint32_t mode;   // Q_*
  uint32_t pad;
};
// Per-frame playback info of the slow track (audio_buffer_source.rs:631-751), already resolved:
struct SlowRec {
  int32_t prev;  // -1: output 0
  int32_t next;  // >=0: index of next sample; -1: next_sample = 0; -2: extrapolate 2*prev - prevprev
  double k;
};
struct SrcSchedule {
  const QRec* qrec;        // [n_quanta]
  const SlowRec* slow;     // [n_quanta*128] or null if no slow quantum
  const uint8_t* tile_fast;  // [n_tiles] 1 = all 16 quanta FAST, contiguous, in range, 16B aligned
};
struct SrcInst {
  const float* base;     // channel 0 of this instance's AudioBuffer
  uint64_t ch_stride;    // floats between channels
  uint64_t frames;
  uint32_t sched;        // index into the schedule table
  uint32_t aligned;      // base and ch_stride multiples of 4 floats
  SrcSchedule sc;        // copy of the schedule entry: saves the kernels one dependent load per wave
  // tiles [0, fast_prefix) are fast AND one linear run: tile t starts at buffer frame linear_start + t * 2048.  The
  // streaming kernels then need neither the per-tile flag nor the per-tile start from the tables — two dependent
  // loads in front of every prefetch (tools/stream_probe.hip: the same copy without them runs at 6.0 TB/s, with
  // them the kernel's copy floor was 4.9 TB/s)
  int64_t linear_start;
  uint32_t fast_prefix;
  uint32_t linear_all;   // 1: every quantum of the render (also those of a partial last tile) belongs to that run
};

// ---- chain kernel description ----------------------------------------------------------
// IN_DELAYED: the output of a DelayNode outside a feedback loop with a constant / k-rate delayTime, read by its
// consumer straight from the delay line (`sig` = the DelayNode's mixed input in absolute time) — the gather of
// waa_delay.hip folded into the consumer's input stage instead of a pass through HBM of its own
enum : int32_t { IN_SILENT = 0, IN_SIGNAL = 1, IN_SOURCE = 2, IN_CONSTANT = 3, IN_DELAYED = 4 };
struct InputRef {
  int32_t kind;
  int32_t nch;            // channels this input delivers
  SignalRef sig;          // IN_SIGNAL; IN_DELAYED: the delay line
  const SrcInst* src;     // IN_SOURCE: [n_inst]
  const SrcSchedule* sched;  // IN_SOURCE: schedule table
  ParamRef offset;        // IN_CONSTANT: the offset param; IN_DELAYED: delayTime (mode 0 or 1), clamped by the host
  const int64_t* active;  // IN_CONSTANT: [n_inst][2] first/last+1 active frame
  double sample_rate;     // IN_DELAYED
  int32_t num_quanta;     // IN_DELAYED: ring capacity - 1 (delay.rs:300-302)
  int32_t feedback;       // IN_DELAYED (host side): the line is written later in the same block of a block-scheduled loop
  uint64_t valid;         // IN_DELAYED: frames of the delay line that may be read (zeros beyond: a source's buffer read in place)
  ParamRef gain;          // has_gain: a GainNode folded into this edge (applied before the mix to the receiver's count)
  int32_t has_gain;
  uint32_t fast_tiles;    // IN_SOURCE (host side): tiles [0, fast_tiles) are fast and one linear run for every instance
  float delay_lo, delay_hi;  // IN_DELAYED (host side): range of the delay over all instances, in frames (hi < lo: not one
                             // host-known value per instance)
};

enum : int32_t {
  OP_GAIN = 1,        // gain.rs:143-199
  OP_BIQUAD = 2,      // biquad_filter.rs:764-899
  OP_WAVESHAPER = 3,  // waveshaper.rs:555-573
  OP_STEREO_PAN = 4,  // stereo_panner.rs:218-317
  OP_PANNER = 5,      // panner.rs:830-897, 988-1057 (equal power)
  OP_MIX = 6,         // quantum.rs:285-505
  OP_IIR = 7,         // iir_filter.rs:323-405 (always cut out of the chain into the streaming IIR kernel)
  OP_PARAM_ADD = 8    // param.rs:737-795: audio-rate input of an AudioParam + intrinsic value, clamped
};
struct OpDesc {
  int32_t kind;
  int32_t nch_in;
  int32_t nch_out;
  int32_t i1;        // PARAM_ADD: max value (float bits)
  int32_t i2;        // PARAM_ADD: default value (float bits)
  int32_t i0;        // MIX: interpretation; WAVESHAPER: curve length; BIQUAD: coef mode (0 per inst, 1 per quantum, 2 per frame); IIR: padded state count, negative = exact lane kernel; PARAM_ADD: min value (float bits)
  ParamRef p0;       // GAIN: gain; STEREO_PAN: pan; PANNER: azimuth (wrapped)
  ParamRef p1;       // STEREO_PAN / PANNER: gain_l
  ParamRef p2;       // STEREO_PAN / PANNER: gain_r
  ParamRef p3;       // PANNER: dist_gain*cone_gain factors (dist), p4 (cone)
  ParamRef p4;
  const void* ptr0;  // WAVESHAPER: curve; BIQUAD: coefficients (double[5] per inst or per inst*quantum); IIR: coef block
  void* ptr1;        // BIQUAD: state double[nch][4] per instance; IIR: state double[nch][ns] per instance
  const void* ptr2;  // IIR: matrix powers
  uint64_t u0;       // BIQUAD: coefficient stride per instance (in doubles)
};

struct ChainDesc {
  int32_t n_inputs;
  int32_t in_nch;          // channel count the summed input is mixed to
  int32_t in_interp;
  int32_t n_ops;
  InputRef in[MAX_INPUTS];
  OpDesc ops[MAX_OPS];
  SignalRef out;
  uint32_t n_inst;
  uint32_t n_tiles;
  uint32_t n_quanta;
  uint32_t xcd_remap;      // set by the launcher: workgroup -> XCD-contiguous ranges of the (instance, sub-tile) order
  uint32_t tile0, tile1;   // tiles [tile0, tile1) of this launch (block-scheduled feedback loops); full range otherwise
  int32_t lds_curve_op;    // set by the launcher: op whose WaveShaper curve is staged in LDS (-1: none)
  int32_t tile_major;      // set by the launcher: wave index -> (sub-tile, instance) instead of (instance, sub-tile)
  // Block-scheduled feedback loop made of this ONE launch (the echo loop  line = source + gain * delayed(line)): > 0 = the
  // loop's block in 256-frame sub-tiles; the kernel is then PERSISTENT — one workgroup per instance walks the blocks of
  // [tile0, tile1) in order, its four wavefronts share a block's sub-tiles, a workgroup barrier between blocks — instead of
  // one launch per block (47 launches of 250 MB each for a 10 s render, at 2 TB/s).  0: a plain launch.
  uint32_t persist_block;
  uint32_t pad;
};

// ---- streaming biquad kernel (the C2 / T1 hot shape) --------------------------------------
// input (source or signal, no input mixing) -> Biquad(per-instance constant coefficients)
// -> up to 2 k-rate-constant gains -> output; one wavefront per (instance, channel).
struct BiquadStreamDesc {
  InputRef in;
  const double* coefs;   // [n_inst][coef_stride]: 5 per instance, 5 per (instance, quantum) when vary == 1, the lane-major
                         // per-frame table of biquad_coef_kernel when vary == 2 (coef_stride 0: one table for all instances)
  uint64_t coef_stride;  // doubles per instance
  int32_t vary;          // 1: per-quantum coefficients (k-rate automation); 2: per-frame coefficients (a-rate params);
                         // 3: per-frame coefficients shared by all instances, with the precomputed `hp` table
  int32_t pad0;
  double* state;         // [n_inst][STATE_STRIDE]
  const double* hp;      // vary == 3: [n_tiles][HP_WORDS][64], see BiquadHpDesc
  ParamRef gain[2];      // mode 0 only
  int32_t n_gain;
  int32_t nch;
  SignalRef out;
  uint32_t n_inst;
  uint32_t n_tiles;
  uint32_t n_quanta;
  uint32_t dup_out;      // mono stream whose only consumer is the speakers up-mix 1 -> 2 of the destination: both channels written here
  uint32_t tile0, tile1;
};
void launch_biquad_stream(const BiquadStreamDesc& d, void* stream);
// waa_biquad_scan.hip: the same shape parallel in time (one unit = one tile of one stream, chained scan over the tiles)
constexpr int BIQUAD_SCAN_PW = 88;   // doubles per instance: A, A^2, A^4, A^8, A^16, A^64 (A = M^32), then A^0 .. A^15
struct BiquadScanCtl {
  uint32_t* counter;      // eight unit counters of the batch, 16 words apart (zeroed at the start of every render)
  uint32_t counter_base[8];  // their values when this launch starts (host bookkeeping)
  double* payload;        // [n_streams][n_tiles][8] self-validating words (filled with 0xFF bytes at the start of every render)
  const double* pw;       // [n_inst][BIQUAD_SCAN_PW]
  uint32_t* error;        // set when a bounded spin gave up
};
void launch_biquad_scan_powers(const double* coefs, uint64_t coef_stride, double* pw, uint32_t n_inst, void* stream);
void launch_biquad_scan(const BiquadStreamDesc& d, const BiquadScanCtl& ctl, uint32_t* issued, void* stream);

// ---- a-rate Biquad with one coefficient table for all instances, one LANE per stream (waa_biquad_lanes.hip) ----
constexpr int BIQUAD_HT_WORDS = 2 * TILE + 8;  // doubles per tile digest: 2048 x (Hx, Hy), Hm1 (2), Hm2 (2), P (4)
struct BiquadLanesDesc {
  InputRef in;            // IN_SIGNAL (valid = 0: the padded length, else that many frames, zeros beyond) or IN_SOURCE
  const double* coefs;    // frame-major [n_tiles * 2048][5], shared by all instances (biquad_coef_kernel, lane_major = 0)
  double* ht;             // [n_tiles][BIQUAD_HT_WORDS] tile digests
  double* z;              // [n_tiles][n_streams][2] zero-state end states of the tiles (pass A)
  double* sin;            // [n_tiles][n_streams][2] y state in front of every tile (chain)
  double* state;          // [n_inst][STATE_STRIDE]
  ParamRef gain[2];       // mode 0 only
  int32_t n_gain, nch;
  SignalRef out;
  uint32_t n_inst, n_tiles, n_quanta, tile0, tile1;
  uint32_t fast_tiles;    // tiles [0, fast_tiles) are one linear run of the input for EVERY stream (host-known)
  uint32_t lt0, lt1;      // (set by the launcher: the tiles of one kernel launch)
  uint32_t debug, pad;    // WAA_LANES_DEBUG (measurement aid: 1 = the rows of a chunk contiguous in memory, results meaningless)
};
void launch_biquad_tile_digest(const BiquadLanesDesc& d, void* stream);
void launch_biquad_lanes(const BiquadLanesDesc& d, void* stream);

// ---- streaming IIR kernel (IIRFilterNode, iir_filter.rs:323-405) -----------------------------
// input (source or signal) -> transposed direct form II with ns state variables -> output; coefficients are
// shared by all instances (they are constructor arguments of the node).
struct IirStreamDesc {
  InputRef in;
  const double* coef;  // [2][ns+1]: normalised feedforward b[0..ns], then feedback a[0..ns], zero padded
  const double* pow;   // [6][ns][ns]: M^(32 * 2^k), M = zero-input state transition
  double* state;       // [n_inst][nch][ns]
  int32_t ns;          // state count (1..19)
  int32_t nch;
  SignalRef out;
  uint32_t n_inst;
  uint32_t n_tiles;
  uint32_t n_quanta;
  uint32_t exact;      // 0: scan kernel; exact kernels (input must be IN_SIGNAL): 1 lane per stream, 2 DPP row per stream
  uint32_t tile0, tile1;
};
int iir_padded_states(int n_states);  // kernel state count for a filter with n_states state variables
void launch_iir_stream(const IirStreamDesc& d, void* stream);

// ---- ConvolverNode (convolver.rs:343-490 + fft-convolver), node-major overlap-save ------------
// out[co] = sum over terms t with t.out_ch == co of  IR[t.ir_ch] * in[t.in_ch]   (linear convolution)
struct ConvTerm {
  int32_t in_ch, ir_ch, out_ch, pad;
};
struct Cplx {
  float re, im;
};
struct ConvDesc {
  SignalRef in, out;
  int32_t n_terms;
  int32_t cin, cout;
  int32_t n;        // complex FFT size = 2 * block
  int32_t block;    // partition size B
  int32_t parts;    // P = ceil(trimmed IR length / B)
  int32_t nb;       // number of output blocks = ceil(frames / B)
  int32_t fft3;     // n == 16384: the three-register-pass transforms (waa_conv3.hip); the spectra are then stored in THEIR
                    // position order (decided when the impulse response's spectra are computed, i.e. at plan time)
  ConvTerm terms[4];
  const Cplx* H;    // [ir_nch][P][n] spectra of the IR partitions (bit-reversed order)
  Cplx* X;          // [n_pairs][cin][nb][n] input spectra of instance pairs (a + i b)
  Cplx* Y;          // [n_pairs][cout][nb][n]
  const Cplx* tw;   // [n] exp(-2 pi i t / n)
  const float* ir;  // [ir_nch][ir_len] scaled, trimmed IR (device) for the IR-spectrum pass
  uint64_t ir_len;
  uint64_t frames;  // valid frames per channel in `out` (padded length)
  uint64_t in_valid;  // frames of `in` that may be read (zeros beyond): the padded length, or the AudioBuffer's length when
                      // `in` is a view of a source's buffer (the source renders its buffer unchanged from frame 0)
  uint32_t n_inst, n_pairs;
  int32_t ir_nch, pad1;
  // blocks [kb0, kb1) of this launch: the full range, or the blocks of one tile range of a block-scheduled feedback loop
  // (the spectra of earlier blocks stay in X between the launches)
  int32_t kb0, kb1;
  int32_t mac_grid_order, pad2;  // (set by the launcher, WAA_CONV_MAC_GRID_ORDER: the product kernel's workgroups in grid order — A/B aid)
  // A BiquadFilterNode with constant coefficients directly in front of the convolver, rendered by the forward transform's
  // input stage (fft3 only; waa_conv3.hip): `in` is then the BIQUAD's input and its filtered signal never crosses HBM.
  const double* pre_coefs;   // [n_inst][pre_coef_stride]: b0 b1 b2 a1 a2 (null: no filter)
  uint64_t pre_coef_stride;
  double* pre_state;         // [n_inst][STATE_STRIDE]: x1 x2 y1 y2 per channel (initial state in, final state out)
};
struct AnalyserDesc {
  SignalRef sig;         // the analyser's (passthrough) signal
  uint32_t n_inst;       // one workgroup per instance
  int32_t fft_size;
  uint64_t frames_written;  // n_quanta * 128
  float smoothing;
  float min_db, max_db;  // byte variant (analysis.rs:371-401)
  int32_t pad;
  const float* window;   // [fft_size] Blackman
  const Cplx* tw;        // [fft_size/2] exp(-2 pi i t / (fft_size/2))
  const Cplx* tw_full;   // [fft_size/2] exp(-2 pi i k / fft_size)
  const float* prev;     // [fft_size/2] previous smoothed spectrum (zeros: one pull per render)
  float* db_out;         // [n_inst][fft_size/2]  20 log10 of the smoothed magnitudes (analysis.rs:365-368)
  uint8_t* byte_out;     // [n_inst][fft_size/2]
  float* time_out;       // [n_inst][fft_size]
  // dynamic-count plans: the per-quantum codes of `sig` (count | silent) — the down-mix to mono follows the count of EVERY
  // quantum (analyser.rs:277-280 mixes the quantum it is handed), not the signal's static width; null: static plans
  const uint8_t* code;
  uint64_t code_stride;
};
void launch_analyser(const AnalyserDesc& d, void* stream);
constexpr int DIRECT_MAX_TAPS = 128;  // trimmed IRs up to this length use the direct FIR kernel
void launch_conv_direct(const ConvDesc& d, void* stream);
void launch_conv_ir_spectra(const ConvDesc& d, void* stream);
void launch_conv_forward(const ConvDesc& d, void* stream);
void launch_conv_mac(const ConvDesc& d, void* stream);
void launch_conv_inverse(const ConvDesc& d, void* stream);
// waa_conv3.hip (n == 16384 and d.fft3)
void launch_conv3_ir_spectra(const ConvDesc& d, void* stream);
void launch_conv3_forward(const ConvDesc& d, void* stream);
void launch_conv3_inverse(const ConvDesc& d, void* stream);

// ---- DelayNode outside a cycle (delay.rs:428-745) as a gather from its materialised input ---------
struct DelayDesc {
  SignalRef in, out;
  ParamRef delay;        // delayTime, clamped to [0, maxDelayTime] by the host
  double sample_rate;
  uint64_t frames;       // padded frames per channel (multiple of TILE)
  int32_t num_quanta;    // ring capacity - 1 = ceil(maxDelayTime * sample_rate / 128), delay.rs:300-302
  int32_t nch;
  uint32_t n_inst;
  uint32_t n_quanta;
  uint32_t tile0, tile1;
  int32_t in_cycle;      // the cycle breaker removed this node's writer->reader edge: delay clamped to one quantum
  int32_t pad;
  double quantum_duration;
};
void launch_delay(const DelayDesc& d, void* stream);


// ---- OscillatorNode (oscillator.rs:323-660): one lane per instance, frames in order ----------------
struct OscDesc {
  ParamRef frequency, detune;
  const double* start;   // [n_inst] start_time (DBL_MAX = never started)
  const double* stop;    // [n_inst]
  const float* table;    // sine (2048 points) or the custom PeriodicWave (8192 points)
  int32_t table_len;
  int32_t type;          // WAA_OSC_*; sine and custom read `table`
  SignalRef out;         // 1 channel
  uint64_t frames;       // padded frames
  uint32_t n_inst;
  uint32_t n_quanta;
  double sample_rate;
  const struct OscQuantum* table_q;  // non-null: time-parallel kernel (host-known frequency), [rows][n_quanta]
  const uint32_t* tq_row;             // [n_inst] row of table_q an instance reads (instances that replay alike share a row)
  const int64_t* active;       // non-null: prefix-sum kernel (a-rate frequency): [n_inst][2] active frames [first, end)
  const double* start_ratio;   // [n_inst] sub-sample start offset in frames (oscillator.rs:516-528)
  double* seg_phase;           // prefix-sum kernel: [n_inst][OSC_SEGMENTS] phase advance of each time segment (scratch)
  // what stands between the oscillator and a consumer it alone feeds, folded into its store (time-parallel and prefix-sum
  // kernels): up to two constant GainNodes (gain.rs:163-179 fast paths) and the speakers up-mix 1 -> 2 (`out` then has 2 channels)
  ParamRef post_gain[2];
  int32_t n_post;
  int32_t post_dup;
  // Two-operator FM folded into the carrier (prefix-sum kernel, round 4): the `frequency` AudioParam's only input is ONE
  // oscillator with a host-known frequency (its time-parallel table `fm_q`), through at most one edge gain — the carrier
  // evaluates modulator, gain and AudioParamProcessor::mix_to_output (param.rs:737-795) per frame itself instead of reading a
  // per-frame table three launches wrote and read back (fm_q == nullptr: not folded; `frequency` then is the intrinsic value).
  const struct OscQuantum* fm_q;  // [rows][n_quanta] of the MODULATOR
  const uint32_t* fm_row;         // [n_inst] its row table
  const float* fm_table;
  int32_t fm_table_len, fm_type;
  ParamRef fm_gain;               // mode 0 / 1 (fm_has_gain)
  int32_t fm_has_gain;
  float fm_min, fm_max, fm_default;
  // An LFO that only drives ONE AudioParam (time-parallel kernel, round 4): the param's summing chain — intrinsic value + input,
  // NaN -> default, clamp (param.rs:737-795) — in the oscillator's store; `out` then IS the param's per-frame table
  int32_t pa_on;
  float pa_min, pa_max, pa_default;
  ParamRef pa_intrinsic;          // mode 0 / 1
};
constexpr int OSC_SEGMENTS = 8;  // time segments per instance of the prefix-sum oscillator (one wavefront each)
// Per-(instance, quantum) record of the time-parallel oscillator: frames [first, end) of the quantum are active,
// the phase of frame `first` is `phase`, every further frame advances by `incr` (oscillator.rs:395-440).
struct OscQuantum {
  double phase;
  double incr;
  int16_t first, end;
  int32_t outside_nyquist;
};
void launch_osc(const OscDesc& d, void* stream);

// ---- feedback loops (graph.rs:323-487 cycle breaker): quantum-serial rendering of a strongly connected group ----
// The members of a loop are rendered render quantum by render quantum, in the reference's processing order, by
// one wavefront per instance; every member writes its own output signal.  A DelayNode contributes two items
// (delay.rs:283-366): the writer stores the node's mixed input ("history"), the reader gathers from it.
enum : int32_t { LI_NODE = 0, LI_DELAY_W = 1, LI_DELAY_R = 2 };
constexpr int LOOP_MAX_ITEMS = 24;
struct LoopItem {
  int32_t kind;
  int32_t nch_in;       // channel count the summed input is mixed to
  int32_t nch_out;
  int32_t interp;
  int32_t n_in;
  int32_t in_item[MAX_INPUTS];  // >= 0: output of that item of the same quantum (LDS); -1: external signal
  int32_t in_nch[MAX_INPUTS];
  SignalRef in_sig[MAX_INPUTS];
  SignalRef out;        // LI_NODE / LI_DELAY_R: the node's output; LI_DELAY_W: the delay line (absolute time)
  OpDesc op;            // LI_NODE: kind 0 = pass-through; LI_DELAY_R: p0 = delayTime
  int32_t writer_item;  // LI_DELAY_R: item whose `out` is the delay line
  int32_t in_cycle;     // LI_DELAY_R: the reader renders before its writer: delay clamped to one quantum
  int32_t num_quanta;   // LI_DELAY_R: ring capacity - 1
  int32_t pad;
};
struct LoopDesc {
  const LoopItem* items;  // device memory
  int32_t n_items;
  uint32_t n_inst;
  uint32_t n_quanta;
  uint32_t no_scan;       // set by the launcher (WAA_DYN_NO_SCAN): biquad members serially on one lane per channel
  double sample_rate;
  double quantum_duration;  // 128 * (1 / sample_rate), delay.rs:546-548
};
void launch_loop(const LoopDesc& d, void* stream);

// ---- exact dynamic channel counts (quantum.rs:109-120,532-569): quantum-serial rendering with per-quantum codes ----
// The reference's AudioRenderQuantum carries a channel count and a silent flag that change from quantum to quantum
// (a silent quantum is mono); count-sensitive nodes (filters with per-channel state, the panners' mono / stereo laws,
// the delay line's re-mix, the convolver's routing) render differently depending on it.  When the planner's replay
// finds such a change, every node that is not a source or a convolver is rendered by dyn_kernel: one wavefront per
// instance walks the render quanta in order, processes the items (nodes) in the reference's processing order and
// carries, next to every signal, a per-quantum CODE = channel count | CODE_SILENT.  Signals are published to HBM in
// their native layout (the first `count` channels) together with their code table, so that later launches (a
// second dyn group behind a convolver) consume them with the same rules.
constexpr int DYN_MAX_IN = 8;
constexpr int DYN_MAX_ITEMS = 48;
constexpr int DYN_STATE = 20;             // doubles of filter state per channel (biquad 4, IIR <= 20)
constexpr uint32_t CODE_SILENT = 0x80u;   // quantum.rs:254-256 is_silent; low bits = number_of_channels
enum : int32_t { DI_NODE = 0, DI_DELAY_W = 1, DI_DELAY_R = 2 };
enum : int32_t {
  DK_PASS = 0,        // destination / analyser / convolver or waveshaper without payload: output = input
  DK_GAIN = 1, DK_BIQUAD = 2, DK_IIR = 3, DK_WAVESHAPER = 4, DK_STEREO_PAN = 5, DK_PANNER = 6,
  DK_CONV_IN = 7      // the mixed input of a ConvolverNode with an impulse response (rendered node-major)
};
struct DynInput {
  int32_t item;             // >= 0: output of that item of this launch (same quantum, through LDS); -1: external
  int32_t nch;              // external: static width of the signal
  SignalRef sig;            // external
  const uint8_t* code;      // external: [n_inst][code_stride] per-quantum codes; null: always active, `nch` channels
  const uint32_t* remap;    // external: channel 1 of quantum q lives in quantum slot remap[inst][q] (mono-IR convolver)
  uint64_t code_stride;
};
struct DynItem {
  int32_t kind;             // DI_*
  int32_t dk;               // DK_* (DI_NODE)
  int32_t cc, mode, interp; // channel config: computed number of channels of the mixed input (quantum.rs:543-547)
  int32_t n_in;
  int32_t nch_pub;          // static width of the published signal
  int32_t publish_upmix;    // published mono quanta are duplicated into channel 1 (consumers outside dyn_kernel)
  int32_t compact_ch1;      // DK_CONV_IN with a mono impulse response: channel 1 is published in compacted time
  int32_t flags;            // DK_WAVESHAPER: bit 0 = can_propagate_silence (waveshaper.rs:498-509)
  int32_t writer_item;      // DI_DELAY_R
  int32_t in_cycle;         // DI_DELAY_R: renders before its writer (delay clamped to one quantum, delay.rs:693-701)
  int32_t num_quanta;       // DI_DELAY_R: ring capacity - 1
  int32_t pad;
  DynInput in[DYN_MAX_IN];
  OpDesc op;                // coefficients / params of the node (same meaning as in ChainDesc)
  ParamRef alt1, alt2;      // DK_STEREO_PAN / DK_PANNER: left / right gain of the MONO law (op.p1 / op.p2: stereo law)
  SignalRef out;            // published signal (DI_DELAY_W: the delay line, absolute time)
  uint8_t* out_code;        // [n_inst][code_stride]
  uint32_t* aux32;          // DI_DELAY_W: line codes as 32-bit words (read back by the reader of the same launch);
                            // DK_CONV_IN + compact_ch1: remap table [n_inst][code_stride]
  uint64_t code_stride;
  // A DelayNode whose writer and reader sit in DIFFERENT launches (a feedback loop cut at a frozen-state node and rendered quantum
  // by quantum, round 5): the reader (writer_item = -1) finds the writer's line, its codes and the ring's state here; the writer
  // publishes that state per instance after every quantum as (count - 1, last mono quantum + 1), zero-initialised per render
  SignalRef xline;          // DI_DELAY_R
  const uint32_t* xaux32;   // DI_DELAY_R
  int32_t* xstate;          // DI_DELAY_R: read; DI_DELAY_W: written (null: nobody outside the launch looks)
  // (round 6) DK_GAIN whose `gain` AudioParam is modulated from INSIDE the node's own feedback loop (param.rs:686-795): the param's
  // inputs are items of this launch — their channel 0 of this quantum (count 1, explicit, discrete), summed in edge order, plus the
  // intrinsic value op.p0, NaN -> default, clamped to [min, max]; pmod_n = 0: not modulated this way
  int32_t pmod_n;
  int32_t pmod_item[4];
  float pmod_min, pmod_max, pmod_def;
};
struct DynDesc {
  const DynItem* items;     // device memory
  int32_t n_items;
  uint32_t n_inst;
  uint32_t n_quanta;
  uint32_t no_scan;         // set by the launcher (WAA_DYN_NO_SCAN): biquad items on two lanes, serially (cross-check)
  int32_t cmax;             // widest signal of the group: <= 2 -> dyn_kernel<2>, else dyn_kernel<6> (layouts up to 5.1)
  int32_t n_stages;         // 1 .. DYN_MAX_STAGES: the item THIRDS [stage_begin[w], stage_begin[w + 1]) are stage w of the quantum pipeline
  int32_t stage_begin[9];   // (unit 3 i: item i's gather + mix, 3 i + 1: its node, 3 i + 2: the publication of its result)
  int32_t pad;
  double sample_rate;
  double quantum_duration;
  unsigned long long* cycles;  // measurement build, WAA_DYN_CYCLES: [items][3] shader-clock ticks of instance 0 (gather, node, hand-over)
  // launches over a RANGE of quanta (a loop rendered block by block, round 5): quanta [q0, q1) (q1 = 0: all), the items' filter
  // and integer state kept in memory between launches ([n_inst][n_items * CM * DYN_STATE] doubles, [n_inst][n_items * 4] ints)
  uint32_t q0, q1;
  double* save_f;
  int32_t* save_i;
  // (round 6) a ranged launch whose items carry NO state from quantum to quantum — DelayNode readers / writers whose partner sits
  // in another launch (the ring's channel count is the block's optimistic constant), gains, mixes, curves: nothing of a block's
  // quantum depends on an earlier one of the same launch, so the block's quanta are spread over `q_split` workgroups per instance
  // (set by the launcher from split_ok and the range's length) instead of walked by one wavefront at ~6 us per quantum
  uint32_t split_ok, q_split;
};
void launch_dyn(const DynDesc& d, void* stream);
// register planes of the kernel instantiation that renders signals of up to `cmax` channels (mono / stereo, 5.1, 7.1, 16, 32)
inline int dyn_planes(int cmax) { return cmax > 16 ? 32 : cmax > 8 ? 16 : cmax > 6 ? 8 : cmax > 2 ? 6 : 2; }
constexpr int DYN_MAX_STAGES = 8;
size_t dyn_lds_bytes(int n_items, int cmax, int stages = 1);  // dynamic LDS of the launch (<= 160 KB: the planner checks)
// ConvolverNode tail / routing on codes (convolver.rs:343-392): input codes -> output codes
struct ConvCodeDesc {
  const uint8_t* in_code;
  uint8_t* out_code;
  uint64_t code_stride;
  uint64_t impulse_length;  // frames of the AudioBuffer (untrimmed), convolver.rs:357-366
  int32_t ir_nch;
  uint32_t n_inst, n_quanta;
  int32_t cout;             // channels of `out`
  // The FFT path packs two instances into one complex transform: an instance whose input has been silent since the start
  // gets ~1e-9 of its partner's signal through the roundoff of the complex arithmetic — where the reference's convolver (one
  // per context) puts out exact zeros.  Quanta the reference still processes there (tail counter < impulse length: coded
  // active) must BE zeros, or every filter behind them starts a "tail" on that noise and the silence flags drift apart
  // (fuzz seed 232847 of the frozen-state generator).  `clean` marks them, conv_floor_kernel clears them in `out`.
  uint8_t* clean;           // [n_inst][code_stride]
  SignalRef out;
  // ranged form (the convolver sits in a loop rendered quantum by quantum, round 5): quanta [q0, q1), the tail counter and the
  // "has ever been active" flag per instance in memory between launches ({tail lo, tail hi, ever_active, initialised}, zero-initialised)
  uint32_t q0, q1;
  int32_t* state;
  // Round 5: the reference's FFT convolver leaves roundoff noise — never exact zeros — for up to two 1024-frame
  // blocks around anything non-zero in its input, and data-dependent silence behind it (a DelayNode that "read nothing but zeros")
  // follows that noise.  With `noise` set the code kernel runs waa_conv_noise.hpp's automaton per FFTConvolver of the node over
  // the non-zero flags of the input quanta (conv_nz_kernel: `in`, `in_test_nch` channels of it are in absolute time) and marks in
  // `clean`, per output channel c, bit c = "exact zeros in the reference: clear" / bit 2 + c = "noise in the reference: raise
  // |samples| below CONV_NOISE_FLOOR to it" (conv_floor_kernel).  The ranged form keeps the four automata behind the tail counter:
  // CONV_CODE_STATE_INTS ints per instance.
  int32_t noise, in_test_nch;
  SignalRef in;
  ConvNoiseIr nir[4];
};
constexpr int CONV_CODE_STATE_INTS = 4 + 4 * 4;
void launch_conv_codes(const ConvCodeDesc& d, void* stream);

// ---- per-frame biquad coefficients for a-rate params (biquad_filter.rs:837-855) -------------
struct BiquadCoefDesc {
  ParamRef frequency, detune, q, gain;
  double* coefs;        // frame-major [rows][frames_padded][5], or lane-major [rows][n_tiles][32][5][64] (streaming kernel)
  uint64_t n_frames;    // n_quanta * 128 (params are read clamped to it)
  uint64_t frames_padded;  // n_tiles * 2048: frames of one table row
  uint32_t rows;        // n_inst, or 1 when the four params are the same for every instance (one shared table)
  int32_t type;
  float sample_rate;
  int32_t lane_major;   // element (tile, k, coef, lane) of frame tile * 2048 + lane * 32 + k
};
void launch_biquad_coefs(const BiquadCoefDesc& d, void* stream);
// A per-frame coefficient table that is the same for every instance is digested once per plan: for the 32 frames a
// lane of the streaming kernel owns in a tile, the zero-state end state is LINEAR in the lane's samples,
//   (y_31, y_30) = sum_i H_i x_i + Hm1 x_{-1} + Hm2 x_{-2},   H_i = G_i b0_i + G_{i+1} b1_{i+1} + G_{i+2} b2_{i+2},
// G_i = first column of M_31 ... M_{i+1}, and the transition is P = M_31 ... M_0 (M_i = [[-a1_i, -a2_i], [1, 0]]).
// The streaming kernel then forms every lane's end state as a 34-tap dot product (no dependent chain, 16 B per frame
// instead of 40 B) before the exact-order pass.  hp[tile][e][lane]: e = 2i, 2i+1: H_i; 64, 65: Hm1; 66, 67: Hm2; 68..71: P.
constexpr int HP_WORDS = 72;
struct BiquadHpDesc {
  const double* coefs;  // lane-major per-frame table, one row
  double* hp;
  uint32_t n_tiles;
  uint32_t pad;
};
void launch_biquad_hp(const BiquadHpDesc& d, void* stream);

// ---- per-frame panner geometry for an audio-rate AudioListener (waa_panner.hip; panner.rs:720-897) ----
struct PannerGeomDesc {
  ParamRef p[15];          // WAA_PARAM_PANNER_* / WAA_PARAM_LISTENER_* in id order
  const uint8_t* single;   // [rows][single_stride]: 1 = all nine listener params are single-valued in this quantum
  uint64_t single_stride;
  const uint8_t* dev_len[9];  // listener params evaluated by timeline_kernel: [n_inst][single_stride] slice lengths, else null
  float *az, *gl_mono, *gr_mono, *gl_stereo, *gr_stereo, *dg, *cg;  // [rows][n_frames]
  uint64_t n_frames;       // n_quanta * 128
  uint32_t rows;           // n_inst, or 1 when nothing depends on the instance
  int32_t distance_model;
  double ref_distance, max_distance, rolloff;
  float cone_inner, cone_outer, cone_outer_gain;
  int32_t pad;
};
void launch_panner_geom(const PannerGeomDesc& d, void* stream);

// ---- AudioParam automation on the device (waa_timeline.hip; param.rs:1049-1584) ----
struct TlEvent {            // one queued AudioParamEvent (param.rs:162-171), as the render side holds it
  int32_t type;             // WAA_EVENT_*
  float value;
  double time;
  double time_constant;     // SetTarget
  double cancel_time;       // CancelAndHold rewrote the end of this event
  double duration;          // SetValueCurve
  int32_t cancelled;
  int32_t curve_off;        // SetValueCurve: offset / length of its values in the curve pool
  int32_t curve_len;
  int32_t pad;
};
struct TlHeader {           // one timeline = one (param, instance)
  float minv, maxv, defv, intrinsic;
  int32_t a_rate;
  int32_t ev_off;           // first event of this timeline in the event arrays
  int32_t n_events;
  int32_t pad;
};
struct TimelineDesc {
  const TlHeader* hdr;      // [rows]
  const TlEvent* events;    // scheduled queues (read-only)
  TlEvent* work;            // same size: the queue as the replay consumes / rewrites it
  const float* curves;
  float* out;               // [rows][out_stride] per-frame values
  uint8_t* lens;            // [rows][n_quanta]: 1 or 128, the length of the reference's slice in that quantum
  uint64_t out_stride;
  uint32_t rows;
  uint32_t n_quanta;
  double sample_rate;
};
void launch_timeline(const TimelineDesc& d, void* stream);

// ---- nodes whose state FREEZES while they do not process: WaveShaper 2x / 4x (waveshaper.rs:395-400: silent input and a
// curve that maps 0 to 0 -> the resamplers are not run) and the HRTF panner (panner.rs:697-711: silent input after the
// tail counter ran out).  Rendered node-major; a serial pass over the per-quantum codes of the node's mixed input
// (link_kernel, one thread per instance) turns the reference's control flow into a table the time-parallel kernels
// follow: prev[inst][q] = LINK_SKIP (the node does not process quantum q: silent output), LINK_FRESH (it processes q
// with fresh state) or the previous quantum it processed (where its state comes from).
constexpr int32_t LINK_SKIP = -2, LINK_FRESH = -1;
struct LinkDesc {
  const uint8_t* in_code;   // [n_inst][code_stride] count | CODE_SILENT of the node's mixed input per quantum
  uint8_t* out_code;        // [n_inst][code_stride] codes of the node's output (null in static plans)
  int32_t* prev;            // [n_inst][prev_stride]
  uint64_t code_stride, prev_stride;
  uint32_t n_inst, n_quanta;
  int32_t kind;             // 0: oversampled WaveShaper, 1: HRTF panner
  int32_t can_propagate_silence;  // kind 0 (waveshaper.rs:498-509)
  uint32_t tail_frames;     // kind 1: HRIR length (panner.rs:270-272)
  int32_t pad;
  // ranged form (link_range_kernel: the node sits in a loop rendered block by block): quanta [q0, q1), the automaton's state
  // per instance in memory between launches: {last, cur_ch, tail_counter lo, hi}, zero = the initial state shifted (see the kernel)
  uint32_t q0, q1;
  int32_t* state;
};
void launch_link(const LinkDesc& d, void* stream);

// One resampling stage of the oversampled WaveShaper as a matrix product over render quanta: column (inst, ch, q) of
// the result = A[:, 0:Kh] * src(inst, ch, q) + A[:, Kh:2Kh] * src(inst, ch, prev(q)) — the block's own response plus
// the overlap the previous processed block left behind (rubato FftFixedInOut's overlap-add) — followed
// by the WaveShaper curve when `curve` is set (the up-sampling stage).
struct QGemmDesc {
  const float* A;           // [2 * Kh][M], k-major: row k holds the M coefficients that multiply source element k
  const uint16_t* A16;      // the same matrix as three bf16 planes (A = hi + mid + lo exactly), tiled [k / 16][plane][M][16]
  int32_t M, Kh;
  const float* src;         // element k of column (inst, ch, q): src[inst * src_inst + ch * src_ch + q * src_q + k]
  uint64_t src_inst, src_ch, src_q;
  float* dst;               // M contiguous floats per column
  uint64_t dst_inst, dst_ch, dst_q;
  const int32_t* prev;
  uint64_t prev_stride;
  const float* curve;       // epilogue: waveshaper.rs:555-573 (null: none)
  int32_t curve_n;
  int32_t nch;
  uint32_t n_inst, n_quanta;
};
void launch_qgemm(const QGemmDesc& d, void* stream);

// The same node in ONE launch as 256-point transforms (waa_osfft.hip / waa_osfft.hpp): both resampling stages and the
// curve; groups of 16 lanes walk runs of `seg_len` quanta with the stages' overlaps in registers.
struct OsFftDesc {
  const float* src;         // the node's mixed input: frame f of (inst, ch) at src[inst * src_inst + ch * src_ch + f]
  uint64_t src_inst, src_ch;
  float* dst;
  uint64_t dst_inst, dst_ch;
  const int32_t* prev;      // [n_inst][prev_stride]: LINK_SKIP / LINK_FRESH / previous processed quantum
  uint64_t prev_stride;
  const float* curve;
  int32_t curve_n;
  int32_t R;                // 2 or 4
  const float* tables;      // 2 R tables of osfft::TAB_SLOTS complex values (U_r, then V_r), lane-major rows
  const float* tw256;       // exp(-2 pi i j / 256)
  float* trash;             // 64 floats nobody reads: where lanes that do not store a quantum put their values
  int32_t nch;              // 1 or 2
  uint32_t n_inst, n_quanta;
  uint32_t seg_len, n_seg;  // quanta per run, runs per instance (n_seg * seg_len >= n_quanta)
  uint32_t q0, q1;          // the quanta this launch renders (q1 = 0: all); runs start at q0
};
void launch_osfft(const OsFftDesc& d, void* stream);
size_t osfft_lds_bytes(int R, int curve_n);

// HRTF panner (panner.rs:781-829 + crate hrtf): per render quantum the HRIR pair of the direction
// (barycentric mix of three measured HRIRs) is convolved with the mono input continued into the previously processed
// quanta; direct-form FIR, one wavefront per (instance, quantum).
struct HrtfQ {              // host-evaluated geometry of one (instance, quantum) — or one per instance when static
  int32_t v[3];             // vertices of the triangle the direction pierces
  float w[3];               // barycentric weights
  float gain;               // cone_gain * dist_gain
  int32_t pad;
};
constexpr int HRTF_MAX_TAPS = 1280;
struct HrtfDesc {
  SignalRef in, out;        // in: the node's mixed input (1 or 2 channels); out: 2 channels
  const uint8_t* in_code;   // [n_inst][code_stride]
  uint64_t code_stride;
  const int32_t* prev;
  uint64_t prev_stride;
  const float* hrir;        // [n_vertices][2][taps]: left, right
  const float* hstatic;     // one HRIR pair for the whole batch, interpolated on the host: [taps rounded up to 4][2]; or null
  const HrtfQ* table;       // [rows][per_row]
  uint32_t rows, per_row;   // rows: n_inst or 1 (nothing depends on the instance); per_row: n_quanta or 1 (static)
  int32_t taps;
  uint32_t n_inst, n_quanta;
  uint32_t q0, q1;          // the quanta this launch renders (q1 = 0: all)
  int32_t pad;
  // the transform form (waa_hrtf_fft.hip; round 6): directions that do not change during the render (per_row == 1) — one for the
  // whole batch (rows == 1) or one per context (rows == n_inst) —, taps <= 512
  const float* fft_tables;  // per row: hrtffft::PARTS tables of osfft::TAB_SLOTS complex values, lane-major rows; null: direct form only
  const float* tw256;       // exp(-2 pi i j / 256)
  float* trash;             // 64 floats nobody reads
  uint32_t seg_len, n_seg;  // quanta per run, runs per instance
  uint32_t n_seg_pad, pad2; // set by the launcher: runs per instance in the workgroup mapping (rows > 1: a multiple of 16)
  const int32_t* jmax;      // [rows][2] last non-zero tap per ear: the exact-zeros form (dynamic plans); null: plain
};
void launch_hrtf(const HrtfDesc& d, void* stream);
void launch_hrtf_fft(const HrtfDesc& d, void* stream);
void launch_hrtf_fft_tables(const float* pairs, uint32_t pair_stride, int taps, uint32_t rows, const double* cs_sn, float* out, void* stream);

// ---- input preparation on the device (waa_decode.hip): decoded 16-bit PCM -> f32 planes at the context rate ----
struct DecodeDesc {
  const int16_t* pcm;        // [n_items][frames][nch] interleaved
  float* out;                // planes: out[item * out_item_stride + ch * out_ch_stride + frame]
  uint64_t frames;           // source frames per item
  uint64_t target_frames;    // ceil(frames * ratio), or frames when the rates match (buffer.rs:315-318)
  uint64_t out_item_stride, out_ch_stride;
  uint32_t nch, n_items;
  int32_t resample;
  float scale;               // 1 (kept for callers that fold a constant gain in)
};
void launch_pcm16_resample(const DecodeDesc& d, void* stream);
struct EncodeDesc {
  const float* in;           // planes: in[item * in_item_stride + ch * in_ch_stride + frame]
  int16_t* pcm;              // [n_items][frames][nch_out] interleaved
  uint64_t frames, in_item_stride, in_ch_stride;
  uint32_t nch_in, nch_out, n_items, pad;
};
void launch_pcm16_pack(const EncodeDesc& d, void* stream);

// launchers implemented in waa_kernels.hip
void launch_chain(const ChainDesc& d, int cmax, void* stream);
// the echo loop with its delay line in LDS (waa_echo.hip): qualification (delay range in frames over all instances) and launch
// EchoTail: the one reader of the line outside the loop (a sum of the delayed line and of signals the loop reads too),
// rendered by the same launch; store_line == 0: nothing else reads the line, it stays in LDS
struct EchoTail {
  int32_t n_inputs, in_nch, in_interp, store_line;
  int32_t ring_frames, sub_frames;  // frames per channel of the LDS ring (a power of two) / per wavefront and chunk (256 or 128): set by the launcher
  InputRef in[MAX_INPUTS];
  int32_t alias[MAX_INPUTS];  // -2: the delayed line; s >= 0: the same signal as the loop step's s-th input from outside the loop
  SignalRef out;
  ParamRef delay;             // the DelayNode's delayTime (mode 0 or 3) and rate: set by the launcher from the feedback
  double sample_rate;         // input, or by echo_feed_forward for a line that is not fed back
};
// A Biquad (constant coefficients) between the delayed read and the loop's sum, rendered by the same launch (the BQ form):
// `y` = the filter's output signal — the operand of the loop stage's feedback edge and of the tail; stored when store_y.
struct EchoBq {
  const double* coefs;        // [n_inst][coef_stride]: b0 b1 b2 a1 a2; nullptr: no filter in the loop
  uint64_t coef_stride;
  SignalRef y;
  int32_t store_y, store_line;  // readers outside the launch (fuse_echo_tails decides)
  ParamRef delay;             // the DelayNode's delayTime and rate (the delayed read is not an input of the sum here)
  double sample_rate;
};
int echo_ring_applicable(const ChainDesc& d, const float* delay_min_max_frames, int* chunk_frames);
// the three body launches of a block-scheduled loop  delayed read -> streaming biquad -> sum into the line  as the BQ form?
// Returns the index of the sum's input that reads the filter's output (and fills bq / the chunk size), or -1.
struct BiquadStreamDesc;
int echo_bq_applicable(const ChainDesc& read, const BiquadStreamDesc& filter, const ChainDesc& sum, const float* delay_min_max_frames,
                       int* chunk_subtiles, EchoBq* bq);
// a chain step OUTSIDE any loop that sums delayed(X) with X itself and at most one other signal, no ops (the feed-forward
// echo): the same kernel with nothing fed back — X goes through the ring instead of being read twice.  Fills the stand-in
// loop stage (`line` = X) and the tail; returns the chunk size or 0.
int echo_feed_forward(const ChainDesc& step, ChainDesc* line, EchoTail* tail, const char** why);
int echo_tail_applicable(const ChainDesc& d, int fb, const ChainDesc& tail, EchoTail* t, const char** why, const EchoBq* bq = nullptr);
// frames per channel the ring needs for delays up to dmax with this chunk size: a power of two in [1024, 16384]
int echo_ring_frames(float dmax_frames, int chunk_frames);
void launch_echo_ring(const ChainDesc& d, int fb, int chunk_frames, int ring_frames, const EchoTail* tail, void* stream, const EchoBq* bq = nullptr);
// dst[inst][q] = src[inst * inst_stride + q * 128]: the first frame of every render quantum of a per-frame table
void launch_quantum_heads(const float* src, uint64_t inst_stride, uint32_t n_inst, uint32_t n_quanta, float* dst, void* stream);
// waa_resample.hip: AudioBufferSource [-> WaveShaper] -> signal without the op interpreter (the C5 shape)
bool resample_shape(const ChainDesc& d, int* curve_op);
void launch_resample(const ChainDesc& d, int curve_op, void* stream);

#ifdef __HIPCC__
// A pointer LOADED from memory (the per-instance source records) is a generic pointer to the compiler: every access
// through it becomes a flat_load, which counts on lgkmcnt as well as vmcnt — so the next wait on an LDS read also
// waits for the prefetched tile, and the software pipeline of the streaming kernels collapses.  The loads below go
// through explicit global-address-space pointers instead.
typedef float f4v __attribute__((ext_vector_type(4)));
#define WAA_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ f4v load_global_f4(const float* p) { return *(const WAA_GLOBAL_AS f4v*)p; }
template <class T>
__device__ __forceinline__ T load_global(const T* p) {
  static_assert(sizeof(T) <= 8, "scalar types only");
  return *(const WAA_GLOBAL_AS T*)p;
}
template <class T>
__device__ __forceinline__ void store_global(T* p, T v) {
  static_assert(sizeof(T) <= 8, "scalar types only");
  *(WAA_GLOBAL_AS T*)p = v;
}
typedef int i4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ QRec load_global(const QRec* p) {
  const i4v t = *(const WAA_GLOBAL_AS i4v*)p;
  QRec r;
  r.start = (int64_t)(((uint64_t)(uint32_t)t.y << 32) | (uint32_t)t.x);
  r.mode = (uint32_t)t.z;
  r.pad = 0;
  return r;
}
__device__ __forceinline__ SlowRec load_global(const SlowRec* p) {
  const i4v t = *(const WAA_GLOBAL_AS i4v*)p;
  SlowRec r;
  r.prev = t.x;
  r.next = t.y;
  r.k = __longlong_as_double((long long)(((uint64_t)(uint32_t)t.w << 32) | (uint32_t)t.z));
  return r;
}

#endif

#ifdef __HIPCC__
// Four consecutive frames (i0 .. i0 + 3 of quantum q) of a DelayNode's output for a delay that is ONE value in this
// quantum (constant or k-rate): delay.rs:560-590 (infos[0] from the value, then one frame per frame), :642 (the f32 fma),
// :622-626 (the sample after frame 127 of the newest block is frame 0 of the oldest ring block).  `in` is the delay line
// of this (instance, channel) in absolute time; frames before 0 read silence.
__device__ __forceinline__ void delay_read4(const float* in, uint64_t valid, float dv, double sample_rate, int32_t num_quanta,
                                            bool in_cycle, double quantum_duration, uint32_t q, int i0, float (&r)[4]) {
  double dd = (double)dv;
  if (in_cycle) dd = fmax(dd, quantum_duration);  // delay.rs:693-701
  const double position = 0. - dd * sample_rate;
  const double fl = floor(position);
  const int64_t pf0 = (int64_t)fl;
  const float k = (float)(position - fl);
  const int64_t qstart = (int64_t)q * RQ;
  {
    // fast path: the five samples in[first .. first + 4] out of two aligned 16-byte loads (instead of eight scattered
    // 4-byte ones); taken when no frame of the group is frame 127 of the newest block (the wrap rule below) and the
    // aligned window lies inside the line
    const int64_t first = qstart + pf0 + i0;
    const int64_t a = first & ~(int64_t)3;
    if (pf0 + i0 + 3 < RQ - 1 && !in_cycle && a >= 0 && (uint64_t)a + 8 <= valid && ((uintptr_t)in & 15) == 0) {
      const f4v v0 = load_global_f4(in + a), v1 = load_global_f4(in + a + 4);
      const float w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const int sh = (int)(first - a);
      float x[5];
#pragma unroll
      for (int e = 0; e < 5; e++) x[e] = sh == 0 ? w[e] : sh == 1 ? w[e + 1] : sh == 2 ? w[e + 2] : w[e + 3];
#pragma unroll
      for (int e = 0; e < 4; e++) r[e] = __builtin_fmaf(1.f - k, x[e], k * x[e + 1]);
      return;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int64_t pf = pf0 + i0 + e;
    const int64_t prev = qstart + pf;
    int64_t next = pf == RQ - 1 ? (int64_t)((int64_t)q - (int64_t)num_quanta) * RQ : prev + 1;
    // a reader that renders before its writer finds, in the slot of the current quantum, the block written
    // ring-capacity quanta ago (only reachable with k == 0)
    if (in_cycle && next >= qstart) next -= ((int64_t)num_quanta + 1) * RQ;
    const float ps = prev >= 0 && (uint64_t)prev < valid ? in[prev] : 0.f;
    const float nsamp = next >= 0 && (uint64_t)next < valid ? in[next] : 0.f;
    r[e] = __builtin_fmaf(1.f - k, ps, k * nsamp);
  }
}
#endif
}  // namespace waa
