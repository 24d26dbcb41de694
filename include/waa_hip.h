/*
 * waa_hip.h — C ABI of the MI355X batched offline render engine.
 *
 * One `waa_batch` = N independent, identically-shaped OfflineAudioContexts
 * ("instances") rendered together on one GPU.  The entry points replace, for the
 * per-quantum DSP path only, what a Rust FFI shim would bind behind these
 * reference interfaces (paths relative to the reference crate web-audio-api 1.6.0):
 *
 *   waa_batch_create            OfflineAudioContext::new                     src/context/offline.rs:78-143
 *                               + ConcreteBaseAudioContext::register          src/context/concrete_base.rs:232-270
 *                               + AudioNode::connect (ControlMessage::ConnectNode) src/node/audio_node.rs:247-289
 *   waa_source_set_buffer*      AudioBufferSourceNode::set_buffer             src/node/audio_buffer_source.rs:853-866 (onmessage)
 *   waa_source_ended            AudioScheduledSourceNode `ended` event        src/render/processor.rs:53-58, src/render/thread.rs:398-411
 *   waa_source_start/stop/loop  AudioBufferSourceNode::start_at_with_offset_and_duration / stop_at / set_loop*
 *                                                                             src/node/audio_buffer_source.rs:388-398
 *   waa_convolver_set_buffer    ConvolverNode::set_buffer                     src/node/convolver.rs:259-317
 *   waa_*_set_buffer_pcm16*     BaseAudioContext::decode_audio_data_sync      src/context/base.rs:68-73, src/decoding.rs:15-54
 *                               (+ AudioBuffer::resample src/buffer.rs:311-363) feeding the two set_buffer calls above
 *   waa_waveshaper_set_curve    WaveShaperNode::set_curve                     src/node/waveshaper.rs:489-509 (onmessage)
 *   waa_hrtf_load_sphere        load_hrtf_processor (include_bytes!(IRC_1003_C.bin) -> hrtf::HrirSphere::new)
 *                                                                             src/node/panner.rs:39-68
 *   waa_hrtf_hrir_length        HrirSphere::len (the HRTF panner's tail time)  src/node/panner.rs:56,270-272
 *   waa_hrtf_sample             hrtf::HrirSphere::sample_bilinear (test hook)  src/node/panner.rs:261 (process_samples)
 *   waa_oscillator_set_periodic_wave  OscillatorNode::set_periodic_wave     src/node/oscillator.rs:318-321
 *   waa_oscillator_set_wavetable   OscillatorRenderer::onmessage(PeriodicWave)  src/node/oscillator.rs:487-493
 *   waa_iir_set_coefficients    IIRFilterNode::new(IIRFilterOptions)          src/node/iir_filter.rs:163-189
 *   waa_iir_frequency_response  IIRFilterNode::get_frequency_response         src/node/iir_filter.rs:218-262
 *   waa_set_param_const/block   AudioParamValues::get (len 1 / len 128)       src/render/processor.rs:186-229
 *                               (the automation timeline itself stays on the host: src/param.rs:685-797)
 *   waa_render                  OfflineAudioContext::start_rendering_sync     src/context/offline.rs:157-185
 *                               -> RenderThread::render_audiobuffer_sync      src/render/thread.rs:260-302
 *                               -> Graph::render                              src/render/graph.rs:490-591
 *                               -> every AudioProcessor::process              src/render/processor.rs:113-139
 *   waa_render_range            OfflineAudioContext::suspend_sync             src/context/offline.rs:359-397
 *                               + the suspend points of the quantum loop      src/render/thread.rs:277-294
 *   waa_connect / waa_disconnect  AudioNode::connect / disconnect_*           src/node/audio_node.rs:247-289, 341-420
 *   waa_download*               AudioBuffer returned by start_rendering_sync  src/render/thread.rs:384-395
 *   waa_analyser_*              AnalyserNode::get_*_data                      src/node/analyser.rs:228-258, src/analysis.rs:261-401
 *
 * Conventions: plain pointers and sizes only; every pointer argument is caller-owned
 * and copied before the call returns (except *_adopt_device, documented below); every
 * call returns a waa_status (0 = ok) and leaves a message readable with
 * waa_last_error() (thread-local).  A batch is single-threaded; different batches are
 * independent (one per GPU for multi-GPU sharding, no collectives).
 *
 * The oracle (oracle/waa_oracle.c, test infrastructure only) exports the same entry
 * points with the prefix orc_ so the parity tests can drive both with one harness.
 */
#ifndef WAA_HIP_H
#define WAA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAA_RENDER_QUANTUM_SIZE 128 /* src/lib.rs:18 */
#define WAA_MAX_CHANNELS 32         /* src/lib.rs:21 */
#define WAA_ALL_INSTANCES 0xFFFFFFFFu

typedef struct waa_batch waa_batch;
typedef int32_t waa_status;

enum {
  WAA_OK = 0,
  WAA_ERR_INVALID_ARGUMENT = 1, /* reference: assert!/panic with "…Error - …" message */
  WAA_ERR_NOT_SUPPORTED = 2,    /* reference: "NotSupportedError - …"                 */
  WAA_ERR_INVALID_STATE = 3,    /* reference: "InvalidStateError - …"                 */
  WAA_ERR_OUT_OF_SCOPE = 4,     /* legal in the reference, not on this hot path (worklets, a long ConvolverNode in a short feedback loop, …) */
  WAA_ERR_DEVICE = 5            /* HIP runtime failure                                 */
};

/* Node kinds: one per *Renderer on the path. */
enum {
  WAA_NODE_DESTINATION = 0,     /* src/node/destination.rs:140-163   */
  WAA_NODE_BUFFER_SOURCE = 1,   /* src/node/audio_buffer_source.rs:422-845 */
  WAA_NODE_BIQUAD = 2,          /* src/node/biquad_filter.rs:764-899 */
  WAA_NODE_GAIN = 3,            /* src/node/gain.rs:143-199          */
  WAA_NODE_CONVOLVER = 4,       /* src/node/convolver.rs:343-490     */
  WAA_NODE_STEREO_PANNER = 5,   /* src/node/stereo_panner.rs:218-317 */
  WAA_NODE_PANNER = 6,          /* src/node/panner.rs:685-904 (equal-power and HRTF panning models, all distance models, cone) */
  WAA_NODE_ANALYSER = 7,        /* src/node/analyser.rs:265-290      */
  WAA_NODE_WAVESHAPER = 8,      /* src/node/waveshaper.rs:383-487 (oversample None, 2x, 4x) */
  WAA_NODE_CONSTANT_SOURCE = 9, /* src/node/constant_source.rs:190-275 */
  WAA_NODE_IIR_FILTER = 10,     /* src/node/iir_filter.rs:323-405 */
  WAA_NODE_DELAY = 11,          /* src/node/delay.rs:428-745 incl. the cycle breaker of graph.rs:323-487 */
  WAA_NODE_OSCILLATOR = 12,     /* src/node/oscillator.rs:323-660 (on-device source) */
  WAA_NODE_KIND_COUNT = 13
};

/* src/node/audio_node.rs ChannelCountMode / ChannelInterpretation */
enum { WAA_COUNT_MODE_MAX = 0, WAA_COUNT_MODE_CLAMPED_MAX = 1, WAA_COUNT_MODE_EXPLICIT = 2 };
enum { WAA_INTERP_SPEAKERS = 0, WAA_INTERP_DISCRETE = 1 };

/* src/node/biquad_filter.rs:376-404 BiquadFilterType (same order) */
enum {
  WAA_BIQUAD_LOWPASS = 0, WAA_BIQUAD_HIGHPASS = 1, WAA_BIQUAD_BANDPASS = 2, WAA_BIQUAD_NOTCH = 3,
  WAA_BIQUAD_ALLPASS = 4, WAA_BIQUAD_PEAKING = 5, WAA_BIQUAD_LOWSHELF = 6, WAA_BIQUAD_HIGHSHELF = 7
};
/* src/node/panner.rs PanningModelType / DistanceModelType */
enum { WAA_PANNING_EQUALPOWER = 0, WAA_PANNING_HRTF = 1 };
enum { WAA_DISTANCE_LINEAR = 0, WAA_DISTANCE_INVERSE = 1, WAA_DISTANCE_EXPONENTIAL = 2 };
/* src/node/waveshaper.rs OverSampleType */
enum { WAA_OVERSAMPLE_NONE = 0, WAA_OVERSAMPLE_X2 = 1, WAA_OVERSAMPLE_X4 = 2 };

/* AudioParam ids, per node kind (the reference has one AudioParamId per param node). */
enum { WAA_PARAM_BIQUAD_FREQUENCY = 0, WAA_PARAM_BIQUAD_DETUNE = 1, WAA_PARAM_BIQUAD_Q = 2, WAA_PARAM_BIQUAD_GAIN = 3 };
enum { WAA_PARAM_GAIN_GAIN = 0 };
enum { WAA_PARAM_SOURCE_PLAYBACK_RATE = 0, WAA_PARAM_SOURCE_DETUNE = 1 };
enum { WAA_PARAM_STEREO_PANNER_PAN = 0 };
enum { WAA_PARAM_CONSTANT_OFFSET = 0 };
enum { WAA_PARAM_DELAY_DELAY_TIME = 0 };
enum { WAA_PARAM_OSCILLATOR_FREQUENCY = 0, WAA_PARAM_OSCILLATOR_DETUNE = 1 };
/* src/node/oscillator.rs OscillatorType */
enum { WAA_OSC_SINE = 0, WAA_OSC_SQUARE = 1, WAA_OSC_SAWTOOTH = 2, WAA_OSC_TRIANGLE = 3, WAA_OSC_CUSTOM = 4 };
enum {
  WAA_PARAM_PANNER_POSITION_X = 0, WAA_PARAM_PANNER_POSITION_Y = 1, WAA_PARAM_PANNER_POSITION_Z = 2,
  WAA_PARAM_PANNER_ORIENTATION_X = 3, WAA_PARAM_PANNER_ORIENTATION_Y = 4, WAA_PARAM_PANNER_ORIENTATION_Z = 5,
  /* the AudioListener's 9 params (src/spatial.rs:127-144) are addressed through any panner node */
  WAA_PARAM_LISTENER_POSITION_X = 6, WAA_PARAM_LISTENER_POSITION_Y = 7, WAA_PARAM_LISTENER_POSITION_Z = 8,
  WAA_PARAM_LISTENER_FORWARD_X = 9, WAA_PARAM_LISTENER_FORWARD_Y = 10, WAA_PARAM_LISTENER_FORWARD_Z = 11,
  WAA_PARAM_LISTENER_UP_X = 12, WAA_PARAM_LISTENER_UP_Y = 13, WAA_PARAM_LISTENER_UP_Z = 14
};
#define WAA_MAX_PARAMS 15

/*
 * Static per-node configuration (the *Options structs of the reference).
 *   channel_count / _mode / _interpretation : AudioNodeOptions (0 in channel_count = kind default)
 *   i[] / d[] by kind:
 *     BIQUAD      i[0] = filter type
 *     PANNER      i[0] = panning model, i[1] = distance model,
 *                 d[0] ref_distance, d[1] max_distance, d[2] rolloff_factor,
 *                 d[3] cone_inner_angle, d[4] cone_outer_angle, d[5] cone_outer_gain
 *     ANALYSER    i[0] = fft_size, d[0] smoothing_time_constant, d[1] min_decibels, d[2] max_decibels
 *     WAVESHAPER  i[0] = oversample
 *     CONVOLVER   i[0] = disable_normalization (0/1)
 *     OSCILLATOR  i[0] = type (WAA_OSC_*; CUSTOM needs waa_oscillator_set_periodic_wave); scheduled with
 *                 waa_source_start / waa_source_stop like the other AudioScheduledSourceNodes
 *     DELAY       d[0] = max_delay_time in seconds (0 = the default, 1 s); must be > 0 and < 180
 *                 (NotSupportedError, delay.rs:290-293).  Graph cycles through a DelayNode are rendered (the delay
 *                 is clamped to one render quantum inside a loop, delay.rs:693-701); cycles without one are muted.
 */
typedef struct {
  uint32_t kind;
  uint32_t channel_count;
  uint32_t channel_count_mode;
  uint32_t channel_interpretation;
  int32_t i[4];
  double d[8];
} waa_node_desc;

/* AudioNode::connect_from_output_to_input(from, output, to, input); mirrors graph.rs Edge.
 * `node.connect(&audio_param)` (audio-rate modulation, src/param.rs:686-795: the param is a graph node with
 * channel count 1 / explicit / discrete whose summed input is added to the intrinsic value) is expressed as an
 * edge into the OWNING node with to_input = WAA_PARAM_INPUT(param id).  Supported on the device for a-rate
 * params rendered there (Gain gain, Biquad frequency/detune/Q/gain, Delay delayTime, StereoPanner pan,
 * ConstantSource offset); params evaluated by the host are resolved at plan time by rendering the modulating subgraph
 * first (source playbackRate / detune: k-rate; PannerNode position / orientation next to a single-valued AudioListener:
 * first value per quantum, panner.rs:833-846) — WAA_ERR_OUT_OF_SCOPE next to an audio-rate listener, for nested
 * modulated sources and on plan-only batches. */
#define WAA_PARAM_INPUT(param) (0x80000000u | (uint32_t)(param))
typedef struct {
  uint32_t from;
  uint32_t from_output;
  uint32_t to;
  uint32_t to_input;
} waa_edge_desc;

/* Node 0 must be the destination (DESTINATION_NODE_ID, src/context/mod.rs:24). */
typedef struct {
  uint32_t n_nodes;
  const waa_node_desc* nodes;
  uint32_t n_edges;
  const waa_edge_desc* edges;
} waa_graph_desc;

/* ---- lifecycle ------------------------------------------------------------------------- */

/* device >= 0: that HIP device; -1: the current HIP device; WAA_DEVICE_PLAN_ONLY (-2): no device at all —
 * the batch can be configured and planned (waa_plan_describe) but never rendered (host-logic tests, tooling).
 * length_frames / sample_rate / n_channels_out as in OfflineAudioContext::new(number_of_channels, length,
 * sample_rate). */
#define WAA_DEVICE_PLAN_ONLY (-2)
waa_status waa_batch_create(const waa_graph_desc* graph, uint32_t n_instances, uint32_t n_channels_out,
                            uint64_t length_frames, float sample_rate, int32_t device, waa_batch** out);
void waa_batch_destroy(waa_batch* batch);
/* Optional: reserve ONE slab of device memory per device (device -1: the current one), once, ideally before anything else
 * allocates on it; every batch created afterwards carves its large buffers (>= 1 MB) out of it in 2 MB-aligned pieces and
 * falls back to hipMalloc for what does not fit.  Pieces return to the slab when the last batch holding one is destroyed.
 * bytes = 0 releases the slab (InvalidStateError while batches still use it).  No counterpart in the reference (its buffers
 * are Vec<f32>s): this exists because the same streaming kernel ran 15 % faster or slower depending on which hipMalloc
 * served its output buffer — and because hipMalloc / hipFree synchronise the device: a process that
 * creates and destroys batches while others render (waa_render_sharded's pipeline, a server) should reserve one (DESIGN.md section 7:
 * 112 -> 95 ms for 2 x 3.9 GB through one device). */
waa_status waa_device_arena_reserve(int32_t device, uint64_t bytes);
/* The same slab, GRADED: which physical memory a buffer lies in decides how fast a streaming kernel writes it (MI355X: the copy of
 * C2's footprint takes 1.30-1.37 ms into some regions, 1.49-1.52 ms into others; regions are tens of GB wide, stable, independent
 * of the virtual address — profiles/r06a...r06f).  This call creates candidate_bytes (>= bytes; clamped to
 * what is free) of physical memory in units (hipMemCreate), times the one-wave-per-stream copy INTO every unit, maps the `bytes`
 * fastest-to-write units side by side — fastest first — and releases the rest.  Batches then take what they WRITE (signals,
 * spectra, outputs) from the bottom of the slab and what they only read (source AudioBuffers) from the top.  About 4 ms per
 * GiB of candidates, once per process: what a serving process does at start-up; an offline context that renders once does not
 * win the time back.  Everything else as waa_device_arena_reserve (bytes = 0 releases; InvalidStateError while in use). */
waa_status waa_device_arena_reserve_graded(int32_t device, uint64_t bytes, uint64_t candidate_bytes);
/* The grades of the units of the graded arena of `device`, in address order (ms per copy into the unit: ascending), and what
 * the candidates looked like.  All zero when the device has no graded arena.  unit_ms may be NULL. */
typedef struct waa_arena_grades {
  uint64_t unit_bytes;
  uint32_t n_units, n_candidates;
  float best_ms, worst_kept_ms, worst_candidate_ms, grading_ms;
} waa_arena_grades;
waa_status waa_device_arena_grades(int32_t device, waa_arena_grades* out, float* unit_ms, uint32_t capacity);
/* What the arena of `device` (-1: the current one) has done so far: a request of >= 1 MB the slab could not serve is a MISS
 * (the batch fell back to hipMalloc, which synchronises the device) — a serving process watches `misses`.  All zero when no
 * arena is reserved.  Pieces are handed back individually when their batch is destroyed (first-fit free list, neighbours merge). */
typedef struct waa_arena_stats {
  uint64_t reserved_bytes, in_use_bytes, peak_bytes, largest_free_bytes;
  uint64_t served, misses, miss_bytes;
} waa_arena_stats;
waa_status waa_device_arena_stats(int32_t device, waa_arena_stats* out);
const char* waa_last_error(void);
/* number of visible HIP devices (0 if none / runtime unavailable) */
int32_t waa_device_count(void);

/* ---- node payloads --------------------------------------------------------------------- */

/* AudioBuffer for one instance (or WAA_ALL_INSTANCES: the same buffer shared by all, like a
 * cloned Arc<Vec<f32>>). channels[c] points at `frames` f32. */
waa_status waa_source_set_buffer(waa_batch* batch, uint32_t node, uint32_t instance, const float* const* channels,
                                 uint32_t n_channels, uint64_t frames, float buffer_sample_rate);
/* Whole batch at once: data laid out [instance][channel][frame], distinct per instance. */
waa_status waa_source_set_buffer_batch(waa_batch* batch, uint32_t node, const float* data, uint32_t n_channels,
                                       uint64_t frames, float buffer_sample_rate);
/* Zero-copy: `device_data` is a device pointer with the same [instance][channel][frame] layout that
 * must stay valid until the batch is destroyed (bench: inputs resident in HBM). */
waa_status waa_source_adopt_device(waa_batch* batch, uint32_t node, const float* device_data, uint32_t n_channels,
                                   uint64_t frames, float buffer_sample_rate);
/* AudioScheduledSourceNode::start_at_with_offset_and_duration(when, offset, duration);
 * duration = DBL_MAX for "none" (the reference stores f64::MAX). */
waa_status waa_source_start(waa_batch* batch, uint32_t node, uint32_t instance, double when, double offset,
                            double duration);
waa_status waa_source_stop(waa_batch* batch, uint32_t node, uint32_t instance, double when);
waa_status waa_source_set_loop(waa_batch* batch, uint32_t node, uint32_t instance, int32_t is_looping,
                               double loop_start, double loop_end);

/* BaseAudioContext::decode_audio_data_sync (src/context/base.rs:68-73 -> src/decoding.rs:15-54) for input the decoder
 * delivers as 16-bit PCM (WAV): interleaved i16 frames in, the sample conversion (sample / 32768) and
 * AudioBuffer::resample to the context's sample rate (src/buffer.rs:311-363) run on the device.  The resulting
 * AudioBuffer has the context's rate.  Half the host-to-device bytes of the f32 entry points; bit-identical to
 * converting and calling waa_buffer_resample on the host.  `_batch`: data = [n_instances][frames][n_channels]. */
waa_status waa_source_set_buffer_pcm16(waa_batch* batch, uint32_t node, uint32_t instance, const int16_t* interleaved,
                                       uint32_t n_channels, uint64_t frames, float sample_rate);
waa_status waa_source_set_buffer_pcm16_batch(waa_batch* batch, uint32_t node, const int16_t* data, uint32_t n_channels,
                                             uint64_t frames, float sample_rate);
/* ... followed by ConvolverNode::set_buffer with the decoded buffer (src/node/convolver.rs:259-317) */
waa_status waa_convolver_set_buffer_pcm16(waa_batch* batch, uint32_t node, const int16_t* interleaved, uint32_t n_channels,
                                          uint64_t frames, float sample_rate);

/* Impulse response shared by all instances. sample_rate must equal the context's
 * (NotSupportedError otherwise), n_channels in {1,2,4}. */
waa_status waa_convolver_set_buffer(waa_batch* batch, uint32_t node, const float* const* channels,
                                    uint32_t n_channels, uint64_t frames, float sample_rate);
waa_status waa_waveshaper_set_curve(waa_batch* batch, uint32_t node, const float* curve, uint32_t n);
/* OscillatorNode::set_periodic_wave(PeriodicWave::new(real, imag, disable_normalization))
 * (src/periodic_wave.rs:88-190, src/node/oscillator.rs:318-321): the 8192-point wavetable is generated on the
 * host; n >= 2 (IndexSizeError); real or imag may be NULL (zeros).  Switches the node to the custom type. */
waa_status waa_oscillator_set_periodic_wave(waa_batch* batch, uint32_t node, const float* real, const float* imag,
                                            uint32_t n, int32_t disable_normalization);
/* The same, for a caller that already holds the finished PeriodicWave (the reference's render side: the processor receives
 * the 8192-point wavetable through onmessage, src/node/oscillator.rs:487-493, and PeriodicWave keeps no coefficients,
 * src/periodic_wave.rs:72-74 — this is what the Rust shim forwards).  n must be 8192 (PERIODIC_WAVE_TABLE_LENGTH,
 * periodic_wave.rs:76); the table is used as it is (already normalised or not).  Switches the node to the custom type. */
#define WAA_PERIODIC_WAVE_TABLE_LENGTH 8192
waa_status waa_oscillator_set_wavetable(waa_batch* batch, uint32_t node, const float* table, uint32_t n);
/* IIRFilterOptions{feedforward, feedback} (src/node/iir_filter.rs:63-72), shared by all instances; required
 * before waa_render.  1..20 coefficients each (NotSupportedError otherwise), feedforward not all zero and
 * feedback[0] != 0 (InvalidStateError), iir_filter.rs:17-46. */
#define WAA_MAX_IIR_COEFFS 20
waa_status waa_iir_set_coefficients(waa_batch* batch, uint32_t node, const double* feedforward, uint32_t n_feedforward,
                                    const double* feedback, uint32_t n_feedback);

/* ---- AudioParam values (computed by the host-side automation) -------------------------- */

waa_status waa_set_param_const(waa_batch* batch, uint32_t node, uint32_t param, uint32_t instance, float value);
/* values: n_quanta * values_per_quantum f32, values_per_quantum in {1, 128}; applies to quanta
 * [quantum0, quantum0 + n_quanta); quanta not covered keep the constant value. */
waa_status waa_set_param_block(waa_batch* batch, uint32_t node, uint32_t param, uint32_t instance,
                               uint64_t quantum0, uint32_t n_quanta, uint32_t values_per_quantum,
                               const float* values);

/* ---- AudioParam automation (src/param.rs:386-600 control side, :796-1584 timeline) ---------------- */

/* AudioParamEventType, same order as src/param.rs:151-160 */
enum {
  WAA_EVENT_SET_VALUE = 0,                /* AudioParam::set_value                                      */
  WAA_EVENT_SET_VALUE_AT_TIME = 1,        /* set_value_at_time(value, start_time)                      */
  WAA_EVENT_LINEAR_RAMP = 2,              /* linear_ramp_to_value_at_time(value, end_time)             */
  WAA_EVENT_EXPONENTIAL_RAMP = 3,         /* exponential_ramp_to_value_at_time(value, end_time)        */
  WAA_EVENT_CANCEL_SCHEDULED_VALUES = 4,  /* cancel_scheduled_values(cancel_time)                      */
  WAA_EVENT_SET_TARGET = 5,               /* set_target_at_time(value, start_time, aux = time_constant) */
  WAA_EVENT_CANCEL_AND_HOLD = 6,          /* cancel_and_hold_at_time(cancel_time)                      */
  WAA_EVENT_SET_VALUE_CURVE = 7           /* set_value_curve_at_time(curve, start_time, aux = duration) */
};

/* Schedule an automation event on a param of a node (instance or WAA_ALL_INSTANCES), before waa_render.  The
 * events of a param are applied in call order, exactly like the reference's render thread receives them
 * (handle_incoming_event, param.rs:796-1047), and the timeline is evaluated per render quantum with the
 * reference's arithmetic (compute_buffer, param.rs:1483-1584): once on the HOST when the events are the same for every
 * instance (automation is control-side work), by a device kernel when instances carry different event lists on an a-rate
 * param (waa_timeline.hip); the resulting per-quantum / per-frame values take the place of waa_set_param_block.
 * Errors mirror the reference's panics: RangeError / TypeError for bad values and times, InvalidStateError for a
 * curve shorter than 2, NotSupportedError for events overlapping a value curve. */
waa_status waa_param_schedule_event(waa_batch* batch, uint32_t node, uint32_t param, uint32_t instance, int32_t type,
                                    float value, double time, double aux, const float* curve, uint32_t n_curve);

/* The same timeline as a stand-alone object (no batch, no device): what AudioParamProcessor::
 * compute_intrinsic_values (param.rs:730-735) is to the reference's unit tests. */
typedef struct waa_timeline waa_timeline;
waa_timeline* waa_timeline_create(float default_value, float min_value, float max_value, int32_t a_rate);
void waa_timeline_destroy(waa_timeline* timeline);
waa_status waa_timeline_event(waa_timeline* timeline, int32_t type, float value, double time, double aux,
                              const float* curve, uint32_t n_curve);
/* one block: writes 1 or `count` values to out (capacity >= count), returns how many */
uint32_t waa_timeline_compute(waa_timeline* timeline, double block_time, double dt, uint32_t count, float* out);
/* AudioParam::value(): the clamped intrinsic value at the beginning of the last computed block */
float waa_timeline_value(const waa_timeline* timeline);
/* The same timeline replayed ON THE DEVICE (the kernel that renders per-instance automation inside a batch,
 * AudioParamProcessor::compute_buffer param.rs:1498-1584 per render quantum from time 0): out[n_quanta * 128] with
 * single-valued slices replicated, lens[n_quanta] = 1 or 128 (the slice length AudioParamValues::get would return,
 * processor.rs:186-229).  Does not consume the timeline.  WAA_ERR_DEVICE without a HIP device. */
waa_status waa_timeline_render_device(const waa_timeline* timeline, uint32_t n_quanta, float sample_rate, float* out,
                                      uint8_t* lens);

/* ---- render ---------------------------------------------------------------------------- */

/* start_rendering_sync for every instance: renders all ceil(length/128) quanta. Asynchronous on the
 * batch's stream; waa_download / waa_download_all / waa_sync wait for it. */
waa_status waa_render(waa_batch* batch);
/* The same render with SUSPEND POINTS: OfflineAudioContext::suspend_sync (src/context/offline.rs:359-397) and the quantum loop
 * that honours it (RenderThread::render_audiobuffer_sync, src/render/thread.rs:277-294: "callback -> handle_control_messages ->
 * render the quantum").  Ranges are consecutive and start at quantum 0; after waa_render_range(b, q0, n) the render is suspended
 * in front of quantum q0 + n, and every control call made until the next range — waa_connect / waa_disconnect,
 * waa_set_param_const (AudioParam::set_value), waa_param_schedule_event, waa_source_start / stop — takes effect FROM THAT
 * QUANTUM, exactly where the reference's render thread would have handled the control message: a start time that has passed
 * becomes the block's time (audio_buffer_source.rs:516-518), an automation event enters the param's queue as it is then
 * (param.rs:796-1047).  The range that reaches the last quantum renders (asynchronously, like waa_render); the engine renders
 * node-major, so the earlier ranges only move the control clock — reading rendered audio in between (waa_download, the
 * analyser getters) is an InvalidStateError: a host that needs it renders a shorter batch of the graph so far.  Payload
 * setters (buffers, curves, coefficients) called at a suspend point apply to the whole render. */
waa_status waa_render_range(waa_batch* batch, uint64_t quantum0, uint32_t n_quanta);
/* AudioNode::connect_from_output_to_input / disconnect_dest_from_output_to_input (src/node/audio_node.rs:247-289, 341-420,
 * ControlMessage::ConnectNode / DisconnectNode -> Graph::add_edge / remove_edge, src/render/graph.rs:203-262) on a batch that
 * exists: before the first range they edit the graph, at a suspend point the connection lives from resp. until that quantum.
 * to_input as in waa_edge_desc (WAA_PARAM_INPUT(param) for an AudioParam).  Connecting twice is a no-op; disconnecting what is
 * not connected an InvalidAccessError.  Nodes cannot be added to a batch: a node a callback creates is declared in the graph
 * description and left unconnected (an unconnected node renders nothing anybody hears) until the callback's connect. */
waa_status waa_connect(waa_batch* batch, uint32_t from, uint32_t from_output, uint32_t to, uint32_t to_input);
waa_status waa_disconnect(waa_batch* batch, uint32_t from, uint32_t from_output, uint32_t to, uint32_t to_input);
/* "Same graph, new audio": put a planned (usually rendered) batch back in front of its render WITHOUT planning or allocating
 * again.  An OfflineAudioContext renders once (src/context/offline.rs:157-185: the renderer is taken); a host that renders the
 * same graph for one set of AudioBuffers after the other creates context after context — here it keeps the batch: after
 * waa_batch_rearm, waa_source_set_buffer_batch / waa_source_set_buffer_pcm16_batch upload INTO the device buffers the batch was
 * planned with (same channels, frames and rate: anything else is an InvalidStateError), waa_source_adopt_device accepts the
 * pointer it already reads, and waa_render renders from the initial state.  Everything else of the batch is frozen as before.
 * waa_render_sharded does this for the sub-batches of a job when waa_sharded_job.reuse_batches is set. */
waa_status waa_batch_rearm(waa_batch* batch);
waa_status waa_sync(waa_batch* batch);
/* rendered AudioBuffer channel of one instance -> dst[frames] (frames <= length) */
waa_status waa_download(waa_batch* batch, uint32_t instance, uint32_t channel, float* dst, uint64_t frames);
/* all instances: dst laid out [instance][channel][length_frames] */
waa_status waa_download_all(waa_batch* batch, float* dst);
/* all instances as interleaved 16-bit PCM, dst laid out [instance][frame][channel] — the layout of the `_pcm16_batch`
 * uploads: half the device-to-host bytes, which is what bounds the boundary when the caller holds host buffers (DESIGN.md
 * section 6).  sample * 32768 rounded to nearest, saturated to [-32768, 32767], NaN -> 0: the inverse of the decoder's
 * sample / 32768 (src/decoding.rs:15-54).  An extension: an OfflineAudioContext hands back f32 planes
 * (src/context/offline.rs:157-185); use it where the next step is a 16-bit file anyway. */
waa_status waa_download_all_pcm16(waa_batch* batch, int16_t* dst);

/* ---- N devices --------------------------------------------------------------------------
 * BASELINE config 4 (4096 contexts over the 8 GPUs of a node) as ONE call: what a host that holds every context's
 * AudioBuffer does instead of N x start_rendering_sync (src/context/offline.rs:157-185) — the reference renders one context
 * per thread on the CPU; here contexts are partitioned into contiguous ranges per device (no
 * collective, nothing crosses devices), each range is cut into `sub_batches` batches and every batch runs on its own host
 * thread: upload of one || render of another || download of a third on each device, one transfer per direction and device
 * at a time.  The graph is the same for every batch; what differs per context is configured by `setup`.            */
#define WAA_NO_NODE 0xFFFFFFFFu
/* called on the sub-batch's thread with the batch just created for the job's contexts [first, first + count): payloads
 * other than the streamed source buffer (impulse responses, curves), AudioParam values / automation, start / stop / loop.
 * A non-zero return aborts that sub-batch and is what waa_render_sharded returns (first error wins). */
typedef int32_t /* waa_status */ (*waa_shard_setup_fn)(waa_batch* batch, uint32_t first, uint32_t count, int32_t device, void* user);
/* called after the sub-batch's render, before its download: control-side pulls (the batched AnalyserNode getters) */
typedef int32_t /* waa_status */ (*waa_shard_pull_fn)(waa_batch* batch, uint32_t first, uint32_t count, int32_t device, void* user);
typedef struct waa_sharded_job {
  const waa_graph_desc* graph;
  uint32_t n_instances;          /* contexts of the whole job */
  uint32_t n_channels_out;
  uint64_t length_frames;
  float sample_rate;
  uint32_t n_devices;
  const int32_t* devices;        /* HIP device ordinals; device d renders the d-th contiguous range (waa_shard_range) */
  uint32_t sub_batches;          /* per device (>= 1; 8 keeps both directions of the link busy) */
  uint32_t source_node;          /* the AudioBufferSourceNode fed from host_in, or WAA_NO_NODE */
  const void* host_in;           /* f32 [n_instances][in_channels][in_frames], or (in_pcm16) i16 [n_instances][in_frames][in_channels] */
  uint32_t in_channels;
  int32_t in_pcm16;
  uint64_t in_frames;
  float in_sample_rate;
  int32_t out_pcm16;
  void* host_out;                /* f32 [n_instances][n_channels_out][length_frames], or (out_pcm16) i16 [n_instances][length_frames][n_channels_out] */
  waa_shard_setup_fn setup;      /* may be NULL */
  waa_shard_pull_fn pull;        /* may be NULL */
  void* user;
  uint32_t reuse_batches;        /* 1: `setup` configures every sub-batch the same way (nothing depends on `first`): a sub-batch that
                                  * has been downloaded is re-armed (waa_batch_rearm) for a later sub-batch of the same size instead
                                  * of being destroyed — the later one skips creation, setup and planning */
} waa_sharded_job;
/* Blocks until every context is rendered and downloaded; *seconds (may be NULL) = wall time.  Pinned host buffers let
 * the transfers run at link speed, and a device arena (waa_device_arena_reserve, once per process) keeps hipMalloc / hipFree —
 * which synchronise the device — out of the pipeline.  WAA_SHARD_TRACE=1: the phase timeline of every sub-batch on stderr.  Under one process per GPU (torch.distributed, MPI) every rank calls this with its own
 * device and its own slice of the contexts. */
waa_status waa_render_sharded(const waa_sharded_job* job, double* seconds);
/* Process-wide: how many sub-batches of ONE device may exist at a time inside waa_render_sharded (created, planned, uploading,
 * rendering or downloading).  Default 4: the pipeline upload(k + 1) || render(k) || download(k - 1) plus one being prepared —
 * device memory holds four sub-batches of a device's range, not all of them.  0 = no bound (everything resident at once, the
 * behaviour before round 5). */
waa_status waa_sharded_in_flight(uint32_t max_sub_batches_per_device);
/* the partition rule: contexts [*first, *end) of n_total belong to part `part` of `n_parts` (sizes differ by at most one) */
waa_status waa_shard_range(uint32_t n_total, uint32_t part, uint32_t n_parts, uint32_t* first, uint32_t* end);
/* device pointer + strides (in floats) of the rendered output, valid until destroy */
waa_status waa_output_device(waa_batch* batch, const float** device_ptr, uint64_t* instance_stride,
                             uint64_t* channel_stride);

/* ---- analyser (control-side pulls after the render; current_time = end of render) ------ */
waa_status waa_analyser_get_float_frequency_data(waa_batch* batch, uint32_t node, uint32_t instance, float* dst,
                                                 uint32_t n);
waa_status waa_analyser_get_byte_frequency_data(waa_batch* batch, uint32_t node, uint32_t instance, uint8_t* dst,
                                                uint32_t n);
waa_status waa_analyser_get_float_time_domain_data(waa_batch* batch, uint32_t node, uint32_t instance, float* dst,
                                                   uint32_t n);
waa_status waa_analyser_get_byte_time_domain_data(waa_batch* batch, uint32_t node, uint32_t instance, uint8_t* dst,
                                                  uint32_t n);
/* The same pulls for EVERY instance of the batch at once: one launch (one workgroup per context), one transfer.  Replaces
 * the loop "for ctx in contexts: analyser.get_float_frequency_data(&mut bins)" a caller of the reference runs after N
 * start_rendering_sync calls (src/node/analyser.rs:228-258 -> src/analysis.rs:261-401; BASELINE config 4 pulls once per
 * context).  dst is [n_instances][n]; row i receives what the per-instance call returns for instance i (min(n,
 * frequencyBinCount) resp. min(n, fftSize) values, the rest of the row untouched).  The per-instance calls above are
 * views into the same cached result. */
waa_status waa_analyser_get_float_frequency_data_batch(waa_batch* batch, uint32_t node, float* dst, uint32_t n);
waa_status waa_analyser_get_byte_frequency_data_batch(waa_batch* batch, uint32_t node, uint8_t* dst, uint32_t n);
waa_status waa_analyser_get_float_time_domain_data_batch(waa_batch* batch, uint32_t node, float* dst, uint32_t n);
waa_status waa_analyser_get_byte_time_domain_data_batch(waa_batch* batch, uint32_t node, uint8_t* dst, uint32_t n);

/* ---- introspection --------------------------------------------------------------------- */

/* Human-readable description of the launch plan the engine derived from the graph (one line per kernel
 * launch / alias / resource: chain fusion, streaming-biquad segments, convolver block size and partitions,
 * source schedules).  Builds the plan if necessary.  Writes at most cap-1 bytes + NUL; returns the full
 * length through *needed (may be NULL). */
waa_status waa_plan_describe(waa_batch* batch, char* buf, size_t cap, size_t* needed);

/* ---- input prep + pure helpers (no batch) ---------------------------------------------- */

/* The HRIR database of the HRTF panning model.  The reference embeds resources/IRC_1003_C.bin in the crate
 * (`include_bytes!`, src/node/panner.rs:55) and hands it to hrtf::HrirSphere::new; a shim passes the same bytes
 * here ONCE per process, before the first batch with an HRTF PannerNode is created (InvalidStateError otherwise).
 * Format (crate hrtf): "HRIR", then u32 LE sample rate, HRIR length, vertex count, index count; the triangle
 * indices (u32); per vertex x, y, z (f32) and the left and right HRIR (f32 each).  The bytes are copied.  HRIRs
 * for other context sample rates are derived on first use and cached per rate (panner.rs:39-60).
 * PARITY NOTE: the crate resamples the sphere with one chunk of rubato's SincFixedIn per impulse response; neither crate
 * is available to this repository, so at every context rate other than the sphere's own (44.1 kHz for IRC_1003_C; 48 kHz
 * is NOT it) the HRIRs — and through their length the panner's tail time — follow this library's written definition of that
 * pass, shared with the oracle and not pinned against the crates.  Header fields are bounds-checked
 * (HRIR length <= 1280 taps after resampling, non-zero counts). */
waa_status waa_hrtf_load_sphere(const void* data, uint64_t size);
/* HrirSphere::len() at a context sample rate (= PannerRenderer's HRTF tail in frames); 0 without a database */
uint32_t waa_hrtf_hrir_length(float sample_rate);
/* HrirSphere::sample_bilinear: the interpolated left / right HRIR (waa_hrtf_hrir_length(sample_rate) taps each) for
 * a direction in the sphere's coordinates (x, y, z).  Test hook; the render path does this per render quantum. */
void waa_hrtf_sample(float sample_rate, const float* dir, float* left, float* right);

/* AudioBuffer::resample (src/buffer.rs:311-363): returns the target length; writes at most dst_capacity
 * frames to dst (call with dst = NULL to query the length). */
uint64_t waa_buffer_resample(const float* src, uint64_t frames, float source_sample_rate, float target_sample_rate,
                             float* dst, uint64_t dst_capacity);
/* BiquadFilterNode::get_frequency_response (src/node/biquad_filter.rs:670-735) */
waa_status waa_biquad_frequency_response(int32_t type, float sample_rate, float frequency, float detune, float q,
                                         float gain, const float* frequency_hz, float* mag, float* phase,
                                         uint32_t n);
/* AudioScheduledSourceNode `ended` (src/node/scheduled_source.rs:44, src/render/processor.rs:53-58): WHEN the
 * reference dispatches the event for source `node` (AudioBufferSource, ConstantSource, Oscillator) of `instance`.
 *   *quantum >= 0        after rendering that render quantum (the renderer called send_ended_event() in it:
 *                        audio_buffer_source.rs:446-461,834-842, constant_source.rs:204-262, oscillator.rs:382-465);
 *   WAA_ENDED_AT_UNLOAD  when the graph is unloaded after the last quantum (render/thread.rs:398-411 -> before_drop:
 *                        the source had started, or its stop time had passed, but it never finished);
 *   WAA_ENDED_NEVER      not at all (never started).
 * Host-side scheduling only: valid before or after the render. */
#define WAA_ENDED_NEVER (-1)
#define WAA_ENDED_AT_UNLOAD (-2)
waa_status waa_source_ended(waa_batch* batch, uint32_t node, uint32_t instance, int64_t* quantum);

/* IIRFilterNode::get_frequency_response (src/node/iir_filter.rs:218-262); frequencies outside
 * [0, sample_rate/2] give NaN. */
waa_status waa_iir_frequency_response(const double* feedforward, uint32_t n_feedforward, const double* feedback,
                                      uint32_t n_feedback, float sample_rate, const float* frequency_hz, float* mag,
                                      float* phase, uint32_t n);

/* ---- measurement ----------------------------------------------------------------------- */

/* When enabled, every kernel launch of waa_render is bracketed by HIP events on the batch's
 * stream; after waa_sync the per-kernel totals can be read back. */
waa_status waa_profile_enable(waa_batch* batch, int32_t on);
int32_t waa_profile_count(waa_batch* batch);
/* name: kernel label, launches: number of launches since the last reset, total_ms: sum of durations */
waa_status waa_profile_get(waa_batch* batch, int32_t index, const char** name, uint64_t* launches,
                           double* total_ms);
waa_status waa_profile_reset(waa_batch* batch);

#ifdef __cplusplus
}
#endif
#endif /* WAA_HIP_H */
