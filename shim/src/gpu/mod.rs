//! Batched rendering of `OfflineAudioContext`s on an AMD MI355X through `libwaa_hip.so` (feature `hip`).
//!
//! ```ignore
//! let mut contexts: Vec<OfflineAudioContext> = (0..1024).map(|i| build_graph(i)).collect();   // same graph, own buffers / params
//! let buffers: Vec<AudioBuffer> = web_audio_api::gpu::start_rendering_sync_batch(&mut contexts, 0);
//! ```
//!
//! is `contexts.iter_mut().map(|c| c.start_rendering_sync()).collect()` with the render quanta of ALL contexts computed
//! by one device batch.  Nothing about how a graph is built changes: nodes, `connect`, `AudioParam` automation, `start` /
//! `stop`, `onended` handlers are the crate's own control-side code.  What changes is who consumes the control messages:
//!
//! 1. every context hands over its one-shot render thread (`OfflineAudioContext::gpu_take_renderer`, the first half of
//!    `start_rendering_sync`), which applies the queued control messages to its `Graph` exactly as the CPU render would
//!    (`RenderThread::gpu_prepare`) — and then does NOT render;
//! 2. the graph of context 0 is read out (`Graph::gpu_nodes`, `AudioProcessor::gpu_desc`) into a `waa_graph_desc`; the graphs
//!    of the other contexts must have the same shape (`GraphShape::same_shape`), their payloads (AudioBuffers, impulse
//!    responses, curves, schedules, AudioParam values) are forwarded per instance;
//! 3. `waa_render`, download of every context's AudioBuffer, the analysers' ring buffers refilled from the device, `ended`
//!    events dispatched in the order the reference would have, `complete` + state change (`gpu_complete`).
//!
//! Forwarded node kinds: every renderer of SURVEY section 8 — Gain, Biquad, IIRFilter, StereoPanner, Panner (equal-power and
//! HRTF, with the listener's params), Convolver, WaveShaper (all oversample types), Analyser, Delay (the writer / reader pair
//! folded into one library node), AudioBufferSource, ConstantSource, Oscillator (built-in types and custom PeriodicWaves: the
//! finished 8192-point table goes through `waa_oscillator_set_wavetable`), Destination, and every AudioParam.
//!
//! Whatever the device path does not cover — a processor without `gpu_desc` (worklets, media nodes, script processors),
//! contexts with scheduled suspensions, graphs of different shape, or a graph the library refuses with status 4 — is rendered
//! by the render threads taken in step 1 on the CPU, as if this module did not exist: they hold the fully applied graph.
//!
//! STATUS: UNCOMPILED SKETCH.  Written against web-audio-api 1.6.0 and include/waa_hip.h without a Rust toolchain at hand; it
//! has never been through rustc, so borrow / lifetime errors are to be expected on the first build.  What IS checked without
//! one (tests/test_shim_patch.py of the engine's repository): the patch applies, every `extern "C"` declaration of ffi.rs
//! agrees with the header in parameter count AND type, the `#[repr(C)]` structs agree field by field, and every `ffi::waa_*`
//! this file calls is declared.  `sh oracle/build_ref.sh` builds it where cargo exists and compares both paths.

mod ffi;

use std::collections::HashMap;

use crate::analysis::AnalyserRingBuffer;
use crate::buffer::AudioBuffer;
use crate::context::{AudioNodeId, BaseAudioContext, OfflineAudioContext};
use crate::events::{EventDispatch, EventLoop};
use crate::node::{ChannelCountMode, ChannelInterpretation};
use crate::param::{AudioParamProcessor, GpuParamValues};
use crate::render::graph::Graph;
use crate::render::RenderThread;
use crate::RENDER_QUANTUM_SIZE;

/// What a processor reports about itself (`AudioProcessor::gpu_desc`): its kind and the state the control messages left in
/// it.  Param ids are the ids of the AudioParam nodes (`AudioNodeId::from(&AudioParamId).0`).
pub(crate) enum GpuNode<'a> {
    Destination,
    Param(&'a AudioParamProcessor),
    Gain {
        gain: u64,
    },
    Biquad {
        type_: u32,
        frequency: u64,
        detune: u64,
        q: u64,
        gain: u64,
    },
    StereoPanner {
        pan: u64,
    },
    BufferSource {
        buffer: Option<&'a AudioBuffer>,
        start_time: f64,
        stop_time: f64,
        offset: f64,
        duration: f64,
        is_looping: bool,
        loop_start: f64,
        loop_end: f64,
        playback_rate: u64,
        detune: u64,
    },
    ConstantSource {
        offset: u64,
        start_time: f64,
        stop_time: f64,
    },
    Oscillator {
        type_: u32,
        frequency: u64,
        detune: u64,
        start_time: f64,
        stop_time: f64,
        /// `Some` = a custom PeriodicWave: the finished table the renderer received (oscillator.rs:487-493)
        wavetable: Option<&'a [f32]>,
    },
    Convolver {
        impulse: Option<(&'a AudioBuffer, bool)>,
    },
    WaveShaper {
        oversample: u32,
        curve: Option<&'a [f32]>,
    },
    IirFilter {
        feedforward: Vec<f64>,
        feedback: Vec<f64>,
    },
    Analyser {
        ring_buffer: &'a AnalyserRingBuffer,
    },
    /// One half of a DelayNode (delay.rs:316-357 registers a writer and a reader processor that share a ring): the library has ONE
    /// node per DelayNode and breaks cycles itself (graph.rs:323-487 restated in its planner), so the writer half becomes that node
    /// and the reader half — the one with the delayTime param — is folded into it.  `ring` pairs the halves.
    Delay {
        ring: usize,
        delay_time: Option<u64>,
        /// capacity of the ring in render quanta = num_quanta + 1 (delay.rs:300-302)
        capacity: usize,
    },
    /// PannerRenderer (panner.rs:667-683): six a-rate params, the distance / cone model, equal-power or HRTF
    Panner {
        position: [u64; 3],
        orientation: [u64; 3],
        distance_model: u32,
        ref_distance: f64,
        max_distance: f64,
        rolloff_factor: f64,
        cone_inner_angle: f64,
        cone_outer_angle: f64,
        cone_outer_gain: f64,
        hrtf: bool,
    },
}

impl GpuNode<'_> {
    fn kind(&self) -> Option<u32> {
        Some(match self {
            GpuNode::Destination => ffi::WAA_NODE_DESTINATION,
            GpuNode::Param(_) => return None,
            GpuNode::Gain { .. } => ffi::WAA_NODE_GAIN,
            GpuNode::Biquad { .. } => ffi::WAA_NODE_BIQUAD,
            GpuNode::StereoPanner { .. } => ffi::WAA_NODE_STEREO_PANNER,
            GpuNode::BufferSource { .. } => ffi::WAA_NODE_BUFFER_SOURCE,
            GpuNode::ConstantSource { .. } => ffi::WAA_NODE_CONSTANT_SOURCE,
            GpuNode::Oscillator { .. } => ffi::WAA_NODE_OSCILLATOR,
            GpuNode::Convolver { .. } => ffi::WAA_NODE_CONVOLVER,
            GpuNode::WaveShaper { .. } => ffi::WAA_NODE_WAVESHAPER,
            GpuNode::IirFilter { .. } => ffi::WAA_NODE_IIR_FILTER,
            GpuNode::Analyser { .. } => ffi::WAA_NODE_ANALYSER,
            GpuNode::Delay { delay_time: None, .. } => ffi::WAA_NODE_DELAY,
            GpuNode::Delay { delay_time: Some(_), .. } => return None, // (the reader half: GraphShape::from_graph folds it)
            GpuNode::Panner { .. } => ffi::WAA_NODE_PANNER,
        })
    }

    /// (AudioParam node id, param index of include/waa_hip.h) of the params this node owns
    fn params(&self) -> Vec<(u64, u32)> {
        match *self {
            GpuNode::Gain { gain } => vec![(gain, 0)],
            GpuNode::Biquad {
                frequency,
                detune,
                q,
                gain,
                ..
            } => vec![(frequency, 0), (detune, 1), (q, 2), (gain, 3)],
            GpuNode::StereoPanner { pan } => vec![(pan, 0)],
            GpuNode::BufferSource {
                playback_rate,
                detune,
                ..
            } => vec![(playback_rate, 0), (detune, 1)],
            GpuNode::ConstantSource { offset, .. } => vec![(offset, 0)],
            GpuNode::Oscillator {
                frequency, detune, ..
            } => vec![(frequency, 0), (detune, 1)],
            // WAA_PARAM_PANNER_POSITION_X .. ORIENTATION_Z = 0 .. 5 (the listener's nine follow at 6 .. 14, forward_payloads)
            GpuNode::Panner {
                position, orientation, ..
            } => vec![
                (position[0], 0),
                (position[1], 1),
                (position[2], 2),
                (orientation[0], 3),
                (orientation[1], 4),
                (orientation[2], 5),
            ],
            _ => vec![],
        }
    }
}

/// Why a batch was rendered on the CPU instead (returned for logging / tests; the result is the same AudioBuffers)
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Fallback {
    /// a context has suspensions scheduled (`suspend` / `suspend_sync`)
    Suspended,
    /// a processor without a device description: its `AudioProcessor::name()`
    UnsupportedNode(String),
    /// context `usize` does not have the graph of context 0
    DifferentShape(usize),
    /// the library refused the graph (status 4) or failed: its message
    Library(String),
}

/// The graph of one context as the library wants it: nodes in id order (destination first: AudioNodeId(0)), edges with
/// param inputs folded into their owning node.
#[derive(Debug, Clone, PartialEq)]
struct GraphShape {
    nodes: Vec<ffi::waa_node_desc>,
    edges: Vec<ffi::waa_edge_desc>,
    /// AudioNodeId -> index into `nodes` (only non-param nodes)
    index_of: HashMap<u64, u32>,
    /// AudioParam node id -> (owner index, param index)
    param_owner: HashMap<u64, (u32, u32)>,
    /// indices of the PannerNodes: the AudioListener's nine params (node ids 2..=10) are forwarded through each of them
    panners: Vec<u32>,
}

const LISTENER_AND_ITS_PARAMS: std::ops::RangeInclusive<u64> = 1..=10; // LISTENER_NODE_ID = 1, LISTENER_PARAM_IDS = 2..=10 (context/mod.rs)

impl GraphShape {
    fn from_graph(graph: &Graph, sample_rate: f32) -> Result<Self, Fallback> {
        let views = graph.gpu_nodes();
        let mut nodes = Vec::new();
        let mut index_of = HashMap::new();
        let mut param_owner = HashMap::new();
        let mut panners = Vec::new();
        let mut delay_of_ring: HashMap<usize, u32> = HashMap::new(); // ring -> index of the writer half's node
        let mut delay_halves: std::collections::HashSet<u64> = std::collections::HashSet::new();
        // pass 1: nodes that are not AudioParams, in id order (id 0 is the destination: the library wants it first)
        for v in &views {
            if LISTENER_AND_ITS_PARAMS.contains(&v.id.0) {
                continue; // the AudioListener is no node of the library: its params go through the PannerNodes (forward_payloads)
            }
            let desc = v
                .processor()
                .gpu_desc()
                .ok_or_else(|| Fallback::UnsupportedNode(v.processor().name().to_string()))?;
            if let GpuNode::Delay {
                ring,
                delay_time: Some(param_id),
                ..
            } = &desc
            {
                // the reader half: registered after its writer (delay.rs:316-317: the outer `register` takes its id first)
                let &idx = delay_of_ring
                    .get(ring)
                    .ok_or_else(|| Fallback::UnsupportedNode("DelayReader without its DelayWriter".into()))?;
                index_of.insert(v.id.0, idx);
                param_owner.insert(*param_id, (idx, 0)); // WAA_PARAM_DELAY_DELAY_TIME
                delay_halves.insert(v.id.0);
                continue;
            }
            let Some(kind) = desc.kind() else { continue };
            let (count, mode, interp) = v.channel_config();
            let mut d = ffi::waa_node_desc {
                kind,
                channel_count: count as u32,
                channel_count_mode: match mode {
                    ChannelCountMode::Max => ffi::WAA_COUNT_MODE_MAX,
                    ChannelCountMode::ClampedMax => ffi::WAA_COUNT_MODE_CLAMPED_MAX,
                    ChannelCountMode::Explicit => ffi::WAA_COUNT_MODE_EXPLICIT,
                },
                channel_interpretation: match interp {
                    ChannelInterpretation::Speakers => ffi::WAA_INTERP_SPEAKERS,
                    ChannelInterpretation::Discrete => ffi::WAA_INTERP_DISCRETE,
                },
                ..Default::default()
            };
            match &desc {
                GpuNode::Biquad { type_, .. } => d.i[0] = *type_ as i32,
                GpuNode::WaveShaper { oversample, .. } => d.i[0] = *oversample as i32,
                GpuNode::Oscillator { type_, .. } => d.i[0] = *type_ as i32,
                // the library normalises the response itself (convolver.rs:59-90): hand over the flag, not a scaled copy
                GpuNode::Convolver { impulse } => d.i[0] = i32::from(!impulse.as_ref().map(|(_, n)| *n).unwrap_or(true)),
                // the control side keeps fftSize & co.; the device keeps the longest window the getters can ask for
                GpuNode::Analyser { .. } => {
                    d.i[0] = 32768;
                    d.d[0] = 0.8;
                    d.d[1] = -100.;
                    d.d[2] = -30.;
                }
                // max_delay_time: the library derives num_quanta = ceil(max_delay * sample_rate / 128) like delay.rs:300-301; half a
                // quantum below capacity - 1 quanta gives exactly the reference's ring
                GpuNode::Delay { capacity, .. } => {
                    d.d[0] = (capacity.saturating_sub(1) as f64 - 0.5).max(0.25) * RENDER_QUANTUM_SIZE as f64 / sample_rate as f64
                }
                GpuNode::Panner {
                    distance_model,
                    ref_distance,
                    max_distance,
                    rolloff_factor,
                    cone_inner_angle,
                    cone_outer_angle,
                    cone_outer_gain,
                    hrtf,
                    ..
                } => {
                    d.i[0] = i32::from(*hrtf); // WAA_PANNING_EQUALPOWER / WAA_PANNING_HRTF
                    d.i[1] = *distance_model as i32;
                    d.d[0] = *ref_distance;
                    d.d[1] = *max_distance;
                    d.d[2] = *rolloff_factor;
                    d.d[3] = *cone_inner_angle;
                    d.d[4] = *cone_outer_angle;
                    d.d[5] = *cone_outer_gain;
                }
                _ => {}
            }
            let idx = nodes.len() as u32;
            match &desc {
                GpuNode::Delay { ring, .. } => {
                    delay_of_ring.insert(*ring, idx);
                    delay_halves.insert(v.id.0);
                }
                GpuNode::Panner { hrtf, .. } => {
                    if *hrtf {
                        load_hrtf_sphere_once()?;
                    }
                    panners.push(idx);
                }
                _ => {}
            }
            for (param_id, param_index) in desc.params() {
                param_owner.insert(param_id, (idx, param_index));
            }
            index_of.insert(v.id.0, idx);
            nodes.push(d);
        }
        if index_of.get(&0) != Some(&0) {
            return Err(Fallback::UnsupportedNode("graph without a destination at id 0".into()));
        }
        // pass 2: edges.  An AudioParam node has one hidden edge to its owner (other_index == usize::MAX, graph.rs:527):
        // dropped; an edge INTO an AudioParam node is an edge into the owner with to_input = WAA_PARAM_INPUT(index).
        let mut edges = Vec::new();
        for v in &views {
            let Some(&from) = index_of.get(&v.id.0) else { continue };
            for (output, other, input) in v.edges() {
                if input == usize::MAX {
                    continue;
                }
                if let Some(&to) = index_of.get(&other.0) {
                    if to == from && v.id != other && delay_halves.contains(&v.id.0) && delay_halves.contains(&other.0) {
                        continue; // the writer -> reader edge of one DelayNode (delay.rs:365): inside the library's node
                    }
                    edges.push(ffi::waa_edge_desc {
                        from,
                        from_output: output as u32,
                        to,
                        to_input: input as u32,
                    });
                } else if let Some(&(owner, param)) = param_owner.get(&other.0) {
                    edges.push(ffi::waa_edge_desc {
                        from,
                        from_output: output as u32,
                        to: owner,
                        to_input: ffi::waa_param_input(param),
                    });
                } else if LISTENER_AND_ITS_PARAMS.contains(&other.0) {
                    // an AudioListener param driven from the graph only matters when a PannerNode listens
                    // (panner.rs:735-760: the listener's params are read through the panner's inputs)
                    if !panners.is_empty() && other.0 != 1 {
                        return Err(Fallback::UnsupportedNode("AudioListener param with an audio-rate input".into()));
                    }
                } else {
                    return Err(Fallback::UnsupportedNode(format!("edge into unknown node {}", other.0)));
                }
            }
        }
        // (AudioParam nodes that belong to no forwarded node cannot occur: every param is registered by its owner)
        Ok(Self {
            nodes,
            edges,
            index_of,
            param_owner,
            panners,
        })
    }

    fn same_shape(&self, other: &Self) -> bool {
        self.nodes == other.nodes && self.edges == other.edges
    }
}

/// The HRIR database the crate embeds (panner.rs:55) handed to the library once per process
fn load_hrtf_sphere_once() -> Result<(), Fallback> {
    static LOADED: std::sync::OnceLock<Result<(), String>> = std::sync::OnceLock::new();
    LOADED
        .get_or_init(|| {
            let bytes: &[u8] = include_bytes!("../../resources/IRC_1003_C.bin");
            let st = unsafe { ffi::waa_hrtf_load_sphere(bytes.as_ptr().cast(), bytes.len() as u64) };
            if st == ffi::WAA_OK {
                Ok(())
            } else {
                Err(ffi::last_error())
            }
        })
        .clone()
        .map_err(Fallback::Library)
}

fn check(status: i32) -> Result<(), Fallback> {
    if status == ffi::WAA_OK {
        Ok(())
    } else {
        Err(Fallback::Library(ffi::last_error()))
    }
}

fn channel_ptrs(buffer: &AudioBuffer) -> Vec<*const f32> {
    (0..buffer.number_of_channels())
        .map(|c| buffer.get_channel_data(c).as_ptr())
        .collect()
}

/// AudioParam -> the library: a constant, or value blocks in runs of equal slice length (1 or 128 values per quantum)
fn forward_param(
    batch: *mut ffi::waa_batch,
    inst: u32,
    owner: u32,
    param: u32,
    p: &AudioParamProcessor,
    has_graph_input: bool,
    n_quanta: usize,
    sample_rate: f32,
) -> Result<(), Fallback> {
    // without an audio-rate input the reference clamps the intrinsic value itself (param.rs:739-795); with one the library
    // adds the input first and clamps the sum, like mix_to_output
    let fix = |v: f32| if has_graph_input { v } else { p.gpu_clamped(v) };
    match p.gpu_values(n_quanta, sample_rate) {
        GpuParamValues::Constant(v) => check(unsafe { ffi::waa_set_param_const(batch, owner, param, inst, fix(v)) }),
        GpuParamValues::PerQuantum { lens, mut values } => {
            values.iter_mut().for_each(|v| *v = fix(*v));
            let (mut q0, mut offset) = (0usize, 0usize);
            while q0 < lens.len() {
                let len = lens[q0] as usize;
                let mut q1 = q0;
                while q1 < lens.len() && lens[q1] as usize == len {
                    q1 += 1;
                }
                let run = &values[offset..offset + (q1 - q0) * len];
                check(unsafe {
                    ffi::waa_set_param_block(batch, owner, param, inst, q0 as u64, (q1 - q0) as u32, len as u32, run.as_ptr())
                })?;
                offset += run.len();
                q0 = q1;
            }
            Ok(())
        }
    }
}

/// Everything instance `inst` holds that is not topology: AudioBuffers, responses, curves, coefficients, schedules, params.
fn forward_payloads(
    batch: *mut ffi::waa_batch,
    inst: u32,
    graph: &Graph,
    shape: &GraphShape,
    n_quanta: usize,
    sample_rate: f32,
) -> Result<(), Fallback> {
    let param_has_input: std::collections::HashSet<(u32, u32)> = shape
        .edges
        .iter()
        .filter(|e| e.to_input & 0x8000_0000 != 0)
        .map(|e| (e.to, e.to_input & 0x7fff_ffff))
        .collect();
    for v in graph.gpu_nodes() {
        let Some(desc) = v.processor().gpu_desc() else { continue };
        if let GpuNode::Param(p) = &desc {
            if let Some(&(owner, param)) = shape.param_owner.get(&v.id.0) {
                forward_param(batch, inst, owner, param, p, param_has_input.contains(&(owner, param)), n_quanta, sample_rate)?;
            } else if (2..=10).contains(&v.id.0) {
                // LISTENER_AUDIO_PARAM_IDS (context/mod.rs:30-40: position x y z, forward x y z, up x y z) =
                // WAA_PARAM_LISTENER_POSITION_X .. UP_Z = 6 .. 14, addressed through every PannerNode
                for &panner in &shape.panners {
                    forward_param(batch, inst, panner, 6 + (v.id.0 - 2) as u32, p, false, n_quanta, sample_rate)?;
                }
            }
            continue;
        }
        let Some(&node) = shape.index_of.get(&v.id.0) else { continue };
        match desc {
            GpuNode::BufferSource {
                buffer,
                start_time,
                stop_time,
                offset,
                duration,
                is_looping,
                loop_start,
                loop_end,
                ..
            } => {
                if let Some(b) = buffer {
                    let ptrs = channel_ptrs(b);
                    check(unsafe {
                        ffi::waa_source_set_buffer(batch, node, inst, ptrs.as_ptr(), ptrs.len() as u32, b.length() as u64, b.sample_rate())
                    })?;
                }
                // f64::MAX = "never started" / "never stopped" in the renderer (audio_buffer_source.rs:313-318): the library's
                // defaults are the same, only real schedules are sent
                if start_time != f64::MAX {
                    // (duration f64::MAX = none, on both sides)
                    check(unsafe { ffi::waa_source_start(batch, node, inst, start_time, offset, duration) })?;
                }
                if stop_time != f64::MAX {
                    check(unsafe { ffi::waa_source_stop(batch, node, inst, stop_time) })?;
                }
                check(unsafe { ffi::waa_source_set_loop(batch, node, inst, i32::from(is_looping), loop_start, loop_end) })?;
            }
            GpuNode::ConstantSource {
                start_time, stop_time, ..
            } => {
                if start_time != f64::MAX {
                    check(unsafe { ffi::waa_source_start(batch, node, inst, start_time, 0., f64::MAX) })?;
                }
                if stop_time != f64::MAX {
                    check(unsafe { ffi::waa_source_stop(batch, node, inst, stop_time) })?;
                }
            }
            GpuNode::Oscillator {
                start_time,
                stop_time,
                wavetable,
                ..
            } => {
                // a custom PeriodicWave is kept once per batch like curves and impulse responses (instance 0's)
                if let (Some(table), 0) = (wavetable, inst) {
                    check(unsafe { ffi::waa_oscillator_set_wavetable(batch, node, table.as_ptr(), table.len() as u32) })?;
                }
                if start_time != f64::MAX {
                    check(unsafe { ffi::waa_source_start(batch, node, inst, start_time, 0., f64::MAX) })?;
                }
                if stop_time != f64::MAX {
                    check(unsafe { ffi::waa_source_stop(batch, node, inst, stop_time) })?;
                }
            }
            // payloads the library keeps once per batch: taken from instance 0 (the contexts of a batch share them;
            // a batch whose contexts carry different responses / curves is a different batch)
            GpuNode::Convolver { impulse: Some((b, _)) } if inst == 0 => {
                let ptrs = channel_ptrs(b);
                check(unsafe {
                    ffi::waa_convolver_set_buffer(batch, node, ptrs.as_ptr(), ptrs.len() as u32, b.length() as u64, b.sample_rate())
                })?;
            }
            GpuNode::WaveShaper { curve: Some(c), .. } if inst == 0 => {
                check(unsafe { ffi::waa_waveshaper_set_curve(batch, node, c.as_ptr(), c.len() as u32) })?;
            }
            GpuNode::IirFilter { feedforward, feedback } if inst == 0 => {
                check(unsafe {
                    ffi::waa_iir_set_coefficients(
                        batch,
                        node,
                        feedforward.as_ptr(),
                        feedforward.len() as u32,
                        feedback.as_ptr(),
                        feedback.len() as u32,
                    )
                })?;
            }
            _ => {}
        }
    }
    Ok(())
}

struct Taken {
    renderer: RenderThread,
    event_loop: EventLoop,
}

struct BatchGuard(*mut ffi::waa_batch);
impl Drop for BatchGuard {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { ffi::waa_batch_destroy(self.0) }
        }
    }
}

/// The device render of a batch whose render threads have applied their control messages; `Err` = nothing was consumed,
/// the caller renders on the CPU.
/// Round 6 — contexts WITH suspensions on the device (UNCOMPILED SKETCH like the rest of this module; until it is wired in,
/// `start_rendering_sync_batch_with_report` keeps returning `Fallback::Suspended` for them).
///
/// `render_audiobuffer_sync` (thread.rs:277-294) runs, in front of quantum q: the callback registered for q, then
/// `handle_control_messages`, then the quantum.  The library's `waa_render_range` is that loop's skeleton without the quanta: it
/// moves a control clock, and whatever the shim forwards between two ranges takes effect from the suspended quantum on.  So per
/// suspend point q (the union over the batch's contexts — they must agree, like their graphs):
///   1. `ffi::waa_render_range(batch, prev, q - prev)`;
///   2. every context's callback runs (`OfflineAudioContext` is `&mut`, state Suspended / Running around it, offline.rs:383-388);
///   3. every render thread handles the messages it produced (`RenderThread::gpu_prepare`), and the `Graph` is READ again
///      (`GraphShape::from_graph`): edges that appeared -> `waa_connect`, edges that vanished -> `waa_disconnect`; nodes that
///      appeared must have been declared up front — the shim builds the `waa_graph_desc` from the graph AFTER the last callback
///      of a dry pass over the callbacks' node creations (a callback that creates nodes depending on rendered audio is the one
///      case that stays on the CPU: `Fallback::Suspended`);
///   4. the params' values for the quanta up to the next suspend point come from the crate's own processor
///      (`AudioParamProcessor::gpu_values`, which now holds the events the callback scheduled) -> `waa_set_param_block(q, n)`:
///      no automation semantics restated, as in the unsuspended path;
///   5. scheduled sources whose renderer state changed (start / stop times, loop points) -> `waa_source_start` / `_stop`
///      with the times the renderer holds; the library clamps a time that has passed to the block (audio_buffer_source.rs:516-518).
/// The last range renders.  A callback that pulls an AnalyserNode is served by a second, shorter batch of the graph so far
/// (`web-audio-api-rs_amd/api.py::OfflineAudioContext::_prefix_render` does exactly that for the Python mirror).
#[allow(dead_code)]
fn render_with_suspends(batch: *mut ffi::waa_batch, suspend_quanta: &[usize], n_quanta: usize, mut at_suspend: impl FnMut(usize) -> Result<(), Fallback>) -> Result<(), Fallback> {
    let mut prev = 0usize;
    for &q in suspend_quanta {
        if q == 0 || q >= n_quanta {
            continue; // (a suspend in front of quantum 0 edits the graph before the batch exists)
        }
        check(unsafe { ffi::waa_render_range(batch, prev as u64, (q - prev) as u32) })?;
        at_suspend(q)?; // steps 2-5 above
        prev = q;
    }
    check(unsafe { ffi::waa_render_range(batch, prev as u64, (n_quanta - prev) as u32) })
}

fn render_on_device(
    contexts: &[OfflineAudioContext],
    taken: &mut [Taken],
    device: i32,
) -> Result<Vec<AudioBuffer>, Fallback> {
    let length = contexts[0].length();
    let sample_rate = taken[0].renderer.gpu_sample_rate();
    let channels = taken[0].renderer.gpu_number_of_channels();
    let n_quanta = length.div_ceil(RENDER_QUANTUM_SIZE);
    let mut shapes = Vec::with_capacity(taken.len());
    for (i, t) in taken.iter_mut().enumerate() {
        let graph = t
            .renderer
            .gpu_prepare()
            .ok_or_else(|| Fallback::UnsupportedNode("render thread without a graph".into()))?;
        let shape = GraphShape::from_graph(graph, sample_rate)?;
        if i > 0 && !shape.same_shape(&shapes[0]) {
            return Err(Fallback::DifferentShape(i));
        }
        if contexts[i].length() != length || t.renderer.gpu_sample_rate() != sample_rate || t.renderer.gpu_number_of_channels() != channels {
            return Err(Fallback::DifferentShape(i));
        }
        shapes.push(shape);
    }
    let shape = &shapes[0];
    let desc = ffi::waa_graph_desc {
        n_nodes: shape.nodes.len() as u32,
        nodes: shape.nodes.as_ptr(),
        n_edges: shape.edges.len() as u32,
        edges: shape.edges.as_ptr(),
    };
    let mut raw = std::ptr::null_mut();
    check(unsafe { ffi::waa_batch_create(&desc, taken.len() as u32, channels as u32, length as u64, sample_rate, device, &mut raw) })?;
    let batch = BatchGuard(raw);
    for (i, t) in taken.iter_mut().enumerate() {
        // (the messages are applied already: this only hands the graph out again)
        let graph = t.renderer.gpu_prepare().unwrap();
        forward_payloads(batch.0, i as u32, graph, &shapes[i], n_quanta, sample_rate)?;
    }
    // status 4 (a graph outside the device path) shows at the first render: nothing of the contexts has been consumed
    check(unsafe { ffi::waa_render(batch.0) })?;
    check(unsafe { ffi::waa_sync(batch.0) })?;

    let mut results = Vec::with_capacity(taken.len());
    for (i, t) in taken.iter_mut().enumerate() {
        // the AudioBuffer of context i (thread.rs:260-302 collects `length` frames per channel)
        let mut planes = vec![vec![0f32; length]; channels];
        for (c, plane) in planes.iter_mut().enumerate() {
            check(unsafe { ffi::waa_download(batch.0, i as u32, c as u32, plane.as_mut_ptr(), length as u64) })?;
        }
        results.push(AudioBuffer::from(planes, sample_rate));
        let graph = t.renderer.gpu_prepare().unwrap();
        let mut ended: Vec<(i64, u64)> = Vec::new();
        for v in graph.gpu_nodes() {
            let Some(&node) = shapes[i].index_of.get(&v.id.0) else { continue };
            match v.processor().gpu_desc() {
                // AnalyserNode getters read the last fftSize frames of this ring (analysis.rs:354-401): refill it with what
                // the renderer would have written, the mono down-mix of the node's input
                Some(GpuNode::Analyser { ring_buffer }) => {
                    let mut last = vec![0f32; 32768];
                    check(unsafe {
                        ffi::waa_analyser_get_float_time_domain_data(batch.0, node, i as u32, last.as_mut_ptr(), last.len() as u32)
                    })?;
                    for chunk in last.chunks(RENDER_QUANTUM_SIZE) {
                        ring_buffer.write(chunk);
                    }
                }
                // `ended`: when the reference would have sent it (scheduled_source.rs:44; thread.rs:290 after the quantum,
                // thread.rs:299-300 at unload)
                Some(GpuNode::BufferSource { .. }) | Some(GpuNode::ConstantSource { .. }) | Some(GpuNode::Oscillator { .. }) => {
                    let mut q = 0i64;
                    check(unsafe { ffi::waa_source_ended(batch.0, node, i as u32, &mut q) })?;
                    if q != ffi::WAA_ENDED_NEVER {
                        ended.push((if q == ffi::WAA_ENDED_AT_UNLOAD { i64::MAX } else { q }, v.id.0));
                    }
                }
                _ => {}
            }
        }
        ended.sort();
        for (_, id) in ended {
            let _ = contexts[i].base().send_event(EventDispatch::ended(AudioNodeId(id)));
        }
        t.event_loop.handle_pending_events();
    }
    Ok(results)
}

/// `contexts.iter_mut().map(|c| c.start_rendering_sync()).collect()`, rendered as one device batch on HIP device `device`
/// when the graphs allow it and by the crate's own render threads otherwise.
///
/// # Panics
///
/// Like `start_rendering_sync`: when a context has been rendered before.
pub fn start_rendering_sync_batch(contexts: &mut [OfflineAudioContext], device: i32) -> Vec<AudioBuffer> {
    start_rendering_sync_batch_with_report(contexts, device).0
}

/// The same, and why the CPU rendered it when it did.
pub fn start_rendering_sync_batch_with_report(
    contexts: &mut [OfflineAudioContext],
    device: i32,
) -> (Vec<AudioBuffer>, Option<Fallback>) {
    if contexts.is_empty() {
        return (Vec::new(), None);
    }
    let mut taken = Vec::with_capacity(contexts.len());
    for (i, ctx) in contexts.iter_mut().enumerate() {
        match ctx.gpu_take_renderer() {
            Some((renderer, event_loop)) => taken.push(Taken { renderer, event_loop }),
            None => {
                // a context with suspensions: give back what was taken (render it now, on the CPU) and do the rest as before
                let mut out: Vec<AudioBuffer> = Vec::with_capacity(contexts.len());
                for (j, t) in taken.drain(..).enumerate() {
                    out.push(contexts[j].gpu_render_on_cpu(t.renderer, &t.event_loop));
                }
                for ctx in contexts[i..].iter_mut() {
                    out.push(ctx.start_rendering_sync());
                }
                return (out, Some(Fallback::Suspended));
            }
        }
    }
    match render_on_device(contexts, &mut taken, device) {
        Ok(buffers) => {
            for ((ctx, t), buffer) in contexts.iter_mut().zip(taken.iter()).zip(buffers.iter()) {
                ctx.gpu_complete(buffer.clone(), &t.event_loop);
            }
            // (the render threads are dropped here without unload_graph: their processors never ran, before_drop would
            // fire `ended` for sources by the wrong rule — the events were dispatched above)
            (buffers, None)
        }
        Err(why) => {
            log::warn!("web_audio_api::gpu: batch rendered on the CPU: {why:?}");
            let out = contexts
                .iter_mut()
                .zip(taken)
                .map(|(ctx, t)| ctx.gpu_render_on_cpu(t.renderer, &t.event_loop))
                .collect();
            (out, Some(why))
        }
    }
}
