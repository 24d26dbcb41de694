//! Raw bindings of `libwaa_hip.so` (include/waa_hip.h of the MI355X engine).  Hand-written; `bindgen` over the header
//! gives the same items.  Only what `super` uses is declared.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct waa_batch {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug, PartialEq)]
pub struct waa_node_desc {
    pub kind: u32,
    pub channel_count: u32,
    pub channel_count_mode: u32,
    pub channel_interpretation: u32,
    pub i: [i32; 4],
    pub d: [f64; 8],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct waa_edge_desc {
    pub from: u32,
    pub from_output: u32,
    pub to: u32,
    pub to_input: u32,
}

#[repr(C)]
pub struct waa_graph_desc {
    pub n_nodes: u32,
    pub nodes: *const waa_node_desc,
    pub n_edges: u32,
    pub edges: *const waa_edge_desc,
}

pub const WAA_OK: i32 = 0;
pub const WAA_ERR_OUT_OF_SCOPE: i32 = 4;
pub const WAA_ALL_INSTANCES: u32 = 0xFFFF_FFFF;

pub const WAA_NODE_DESTINATION: u32 = 0;
pub const WAA_NODE_BUFFER_SOURCE: u32 = 1;
pub const WAA_NODE_BIQUAD: u32 = 2;
pub const WAA_NODE_GAIN: u32 = 3;
pub const WAA_NODE_CONVOLVER: u32 = 4;
pub const WAA_NODE_STEREO_PANNER: u32 = 5;
pub const WAA_NODE_PANNER: u32 = 6;
pub const WAA_NODE_ANALYSER: u32 = 7;
pub const WAA_NODE_WAVESHAPER: u32 = 8;
pub const WAA_NODE_CONSTANT_SOURCE: u32 = 9;
pub const WAA_NODE_IIR_FILTER: u32 = 10;
pub const WAA_NODE_DELAY: u32 = 11;
pub const WAA_NODE_OSCILLATOR: u32 = 12;

/// `waa_node_desc.i[0]` of an oscillator (WAA_OSC_*); the reference's `OscillatorType as u32` has the same order
pub const WAA_OSC_CUSTOM: u32 = 4;

pub const WAA_COUNT_MODE_MAX: u32 = 0;
pub const WAA_COUNT_MODE_CLAMPED_MAX: u32 = 1;
pub const WAA_COUNT_MODE_EXPLICIT: u32 = 2;
pub const WAA_INTERP_SPEAKERS: u32 = 0;
pub const WAA_INTERP_DISCRETE: u32 = 1;

pub const fn waa_param_input(param: u32) -> u32 {
    0x8000_0000 | param
}

/// `waa_source_ended`: the source never ends inside the render / ends when the graph is unloaded
pub const WAA_ENDED_NEVER: i64 = -1;
pub const WAA_ENDED_AT_UNLOAD: i64 = -2;

extern "C" {
    pub fn waa_last_error() -> *const c_char;
    pub fn waa_batch_create(
        graph: *const waa_graph_desc,
        n_instances: u32,
        n_channels_out: u32,
        length_frames: u64,
        sample_rate: f32,
        device: i32,
        out: *mut *mut waa_batch,
    ) -> i32;
    pub fn waa_batch_destroy(batch: *mut waa_batch);
    pub fn waa_source_set_buffer(
        batch: *mut waa_batch,
        node: u32,
        instance: u32,
        channels: *const *const f32,
        n_channels: u32,
        frames: u64,
        buffer_sample_rate: f32,
    ) -> i32;
    pub fn waa_source_start(batch: *mut waa_batch, node: u32, instance: u32, when: f64, offset: f64, duration: f64) -> i32;
    pub fn waa_source_stop(batch: *mut waa_batch, node: u32, instance: u32, when: f64) -> i32;
    pub fn waa_source_set_loop(
        batch: *mut waa_batch,
        node: u32,
        instance: u32,
        is_looping: i32,
        loop_start: f64,
        loop_end: f64,
    ) -> i32;
    pub fn waa_convolver_set_buffer(
        batch: *mut waa_batch,
        node: u32,
        channels: *const *const f32,
        n_channels: u32,
        frames: u64,
        sample_rate: f32,
    ) -> i32;
    pub fn waa_waveshaper_set_curve(batch: *mut waa_batch, node: u32, curve: *const f32, n: u32) -> i32;
    /// OscillatorRenderer::onmessage(PeriodicWave) (oscillator.rs:487-493): the finished 8192-point table
    pub fn waa_oscillator_set_wavetable(batch: *mut waa_batch, node: u32, table: *const f32, n: u32) -> i32;
    /// load_hrtf_processor's database (panner.rs:39-68): once per process, before the first HRTF PannerNode renders
    pub fn waa_hrtf_load_sphere(data: *const std::ffi::c_void, size: u64) -> i32;
    pub fn waa_iir_set_coefficients(
        batch: *mut waa_batch,
        node: u32,
        feedforward: *const f64,
        n_ff: u32,
        feedback: *const f64,
        n_fb: u32,
    ) -> i32;
    pub fn waa_set_param_const(batch: *mut waa_batch, node: u32, param: u32, instance: u32, value: f32) -> i32;
    pub fn waa_set_param_block(
        batch: *mut waa_batch,
        node: u32,
        param: u32,
        instance: u32,
        quantum0: u64,
        n_quanta: u32,
        values_per_quantum: u32,
        values: *const f32,
    ) -> i32;
    pub fn waa_render(batch: *mut waa_batch) -> i32;
    /// round 6: the quantum loop's suspend points (thread.rs:277-294); see `mod.rs::render_with_suspends`
    pub fn waa_render_range(batch: *mut waa_batch, quantum0: u64, n_quanta: u32) -> i32;
    pub fn waa_connect(batch: *mut waa_batch, from: u32, from_output: u32, to: u32, to_input: u32) -> i32;
    pub fn waa_disconnect(batch: *mut waa_batch, from: u32, from_output: u32, to: u32, to_input: u32) -> i32;
    pub fn waa_sync(batch: *mut waa_batch) -> i32;
    pub fn waa_source_ended(batch: *mut waa_batch, node: u32, instance: u32, quantum: *mut i64) -> i32;
    pub fn waa_download(batch: *mut waa_batch, instance: u32, channel: u32, dst: *mut f32, frames: u64) -> i32;
    pub fn waa_analyser_get_float_time_domain_data(
        batch: *mut waa_batch,
        node: u32,
        instance: u32,
        dst: *mut f32,
        n: u32,
    ) -> i32;
}

/// The library's message for the calling thread's last non-zero status (the reference's panic texts: "NotSupportedError - ...")
pub fn last_error() -> String {
    unsafe {
        let p = waa_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

// (the N-device entry point waa_render_sharded takes host buffers of every context and callbacks per sub-batch; a host that
// already holds its AudioBuffers in Rust Vecs calls GpuOfflineBatch per device range instead, one thread per range)
#[allow(unused)]
type _Unused = c_void;
