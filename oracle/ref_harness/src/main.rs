//! Renders the parity cases with the reference crate and writes every result as raw little-endian f32.
//!
//!     waa-ref-harness <inputs_dir> <out_dir> <reference_dir>
//!
//! Inputs (written by tools/ref_inputs.py with fixed seeds; the SAME files feed the oracle and the HIP library in
//! tests/test_reference_dumps.py):  noise_stereo.f32 (2 x 96000, channel-major), noise_mono.f32 (96000), curve_tanh.f32.
//! Outputs: <case>.f32 = the rendered AudioBuffer, channel-major, plus manifest.txt (case, channels, frames, sample rate).
//! Cases — each is the graph of the same name in tests/test_reference_dumps.py:
//!   c1            48 kHz stereo: BufferSource -> Biquad(lowpass 200 Hz, Q 1) -> destination            (biquad_filter.rs)
//!   c1_arate      the same with frequency.exponential_ramp 10 Hz -> 10 kHz over the render            (examples/biquad.rs)
//!   c2            ... -> Biquad -> Gain(0.5) -> destination
//!   t1            BufferSource -> Biquad -> Convolver(parking-garage response, normalised) -> destination   (P = 175 > 1:
//!                 pins fft-convolver's multi-partition path, which the reference's own tests do not)
//!   os2 / os4     BufferSource -> WaveShaper(tanh curve, oversample 2x / 4x) -> destination            (pins rubato)
//!   hrtf_44k1 / hrtf_48k   BufferSource(mono) -> PannerNode(HRTF, position (1, 0.5, -0.5)) -> destination   (pins hrtf,
//!                 and at 48 kHz the crate's HRIR resampling)
//!   c5            BufferSource(playbackRate 1.5, loop) -> WaveShaper(2048-pt cos curve) -> destination  (pins `almost`)
//!   analyser_db   c1's graph with an AnalyserNode(2048, 0.8) in front of the destination: the 1024 dB values (pins realfft)
use std::fs::File;
use std::io::{Read, Write};
use std::path::Path;

use web_audio_api::context::{BaseAudioContext, OfflineAudioContext};
use web_audio_api::node::{
    AnalyserNode, AnalyserOptions, AudioNode, AudioScheduledSourceNode, BiquadFilterType, ConvolverNode, ConvolverOptions,
    OverSampleType, PannerNode, PannerOptions, PanningModelType, WaveShaperNode, WaveShaperOptions,
};
use web_audio_api::AudioBuffer;

fn read_f32(path: &Path) -> Vec<f32> {
    let mut bytes = Vec::new();
    File::open(path).unwrap_or_else(|e| panic!("{path:?}: {e}")).read_to_end(&mut bytes).unwrap();
    bytes.chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}

fn write_case(out: &Path, manifest: &mut String, name: &str, buf: &AudioBuffer) {
    let mut f = File::create(out.join(format!("{name}.f32"))).unwrap();
    for c in 0..buf.number_of_channels() {
        for v in buf.get_channel_data(c) {
            f.write_all(&v.to_le_bytes()).unwrap();
        }
    }
    manifest.push_str(&format!("{name} {} {} {}\n", buf.number_of_channels(), buf.length(), buf.sample_rate()));
}

fn write_vec(out: &Path, manifest: &mut String, name: &str, v: &[f32], sr: f32) {
    let mut f = File::create(out.join(format!("{name}.f32"))).unwrap();
    for x in v {
        f.write_all(&x.to_le_bytes()).unwrap();
    }
    manifest.push_str(&format!("{name} 1 {} {}\n", v.len(), sr));
}

const FRAMES: usize = 96_000; // 2 s @ 48 kHz = 750 render quanta (12 convolver blocks of 8192 on the device side)

fn stereo_source(ctx: &OfflineAudioContext, noise: &[f32], sr: f32) -> web_audio_api::node::AudioBufferSourceNode {
    let buf = AudioBuffer::from(vec![noise[..FRAMES].to_vec(), noise[FRAMES..2 * FRAMES].to_vec()], sr);
    let mut src = ctx.create_buffer_source();
    src.set_buffer(buf);
    src
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let (inputs, out, reference) = (Path::new(&args[1]), Path::new(&args[2]), Path::new(&args[3]));
    std::fs::create_dir_all(out).unwrap();
    let noise = read_f32(&inputs.join("noise_stereo.f32"));
    let mono = read_f32(&inputs.join("noise_mono.f32"));
    let tanh_curve = read_f32(&inputs.join("curve_tanh.f32"));
    let mut manifest = String::new();

    // ---- c1, c1_arate, c2, analyser_db
    for case in ["c1", "c1_arate", "c2", "analyser_db"] {
        let mut ctx = OfflineAudioContext::new(2, FRAMES, 48_000.);
        let mut src = stereo_source(&ctx, &noise, 48_000.);
        let mut biquad = ctx.create_biquad_filter();
        biquad.set_type(BiquadFilterType::Lowpass);
        biquad.frequency().set_value(200.);
        biquad.q().set_value(1.);
        if case == "c1_arate" {
            biquad.frequency().set_value_at_time(10., 0.);
            biquad.frequency().exponential_ramp_to_value_at_time(10_000., FRAMES as f64 / 48_000.);
        }
        src.connect(&biquad);
        let mut analyser: Option<AnalyserNode> = None;
        if case == "c2" {
            let gain = ctx.create_gain();
            gain.gain().set_value(0.5);
            biquad.connect(&gain);
            gain.connect(&ctx.destination());
        } else if case == "analyser_db" {
            let an = AnalyserNode::new(&ctx, AnalyserOptions { fft_size: 2048, smoothing_time_constant: 0.8, ..AnalyserOptions::default() });
            biquad.connect(&an);
            an.connect(&ctx.destination());
            analyser = Some(an);
        } else {
            biquad.connect(&ctx.destination());
        }
        src.start();
        let rendered = ctx.start_rendering_sync();
        if let Some(mut an) = analyser {
            let mut bins = vec![0.0f32; 1024];
            an.get_float_frequency_data(&mut bins);
            write_vec(out, &mut manifest, "analyser_db", &bins, 48_000.);
        } else {
            write_case(out, &mut manifest, case, &rendered);
        }
    }

    // ---- t1: the parking-garage response, decoded and resampled by the crate itself (decoding.rs:15-54, buffer.rs:311)
    {
        let mut ctx = OfflineAudioContext::new(2, FRAMES, 48_000.);
        let ir_file = File::open(reference.join("samples/parking-garage-response.wav")).unwrap();
        let ir = ctx.decode_audio_data_sync(ir_file).unwrap();
        assert_eq!((ir.number_of_channels(), ir.length()), (2, 178_899));
        let mut src = stereo_source(&ctx, &noise, 48_000.);
        let mut biquad = ctx.create_biquad_filter();
        biquad.frequency().set_value(200.);
        biquad.q().set_value(1.);
        let mut conv = ConvolverNode::new(&ctx, ConvolverOptions::default());
        conv.set_buffer(ir);
        src.connect(&biquad);
        biquad.connect(&conv);
        conv.connect(&ctx.destination());
        src.start();
        write_case(out, &mut manifest, "t1", &ctx.start_rendering_sync());
    }

    // ---- os2 / os4
    for (case, os) in [("os2", OverSampleType::X2), ("os4", OverSampleType::X4)] {
        let mut ctx = OfflineAudioContext::new(2, FRAMES, 48_000.);
        let mut src = stereo_source(&ctx, &noise, 48_000.);
        let shaper = WaveShaperNode::new(&ctx, WaveShaperOptions { curve: Some(tanh_curve.clone()), oversample: os, ..WaveShaperOptions::default() });
        src.connect(&shaper);
        shaper.connect(&ctx.destination());
        src.start();
        write_case(out, &mut manifest, case, &ctx.start_rendering_sync());
    }

    // ---- HRTF at the database's own rate and at 48 kHz
    for (case, sr) in [("hrtf_44k1", 44_100.0f32), ("hrtf_48k", 48_000.0f32)] {
        let mut ctx = OfflineAudioContext::new(2, FRAMES, sr);
        let buf = AudioBuffer::from(vec![mono[..FRAMES].to_vec()], sr);
        let mut src = ctx.create_buffer_source();
        src.set_buffer(buf);
        let panner = PannerNode::new(&ctx, PannerOptions { panning_model: PanningModelType::HRTF, position_x: 1., position_y: 0.5, position_z: -0.5, ..PannerOptions::default() });
        src.connect(&panner);
        panner.connect(&ctx.destination());
        src.start();
        write_case(out, &mut manifest, case, &ctx.start_rendering_sync());
    }

    // ---- c5: playbackRate 1.5, loop over the first 65 536 frames, WaveShaper(2048-pt cos curve, no oversampling)
    {
        let mut ctx = OfflineAudioContext::new(2, FRAMES, 48_000.);
        let buf = AudioBuffer::from(vec![noise[..65_536].to_vec(), noise[FRAMES..FRAMES + 65_536].to_vec()], 48_000.);
        let mut src = ctx.create_buffer_source();
        src.set_buffer(buf);
        src.playback_rate().set_value(1.5);
        src.set_loop(true);
        let curve: Vec<f32> = (0..2048).map(|i| (std::f32::consts::PI + i as f32 * std::f32::consts::PI / 2047.).cos()).collect();
        let shaper = WaveShaperNode::new(&ctx, WaveShaperOptions { curve: Some(curve), ..WaveShaperOptions::default() });
        src.connect(&shaper);
        shaper.connect(&ctx.destination());
        src.start();
        write_case(out, &mut manifest, "c5", &ctx.start_rendering_sync());
    }

    File::create(out.join("manifest.txt")).unwrap().write_all(manifest.as_bytes()).unwrap();
    println!("{manifest}");
}
