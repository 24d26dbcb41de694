//! waa-shim-check <inputs_dir> <reference_dir> [device]
//!
//! For every case: N contexts with the same graph and their own input are rendered (a) one by one with
//! `start_rendering_sync` — the reference's CPU path, untouched by the patch — and (b) as one batch with
//! `web_audio_api::gpu::start_rendering_sync_batch`; every context's AudioBuffer must agree within 1e-6 RMS per channel
//! (the north star's tolerance), `ended` handlers must have fired the same number of times, and the batch must really
//! have been rendered by the device (no fallback).  Inputs: tools/ref_inputs.py (the files oracle/ref_harness reads).
use std::fs::File;
use std::io::Read;
use std::path::Path;
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::Arc;

use web_audio_api::context::{BaseAudioContext, OfflineAudioContext};
use web_audio_api::gpu::start_rendering_sync_batch_with_report;
use web_audio_api::node::{
    AnalyserNode, AnalyserOptions, AudioNode, AudioScheduledSourceNode, BiquadFilterType, ConvolverNode, ConvolverOptions,
    OverSampleType, PanningModelType, WaveShaperNode, WaveShaperOptions,
};
use web_audio_api::AudioBuffer;

const FRAMES: usize = 96_000;
const N: usize = 8;

fn read_f32(path: &Path) -> Vec<f32> {
    let mut bytes = Vec::new();
    File::open(path).unwrap_or_else(|e| panic!("{path:?}: {e}")).read_to_end(&mut bytes).unwrap();
    bytes.chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}

/// context i of a case: the shared noise rotated by i * 977 frames, so that every context renders something else
fn input(noise: &[f32], i: usize) -> AudioBuffer {
    let rot = |ch: &[f32]| -> Vec<f32> { (0..FRAMES).map(|k| ch[(k + i * 977) % FRAMES]).collect() };
    AudioBuffer::from(vec![rot(&noise[..FRAMES]), rot(&noise[FRAMES..2 * FRAMES])], 48_000.)
}

struct Built {
    ctx: OfflineAudioContext,
    ended: Arc<AtomicUsize>,
    analyser: Option<AnalyserNode>,
}

fn build(case: &str, i: usize, noise: &[f32], tanh_curve: &[f32], ir: &AudioBuffer) -> Built {
    let ctx = OfflineAudioContext::new(2, FRAMES, 48_000.);
    let ended = Arc::new(AtomicUsize::new(0));
    let mut src = ctx.create_buffer_source();
    src.set_buffer(input(noise, i));
    let counter = Arc::clone(&ended);
    src.set_onended(move |_| {
        counter.fetch_add(1, Ordering::SeqCst);
    });
    let mut analyser = None;
    match case {
        "c1" | "c1_arate" | "c2" | "c4" | "t1" => {
            let mut biquad = ctx.create_biquad_filter();
            biquad.set_type(BiquadFilterType::Lowpass);
            biquad.frequency().set_value(200. + 25. * i as f32); // per-context AudioParam values
            biquad.q().set_value(1.);
            if case == "c1_arate" {
                biquad.frequency().set_value_at_time(10., 0.);
                biquad.frequency().exponential_ramp_to_value_at_time(10_000., FRAMES as f64 / 48_000.);
            }
            src.connect(&biquad);
            match case {
                "c2" => {
                    let gain = ctx.create_gain();
                    gain.gain().set_value(0.5);
                    biquad.connect(&gain);
                    gain.connect(&ctx.destination());
                }
                "t1" | "c4" => {
                    let mut conv = ConvolverNode::new(&ctx, ConvolverOptions::default());
                    conv.set_buffer(ir.clone());
                    biquad.connect(&conv);
                    if case == "c4" {
                        let pan = ctx.create_stereo_panner();
                        pan.pan().set_value(0.1);
                        let an = AnalyserNode::new(&ctx, AnalyserOptions { fft_size: 2048, smoothing_time_constant: 0.8, ..AnalyserOptions::default() });
                        conv.connect(&pan);
                        pan.connect(&an);
                        an.connect(&ctx.destination());
                        analyser = Some(an);
                    } else {
                        conv.connect(&ctx.destination());
                    }
                }
                _ => {
                    biquad.connect(&ctx.destination());
                }
            }
        }
        "os2" | "os4" => {
            let os = if case == "os2" { OverSampleType::X2 } else { OverSampleType::X4 };
            let shaper = WaveShaperNode::new(&ctx, WaveShaperOptions { curve: Some(tanh_curve.to_vec()), oversample: os, ..WaveShaperOptions::default() });
            src.connect(&shaper);
            shaper.connect(&ctx.destination());
        }
        "echo" => {
            // Delay <-> Gain feedback loop + dry: the crate's writer / reader pair folded into ONE library node (round 4)
            let delay = ctx.create_delay(1.);
            delay.delay_time().set_value(0.05 + 0.01 * i as f32);
            let feedback = ctx.create_gain();
            feedback.gain().set_value(0.5);
            src.connect(&delay);
            delay.connect(&feedback);
            feedback.connect(&delay);
            delay.connect(&ctx.destination());
            src.connect(&ctx.destination());
        }
        "panner" | "panner_hrtf" => {
            // PannerNode with per-context position and a moved AudioListener (its nine params go through the panner)
            let mut panner = ctx.create_panner();
            if case == "panner_hrtf" {
                panner.set_panning_model(PanningModelType::HRTF);
            }
            panner.position_x().set_value(1. + 0.3 * i as f32);
            panner.position_z().set_value(-0.5);
            ctx.listener().position_y().set_value(0.25);
            src.connect(&panner);
            panner.connect(&ctx.destination());
        }
        "c5" => {
            src.playback_rate().set_value(1.5);
            src.set_loop(true);
            let curve: Vec<f32> = (0..2048).map(|k| (std::f32::consts::PI + k as f32 * std::f32::consts::PI / 2047.).cos()).collect();
            let shaper = WaveShaperNode::new(&ctx, WaveShaperOptions { curve: Some(curve), ..WaveShaperOptions::default() });
            src.connect(&shaper);
            shaper.connect(&ctx.destination());
        }
        _ => unreachable!(),
    }
    // (the source ends inside the render in some contexts: `ended` events with a quantum, the others at unload)
    src.start();
    if i % 2 == 1 {
        src.stop_at(1.0 + 0.01 * i as f64);
    }
    Built { ctx, ended, analyser }
}

fn rms(a: &[f32], b: &[f32]) -> f64 {
    (a.iter().zip(b).map(|(x, y)| (*x as f64 - *y as f64).powi(2)).sum::<f64>() / a.len() as f64).sqrt()
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let (inputs, reference) = (Path::new(&args[1]), Path::new(&args[2]));
    let device: i32 = args.get(3).map(|s| s.parse().unwrap()).unwrap_or(0);
    let noise = read_f32(&inputs.join("noise_stereo.f32"));
    let tanh_curve = read_f32(&inputs.join("curve_tanh.f32"));
    let ir = {
        let ctx = OfflineAudioContext::new(2, 128, 48_000.);
        ctx.decode_audio_data_sync(File::open(reference.join("samples/parking-garage-response.wav")).unwrap()).unwrap()
    };
    let mut failures = 0;
    for case in ["c1", "c1_arate", "c2", "t1", "c4", "os2", "os4", "c5", "echo", "panner", "panner_hrtf"] {
        let mut cpu: Vec<Built> = (0..N).map(|i| build(case, i, &noise, &tanh_curve, &ir)).collect();
        let mut gpu: Vec<Built> = (0..N).map(|i| build(case, i, &noise, &tanh_curve, &ir)).collect();
        let want: Vec<AudioBuffer> = cpu.iter_mut().map(|b| b.ctx.start_rendering_sync()).collect();
        let mut contexts: Vec<OfflineAudioContext> = gpu.iter_mut().map(|b| std::mem::replace(&mut b.ctx, OfflineAudioContext::new(1, 128, 48_000.))).collect();
        let (got, fallback) = start_rendering_sync_batch_with_report(&mut contexts, device);
        let mut worst = 0f64;
        for (w, g) in want.iter().zip(&got) {
            for c in 0..w.number_of_channels() {
                worst = worst.max(rms(w.get_channel_data(c), g.get_channel_data(c)));
            }
        }
        let ended_ok = cpu.iter().zip(&gpu).all(|(a, b)| a.ended.load(Ordering::SeqCst) == b.ended.load(Ordering::SeqCst));
        let mut bins_worst = 0f32;
        for (a, b) in cpu.iter_mut().zip(gpu.iter_mut()) {
            if let (Some(x), Some(y)) = (a.analyser.as_mut(), b.analyser.as_mut()) {
                let (mut p, mut q) = (vec![0f32; 1024], vec![0f32; 1024]);
                x.get_float_frequency_data(&mut p);
                y.get_float_frequency_data(&mut q);
                for (u, v) in p.iter().zip(&q) {
                    if u.is_finite() && *u > -120. {
                        bins_worst = bins_worst.max((u - v).abs());
                    }
                }
            }
        }
        let ok = fallback.is_none() && worst <= 1e-6 && ended_ok && bins_worst <= 0.05;
        println!("{case:9} device: {}  worst RMS {worst:.3e}  ended events equal: {ended_ok}  analyser dB diff {bins_worst:.3e}  {}",
                 if fallback.is_none() { "yes".to_string() } else { format!("NO ({fallback:?})") }, if ok { "ok" } else { "FAILED" });
        if !ok {
            failures += 1;
        }
    }
    std::process::exit(if failures == 0 { 0 } else { 1 });
}
