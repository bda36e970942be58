// PiperHIPRuntime — PiperMetalRuntime (Sources/PiperMetal/PiperMetalRuntime.swift:62-115 in the reference) over the voice-level C-ABI:
// one static schedule replayed as a HIP graph per request instead of a 2 755-node interpreted walk. UNTESTED GLUE (see Package.swift).
import CPiperHIP
import Foundation

public final class PiperHIPRuntime {
    private let backend: HIPBackend
    private var voice: OpaquePointer?
    public let sampleRate: Int32
    public let hop: Int

    /// A Piper voice file: the ONNX loader (host-only C++) infers the geometry, folds weight norm and lays the weights out in the order
    /// of include/piper_hip_voice_layout.h; `voice.onnx.json` supplies the sample rate (PiperConfig.swift:3-47).
    public init(modelPath: String, device: Int32 = 0) throws {
        backend = try HIPBackend(device: device)
        var model: OpaquePointer?
        try HIPBackend.check(piper_hip_onnx_open(modelPath, &model))
        defer { piper_hip_onnx_close(model) }
        var cfg = piper_hip_voice_config()
        try HIPBackend.check(piper_hip_onnx_infer_config(model, &cfg))
        if let json = try? String(contentsOfFile: modelPath + ".json", encoding: .utf8) {
            var info = piper_hip_piper_json_info()
            try HIPBackend.check(piper_hip_piper_json(json, &info))
            try HIPBackend.check(piper_hip_voice_check_json(&cfg, &info))       // num_symbols vs n_vocab, single speaker
            cfg.sample_rate = info.sample_rate
        }
        var n = 0
        try HIPBackend.check(piper_hip_voice_blob_floats(&cfg, &n))
        var blob = [Float](repeating: 0, count: n)
        try HIPBackend.check(piper_hip_onnx_build_blob(model, &cfg, &blob, n))
        try HIPBackend.check(piper_hip_voice_create(backend.ctx, &cfg, blob, 0, &voice))
        sampleRate = cfg.sample_rate
        var h = 1                                                              // hop = Π upsample rates (256 for Piper)
        withUnsafeBytes(of: &cfg.up_rates) { r in for i in 0..<Int(cfg.n_ups) { h *= Int(r.load(fromByteOffset: 4 * i, as: Int32.self)) } }
        hop = h
    }
    deinit { piper_hip_voice_destroy(voice) }

    /// PiperMetalRuntime.synthesize(phonemeIDs:noiseScale:lengthScale:noiseW:) — the whole graph on the device: `durations == nil` ⇒ the
    /// voice's stochastic duration predictor runs (part of the plan's HIP graph); `noise_mode = DEVICE` ⇒ both RandomNormalLike tensors are
    /// drawn on the device with the reference's xorshift32 + Box-Muller generator (elementwise.metal:132-163) from `seed`.
    public func synthesize(phonemeIDs: [Int64], noiseScale: Float = 0.667, lengthScale: Float = 1.0, noiseW: Float = 0.8,
                           seed: UInt32 = 1234) throws -> [Float] {
        try phonemeIDs.withUnsafeBufferPointer { ids in
            var u = piper_hip_utterance(phoneme_ids: ids.baseAddress, t: Int32(ids.count), durations: nil, noise: nil,
                                        noise_scale: noiseScale, noise_mode: Int32(PIPER_HIP_NOISE_DEVICE), seed: seed,
                                        length_scale: lengthScale, noise_w: noiseW, dp_noise: nil)
            try HIPBackend.check(piper_hip_voice_prepare(voice, &u, 0))          // predicts the durations, uploads the inputs
            var total: Int64 = 0
            try HIPBackend.check(piper_hip_voice_prepared_samples(voice, 0, nil, 0, &total))   // Σ predicted frames · hop
            var audio = [Float](repeating: 0, count: Int(total))
            try HIPBackend.check(piper_hip_voice_launch(voice, 0))
            try HIPBackend.check(piper_hip_voice_collect(voice, 0, &audio, total))
            return audio
        }
    }

    /// The same without the host round trip between the duration predictor and the flow (piper_hip_voice_prepare_batch_bounded): the caller bounds
    /// the frames (Piper voices stay below ≈ 6 frames per phoneme id at lengthScale 1), the plan is the bucket of that bound, generate_path runs on the
    /// device and the true length comes back with the waveform. Throws ExecutionError.shapeMismatch when the prediction exceeds the bound.
    public func synthesize(phonemeIDs: [Int64], maxFrames: Int, noiseScale: Float = 0.667, lengthScale: Float = 1.0, noiseW: Float = 0.8,
                           seed: UInt32 = 1234) throws -> [Float] {
        try phonemeIDs.withUnsafeBufferPointer { ids in
            var u = piper_hip_utterance(phoneme_ids: ids.baseAddress, t: Int32(ids.count), durations: nil, noise: nil,
                                        noise_scale: noiseScale, noise_mode: Int32(PIPER_HIP_NOISE_DEVICE), seed: seed,
                                        length_scale: lengthScale, noise_w: noiseW, dp_noise: nil)
            try HIPBackend.check(piper_hip_voice_prepare_batch_bounded(voice, &u, 1, 0, Int32(maxFrames)))   // nothing waits for the GPU here
            try HIPBackend.check(piper_hip_voice_launch(voice, 0))
            var cap: Int64 = 0
            try HIPBackend.check(piper_hip_voice_prepared_samples(voice, 0, nil, 0, &cap))       // capacity: bucket(maxFrames) · hop
            var audio = [Float](repeating: 0, count: Int(cap))
            try HIPBackend.check(piper_hip_voice_collect(voice, 0, &audio, cap))
            var total: Int64 = 0
            try HIPBackend.check(piper_hip_voice_prepared_samples(voice, 0, nil, 0, &total))     // the true length now
            audio.removeLast(audio.count - Int(total))
            return audio
        }
    }

    /// The reference's `overrides` (GraphExecutor.swift:101-104): pinned durations and an injected noise tensor — the parity entry.
    public func synthesize(phonemeIDs: [Int64], durations: [Int32], noise: [Float]?, noiseScale: Float) throws -> [Float] {
        var n: Int64 = 0
        return try phonemeIDs.withUnsafeBufferPointer { ids in try durations.withUnsafeBufferPointer { dur in
            try (noise ?? []).withUnsafeBufferPointer { nz in
                var u = piper_hip_utterance(phoneme_ids: ids.baseAddress, t: Int32(ids.count), durations: dur.baseAddress,
                                            noise: noise == nil ? nil : nz.baseAddress, noise_scale: noiseScale,
                                            noise_mode: Int32(PIPER_HIP_NOISE_INJECTED), seed: 1234, length_scale: 1.0, noise_w: 0.8, dp_noise: nil)
                var audio = [Float](repeating: 0, count: Int(piper_hip_voice_num_samples(voice, &u)))
                try HIPBackend.check(piper_hip_voice_synthesize(voice, &u, &audio, Int64(audio.count), &n))
                return audio
            }
        } }
    }

    /// PiperMetalRuntime.synthesizeStream (PiperMetalRuntime.swift:82-115) with a generator that really decodes incrementally:
    /// encoder + flow once, then one HiFi-GAN window (chunk + receptive-field halo) per call.
    public func synthesizeStream(phonemeIDs: [Int64], durations: [Int32], noiseScale: Float, chunkFrames: Int32 = 64,
                                 onChunk: ([Float]) -> Void) throws {
        try phonemeIDs.withUnsafeBufferPointer { ids in try durations.withUnsafeBufferPointer { dur in
            var u = piper_hip_utterance(phoneme_ids: ids.baseAddress, t: Int32(ids.count), durations: dur.baseAddress,
                                        noise: nil, noise_scale: noiseScale, noise_mode: Int32(PIPER_HIP_NOISE_INJECTED), seed: 1234,
                                        length_scale: 1.0, noise_w: 0.8, dp_noise: nil)
            let chunks = piper_hip_voice_stream_begin(voice, &u, 0, chunkFrames)      // encoder + flow run here
            if chunks < 0 { try HIPBackend.check(chunks) }
            var buf = [Float](repeating: 0, count: Int(chunkFrames) * hop)
            var n: Int64 = 0
            repeat {
                try HIPBackend.check(piper_hip_voice_stream_next(voice, 0, &buf, Int64(buf.count), &n))
                if n > 0 { onChunk(Array(buf[0..<Int(n)])) }
            } while n > 0
        } }
    }

    /// WavFileWriter (Sources/PiperCLI/WavFileWriter.swift:20-60): float → int16 with the CLI's x·32767 clamp, RIFF header.
    public func writeWav(_ samples: [Float], to path: String) throws {
        try HIPBackend.check(piper_hip_wav_write(path, samples, samples.count, sampleRate))
    }
}
