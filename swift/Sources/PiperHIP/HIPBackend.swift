// HIPBackend — MetalBackend's per-operator surface (Sources/PiperMetal/Execution/MetalBackend.swift:8-3427 in the reference) over
// the C-ABI of include/piper_hip.h: every method of the Python twin (piper_hip.HipBackend, which the test-suite drives) under the
// reference's method names and argument labels. UNCOMPILED GLUE in this pipeline (no Swift toolchain in the image, SURVEY F5).
import CPiperHIP
import Foundation

/// The reference's error cases (CPUBackend.swift:3-17), so a GraphExecutor arm can catch the same things.
public enum ExecutionError: Error {
    case unsupportedOp(String)
    case typeMismatch(String)
    case shapeMismatch(String)
    case metalUnavailable(String)
}

/// MTLBuffer stand-in: a device pointer owned by the context's pool (freed on deinit, like ARC on MTLBuffer).
public final class HIPBuffer {
    public let ptr: UnsafeMutableRawPointer
    private let ctx: OpaquePointer
    init(_ p: UnsafeMutableRawPointer, ctx: OpaquePointer) { self.ptr = p; self.ctx = ctx }
    deinit { piper_hip_free(ctx, ptr) }
    var f32: UnsafeMutablePointer<Float> { ptr.assumingMemoryBound(to: Float.self) }
}

public final class HIPBackend {
    public let ctx: OpaquePointer

    public init(device: Int32 = 0) throws {
        var c: OpaquePointer?
        try HIPBackend.check(piper_hip_create(device, &c))           // MetalContext.init, MetalContext.swift:9-33
        ctx = c!
    }
    deinit { piper_hip_destroy(ctx) }

    /// int status + thread-local message → the ExecutionError cases of CPUBackend.swift:3-17
    public static func check(_ rc: Int32) throws {
        guard rc != 0 else { return }
        let msg = String(cString: piper_hip_last_error())
        switch rc {
        case -1: throw ExecutionError.shapeMismatch(msg)
        case -2: throw ExecutionError.typeMismatch(msg)
        case -3: throw ExecutionError.unsupportedOp(msg)
        case -4: throw ExecutionError.metalUnavailable(msg)
        default: throw NSError(domain: "HIPBackend", code: Int(rc), userInfo: [NSLocalizedDescriptionKey: msg])
        }
    }

    public func makeCommandBuffer() throws -> UnsafeMutableRawPointer {      // MetalBackend.swift:841-843
        var s: UnsafeMutableRawPointer?
        try Self.check(piper_hip_stream_create(ctx, &s)); return s!
    }
    public func flush(_ cmd: UnsafeMutableRawPointer) throws { try Self.check(piper_hip_stream_sync(ctx, cmd)) }   // :845-855

    public func uploadFloat32(_ data: [Float]) throws -> HIPBuffer {         // MetalBackend.swift:983-993
        var out: UnsafeMutablePointer<Float>?
        try data.withUnsafeBufferPointer { try Self.check(piper_hip_upload_f32(ctx, $0.baseAddress, $0.count, &out)) }
        return HIPBuffer(UnsafeMutableRawPointer(out!), ctx: ctx)
    }
    public func downloadFloat32(_ buf: HIPBuffer, count: Int) throws -> [Float] {   // MetalBackend.swift:963-981
        var host = [Float](repeating: 0, count: count)
        try Self.check(piper_hip_download_f32(ctx, buf.f32, &host, count))
        return host
    }

    /// MetalBackend.conv1dF32 (MetalBackend.swift:1149-1161) — same labels, `commandBuffer:` is a HIP stream or nil.
    public func conv1dF32(input: HIPBuffer, inputShape: [Int], weight: HIPBuffer, weightShape: [Int], bias: HIPBuffer?,
                          stride: Int, dilation: Int, padL: Int, padR: Int, groups: Int,
                          commandBuffer: UnsafeMutableRawPointer? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        guard inputShape.count == 3 else { throw ExecutionError.shapeMismatch("conv1dF32 input must be [N,C,L]") }
        guard weightShape.count == 3 else { throw ExecutionError.shapeMismatch("conv1dF32 weight must be [C_out,C_in,K]") }
        var p = piper_hip_conv1d_params(stride: Int32(stride), dilation: Int32(dilation), pad_l: Int32(padL),
                                        pad_r: Int32(padR), groups: Int32(groups))
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 3)
        try Self.check(piper_hip_conv1d_f32(ctx, input.f32, inputShape.map(Int64.init), weight.f32, weightShape.map(Int64.init),
                                            bias?.f32, &p, &out, &oshape, commandBuffer))
        return (HIPBuffer(UnsafeMutableRawPointer(out!), ctx: ctx), oshape.map(Int.init))
    }

    /// MetalBackend.convTranspose1dF32 (MetalBackend.swift:2812-2895)
    public func convTranspose1dF32(input: HIPBuffer, inputShape: [Int], weight: HIPBuffer, weightShape: [Int], bias: HIPBuffer?,
                                   stride: Int, dilation: Int, padL: Int, padR: Int, outputPadding: Int, groups: Int,
                                   commandBuffer: UnsafeMutableRawPointer? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var p = piper_hip_convtranspose1d_params(stride: Int32(stride), dilation: Int32(dilation), pad_l: Int32(padL), pad_r: Int32(padR),
                                                 output_padding: Int32(outputPadding), groups: Int32(groups))
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 3)
        try Self.check(piper_hip_convtranspose1d_f32(ctx, input.f32, inputShape.map(Int64.init), weight.f32, weightShape.map(Int64.init),
                                                     bias?.f32, &p, &out, &oshape, commandBuffer))
        return (HIPBuffer(UnsafeMutableRawPointer(out!), ctx: ctx), oshape.map(Int.init))
    }

    /// The fused attention core the reference spells as 4 MatMuls + Pad/Reshape/Slice skews + Softmax per layer (GraphExecutor.swift:1862-1929)
    public func relAttentionF32(q: HIPBuffer, k: HIPBuffer, v: HIPBuffer, embRelK: HIPBuffer, embRelV: HIPBuffer, batch: Int, heads: Int,
                                headDim: Int, length: Int, window: Int,
                                commandBuffer: UnsafeMutableRawPointer? = nil) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_rel_attention_f32(ctx, q.f32, k.f32, v.f32, embRelK.f32, embRelV.f32, Int64(batch), Int64(heads), Int64(headDim),
                                                   Int64(length), Int64(window), &out, commandBuffer))
        return HIPBuffer(UnsafeMutableRawPointer(out!), ctx: ctx)
    }
    // ---- shared plumbing of the wrappers below ----
    public typealias Stream = UnsafeMutableRawPointer

    private func wrap(_ p: UnsafeMutablePointer<Float>?) -> HIPBuffer { HIPBuffer(UnsafeMutableRawPointer(p!), ctx: ctx) }
    private static func i64(_ a: [Int]) -> [Int64] { a.map(Int64.init) }

    /// MetalBackend.allocateBuffer(length:) (MetalBackend.swift:34-39): `length` BYTES of device memory (≥ 1 byte, like the reference)
    public func allocateBuffer(length: Int) throws -> HIPBuffer {
        var p: UnsafeMutableRawPointer? = nil
        try Self.check(piper_hip_alloc(ctx, max(1, length), &p))
        return HIPBuffer(p!, ctx: ctx)
    }

    /// MetalBackend.matmulF32 (MetalBackend.swift:1232-1323). `useTiledKernel` is accepted for source compatibility and ignored: there is one
    /// kernel. Equal ranks ≥ 2; lead dims equal or 1-broadcast (the executor's expandF32, GraphExecutor.swift:1870-1899, is not needed).
    public func matmulF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], useTiledKernel: Bool = false,
                          commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        guard aShape.count == bShape.count, aShape.count >= 2 else { throw ExecutionError.shapeMismatch("matmulF32 needs equal ranks >= 2") }
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: aShape.count)
        try Self.check(piper_hip_matmul_f32(ctx, a.f32, Self.i64(aShape), b.f32, Self.i64(bShape), Int32(aShape.count), &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }

    /// MetalBackend.softmaxLastDimF32 (MetalBackend.swift:1326-1355)
    public func softmaxLastDimF32(input: HIPBuffer, shape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_softmax_lastdim_f32(ctx, input.f32, Self.i64(shape), Int32(shape.count), &out, commandBuffer))
        return (wrap(out), shape)
    }

    /// MetalBackend.reduceMeanLastDimF32 (MetalBackend.swift:1357-1390): [.., C] → [.., 1]
    public func reduceMeanLastDimF32(input: HIPBuffer, shape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_reduce_mean_lastdim_f32(ctx, input.f32, Self.i64(shape), Int32(shape.count), &out, commandBuffer))
        return (wrap(out), Array(shape.dropLast()) + [1])
    }

    // ---- unary (MetalBackend.swift:1501-1586) ----
    private func unaryF32(_ op: piper_hip_unary_op, input: HIPBuffer, count: Int, alpha: Float = 0, commandBuffer: Stream?) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_unary_f32(ctx, op, input.f32, count, alpha, &out, commandBuffer))
        return wrap(out)
    }
    public func reluF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_RELU, input: input, count: count, commandBuffer: commandBuffer) }
    public func erfF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_ERF, input: input, count: count, commandBuffer: commandBuffer) }
    public func softplusF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_SOFTPLUS, input: input, count: count, commandBuffer: commandBuffer) }
    public func negF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_NEG, input: input, count: count, commandBuffer: commandBuffer) }
    public func expF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_EXP, input: input, count: count, commandBuffer: commandBuffer) }
    public func ceilF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_CEIL, input: input, count: count, commandBuffer: commandBuffer) }
    public func tanhF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_TANH, input: input, count: count, commandBuffer: commandBuffer) }
    public func sigmoidF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_SIGMOID, input: input, count: count, commandBuffer: commandBuffer) }
    public func sqrtF32(input: HIPBuffer, count: Int, commandBuffer: Stream? = nil) throws -> HIPBuffer { try unaryF32(PIPER_HIP_SQRT, input: input, count: count, commandBuffer: commandBuffer) }
    public func leakyReluF32(input: HIPBuffer, count: Int, alpha: Float, commandBuffer: Stream? = nil) throws -> HIPBuffer {
        try unaryF32(PIPER_HIP_LEAKYRELU, input: input, count: count, alpha: alpha, commandBuffer: commandBuffer)
    }

    // ---- binary with NumPy broadcasting, output rank ≤ 4 (MetalBackend.swift:2099-2134, 2592-2610) ----
    private func binaryBroadcastF32(_ op: piper_hip_binary_op, a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int],
                                    commandBuffer: Stream?) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 4)
        var orank: Int32 = 0
        try Self.check(piper_hip_binary_broadcast_f32(ctx, op, a.f32, Self.i64(aShape), Int32(aShape.count), b.f32, Self.i64(bShape), Int32(bShape.count),
                                                      &out, &oshape, &orank, commandBuffer))
        return (wrap(out), oshape.prefix(Int(orank)).map(Int.init))
    }
    public func addF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try binaryBroadcastF32(PIPER_HIP_ADD, a: a, aShape: aShape, b: b, bShape: bShape, commandBuffer: commandBuffer)
    }
    public func subF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try binaryBroadcastF32(PIPER_HIP_SUB, a: a, aShape: aShape, b: b, bShape: bShape, commandBuffer: commandBuffer)
    }
    public func mulF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try binaryBroadcastF32(PIPER_HIP_MUL, a: a, aShape: aShape, b: b, bShape: bShape, commandBuffer: commandBuffer)
    }
    public func divF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try binaryBroadcastF32(PIPER_HIP_DIV, a: a, aShape: aShape, b: b, bShape: bShape, commandBuffer: commandBuffer)
    }
    public func powF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try binaryBroadcastF32(PIPER_HIP_POW, a: a, aShape: aShape, b: b, bShape: bShape, commandBuffer: commandBuffer)
    }

    // ---- layout ops of the skew and of the flow coupling ----
    /// MetalBackend.padConstantF32 (MetalBackend.swift:780-839): pads = [begin_0 … begin_{r-1}, end_0 … end_{r-1}], rank ≤ 4
    public func padConstantF32(input: HIPBuffer, shape: [Int], pads: [Int], constant: Float, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        guard pads.count == 2 * shape.count else { throw ExecutionError.shapeMismatch("padConstantF32: pads must hold 2·rank entries") }
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: shape.count)
        try Self.check(piper_hip_pad_constant_f32(ctx, input.f32, Self.i64(shape), Int32(shape.count), Self.i64(pads), constant, &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }
    /// MetalBackend.transposeF32 (MetalBackend.swift:995-1060)
    public func transposeF32(input: HIPBuffer, shape: [Int], perm: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: shape.count)
        try Self.check(piper_hip_transpose_f32(ctx, input.f32, Self.i64(shape), Int32(shape.count), perm.map(Int32.init), &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }
    /// One-axis slice with any non-zero step; the reference's family sliceAxis1NCLF32 / sliceAxis2NCLF32 / slice2DAxis1F32Step1 /
    /// sliceRank4Axis3F32Step1 / slice1DF32Step1 / reverseRank3Axis1F32 (MetalBackend.swift:1702-1980) are calls of this with their axis.
    public func sliceF32(input: HIPBuffer, shape: [Int], axis: Int, start: Int, end: Int, step: Int = 1, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: shape.count)
        try Self.check(piper_hip_slice_f32(ctx, input.f32, Self.i64(shape), Int32(shape.count), Int32(axis), Int64(start), Int64(end), Int64(step), &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }
    public func sliceAxis1NCLF32(input: HIPBuffer, shape: [Int], start: Int, step: Int, count: Int, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try sliceF32(input: input, shape: shape, axis: 1, start: start, end: start + step * count, step: step, commandBuffer: commandBuffer)   // :1702
    }
    public func sliceAxis2NCLF32(input: HIPBuffer, shape: [Int], start: Int, step: Int, count: Int, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try sliceF32(input: input, shape: shape, axis: 2, start: start, end: start + step * count, step: step, commandBuffer: commandBuffer)   // :1730
    }
    public func sliceRank4Axis3F32Step1(input: HIPBuffer, shape: [Int], start: Int, end: Int, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try sliceF32(input: input, shape: shape, axis: 3, start: start, end: end, commandBuffer: commandBuffer)                               // :1873
    }
    /// VITS `Flip`: reverse the channel axis of [N, C, L] (MetalBackend.swift:1803-1871)
    public func reverseRank3Axis1F32(input: HIPBuffer, shape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        try sliceF32(input: input, shape: shape, axis: 1, start: shape[1] - 1, end: -1, step: -1, commandBuffer: commandBuffer)
    }
    /// MetalBackend.concat2Axis1NCLF32 (MetalBackend.swift:1597-1641)
    public func concat2Axis1NCLF32(a: HIPBuffer, aShape: [Int], b: HIPBuffer, bShape: [Int], commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 3)
        try Self.check(piper_hip_concat2_axis1_f32(ctx, a.f32, Self.i64(aShape), b.f32, Self.i64(bShape), &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }
    /// MetalBackend.split2Axis1NCLF32 (MetalBackend.swift:1643-1700)
    public func split2Axis1NCLF32(input: HIPBuffer, inputShape: [Int], c0: Int, c1: Int, commandBuffer: Stream? = nil) throws
        -> (out0: HIPBuffer, out0Shape: [Int], out1: HIPBuffer, out1Shape: [Int]) {
        guard inputShape.count == 3, c0 + c1 == inputShape[1] else { throw ExecutionError.shapeMismatch("split2Axis1NCLF32: c0 + c1 must equal C") }
        var o0: UnsafeMutablePointer<Float>? = nil, o1: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_split2_axis1_f32(ctx, input.f32, Self.i64(inputShape), Int64(c0), &o0, &o1, commandBuffer))
        return (wrap(o0), [inputShape[0], c0, inputShape[2]], wrap(o1), [inputShape[0], c1, inputShape[2]])
    }
    /// MetalBackend.expandF32 (MetalBackend.swift:2438-2458)
    public func expandF32(input: HIPBuffer, inShape: [Int], outShape: [Int], commandBuffer: Stream? = nil) throws -> HIPBuffer {
        guard inShape.count == outShape.count else { throw ExecutionError.shapeMismatch("expandF32 needs equal ranks") }
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_expand_f32(ctx, input.f32, Self.i64(inShape), Self.i64(outShape), Int32(inShape.count), &out, commandBuffer))
        return wrap(out)
    }

    /// MetalBackend.randomNormalLike(shape:seed:) (MetalBackend.swift:3398-3426): the reference's xorshift32 + Box-Muller generator
    public func randomNormalLike(shape: [Int], seed: UInt64 = 1234, commandBuffer: Stream? = nil) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_random_normal_like_f32(ctx, shape.reduce(1, *), seed, &out, commandBuffer))
        return wrap(out)
    }

    // ---- bf16-operand contractions (build extension; same contract as the f32 ones) ----
    public func conv1dBF16(input: HIPBuffer, inputShape: [Int], weight: HIPBuffer, weightShape: [Int], bias: HIPBuffer?, stride: Int, dilation: Int,
                           padL: Int, padR: Int, groups: Int, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var p = piper_hip_conv1d_params(stride: Int32(stride), dilation: Int32(dilation), pad_l: Int32(padL), pad_r: Int32(padR), groups: Int32(groups))
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 3)
        try Self.check(piper_hip_conv1d_bf16(ctx, input.f32, Self.i64(inputShape), weight.f32, Self.i64(weightShape), bias?.f32, &p, &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }
    public func convTranspose1dBF16(input: HIPBuffer, inputShape: [Int], weight: HIPBuffer, weightShape: [Int], bias: HIPBuffer?, stride: Int, dilation: Int,
                                    padL: Int, padR: Int, outputPadding: Int, groups: Int, commandBuffer: Stream? = nil) throws -> (out: HIPBuffer, outShape: [Int]) {
        var p = piper_hip_convtranspose1d_params(stride: Int32(stride), dilation: Int32(dilation), pad_l: Int32(padL), pad_r: Int32(padR),
                                                 output_padding: Int32(outputPadding), groups: Int32(groups))
        var out: UnsafeMutablePointer<Float>? = nil
        var oshape = [Int64](repeating: 0, count: 3)
        try Self.check(piper_hip_convtranspose1d_bf16(ctx, input.f32, Self.i64(inputShape), weight.f32, Self.i64(weightShape), bias?.f32, &p, &out, &oshape, commandBuffer))
        return (wrap(out), oshape.map(Int.init))
    }

    // ---- fused blocks (results equal the composition of the ops above; tests/test_gpu_ops.py) ----
    /// out = LN_c(x + y)·gamma + beta (the Add + ReduceMean … Div chain, GraphExecutor.swift:2071-2125); y may be nil
    public func addLayerNormF32(x: HIPBuffer, y: HIPBuffer?, gamma: HIPBuffer, beta: HIPBuffer, batch: Int, channels: Int, length: Int, eps: Float = 1e-5,
                                commandBuffer: Stream? = nil) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_add_layernorm_f32(ctx, x.f32, y?.f32, gamma.f32, beta.f32, Int64(batch), Int64(channels), Int64(length), eps, &out, commandBuffer))
        return wrap(out)
    }
    /// attention core + output projection + residual + LayerNorm of an encoder layer in one launch (PIPER_HIP_ERR_UNSUPPORTED outside its geometry)
    public func attentionBlockF32(q: HIPBuffer, k: HIPBuffer, v: HIPBuffer, embRelK: HIPBuffer, embRelV: HIPBuffer, wO: HIPBuffer, bO: HIPBuffer, x: HIPBuffer,
                                  gamma: HIPBuffer, beta: HIPBuffer, batch: Int, heads: Int, headDim: Int, length: Int, window: Int, eps: Float = 1e-5,
                                  commandBuffer: Stream? = nil) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_attention_block_f32(ctx, q.f32, k.f32, v.f32, embRelK.f32, embRelV.f32, wO.f32, bO.f32, x.f32, gamma.f32, beta.f32, Int64(batch),
                                                     Int64(heads), Int64(headDim), Int64(length), Int64(window), eps, &out, commandBuffer))
        return wrap(out)
    }
    /// One WaveNet layer of the flow: gate conv + tanh·sigmoid + res/skip conv. Returns (x_out — nil when `last` —, skip_out).
    public func wavenetLayerF32(x: HIPBuffer, skipIn: HIPBuffer?, wIn: HIPBuffer, bIn: HIPBuffer, wRs: HIPBuffer, bRs: HIPBuffer, batch: Int, channels: Int,
                                length: Int, kernel: Int, dilation: Int, last: Bool, commandBuffer: Stream? = nil) throws -> (xOut: HIPBuffer?, skipOut: HIPBuffer) {
        var xo: UnsafeMutablePointer<Float>? = nil, so: UnsafeMutablePointer<Float>? = nil
        try Self.check(piper_hip_wavenet_layer_f32(ctx, x.f32, skipIn?.f32, wIn.f32, bIn.f32, wRs.f32, bRs.f32, Int64(batch), Int64(channels), Int64(length),
                                                   Int64(kernel), Int64(dilation), last ? 1 : 0, &xo, &so, commandBuffer))
        return (last ? nil : wrap(xo), wrap(so))
    }
    /// HiFi-GAN ResBlock1 (type 1: weights c1_0, c2_0, c1_1, …) / ResBlock2 (type 2) over all its dilations
    public func hifiganResblockF32(type: Int, x: HIPBuffer, batch: Int, channels: Int, length: Int, kernel: Int, dilations: [Int], weights: [HIPBuffer],
                                   biases: [HIPBuffer], slope: Float = 0.1, commandBuffer: Stream? = nil) throws -> HIPBuffer {
        var out: UnsafeMutablePointer<Float>? = nil
        let w: [UnsafePointer<Float>?] = weights.map { UnsafePointer($0.f32) }
        let b: [UnsafePointer<Float>?] = biases.map { UnsafePointer($0.f32) }
        try Self.check(piper_hip_hifigan_resblock_f32(ctx, Int32(type), x.f32, Int64(batch), Int64(channels), Int64(length), Int64(kernel), dilations.map(Int32.init),
                                                      Int32(dilations.count), w, b, slope, &out, commandBuffer))
        return wrap(out)
    }
}
