// HIPBackend — MetalBackend's per-operator surface (Sources/PiperMetal/Execution/MetalBackend.swift:8-3427 in the reference) over
// the C-ABI of include/piper_hip.h. UNTESTED GLUE (see Package.swift); the Python twin with the same method names is tested.
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
    // matmulF32 (:1232), softmaxLastDimF32 (:1326), reluF32/tanhF32/sigmoidF32/leakyReluF32 (:1577-1586 → piper_hip_unary_f32),
    // addF32/subF32/mulF32/divF32 (:2592-2610 → piper_hip_binary_broadcast_f32), padConstantF32 (:780), transposeF32 (:995),
    // expandF32 (:2438), concat/split (tensorops) follow the same pattern: shapes as [Int64], optional stream last.
}
