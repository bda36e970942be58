// One-shot weight broadcast for N-GPU replica serving (one process per GPU) over the library's rccl.h wrapper. UNTESTED GLUE.
import CPiperHIP

public final class PiperHIPComm {
    private var comm: OpaquePointer?

    /// rank 0 calls `makeID()` and hands the 128 bytes to the other ranks by any side channel; every rank then constructs the communicator.
    public static func makeID() throws -> [UInt8] {
        var id = [UInt8](repeating: 0, count: Int(PIPER_HIP_COMM_ID_BYTES))
        try HIPBackend.check(piper_hip_comm_unique_id(&id))
        return id
    }
    public init(backend: HIPBackend, id: [UInt8], rank: Int32, world: Int32) throws {
        try HIPBackend.check(piper_hip_comm_create(backend.ctx, id, rank, world, &comm))       // collective
    }
    deinit { piper_hip_comm_destroy(comm) }

    public var world: Int32 { piper_hip_comm_world(comm) }                                       // ncclCommCount
    /// in place; returns when the data is in this rank's HBM
    public func broadcast(_ device: UnsafeMutablePointer<Float>, count: Int, root: Int32 = 0) throws {
        try HIPBackend.check(piper_hip_comm_broadcast_f32(comm, device, count, root))
    }
    public func max(_ value: Double) throws -> Double {
        var v = value
        try HIPBackend.check(piper_hip_comm_max_f64(comm, &v))
        return v
    }
    public func barrier() throws { try HIPBackend.check(piper_hip_comm_barrier(comm)) }
}
