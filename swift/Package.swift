// swift-tools-version:5.9
// UNTESTED GLUE: no Swift toolchain exists in the image this repository is built and tested in (SURVEY.md F5). The C header these
// sources bind (include/piper_hip.h) and its Python twin (piper-swift_amd/python/piper_hip) are what the test-suite exercises;
// this package states the same binding in the reference's own language, next to where `Sources/PiperMetal` sits there.
import PackageDescription

let package = Package(
    name: "PiperHIP",
    products: [.library(name: "PiperHIP", targets: ["PiperHIP"])],
    targets: [
        .systemLibrary(name: "CPiperHIP", path: "Sources/CPiperHIP"),
        .target(name: "PiperHIP", dependencies: ["CPiperHIP"],
                linkerSettings: [.unsafeFlags(["-L", "../piper-swift_amd/lib", "-Xlinker", "-rpath", "-Xlinker", "../piper-swift_amd/lib"])]),
    ]
)
