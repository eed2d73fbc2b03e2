// Links libwhisper_hip.so (built by `make -C whisper-burn_amd/csrc`, hipcc --offload-arch=gfx950).
// WHISPER_HIP_LIB_DIR overrides the search path; the default is the in-tree build output.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("WHISPER_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../whisper-burn_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=whisper_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=WHISPER_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/whisper_hip.h");
}
