// UNVERIFIED (no Rust toolchain in the build image).  CFTRT_LIB_DIR = the directory that holds libcftrt.so
// (ray-tracing_amd/csrc after `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    if let Ok(dir) = std::env::var("CFTRT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=cftrt");
    println!("cargo:rerun-if-env-changed=CFTRT_LIB_DIR");
}
