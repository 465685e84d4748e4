// Link against libminilp_hip.so.  MINILP_HIP_LIB_DIR points at the directory that holds it
// (the `minilp_amd/` directory of the build tree after `python -m minilp_amd.build`).
fn main() {
    if let Ok(dir) = std::env::var("MINILP_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=minilp_hip");
    println!("cargo:rerun-if-env-changed=MINILP_HIP_LIB_DIR");
}
