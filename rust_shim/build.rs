// SPH_HIP_LIB_DIR = directory that holds libsph_hip.so (adaptive_sph_amd/csrc after `python -c "import __graft_entry__ as g; g.build()"`)
fn main() {
    let dir = std::env::var("SPH_HIP_LIB_DIR").unwrap_or_else(|_| "../adaptive_sph_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=sph_hip");
    println!("cargo:rerun-if-env-changed=SPH_HIP_LIB_DIR");
}
