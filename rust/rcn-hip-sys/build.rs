// Points rustc at librcn_hip.so (built by `python -m mercer_research_amd.build`, i.e. one hipcc command).
fn main() {
    let dir = std::env::var("RCN_HIP_LIB_DIR").unwrap_or_else(|_| "../../mercer_research_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rcn_hip");
    println!("cargo:rerun-if-env-changed=RCN_HIP_LIB_DIR");
}
