// Links libballista_hip.so (built by `make -C ballista_amd/csrc`); BALLISTA_HIP_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("BALLISTA_HIP_LIB_DIR").unwrap_or_else(|_| "../../ballista_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=ballista_hip");
    println!("cargo:rerun-if-env-changed=BALLISTA_HIP_LIB_DIR");
}
