// Links libsassy_hip.so: SASSY_HIP_LIB_DIR, or ../../sassy_amd/lib relative to this crate.
fn main() {
    let dir = std::env::var("SASSY_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../sassy_amd/lib")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sassy_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
