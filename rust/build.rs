// Links libbitnuc_hip.so.  Set BITNUC_HIP_LIB_DIR to the directory holding it
// (in this repo: bitnuc_amd/).
fn main() {
    if let Ok(dir) = std::env::var("BITNUC_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=bitnuc_hip");
}
