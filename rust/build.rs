// NOT COMPILED HERE.  Links the gfx950 library built by `python -m basisu_rs_amd.build` (or the hipcc command in INTEGRATION.md).
fn main() {
    if let Ok(dir) = std::env::var("BASISU_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=basisu_hip");
    println!("cargo:rerun-if-env-changed=BASISU_HIP_LIB_DIR");
}
