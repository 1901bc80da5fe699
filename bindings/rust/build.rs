// Links the prebuilt HIP library (make -C raytracer-rs_amd builds lib/librtamd.so with hipcc for gfx950).
fn main() {
    let dir = std::env::var("RTAMD_LIB_DIR").expect("set RTAMD_LIB_DIR to <repo>/raytracer-rs_amd/lib");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rtamd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RTAMD_LIB_DIR");
}
