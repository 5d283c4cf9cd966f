// Where libsalva_hip.so lives: $SALVA_HIP_LIB_DIR, or salva_amd/csrc of this repository (`make -C salva_amd/csrc`).
fn main() {
    let dir = std::env::var("SALVA_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../../salva_amd/csrc", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=SALVA_HIP_LIB_DIR");
}
