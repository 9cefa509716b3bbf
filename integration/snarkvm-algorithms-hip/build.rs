// Link against the in-tree build of the library (python -m snarkos_amd.build), or MI355X_SNARK_LIB_DIR.
fn main() {
    let dir = std::env::var("MI355X_SNARK_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../snarkos_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mi355x_snark");
    println!("cargo:rerun-if-env-changed=MI355X_SNARK_LIB_DIR");
}
