// Links libbinius_amd.so (the gfx950 kernels + C ABI, include/binius_amd.h).
fn main() {
	println!("cargo:rerun-if-env-changed=BINIUS_AMD_LIB_DIR");
	if let Ok(dir) = std::env::var("BINIUS_AMD_LIB_DIR") {
		println!("cargo:rustc-link-search=native={dir}");
		println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
	}
	println!("cargo:rustc-link-lib=dylib=binius_amd");
	// the compiled provers behind src/mlecheck.rs (include/binius_amd_host.h)
	if std::env::var("CARGO_FEATURE_PROVERS").is_ok() {
		println!("cargo:rustc-link-lib=dylib=binius_amd_host");
	}
}
