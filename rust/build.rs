// crates/orchestrator/build.rs — link libpm_engine.so (SOURCE ONLY, never compiled here).
fn main() {
    let dir = std::env::var("PM_ENGINE_LIB_DIR").unwrap_or_else(|_| "/opt/pm_engine/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pm_engine");
    println!("cargo:rerun-if-env-changed=PM_ENGINE_LIB_DIR");
}
