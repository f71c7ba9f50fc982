// UNCOMPILED DOCUMENTATION -- there is no Rust toolchain in the image this repository is built in (no cargo, no
// rustc), so nothing in this file has ever been through a compiler.  It is what a maintainer of wathiede/pbrt would
// add to call libpbrt_hip.so from `PbrtAPI::world_end` (src/core/api.rs:432-473): the `extern "C"` mirror of
// include/pbrt_hip.h and the body of `world_end`.  INTEGRATION.md explains each piece; the C++ program
// pbrt_amd/csrc/pbrt_main.cpp does the same through the same C ABI and IS compiled and tested.

// ---------------------------------------------------------------- build.rs
fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("PBRT_HIP_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=pbrt_hip");        // pbrt_amd/lib/libpbrt_hip.so
}

// ---------------------------------------------------------------- src/core/hip.rs
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct HipMaterial { pub kind: u32, pub k: [f32; 3], pub le: [f32; 3], pub kd_tex: u32 }  // kd_tex: 0, or 1 + the index of the checkerboard that is the matte Kd
#[repr(C)] pub struct HipTexture  { pub kind: u32, pub tex1: [f32; 3], pub tex2: [f32; 3], pub su: f32, pub sv: f32, pub du: f32, pub dv: f32, pub pad: [u32; 5] }  // Texture "..." "spectrum" "checkerboard" (check-sphere.pbrt:24-25)
#[repr(C)] pub struct HipLight    { pub kind: u32, pub p: [f32; 3], pub c: [f32; 3], pub pad: f32 }
#[repr(C)] pub struct HipSphere   { pub c: [f32; 3], pub r: f32, pub mat: u32, pub pad: [u32; 3] }
#[repr(C)] pub struct HipSceneDesc {
    pub p: *const f32, pub idx: *const u32, pub mat_id: *const u16,
    pub mats: *const HipMaterial, pub lights: *const HipLight, pub spheres: *const HipSphere,
    pub n_verts: u32, pub n_tris: u32, pub n_mats: u32, pub n_lights: u32, pub n_spheres: u32,
    pub cam_to_world: [f32; 16], pub fov: f32, pub xres: i32, pub yres: i32, pub crop: [f32; 4],
    pub tri_uv: *const f32,           // 6 per triangle: corner (u, v) ("float st" / "uv"); NULL unless a triangle's material is textured
    pub textures: *const HipTexture, pub n_textures: u32,
}
#[repr(C)] pub struct HipRenderDesc {
    pub integrator: u32, pub max_depth: u32, pub spp_x: u32, pub spp_y: u32,
    pub seed: u64, pub rank: u32, pub world_size: u32, pub flags: u32,
    pub sampler: u32,                                // 0 stratified, 1 the (0,2)-sequence sampler, 2 Sobol' proper, 3 Halton proper (the default NAME, api.rs:235)
    pub filter_xwidth: f32, pub filter_ywidth: f32,  // box filter radii (box.rs:57-61); 0 = 0.5; any radius in (0, 16]
    pub max_sample_luminance: f32,                   // Film "maxsampleluminance" (film.rs:75,279); 0 = no bound
}
#[repr(C)] #[derive(Default)] pub struct HipStats {
    pub camera_rays: u64, pub bounce_rays: u64, pub shadow_rays: u64,
    pub nodes_visited: u64, pub tris_tested: u64, pub kernel_ms: f64, pub samples: u64,
}
#[repr(C)] pub struct HipScene { _private: [u8; 0] }
#[repr(C)] pub struct HipMulti { _private: [u8; 0] }
#[repr(C)] pub struct HipLoaded { _private: [u8; 0] }

extern "C" {
    pub fn pbrt_hip_device_count() -> c_int;
    pub fn pbrt_hip_last_error() -> *const c_char;
    pub fn pbrt_hip_scene_create(desc: *const HipSceneDesc, device: c_int, out: *mut *mut HipScene) -> c_int;
    pub fn pbrt_hip_scene_create_ex(desc: *const HipSceneDesc, device: c_int, flags: u32, out: *mut *mut HipScene) -> c_int; // flags: 0 = the default = what pbrt_hip_scene_create and pbrt_hip_render_multi do: the BVH built AND optimised on the GPU (binned SAH + parallel re-insertion: 0.1 s per 1 M triangles); 1 = the same, said explicitly; 4 = without the optimisation; 8 = the host's binned-SAH builder (0.7 s, ~4.5 % more node fetches per ray); 2 = host build + the optimisation run on one host core (seconds)
    pub fn pbrt_hip_scene_destroy(scene: *mut HipScene);
    pub fn pbrt_hip_render(scene: *mut HipScene, desc: *const HipRenderDesc,
                           film_xyzw: *mut f32, stats: *mut HipStats) -> c_int;
    pub fn pbrt_hip_render_device(scene: *mut HipScene, desc: *const HipRenderDesc,
                                  d_slab: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn pbrt_hip_render_wait(scene: *mut HipScene, stats: *mut HipStats) -> c_int;
    // every GPU of the node from this one process: scene copied device to device, one host thread + stream per GPU,
    // one ncclGather to GPU 0 (create + render + destroy in one call; pbrt_hip_multi_* keep the handle)
    pub fn pbrt_hip_render_prepare(scene: *mut HipScene, render: *const HipRenderDesc) -> c_int; // a frame's device scratch, allocated before any launch (multi-GPU hosts: every GPU first)
    pub fn pbrt_hip_render_multi(desc: *const HipSceneDesc, render: *const HipRenderDesc, n_gpus: c_int,
                                 film_xyzw: *mut f32, per_gpu: *mut HipStats) -> c_int;
    pub fn pbrt_hip_intersect(scene: *mut HipScene, n: i64, o: *const f32, d: *const f32, tmax: *const f32,
                              t: *mut f32, prim: *mut u32, b1: *mut f32, b2: *mut f32, counters: *mut u64) -> c_int;
    // scene ingestion, should the crate prefer this library's parser to completing its own (parser.rs:226-313):
    pub fn pbrt_hip_load_file(path: *const c_char, out: *mut *mut HipLoaded) -> c_int;
    pub fn pbrt_hip_loaded_get(loaded: *const HipLoaded, desc: *mut HipSceneDesc, render: *mut HipRenderDesc,
                               filename: *mut c_char, cap: usize) -> c_int;
    pub fn pbrt_hip_loaded_free(loaded: *mut HipLoaded);
    // image output / input with the reference's conventions (imageio.rs:66-68,87-283):
    pub fn pbrt_hip_film_to_rgb(film_xyzw: *const f32, n_pixels: i64, scale: f32, rgb: *mut f32);
    pub fn pbrt_hip_write_image(name: *const c_char, rgb: *const f32, width: i32, height: i32) -> c_int;
    pub fn pbrt_hip_read_image(name: *const c_char, rgb: *mut f32, width: *mut i32, height: *mut i32) -> c_int;
    // ... the remaining entry points of include/pbrt_hip.h bind the same way
}

// ---------------------------------------------------------------- src/core/api.rs (world_end)
fn world_end(&mut self) {
    verify_world!(self, "WorldEnd");
    // ... existing attribute / transform stack checks (api.rs:434-444) ...
    let ro = &self.render_options;
    let desc = ro.to_hip_scene_desc();              // section 1: arrays collected while parsing

    // Film::new (film.rs:82-137) keeps owning the pixels; the library fills them.
    let film = Film::new(res, crop, filter, diagonal, filename, scale, max_lum);
    let b = film.cropped_pixel_bounds;
    let mut xyzw = vec![0f32; (b.area() * 4) as usize];
    let rd = HipRenderDesc { integrator: if ro.integrator_name == "directlighting" { 1 } else if ro.integrator_params.find_one_bool("mis", true) { 2 } else { 0 },  // "path" = pbrt-v3's: MIS (2); "bool mis" "false": 0
                             max_depth: ro.integrator_params.find_one_int("maxdepth", 5) as u32,
                             spp_x, spp_y, seed: 0, rank: 0, world_size: 1, flags: 0,
                             sampler: match ro.sampler_name.as_str() { "stratified" => 0, "sobol" => 2, "halton" => 3, _ => 1 },
                             filter_xwidth: filter.radius.x, filter_ywidth: filter.radius.y,
                             max_sample_luminance: if max_lum.is_finite() { max_lum } else { 0. } };
    let n_gpus = unsafe { pbrt_hip_device_count() };   // all of them: 8 on an MI355X node
    let mut stats = vec![HipStats::default(); n_gpus.max(1) as usize];
    // n_gpus = 0: every visible GPU (no more than the film has 64x64 super-tiles); the accelerator is built on the device
    let rc = unsafe { pbrt_hip_render_multi(&desc, &rd, 0, xyzw.as_mut_ptr(), stats.as_mut_ptr()) };
    if rc != 0 { error!("pbrt_hip_render_multi: {}", hip_last_error()); return; }   // api.rs:291-332 style: log, continue

    // xyzw is Film.pixels AFTER merge_film_tile (film.rs:313-326): {xyz, filter_weight_sum}
    film.set_pixels_xyzw(&xyzw);     // new: Pixel.xyz / filter_weight_sum from the buffer (film.rs:47-55)
    film.write_image(1.);            // unchanged: film.rs:340-383 -> imageio::write_image
    self.current_api_state = APIState::OptionsBlock;   // api.rs:458
}

