/*
 * pbrt_hip.h -- C ABI of the MI355X render path ("what `PbrtAPI::world_end` calls").
 *
 * The reference (wathiede/pbrt, Rust) has no FFI layer and no renderer: the seam this
 * library fills is the body of `fn world_end(&mut self)` (/root/reference/src/core/api.rs:432-473,
 * whose render call survives only as the comment at :446-453), reached from the parser's
 * "WorldEnd" arm (src/core/parser.rs:312).  Its inputs are the fields of `RenderOptions`
 * (api.rs:201-224) -- camera_to_world (api.rs:813-820), film/sampler/integrator parameters --
 * plus the geometry the reference never stores (api.rs:220-223 TODO).  INTEGRATION.md shows the
 * `extern "C"` block and the `world_end` body a maintainer of the Rust crate would add.
 *
 * This header is what a host binds.  The hooks of tests and measurements (tree exports, walk / stack-plan introspection, the
 * host-only builders, the tokenizer alone, parser state) live in pbrt_hip_debug.h: same library, not part of the stable ABI.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every HOST pointer it passes, the library
 *    copies what it needs during the call and keeps nothing;
 *  - every function returns 0 on success or a negative pbrt_hip_status; the message is
 *    available from pbrt_hip_last_error() (thread-local); no C++ exception crosses the ABI
 *    (reference error style: api.rs:50-63 `Error`, api.rs:291-332 verify_* = log and continue);
 *  - there is NO CPU fallback: without a HIP device every entry point that computes returns
 *    PBRT_HIP_ERR_NO_DEVICE;
 *  - a scene handle is used by one host thread at a time (api.rs is single-threaded).
 */
#ifndef PBRT_HIP_H
#define PBRT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PBRT_HIP_OK = 0,
  PBRT_HIP_ERR_INVALID = -1,   /* bad argument */
  PBRT_HIP_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime error at init */
  PBRT_HIP_ERR_HIP = -3,       /* a HIP call failed */
  PBRT_HIP_ERR_LIMIT = -4,     /* scene exceeds a compiled-in limit (BVH depth, material count) */
  PBRT_HIP_ERR_INTERNAL = -5
} pbrt_hip_status;

/* Material "matte" / "mirror" (scene files name them: scenes/check-sphere.pbrt:21,29; the
 * reference only has the TODO at api.rs:255-269).  32 bytes. */
typedef struct {
  uint32_t type; /* 0 = matte (Lambertian, k = Kd), 1 = mirror (perfect specular, k = Kr) */
  float k[3];
  float le[3]; /* emitted radiance; non-zero turns every triangle using it into an area light
                  (replaces api.rs:476-478 area_light_source -> todo!()) */
  uint32_t kd_tex; /* matte only: 0 = Kd is k; t > 0 = Kd is textures[t - 1] evaluated at the hit's (u, v): a triangle's corner (u, v)
                      (tri_uv) interpolated like the hit point; a SPHERE's own (u, v) = (phi / 2 pi, 1 - theta / pi) with phi = atan2(n.y, n.x)
                      in [0, 2 pi), theta = acos(n.z) of the unit normal n about the WORLD's z axis (pbrt-v3 Sphere::Intersect for an
                      unrotated sphere: a sphere here is a centre and a radius, its CTM's rotation is not carried -- the parser warns) */
} pbrt_hip_material;

/* Texture "name" "spectrum" "checkerboard" (check-sphere.pbrt:24-25; the reference: api.rs:524-580 stores nothing, texture.rs is an
 * empty marker): pbrt-v3's Checkerboard2DTexture over a UVMapping2D, point-sampled (aamode "none": no ray differentials here):
 * (s, t) = (su u + du, sv v + dv); tex1 where floor(s) + floor(t) is even, tex2 where odd.  64 bytes. */
typedef struct {
  uint32_t type;  /* 0 = checkerboard, dimension 2, "uv" mapping */
  float tex1[3], tex2[3];
  float su, sv, du, dv; /* "float uscale" / "vscale" / "udelta" / "vdelta" */
  uint32_t pad[5];
} pbrt_hip_texture;

/* LightSource "point" / "distant" / "infinite" (api.rs:334-351 make_light: todo!() for all). */
typedef struct {
  uint32_t type; /* 0 point, 1 distant, 2 infinite with constant radiance */
  float p[3];    /* point: position; distant: unit direction TOWARDS the light */
  float c[3];    /* point: intensity I; distant / infinite: radiance L */
  float pad;
} pbrt_hip_light;

/* Shape "sphere" (check-sphere.pbrt:22), world space.  A primitive of the BVH like a triangle (primitive number n_tris + index, which is
 * what pbrt_hip_intersect reports), bounded by [c - r, c + r]: thousands of spheres cost a ray what thousands of triangles do. */
typedef struct {
  float c[3];
  float r;
  uint32_t mat;
  uint32_t pad[3];
} pbrt_hip_sphere;

/* Everything RenderOptions would hold at WorldEnd (api.rs:201-224), as arrays. */
typedef struct {
  const float *P;         /* "point P": 3 * n_verts (parser.rs:821-839 shows the param shape) */
  const uint32_t *idx;    /* "integer indices": 3 * n_tris */
  const uint16_t *mat_id; /* n_tris, index into mats */
  const pbrt_hip_material *mats;
  const pbrt_hip_light *lights;
  const pbrt_hip_sphere *spheres;
  uint32_t n_verts, n_tris, n_mats, n_lights, n_spheres;
  float cam_to_world[16]; /* RenderOptions.camera_to_world, row-major Matrix4x4 (transform.rs:75-77) */
  float fov;              /* Camera "perspective" "float fov" (degrees, shorter axis) */
  int32_t xres, yres;     /* Film "integer xresolution" / "yresolution" */
  float crop[4];          /* Film "float cropwindow" x0 x1 y0 y1 (film.rs:92-101) */
  /* textured materials (all three may be 0 / NULL): */
  const float *tri_uv;    /* 6 * n_tris: (u, v) of the three corners of every triangle -- Shape "trianglemesh" "float st" / "uv"
                             (check-sphere.pbrt:31), or pbrt-v3's default (0,0) (1,0) (1,1) for a mesh without them.  Needed
                             (PBRT_HIP_ERR_INVALID otherwise) when a material of a triangle has kd_tex != 0 */
  const pbrt_hip_texture *textures;
  uint32_t n_textures;
} pbrt_hip_scene_desc;

#define PBRT_HIP_INTEGRATOR_PATH 0   /* Integrator "path" (default name, api.rs:239) */
#define PBRT_HIP_INTEGRATOR_DIRECT 1 /* Integrator "directlighting" */
#define PBRT_HIP_INTEGRATOR_PATH_MIS 2 /* Integrator "path" "bool mis" "true": the path integrator with the direct-light estimate
                                          MULTIPLE-IMPORTANCE-SAMPLED as pbrt-v3's path integrator does (power heuristic between the
                                          one light sample and the BSDF-sampled bounce ray: DESIGN.md 3.14; integrator 0 is SURVEY A8's
                                          "no MIS in v1").  Not with the counter flags */
#define PBRT_HIP_FLAG_COUNTERS 1u    /* count nodes visited / triangles tested of the canonical walk:
                                        runs the exact-order instantiation of the kernel, equal to the oracle's counters */
#define PBRT_HIP_FLAG_WALK_COUNTERS 2u /* count what the production kernel itself does instead: nodes_visited = 64-byte
                                        child-pair records fetched, tris_tested = triangles tested */

#define PBRT_HIP_SAMPLER_STRATIFIED 0 /* Sampler "stratified" (north_star's sampler) */
#define PBRT_HIP_SAMPLER_SOBOL 1      /* Sampler "02sequence" / "lowdiscrepancy": the (0,2)-sequence sampler of DESIGN.md 3.10
                                         (the reference holds only names, api.rs:235, and the generator matrices,
                                         sobolmatrices.rs:81) */
#define PBRT_HIP_SAMPLER_SOBOL_ND 2   /* Sampler "sobol": Sobol' proper -- request j of a sample takes its own dimensions (2j, 2j + 1)
                                         of the first 128 of the reference's table (sobolmatrices.rs:81): 64 requests, every one a
                                         path of maxdepth 16 can make; later requests the padded scheme of sampler 1 (DESIGN.md
                                         3.12).  Not with the counter flags */
#define PBRT_HIP_SAMPLER_HALTON 3     /* Sampler "halton" -- the reference's DEFAULT sampler name (api.rs:235): scrambled radical
                                         inverses in the prime bases 2 .. 719, request j of a sample on the dimensions (2j, 2j + 1)
                                         for 64 requests.  Not with the counter flags */
#define PBRT_HIP_MAX_SPP (1u << 20)   /* spp_x * spp_y: the kernels pack the sample index into 20 bits */
#define PBRT_HIP_MAX_DEPTH 1023u      /* max_depth: the bounce count is packed into 10 bits */

typedef struct {
  uint32_t integrator;
  uint32_t max_depth;        /* Integrator "integer maxdepth" (<= PBRT_HIP_MAX_DEPTH, else PBRT_HIP_ERR_LIMIT) */
  uint32_t spp_x, spp_y;     /* Sampler "stratified": pixelsamples = spp_x * spp_y (<= PBRT_HIP_MAX_SPP, else PBRT_HIP_ERR_LIMIT) */
  uint64_t seed;
  uint32_t rank, world_size; /* this process renders the 64x64 super-tiles t with t % world_size == rank */
  uint32_t flags;
  uint32_t sampler;          /* PBRT_HIP_SAMPLER_* */
  float filter_xwidth, filter_ywidth; /* PixelFilter "box" "float xwidth" / "ywidth" (box.rs:57-61): the filter's radii in
                                         pixels.  0 = the default 0.5 (a sample lands in its own pixel, film.rs:264-273 needs
                                         no tile overlap).  Any other radius in (0, 16]: every sample inside the sample bounds
                                         (film.rs:166-175) adds, weight 1, to all pixels within the radius; the film is then
                                         accumulated in 64-bit fixed point -- see pbrt_hip_render_acc below */
  float max_sample_luminance;         /* Film "float maxsampleluminance" (film.rs:75,279; pbrt-v3 FilmTile::AddSample): a
                                         sample whose luminance exceeds it is scaled down to it.  0 = no bound */
} pbrt_hip_render_desc;

typedef struct {
  uint64_t camera_rays, bounce_rays, shadow_rays; /* filled only with PBRT_HIP_FLAG_COUNTERS */
  uint64_t nodes_visited, tris_tested;            /* " */
  double kernel_ms;                               /* HIP-event time of the render kernel on its stream */
  uint64_t samples;                               /* camera samples this call rendered */
} pbrt_hip_stats;

typedef struct pbrt_hip_scene pbrt_hip_scene;

/* number of visible HIP devices (0 when there is none; never fails) */
int pbrt_hip_device_count(void);
const char *pbrt_hip_last_error(void);
const char *pbrt_hip_version(void); /* "pbrt_hip 0.5 (gfx950)": 0.5 = round 5's ABI (pbrt_hip_scene_desc gained tri_uv / textures / n_textures at its
                                       end, pbrt_hip_material.pad became kd_tex, flags 0 of scene_create_ex = the device builder) */
/* identity of the build: a hash of the library's sources and kernel-shaping flags (pbrt_amd/build.py source_id).  A
 * profile taken on one build must not price another: bench.py compares this with the id stored beside the counters */
const char *pbrt_hip_build_id(void);

/* ---- scene: upload to HBM, accelerator build.  device < 0: current device.
 * Input is validated before any device work: indices in range, vertices / spheres / camera matrix finite, sphere radii
 * positive, 0 < fov < 180 (PBRT_HIP_ERR_INVALID otherwise).
 * ONE default, whoever builds (this call, pbrt_hip_scene_create_ex with flags 0, pbrt_hip_multi_create, pbrt_hip_render_multi,
 * the command line): the accelerator is built ON THE DEVICE from the uploaded vertex / index buffers -- Morton order,
 * level-synchronous binned SAH, the tree (from 1024 triangles on) optimised by parallel re-insertion, collapse into the
 * quantised 4-wide nodes; stands in for what core/api.rs:237 names "bvh" and api.rs:446-453 would have
 * built: 0.1 s for 1M triangles.  So a `world_end` that calls pbrt_hip_scene_create + pbrt_hip_render gets the same tree as
 * one that calls pbrt_hip_render_multi.  PBRT_HIP_BUILDER=host in the environment turns the default into
 * PBRT_HIP_SCENE_HOST_BUILD for callers that left the choice open (flags 0). ---- */
int pbrt_hip_scene_create(const pbrt_hip_scene_desc *desc, int device, pbrt_hip_scene **out);
/* The same with options (0 = the default above).
 * PBRT_HIP_SCENE_GPU_BUILD: the default, said explicitly (PBRT_HIP_BUILDER is then not consulted).  The SAME film and hit
 * records bit for bit whichever builder is used (a hit does not depend on the tree).  The counter flags of
 * render / intersect count the oracle's canonical walk: for a device-built scene the canonical tree is built on the host at
 * the first call that asks for them (vertex / index buffers read back; about a second for 1M triangles, never on the
 * render path). */
#define PBRT_HIP_SCENE_GPU_BUILD 1u
/* PBRT_HIP_SCENE_OPTIMIZED_TREE: the HOST builder followed by the tree optimisation the device builder applies -- parallel
 * re-insertion (pbrt_amd/csrc/reinsert_core.hpp), run on one host core: a few seconds per million triangles, 4-6 % fewer node
 * fetches per ray than the plain host tree.  For hosts that must build on the CPU.  Not combined with PBRT_HIP_SCENE_GPU_BUILD /
 * PBRT_HIP_SCENE_PLAIN_TREE (PBRT_HIP_ERR_INVALID). */
#define PBRT_HIP_SCENE_OPTIMIZED_TREE 2u
/* PBRT_HIP_SCENE_PLAIN_TREE: the device's binned-SAH tree as built, without the re-insertion passes (A-B measurements). */
#define PBRT_HIP_SCENE_PLAIN_TREE 4u
/* PBRT_HIP_SCENE_HOST_BUILD: the host's binned-SAH builder (the canonical tree of the oracle and of the counter
 * flags, collapsed into the 4-wide nodes as it is): 0.7 s per million triangles on one core and a tree rays cross in ~4.5 % more
 * steps -- kept for A-B runs and for tests of that builder.  Not combined with PBRT_HIP_SCENE_GPU_BUILD / _PLAIN_TREE. */
#define PBRT_HIP_SCENE_HOST_BUILD 8u
int pbrt_hip_scene_create_ex(const pbrt_hip_scene_desc *desc, int device, uint32_t flags, pbrt_hip_scene **out);
/* how the accelerator was built: *gpu_built 0 / 1, *build_ms = host build time (wall) or device build time (events) */
int pbrt_hip_scene_build_info(const pbrt_hip_scene *scene, uint32_t *gpu_built, double *build_ms);
/* the device build's tree optimisation (parallel re-insertion, pbrt_amd/csrc/reinsert_core.hpp): passes run, nodes moved, and
 * its share of build_ms; zeros for a host-built scene or PBRT_HIP_SCENE_PLAIN_TREE.  Any pointer may be NULL. */
int pbrt_hip_scene_optimize_info(const pbrt_hip_scene *scene, uint32_t *passes, uint32_t *moves, double *ms);
void pbrt_hip_scene_destroy(pbrt_hip_scene *scene);
/* scene introspection: nodes / depth of the canonical tree (0 for a device-built scene until a counter flag built it), lights
 * (explicit + emissive triangles), bytes held in HBM */
int pbrt_hip_scene_info(const pbrt_hip_scene *scene, uint32_t *n_nodes, uint32_t *depth, uint32_t *n_lights,
                        uint64_t *device_bytes);
/* ---- the hot path ---- */
/* Render this rank's super-tiles and return the assembled film in HOST memory:
 * film_xyzw = (crop_w * crop_h * 4) floats, row-major over the cropped pixel bounds,
 * {X, Y, Z, filter_weight_sum} per pixel = Film.pixels after merge_film_tile (film.rs:47-55,313-326).
 * Pixels of super-tiles owned by other ranks are written as zeros.  A crop window that holds no pixel (crop_w or crop_h
 * 0) is a film of no floats: nothing is sampled, nothing is written, PBRT_HIP_OK (film_xyzw must still be non-NULL). */
int pbrt_hip_render(pbrt_hip_scene *scene, const pbrt_hip_render_desc *desc, float *film_xyzw, pbrt_hip_stats *stats);
/* Allocates (or grows) the device scratch a render of `scene` with this description needs -- the lanes' path-state records, the
 * partial film sums of the work items, the overflow area of the walk's stack, sampler 2's matrices -- without launching anything.
 * pbrt_hip_render_device does the same on demand; a host about to launch one frame on SEVERAL GPUs calls this for every GPU first,
 * so that no allocation (a synchronising call) separates the launches (pbrt_hip_multi_render does).  Stands in for nothing in the
 * reference (core/api.rs:432-473 renders nothing); part of the boundary's launch plumbing. */
int pbrt_hip_render_prepare(pbrt_hip_scene *scene, const pbrt_hip_render_desc *r);
/* Same, but asynchronous on `stream` (a hipStream_t, may be NULL) into a DEVICE slab of
 * pbrt_hip_slab_floats() floats: this rank's super-tiles back to back, 64*64 float4 each.
 * No synchronisation is done; stats->kernel_ms is filled by pbrt_hip_render_wait(). */
int pbrt_hip_render_device(pbrt_hip_scene *scene, const pbrt_hip_render_desc *desc, void *d_slab, void *stream);
int pbrt_hip_render_wait(pbrt_hip_scene *scene, pbrt_hip_stats *stats);
/* scatter one rank's slab (device) into a row-major film (device), both float4 per pixel */
int pbrt_hip_film_assemble_device(const pbrt_hip_scene *scene, const void *d_slab, uint32_t rank, uint32_t world_size,
                                  void *d_film_xyzw, void *stream);
/* ---- box filter radii other than 0.5 (pbrt_hip_render_desc.filter_xwidth / ywidth) ----
 * The reference's Film keeps a filter table for any radius (film.rs:113-123), expands a tile by it (film.rs:264-273) and
 * merges tiles under a mutex (film.rs:313-326); it has no add_sample.  Here such a film is accumulated in 64-bit fixed
 * point -- four int64 per pixel of the cropped window, {r, g, b} in units of 2^-24 (a component of a sample clamped to
 * [0, 2^15]) and the sample count -- with integer atomics: integer sums do not depend on the order of arrival, so the
 * image is reproducible and the accumulators of several ranks ADD to those of one rank exactly (multi-GPU: one sum
 * reduction instead of the gather).  pbrt_hip_render / pbrt_hip_render_multi accept such a desc like any other (with
 * world_size > 1, pbrt_hip_render's film then holds the rank's OWN samples only: combine ranks by adding accumulators).
 *  - pbrt_hip_render_buffer_bytes: bytes pbrt_hip_render_device writes for this desc -- the rank's slab (default filter)
 *    or the accumulators of the whole cropped window, 32 bytes per pixel (zeroed by the call);
 *  - pbrt_hip_render_acc: this rank's accumulators in HOST memory (crop_w * crop_h * 4 int64);
 *  - pbrt_hip_film_from_acc_device / pbrt_hip_film_from_acc: accumulators -> Film pixels {X, Y, Z, weight}
 *    (= Film::merge_film_tile's arithmetic on the radiance sum, film.rs:313-326), on the device or on the host. */
int64_t pbrt_hip_render_buffer_bytes(const pbrt_hip_scene *scene, const pbrt_hip_render_desc *desc);
int pbrt_hip_render_acc(pbrt_hip_scene *scene, const pbrt_hip_render_desc *desc, int64_t *acc, pbrt_hip_stats *stats);
int pbrt_hip_film_from_acc_device(const pbrt_hip_scene *scene, const void *d_acc, void *d_film_xyzw, void *stream);
void pbrt_hip_film_from_acc(const int64_t *acc, int64_t n_pixels, float *film_xyzw);
/* host-side geometry of the sharding (no device needed) */
int64_t pbrt_hip_slab_floats(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world_size);
/* for every float4 slot of a rank's slab the row-major pixel index inside the cropped film, or -1 */
int pbrt_hip_slab_pixel_index(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world_size,
                              int64_t *out);

/* ---- every GPU of the node behind one call: what a single-process host (the reference's world_end, api.rs:432-473,
 * reached from bin/pbrt.rs:72-83) needs to render on 8 MI355X.  The scene is built once on GPU 0 and copied device to
 * device (xGMI); GPU g renders the super-tiles t with t % n_gpus == g on its own stream from its own host thread; ONE
 * RCCL gather (ncclGather, communicators from ncclCommInitAll) brings the slabs to GPU 0, where they are assembled into
 * the film.  n_gpus <= 0: all visible devices.  desc->rank / world_size are ignored (the library sets them).
 * film_xyzw (host, crop_w * crop_h * 4 floats) may be NULL when only the device film is wanted;
 * per_gpu (n_gpus entries) may be NULL.  flags as pbrt_hip_scene_create_ex. ---- */
typedef struct pbrt_hip_multi pbrt_hip_multi;
int pbrt_hip_multi_create(const pbrt_hip_scene_desc *desc, int n_gpus, uint32_t flags, pbrt_hip_multi **out);
int pbrt_hip_multi_gpus(const pbrt_hip_multi *multi);
int pbrt_hip_multi_render(pbrt_hip_multi *multi, const pbrt_hip_render_desc *desc, float *film_xyzw, pbrt_hip_stats *per_gpu);
/* the assembled film of the last pbrt_hip_multi_render on GPU 0 (float4 per pixel, row-major) */
int pbrt_hip_multi_film_device(pbrt_hip_multi *multi, void **d_film_xyzw);
void pbrt_hip_multi_destroy(pbrt_hip_multi *multi);
/* create + render + destroy: the whole of `world_end` in one call */
int pbrt_hip_render_multi(const pbrt_hip_scene_desc *desc, const pbrt_hip_render_desc *render, int n_gpus, float *film_xyzw,
                          pbrt_hip_stats *per_gpu);

/* Ray-batch entry points: the traversal kernels on their own (parity + roofline of the
 * dominant loop).  Host SoA-of-xyz arrays, n rays.  prim = 0xffffffff on a miss.  A ray whose origin or direction has a component
 * that is not finite, or whose tmax is NaN, is a miss (and is not walked); direction components that are exactly 0 are fine.
 * What counts as a hit (the own-box rule): the triangle / sphere test's candidate, provided the ray meets the primitive's own
 * bounding box; t is at least the distance at which it enters that box.  Whether and where a ray hits a primitive depends on the ray and the
 * primitive alone -- never on the tree, the builder or the walk -- and ties on t go to the lower primitive number (spheres: n_tris + index). */
int pbrt_hip_intersect(pbrt_hip_scene *scene, int64_t n, const float *o, const float *d, const float *tmax, float *t,
                       uint32_t *prim, float *b1, float *b2, uint64_t *counters /* 2, may be NULL */);
int pbrt_hip_occluded(pbrt_hip_scene *scene, int64_t n, const float *o, const float *d, const float *tmax,
                      uint8_t *hit);

/* ---- host pieces either side of the path that the reference does implement ---- */
/* Film::new / get_sample_bounds / get_film_tile (film.rs:82-137,166-175,264-281); bounds are x0 y0 x1 y1 */
void pbrt_hip_film_cropped_bounds(int32_t xres, int32_t yres, const float crop[4], int32_t out[4]);
void pbrt_hip_film_sample_bounds(int32_t xres, int32_t yres, const float crop[4], float rx, float ry, int32_t out[4]);
void pbrt_hip_film_tile_bounds(int32_t xres, int32_t yres, const float crop[4], float rx, float ry,
                               const int32_t sample_bounds[4], int32_t out[4]);
/* Film::write_image's pixel arithmetic (film.rs:340-372): xyzw -> linear RGB, 3 floats per pixel */
void pbrt_hip_film_to_rgb(const float *film_xyzw, int64_t n_pixels, float scale, float *rgb);
/* imageio::write_image (imageio.rs:235-283): ".png" (8-bit sRGB via to_byte, imageio.rs:66-68) or ".pfm" */
int pbrt_hip_write_image(const char *name, const float *rgb, int32_t width, int32_t height);
/* imageio::read_image (imageio.rs:87-184): ".png" (8-bit, value = byte / 255) or ".pfm".  Call with rgb == NULL to get
 * the size, then with a width*height*3 buffer (width / height must then hold that size). */
int pbrt_hip_read_image(const char *name, float *rgb, int32_t *width, int32_t *height);
/* Transform::look_at (transform.rs:485-520): m = world->camera, m_inv = camera->world */
void pbrt_hip_look_at(const float pos[3], const float look[3], const float up[3], float m[16], float m_inv[16]);

/* ---- scene ingestion: the reference's parser + API state machine, completed for this path ----
 * (parser.rs:205-317 handles 12 of 37 directive arms; api.rs stores no geometry.)  Host only.
 * Status codes: 0 ok; PBRT_HIP_ERR_INVALID with last_error() = "<kind>: <detail>", kind in
 * {Eof, UnterminatedString, MixedParameters, Unquoted, Syntax, NotImplemented, Io} (parser.rs:31-58). */
typedef struct pbrt_hip_loaded pbrt_hip_loaded;
int pbrt_hip_load_file(const char *path, pbrt_hip_loaded **out);
int pbrt_hip_load_string(const char *text, size_t len, const char *base_dir /* for Include, may be NULL */,
                         pbrt_hip_loaded **out);
void pbrt_hip_loaded_free(pbrt_hip_loaded *loaded);
/* desc / render are filled with pointers INTO `loaded` (valid until it is freed); filename = Film "string filename" */
int pbrt_hip_loaded_get(const pbrt_hip_loaded *loaded, pbrt_hip_scene_desc *desc, pbrt_hip_render_desc *render,
                        char *filename, size_t filename_cap);
/* Film "float scale" (film.rs:368-371: every pixel is multiplied by it in Film::write_image): pass it to
 * pbrt_hip_film_to_rgb.  1 for a NULL handle. */
float pbrt_hip_loaded_film_scale(const pbrt_hip_loaded *loaded);
/* warnings (ignored directives / parameters, api.rs:291-332 "log and continue"), '\n'-separated; returns their count */
int pbrt_hip_loaded_warnings(const pbrt_hip_loaded *loaded, char *buf, size_t cap);
#ifdef __cplusplus
}
#endif
#endif
