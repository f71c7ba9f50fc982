/*
 * pbrt_oracle.h -- C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and there only as the checker / reported baseline.  The product path
 * (pbrt_amd/, include/pbrt_hip.h) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the rendered pixels.  The reference
 * (wathiede/pbrt) has no renderer; this oracle is pinned
 * against the reference's own known-answer vectors only where the reference has
 * code: PCG32 (src/core/rng.rs:46-93, vectors :131-175), Film geometry
 * (src/core/film.rs:82-175,264-281, vectors :151-164,251-262,186-216), RGB<->XYZ
 * (src/core/spectrum.rs:129-145), look_at / Gauss-Jordan inverse
 * (src/core/transform.rs:162-234,485-520), quadratic (src/lib.rs:181-203, vectors
 * :171-180), gamma / to_byte (src/lib.rs:93-99, src/core/imageio.rs:66-68).
 * Everything in the integrator itself follows the written spec in DESIGN.md
 * section 3 (derived from pbrt-v3, which the reference declares as its model,
 * README.md:12-13).
 */
#ifndef PBRT_ORACLE_H
#define PBRT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Scene description: plain pointers + counts.  Field-for-field the same layout as
 * include/pbrt_hip.h's pbrt_hip_scene_desc so a test can hand both the same bytes,
 * but declared independently. */
typedef struct {
  uint32_t type;  /* 0 = matte, 1 = mirror */
  float k[3];     /* Kd (matte) or Kr (mirror) */
  float le[3];    /* emitted radiance (one-sided, along the geometric normal) */
  uint32_t kd_tex; /* matte: 0 = Kd is k, t > 0 = textures[t - 1] at the hit's (u, v) on triangles */
} orc_material;

typedef struct {
  uint32_t type;  /* 0 = checkerboard 2D over (u, v): pbrt-v3 Checkerboard2DTexture, aamode none */
  float tex1[3], tex2[3];
  float su, sv, du, dv;
  uint32_t pad[5];
} orc_texture;

typedef struct {
  uint32_t type;  /* 0 point, 1 distant, 2 infinite(constant) */
  float p[3];     /* point: position; distant: direction TOWARDS the light (normalised by caller) */
  float c[3];     /* point: intensity I; distant/infinite: radiance L */
  float pad;
} orc_light;

typedef struct {
  float c[3];
  float r;
  uint32_t mat;
  uint32_t pad[3];
} orc_sphere;

typedef struct {
  const float *P;          /* 3*n_verts */
  const uint32_t *idx;     /* 3*n_tris */
  const uint16_t *mat_id;  /* n_tris */
  const orc_material *mats;
  const orc_light *lights;
  const orc_sphere *spheres;
  uint32_t n_verts, n_tris, n_mats, n_lights, n_spheres;
  float cam_to_world[16];  /* row-major, as transform.rs Matrix4x4 */
  float fov;               /* degrees, on the shorter image axis */
  int32_t xres, yres;
  float crop[4];           /* x0 x1 y0 y1 in [0,1], film.rs crop_window */
  const float *tri_uv;     /* 6 * n_tris corner (u, v), or NULL */
  const orc_texture *textures;
  uint32_t n_textures;
} orc_scene_desc;

typedef struct {
  uint32_t integrator; /* 0 = path, 1 = directlighting */
  uint32_t max_depth;
  uint32_t spp_x, spp_y;
  uint64_t seed;
  uint32_t rank, world_size; /* which 64x64 super-tiles to render (t % world == rank) */
  uint32_t flags;
  uint32_t sampler;    /* 0 = stratified, 1 = padded (0,2)-sequence (3.10), 2 = Sobol' with its own dimensions per request (3.12) */
  float filter_xwidth, filter_ywidth; /* box filter radii (box.rs:57-61); 0 = the default 0.5.  Other radii: DESIGN.md 3.11 */
  float max_sample_luminance;         /* Film "maxsampleluminance" (film.rs:75,279); 0 = infinity */
} orc_render_desc;

typedef struct {
  uint64_t camera_rays, bounce_rays, shadow_rays;
  uint64_t nodes_visited, tris_tested; /* exact counters: define the algorithmic bytes */
  double seconds;
} orc_stats;

typedef struct orc_scene orc_scene;

/* ---- Sobol' generator matrices from the Joe-Kuo direction numbers: 52 columns per dimension, the layout of the
 * reference's SOBOL_MATRICES32 (sobolmatrices.rs:81) ---- */
int orc_sobol_dims(void);
void orc_sobol_matrix(int dim, uint32_t *out52);
void orc_sobol_points(uint32_t key_seed, uint32_t n, float *out2n);
/* Halton sampler (sampler 3): value u[i] and integer head v[i] of dimension d (base b = the d-th prime) for point indices 0 .. n - 1 of a
 * frame whose largest sample index is spp_mask; returns b^D, the modulus of the head (0 for base 2) */
uint32_t orc_halton_points(uint32_t d, uint32_t key, uint32_t spp_mask, uint32_t n, float *u, uint32_t *v);

/* ---- reference-pinned primitives ---- */
void orc_rng_default_u32(uint32_t *out, int n);
void orc_rng_default_float(float *out, int n);
void orc_rng_default_threshold(uint32_t b, uint32_t *out, int n);
void orc_rng_seq_u32(uint64_t seq, uint32_t *out, int n);
void orc_rng_seq_float(uint64_t seq, float *out, int n);
void orc_film_cropped_bounds(int xres, int yres, const float crop[4], int32_t out[4]);
void orc_film_sample_bounds(int xres, int yres, const float crop[4], float rx, float ry, int32_t out[4]);
void orc_film_tile_bounds(int xres, int yres, const float crop[4], float rx, float ry,
                          const int32_t sample_bounds[4], int32_t out[4]);
void orc_film_physical_extent(int xres, int yres, float diagonal_mm, float out[4]);
void orc_rgb_to_xyz(const float rgb[3], float xyz[3]);
void orc_xyz_to_rgb(const float xyz[3], float rgb[3]);
/* film.rs:340-372: xyzw (4 floats per pixel) -> rgb (3 floats per pixel) */
void orc_film_write_rgb(const float *xyzw, int64_t n_px, float scale, float *rgb);
void orc_look_at(const float pos[3], const float look[3], const float up[3], float m[16], float m_inv[16]);
void orc_matrix_inverse(const float m[16], float out[16]);
void orc_matrix_mul(const float a[16], const float b[16], float out[16]);
void orc_matrix_transpose(const float m[16], float out[16]);               /* transform.rs:129-139 */
int orc_quadratic(float a, float b, float c, float *t0, float *t1);
/* lib.rs:115-141 / transform.rs:59-71 (their doc-tests are among the reference's known-answer vectors) */
float orc_clamp_f(float v, float lo, float hi);
long orc_clamp_i(long v, long lo, long hi);
float orc_lerp(float t, float v1, float v2);
int orc_solve_2x2(const float a[4], const float b[2], float x[2]);
float orc_gamma_correct(float v);
uint8_t orc_to_byte(float v);

/* ---- scene / BVH / rays ---- */
orc_scene *orc_scene_create(const orc_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
uint32_t orc_bvh_node_count(const orc_scene *s);
uint32_t orc_bvh_depth(const orc_scene *s);
/* nodes: 8 x uint32 per node {min xyz, max xyz, offset, nprims | axis<<16}; order: n_tris uint32 */
void orc_bvh_export(const orc_scene *s, uint32_t *nodes, uint32_t *order);
uint32_t orc_light_count(const orc_scene *s);
/* closest hit.  prim = 0xffffffff on miss.  counters (2 x uint64: nodes, tris) may be NULL. */
void orc_intersect(const orc_scene *s, int64_t n, const float *o, const float *d, const float *tmax,
                   float *t, uint32_t *prim, float *b1, float *b2, uint64_t *counters, int brute_force);
void orc_occluded(const orc_scene *s, int64_t n, const float *o, const float *d, const float *tmax,
                  uint8_t *hit, int brute_force);
/* camera ray for film point (fx, fy) */
void orc_camera_ray(const orc_scene *s, float fx, float fy, float o[3], float d[3]);
/* radiance of ONE pixel's samples (debug / fixtures): out = spp*3 floats */
void orc_pixel_samples(const orc_scene *s, const orc_render_desc *r, int x, int y, float *out);
/* film_xyzw: row-major over the cropped pixel bounds, 4 floats/pixel {X,Y,Z,weight};
 * pixels of super-tiles owned by other ranks are left untouched. */
int orc_render(const orc_scene *s, const orc_render_desc *r, float *film_xyzw, orc_stats *st, int n_threads);
/* Box filter radii other than 0.5: the film is accumulated in 64-bit fixed point (2^-24 units), four
 * int64 per pixel of the cropped window {r, g, b, samples}; integer sums do not depend on the order of addition, so the
 * accumulators of several ranks ADD to the single-rank ones exactly.  orc_render_acc ADDS this rank's samples into `acc`
 * (zero it first); orc_film_from_acc converts accumulators into film pixels {X, Y, Z, weight}. */
int orc_render_acc(const orc_scene *s, const orc_render_desc *r, int64_t *acc, orc_stats *st, int n_threads);
void orc_film_from_acc(const int64_t *acc, int64_t n_px, float *film_xyzw);

/* ---- the production walk of the render kernel (pbrt_amd/csrc/kernels.hip, quantised 4-wide nodes of DESIGN.md section
 * 4) restated one ray at a time (quad_walk.cpp): checks the trees the product's builders emit without a GPU, and counts
 * node steps / triangle tests per ray.  quads: 16 words per node; root_ref: 0 = quad 0, a leaf ref for a tree that is one
 * leaf, 0xffffffff = no tree; order: leaf slot -> triangle id.  Every output may be NULL. ---- */
void orc_quad_walk(const uint32_t *quads, uint32_t n_quads, uint32_t root_ref, const float root_box[6], const float *P,
                   const uint32_t *idx, const uint32_t *order, int64_t n, const float *o, const float *d, const float *tmax,
                   int any_hit, float *t, uint32_t *prim, float *b1, float *b2, uint8_t *occluded, uint32_t *steps,
                   uint32_t *tris, uint32_t *max_stack, int n_threads, const float *exact_boxes /* diagnostics: NULL */);
/* measurement aid: the walks that follow add their node steps to per_node[node] (n_quads words; NULL stops counting) */
void orc_quad_walk_count_visits(uint64_t *per_node);
/* tests only: 0 switches the own-box rule of Triangle::Intersect off in the oracle's BVH walk and brute force -- the
 * spec as it was until round 5, where the two can disagree --, anything else switches it back on (the default) */
void orc_debug_own_box_rule(int on);
/* tests: Triangle::Intersect (Moeller-Trumbore + the own-box rule) of ray i against triangle tri[i] alone -> ok[i], t[i] */
void orc_tri_accepts(const orc_scene *s, int64_t n, const float *o, const float *d, const float *tmax, const uint32_t *tri, uint8_t *ok, float *t);
/* tests: what the own-box rule promises, node by node on a product tree -- fails[i] = the node tests on the way from the root to triangle
 * tri[i]'s leaf slot that do NOT pass for ray i with tfar = th[i] (oracle/quad_walk.cpp) */
void orc_quad_path_check(const uint32_t *quads, uint32_t n_quads, const float root_box[6], const uint32_t *order, uint32_t n_tris, int64_t n,
                         const float *o, const float *d, const uint32_t *tri, const float *th, uint32_t *fails);

#ifdef __cplusplus
}
#endif
#endif
