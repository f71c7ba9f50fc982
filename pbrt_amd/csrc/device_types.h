// device_types.h -- kernel argument blocks shared by the host launcher and the HIP kernels.
// HBM layout:
//   nodes   : 4 x 16 B per INTERIOR node of the binary BVH, holding the boxes of its two children
//             ("children in parent": one fetch tests both children, leaves need no fetch at all)
//             {c0.lo.xyz c0.hi.x} {c0.hi.yz c1.lo.xy} {c1.lo.z c1.hi.xyz} {ref0 ref1 axis 0}
//             ref = interior index, or 0x80000000 | n_prims << 24 | first leaf slot
//   tris    : 3 x 16 B per LEAF SLOT  {p0.xyz, triangle id} {p1.xyz, material id} {p2.xyz, 0}
//             (leaf order, so the <=4 triangles of a leaf are one contiguous 48..192 B run)
//   mats    : 2 x 16 B per material   {type, k.xyz} {le.xyz, 0}
//   lights  : 5 x 16 B per light      {type, p0.xyz} {p1.xyz, area} {p2.xyz, 0} {c.xyz, 0} {n.xyz, 0}
//   spheres : 2 x 16 B per sphere     {c.xyz, r} {material id, 0, 0, 0}   (shading; in the tree a sphere is a leaf record of `tris`:
//             {c.xyz, primitive id} {r, 0, 0, material id} {0, 0, 0, 1} -- word 3 of the third float4 flags it)
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace pbrt_hip {

// Tuning / A-B knobs (PBRT_HIP_MIN_WALKERS, PBRT_HIP_COLLAPSE, PBRT_HIP_TREE, ...) are read from the environment
// only when PBRT_HIP_DEBUG_KNOBS is set (to anything but "0"): a production process is not steered by stray variables.
inline const char *debug_knob(const char *name) {
  const char *on = std::getenv("PBRT_HIP_DEBUG_KNOBS");
  if (!on || !*on || (on[0] == '0' && !on[1])) return nullptr;
  return std::getenv(name);
}

// LDS stack of the production (quad) walk: one row = 64 lanes x 4 bytes.  A row count r serves trees whose worst-case stack
// bound is r - 2 (the sentinel and one scratch row).  The render kernel takes its rows as DYNAMIC shared memory, sized per
// scene (render_stack_plan below).  A CU has 160 KB of LDS, handed out in granules of 1280 bytes (measured: 31 and 32 rows run
// like 35, 30 rows 3 % faster): up to 30 rows (6 granules) a CU holds 20 one-wave workgroups -- 5 per SIMD, which is also
// what the kernel's 96 VGPRs allow --, 31..35 rows (7 granules) 18, 36..40 rows 16.  Trees that need more than 30 rows run
// the overflow variant: kQuadLdsStackOvf rows in LDS at 20 waves, deeper entries in HBM (kQuadLdsStack = the rows of the
// ray-batch kernel, and the most the render kernel takes with PBRT_HIP_PREFER_LDS_STACK).  -DPBRT_QUAD_LDS_STACK=12 forces the overflow variant on nearly every scene (tests of that path).
#ifdef PBRT_QUAD_LDS_STACK
constexpr uint32_t kQuadLdsStack = PBRT_QUAD_LDS_STACK, kQuadLdsStackOvf = PBRT_QUAD_LDS_STACK, kQuadLdsStackOvfDeep = PBRT_QUAD_LDS_STACK;
#else
constexpr uint32_t kQuadLdsStack = 40, kQuadLdsStackOvf = 30, kQuadLdsStackOvfDeep = 30;
#endif
// Round 2 gave very deep trees (stack bound >= 42: the 12 M-triangle `big` workload has 48) a 35-row variant at 18 waves per
// CU, on the assumption that they reach the end of a 30-row LDS part often.  They do not: the bound is a worst case that
// walks do not come near (tools/walk_sim.py with ORC_WALK_STACK_IN_TRIS: a ray's deepest stack in the 12 M-triangle tree is
// 8.5 entries on average, 19 at the 99.9th percentile, 23 at most over 40 000 rays; 1 M triangles: 7.6 / 17 / 20), so the
// HBM overflow area is a safety net, not a path; and more resident waves help this latency-bound workload (r03q / r03r:
// 20 waves with 30 rows 217 ms, 18 waves with 35 rows 219 ms, 16 waves 224 ms).  The deep variant is therefore the same
// 30 rows (kept as a name so that a different choice stays a one-line change).
constexpr uint32_t kOvfDeepNeed = 42;
constexpr uint32_t kLdsGranule = 1280u;
constexpr uint32_t kLdsBytesPerCu = 160u * 1024u;
// float4 per triangle record in `tris`: {p0, id}{p1, material}{p2, 0} + one of padding -- a 64-byte record never straddles
// a 128-byte line (a 48-byte one does so two times in eight): C3 +0.9 %, C2 +0.3 % for 16 bytes per triangle (3 = packed, A-B)
#ifndef PBRT_TRI_STRIDE
#define PBRT_TRI_STRIDE 4
#endif
constexpr uint32_t kTriStride = PBRT_TRI_STRIDE;
#ifndef PBRT_RENDER_MAX_WAVES_PER_CU  // 4 SIMDs x the waves per SIMD the render kernel's register budget allows (kernels.hip)
#define PBRT_RENDER_MAX_WAVES_PER_CU 20
#endif
constexpr uint32_t kRenderMaxWavesPerCu = PBRT_RENDER_MAX_WAVES_PER_CU;

// An unused child slot of a quad node holds the ref of a leaf with no triangles (and an inverted box: lower planes 255,
// upper planes 0).  The box rejects it except for degenerate rays / flat nodes, and then the walk parks at an empty leaf and
// pops: harmless, so the node step needs no "is this slot used" test (it had two compares per step for it until r02c).
constexpr uint32_t kEmptyLeafRef = 0x80000000u;

// A pixel's spp samples are cut into K = sample_chunks(spp) chunks, each a work item with its own RNG stream and partial
// film sum (oracle/oracle.cpp has the same rule): the largest power of two <= 16 that leaves a chunk at least
// 32 samples (1 below 64 spp).  Item number: pixel-in-block (6 bits) | chunk << 6 | 8x8 block << (6 + log2 K).
constexpr uint32_t kMaxSampleChunks = 16u, kMinSamplesPerChunk = 32u;
inline uint32_t sample_chunk_shift(uint32_t spp) {  // log2 K
  uint32_t kb = 0;
  while ((2u << kb) <= kMaxSampleChunks && (2u << kb) * kMinSamplesPerChunk <= spp) kb++;
  return kb;
}

// Sampler 2: the first kSobolNdRequests requests of a sample take their own pair of Sobol' dimensions (host_math.hpp
// builds the 2 x kSobolNdRequests generator matrices; capi.cpp asserts the two constants agree)
constexpr uint32_t kSobolNdRequests = 64u;

struct RenderStackPlan {
  uint32_t rows;           // LDS rows per wave
  bool overflow;           // the overflow variant (rows == kQuadLdsStackOvf or kQuadLdsStackOvfDeep)
  uint32_t waves_per_cu;   // one-wave workgroups a CU holds at once with these rows (at most 20: 5 per SIMD by registers)
  uint32_t extra_entries;  // HBM entries per lane beyond the LDS part (overflow variant)
};
inline RenderStackPlan render_stack_plan(uint32_t quad_stack_need, bool force_overflow, bool prefer_lds = false) {
  RenderStackPlan p;
  const uint32_t need_rows = quad_stack_need + 2u;
  auto waves = [](uint32_t rows) {
    const uint32_t bytes = (rows * 256u + kLdsGranule - 1u) / kLdsGranule * kLdsGranule, w = kLdsBytesPerCu / bytes;
    return w > kRenderMaxWavesPerCu ? kRenderMaxWavesPerCu : w;
  };
  p.rows = need_rows < 8u ? 8u : need_rows;
  // (20 waves with the overflow variant beat 18 or 16 waves with the whole stack in LDS: C2 +1 %, C3 +4 %)
  p.overflow = force_overflow || need_rows > kQuadLdsStack || (waves(p.rows) < waves(kQuadLdsStackOvf) && !prefer_lds);
  if (p.overflow) p.rows = quad_stack_need >= kOvfDeepNeed ? kQuadLdsStackOvfDeep : kQuadLdsStackOvf;
  p.waves_per_cu = waves(p.rows);
  p.extra_entries = p.overflow && need_rows > p.rows ? need_rows - p.rows : 0u;
  return p;
}
inline bool render_prefer_lds() {  // A-B runs: the whole stack in LDS whenever it fits kQuadLdsStack rows, whatever the occupancy
  static const bool f = debug_knob("PBRT_HIP_PREFER_LDS_STACK") != nullptr;
  return f;
}
inline bool render_force_overflow() {  // A-B runs
  static const bool f = debug_knob("PBRT_HIP_FORCE_OVERFLOW_VARIANT") != nullptr;
  return f;
}
// the traversal kernel over ray batches keeps static rows
constexpr uint32_t kQuadLdsEntries = kQuadLdsStack - 1u;
// The ray-batch kernel (pbrt_hip_intersect / pbrt_hip_occluded: the walk alone, ~64 VGPRs) runs 8 waves per SIMD with 20 rows of a
// lane's stack in LDS (20 KB per 4-wave workgroup, 8 per CU) and deeper entries in HBM: +9 % on 4 waves with 40 rows, +12 % at equal
// rows (tools/experiments/dual_ray, profiles/r03y_traversal_occupancy_1m.txt)
constexpr uint32_t kIntersectLdsStack = 20;

struct DevScene {
  const uint4 *nodes;
  const uint4 *quads;  // 4 x 16 B per quantised quad node (capi.cpp QuadNodes)
  uint32_t quad_stack_need;
  const float4 *tris;
  const float4 *mats;
  const float4 *lights;
  const float4 *spheres;
  uint32_t n_nodes, n_tris, n_spheres, n_lights;  // n_nodes: nodes of the binary tree (0 = no triangles)
  float n_lights_f;                                 // (float)n_lights
  uint32_t root_ref;                                // ref of the root (a leaf ref for tiny scenes)
  float root_lo[3], root_hi[3];                     // its box
  float le_inf[3];
  uint32_t has_inf;
  float c2w[12];  // rows 0..2 of camera_to_world
  float cam_ax, cam_bx, cam_ay, cam_by;
  int32_t xres, yres;
  int32_t cx0, cy0, cx1, cy1;  // cropped pixel bounds
  // The production walk's stand-in for 1 / 0 (a ray PARALLEL to a slab; kernels.hip trav_run): a power of two so large that every t it
  // makes lies outside any ray interval, yet small enough that (o - origin) x it stays finite for every origin inside the root box
  // (host_math.hpp inv_parallel_for_extent).
  float inv_parallel;
};

struct RenderParams {
  uint32_t integrator, max_depth, spp_x, spp_y;
  uint64_t seed;
  uint32_t rank, world;
  float inv_nx, inv_ny;
  unsigned long long *counters;  // 5: camera, bounce, shadow rays, nodes visited, triangles tested
  uint32_t min_walkers, min_parked;  // traversal scheduling thresholds (kernels.hip trav_run)
  float4 *lane_state;                // 5 x 64 float4 per workgroup: path state parked in HBM (kernels.hip PathState)
  float4 *wide_slots;                // wide box filter: 16 x 2 x 64 float4 per workgroup, a chunk's sums per footprint (kernels.hip)
  uint32_t *stack_overflow;          // [workgroup][entry][lane]: stack entries beyond the LDS part
  uint32_t stack_overflow_entries;
  uint32_t *next_item;    // hand-out counters of the render kernel's item list, one per region, 16 words apart (zeroed before the launch)
  uint32_t n_regions;     // contiguous parts of the list, one per XCD (kernels.hip fetch step)
  uint32_t n_items;       // n_local_super * 4096 pixels * K chunks (kernels.hip: item = block, chunk, pixel in block)
  uint32_t chunk_shift;   // log2 K, K = sample_chunks(spp)
  uint32_t n_workgroups;  // one-wave workgroups launched: what the device holds at once, not one per tile
  float4 *partials;       // [slab position][K]: the partial film sums of the chunks (merge_kernel adds them in order)
  uint32_t sampler;       // PBRT_HIP_SAMPLER_*
  uint32_t spp_mask;      // Sobol sampler: 2^ceil(log2(spp)) - 1
  uint32_t stx_recip, spp_x_recip;  // ceil(2^32 / super-tiles per row), ceil(2^32 / spp_x): the kernel's divisions by these two
  // The pixels that are SAMPLED (Film::get_sample_bounds, film.rs:166-175): the cropped window for the default box filter,
  // `pad` = ceil(radius - 0.5) pixels more on every side for a wider one.  The 64x64 super-tiles cover them.
  int32_t sx0, sy0, sw, sh;  // origin and size of the sample bounds
  int32_t seq_x0, seq_y0;    // sx0 + pad_x, sy0 + pad_y: a sampled pixel's coordinates in the numbering of the RNG streams ...
  uint32_t seq_w, seq_h;     // ... whose rows are xres + 2 pad_x long (yres + 2 pad_y of them): the image plus its halo
  float max_lum;             // Film "maxsampleluminance" (film.rs:75,279; pbrt-v3 FilmTile::AddSample); +inf = none
  // box filter radii other than 0.5 (render_kernel<..., WIDE>): fixed-point film accumulators, int64 {r, g, b, samples} per
  // pixel of the cropped window, added to with atomics
  float filter_rx, filter_ry;
  unsigned long long *acc;
  // sampler 2 (render_kernel<..., SND>): generator matrices of the first ten Sobol' dimensions, 32 columns each
  const uint32_t *sobol_mat;
  // textured materials (render_kernel_x<..., TEX>): corner (u, v) per leaf slot -- 3 x float2 --, and 3 x 16 B per
  // texture {type, tex1.rgb}{tex2.rgb, su}{sv, du, dv, 0}; a material's texture number rides in mats[2 i + 1].w (0 = none).  (Scene
  // data -- but appended HERE, to the kernel's last argument: a field added to DevScene would move every offset of this block and
  // with them the machine code of the instantiations the counter profiles are keyed by.)
  const float2 *tri_uv;
  const float4 *textures;
};
// 2^24 fixed-point units per unit of radiance, a component clamped to [0, 2^15]
constexpr float kFixedOne = 16777216.0f, kFixedMax = 32768.0f;

struct RayBatch {
  const float *o, *d, *tmax;  // 3n, 3n, n
  int64_t n;
  float *t;
  uint32_t *prim;
  float *b1, *b2;
  uint8_t *occluded;
  unsigned long long *counters;  // 2: nodes, tris (may be null)
  uint32_t min_walkers, min_parked;
  uint32_t *stack_overflow;
  uint32_t stack_overflow_entries;
};

// launchers (kernels.hip)
hipError_t launch_render(const DevScene &S, const RenderParams &R, uint32_t n_local_super, uint32_t bvh_depth,
                         int counters /* 0 none, 1 exact walk, 2 production walk */, bool wide_filter, bool sobol_nd, hipStream_t stream,
                         bool mis = false, bool textured = false);
// fixed-point accumulators -> film pixels {X, Y, Z, weight}
hipError_t launch_film_from_acc(const unsigned long long *acc, float4 *film, size_t n_px, hipStream_t stream);
hipError_t launch_acc_add(unsigned long long *dst, const unsigned long long *src, size_t n, hipStream_t stream);  // dst[i] += src[i]
hipError_t launch_intersect(const DevScene &S, const RayBatch &B, bool any_hit, uint32_t bvh_depth, hipStream_t stream);
// (n_prims = n_tris + the spheres, whose records come from the `spheres` table: kernels.hip pack_tris_kernel)
hipError_t launch_pack_tris(const float *P, const uint32_t *idx, const uint16_t *mat_id, const uint32_t *order,
                            uint32_t n_prims, uint32_t n_tris, const float4 *spheres, float4 *tris, hipStream_t stream);
// corner (u, v) of every triangle (6 floats, triangle order) -> leaf-slot order
hipError_t launch_pack_uv(const float *tri_uv, const uint32_t *order, uint32_t n_prims, uint32_t n_tris, float2 *out, hipStream_t stream);
// adds the K partial sums of every pixel of a rank's slab in chunk order and converts to XYZ (Film::merge_film_tile)
hipError_t launch_merge(const float4 *partials, float4 *slab, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                        uint32_t n_local_super, uint32_t spp, hipStream_t stream, uint32_t j0 = 0, uint32_t jstride = 1);
// bvh_gpu.hip: the accelerator built on the device.  d_order (n_tris) and d_quads (>= n_tris nodes of 4 uint4) are outputs.
struct GpuBuildInfo {
  uint32_t n_quads, stack_need, levels;
  float root_lo[3], root_hi[3];
  float build_ms;  // everything, the optimisation included
  uint32_t reinsert_passes, reinsert_moves;  // the tree's optimisation by parallel re-insertion (reinsert_core.hpp)
  float reinsert_ms;
  double reinsert_cost_before, reinsert_cost_after;  // summed half surface area of the interior nodes before the first and after the last kept pass
  uint32_t reinsert_undone;                           // 1: a pass raised that sum, was undone and ended the passes
};
constexpr uint32_t kGpuBuildReinsert = 1u;  // flags of gpu_build_quads: optimise the binary tree by re-insertion before the collapse
hipError_t gpu_build_quads(const float *d_P, const uint32_t *d_idx, uint32_t n_tris, uint32_t *d_order, uint4 *d_quads,
                           uint32_t quad_capacity, uint32_t flags, GpuBuildInfo *info, hipStream_t stream);
hipError_t launch_assemble(const float4 *slab, float4 *film, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                           uint32_t n_local_super, hipStream_t stream);

}  // namespace pbrt_hip
