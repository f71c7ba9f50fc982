/*
 * pbrt_hip_debug.h -- test and measurement hooks of libpbrt_hip.so.  NOT part of the boundary a host binds (that is
 * pbrt_hip.h: SURVEY.md 8(b)'s calls, the multi-GPU entry points, the loader, image I/O, the Film helpers): these entry
 * points exist so that tests/ can check the trees the builders make, tools/ can cost them without a GPU, and the parser's
 * conformance tests can look at its state.  They may change between versions.  Each cites what it stands in for, as there.
 */
#ifndef PBRT_HIP_DEBUG_H
#define PBRT_HIP_DEBUG_H
#include "pbrt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- the accelerator as built (the reference names one, core/api.rs:237 "bvh", and builds none) ---- */
/* the canonical tree behind the counter flags: *ready = it exists (always for a host-built scene; for a device-built one
 * after the first call that counted), *build_ms = the host builder's time for it */
int pbrt_hip_scene_canonical_info(const pbrt_hip_scene *scene, uint32_t *ready, double *build_ms);
/* the objective of the device build's tree optimisation (pbrt_hip_scene_optimize_info: passes, moves, time): the summed half surface
 * area of the interior nodes of the binary tree before the first pass and after the last one kept; *undone = 1 when a pass raised
 * it, was undone and ended the passes (pbrt_amd/csrc/reinsert_core.hpp).  Zeros for a host-built scene.  Any pointer may be NULL. */
int pbrt_hip_scene_optimize_cost(const pbrt_hip_scene *scene, double *before, double *after, uint32_t *undone);
/* the production walk's tree as it sits in HBM: quads = 16 words per node (cap_nodes of them), order = leaf slot ->
 * triangle id (n_tris words); either may be NULL */
int pbrt_hip_scene_export_quads(const pbrt_hip_scene *scene, uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads,
                                uint32_t *order);
/* the canonical binary tree, host copy: 8 words per node; order maps leaf slot -> triangle id */
int pbrt_hip_scene_export_bvh(const pbrt_hip_scene *scene, uint32_t *nodes /* 8 words each */, uint32_t *order);
/* the production walk's own structure: number of 64-byte quantised 4-wide nodes, and the most stack entries a
 * walk can hold (see pbrt_hip_render_stack_plan for where they live) */
int pbrt_hip_scene_walk_info(const pbrt_hip_scene *scene, uint32_t *quad_nodes, uint32_t *stack_need);
/* How the render kernel is launched for a tree with that stack bound (pure function, no device touched): LDS rows of
 * 64 x 4 bytes per wave, one-wave workgroups a CU holds at once with them (the grid is this x the CU count), and the
 * entries per lane kept in an HBM overflow area (non-zero = the overflow variant of the kernel). */
int pbrt_hip_render_stack_plan(uint32_t stack_need, uint32_t *lds_rows, uint32_t *waves_per_cu, uint32_t *overflow_entries);
/* the librccl the in-library multi-GPU path uses (pbrt_hip_multi_*: loaded on first use by dlopen): its file name -- it must be the one
 * that belongs to the HIP runtime of the process (beside the loaded libamdhip64), whichever host loaded that.  No device is touched. */
int pbrt_hip_rccl_library(char *path, size_t cap);
/* host-only variant for CPU-side tests of the builder: no device is touched */
int pbrt_hip_bvh_build_host(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
                            uint32_t *nodes /* 8*(2*n_tris) words cap */, uint32_t *order, uint32_t *n_nodes,
                            uint32_t *depth);

/* host-only: the production walk's quantised 4-wide tree from a triangle soup, for CPU-side
 * tests of its invariants.  quads: 16 words per node (cap_nodes nodes of room); split_leaves as the library default. */
int pbrt_hip_quad_build_host(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris, int split_leaves,
                             uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads, uint32_t *stack_need);
/* The same with the binary tree the 4-wide nodes are collapsed from chosen explicitly -- PBRT_HIP_TREE_SAH: the canonical
 * binned-SAH tree of DESIGN.md 3.3; PBRT_HIP_TREE_REINSERT: that tree with its leaves opened into single triangles and optimised
 * by the device builder's parallel re-insertion pass RUN ON THE HOST (the same functions, pbrt_amd/csrc/reinsert_core.hpp: what
 * lets the CPU tests walk the trees that pass makes; PBRT_HIP_SCENE_OPTIMIZED_TREE's tree) -- and more outputs, each of which may
 * be NULL: order = leaf slot -> triangle id (n_tris words; what a leaf child's slot refers to), root_box = lo xyz, hi xyz, n_refs
 * = references in the tree (n_tris), exact_boxes = the children's boxes before quantisation (24 floats per node of `quads`: lo
 * xyz, hi xyz of child 0..3; diagnostics).  (Round 3's PBRT_HIP_TREE_SBVH = 1, a spatial-split builder measured negative on
 * BASELINE's meshes, is no longer in the library: tools/experiments/r03_host_tree_builders/.) */
#define PBRT_HIP_TREE_SAH 0u
#define PBRT_HIP_TREE_REINSERT 2u
#define PBRT_HIP_TREE_DEFAULT 0xffffffffu
int pbrt_hip_quad_build_host_ex(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris, int split_leaves,
                                uint32_t tree, uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads, uint32_t *stack_need,
                                uint32_t *order, float *root_box, uint32_t *n_refs, float *exact_boxes);

/* ---- samplers: the reference holds the Sobol' generator matrices (sobolmatrices.rs:81) and no sampler ---- */
/* host only: the generator matrices sampler 2 uses -- 128 dimensions x 32 columns, rows 0 .. 127 of the reference's
 * SOBOL_MATRICES32 (sobolmatrices.rs:81), the first 32 of its 52 columns -- for tests of that claim */
void pbrt_hip_sobol_matrices(uint32_t *out_4096_words);
/* ---- the parser's state and its tokenizer alone (conformance tests replay parser.rs:778-880, api.rs:979-1045) ---- */
/* CTM (current_transform[0].m) when parsing stopped, and the directive names stored by the option setters
 * (api.rs:778-820) as "camera sampler integrator filter accelerator film" */
int pbrt_hip_loaded_state(const pbrt_hip_loaded *loaded, float ctm[16], char *names, size_t names_cap);
/* the tokenizer alone (parser.rs:61-170): tokens '\n'-separated into buf; returns the token count, or
 * -(1 + count) when the stream ends in an error (EOF / newline inside a quoted string) after `count` tokens */
int pbrt_hip_tokenize(const char *text, size_t len, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
