/*
 * rt_oracle.h -- CPU restatement of the reference ray tracer's ray-scene query
 * path (octree Build/Search, ray/triangle and ray/box tests, and the renderer
 * loops that call them).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library or package may
 * include, link or call this code; it exists so that tests, the smoke check and
 * the benchmark's CPU-baseline leg have an independent statement of what the
 * reference computes.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout).
 *
 * Parity pin: in "legacy rand" mode this restatement reproduces, byte for byte,
 * the PPM images and all Search / ray-triangle counters of the unmodified
 * reference binary on the known-answer configurations in
 * tests/golden/ref_kats.json (checked by tests/test_oracle_pin.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;
typedef struct orc_octree orc_octree;

/* Material record: 18 floats, order of the reference's Material (src/material.h:7-16):
 * ambient[3] diffuse[3] specular[3] transmittance[3] emission[3] shininess ior dissolve */
#define ORC_MAT_FLOATS 18

enum { ORC_RENDER_DISTANCE = 0, ORC_RENDER_DIFFUSION = 1, ORC_RENDER_PHONG = 2, ORC_RENDER_BLINN_PHONG = 3 };
enum { ORC_SHADING_FLAT = 0, ORC_SHADING_SMOOTH = 1 };
enum { ORC_ADS_NONE = 0, ORC_ADS_OCTREE = 1 };
enum { ORC_RNG_LEGACY_RAND = 0, ORC_RNG_COUNTER = 1 };
enum { ORC_ASPECT_COMPAT = 0, ORC_ASPECT_TRUE = 1 };

typedef struct {
    /* camera (src/camera.h:15-21) */
    float cam_pos[3], cam_up[3], cam_dir[3];
    float fov;
    int32_t width, height, samples_per_pixel;
    /* renderer (src/Renderer/renderer.cc:337-379) */
    int32_t max_depth;
    int32_t samples_per_triangle;
    int32_t render_type;
    int32_t shading_type;
    int32_t cull_backfaces;
    float max_distance;
    float background[3];
    /* extras of this build */
    int32_t ads;         /* ORC_ADS_* */
    int32_t rng_mode;    /* ORC_RNG_* */
    int32_t aspect_mode; /* ORC_ASPECT_* */
    uint32_t seed;
    /* tile subset: render only rows [row_begin,row_end) ; 0,0 = all */
    int32_t row_begin, row_end;
} orc_render_cfg;

typedef struct {
    uint64_t search_calls;    /* octree.cc:174 */
    uint64_t search_nodes;    /* octree.cc:185 */
    uint64_t search_leaves;   /* octree.cc:187 */
    uint64_t search_returned; /* octree.cc:204 */
    uint64_t tri_tests;       /* renderer.cc:137,305,311 */
    uint64_t slab_tests;      /* collision_ray_aabb calls (octree.cc:195) */
    uint64_t primary_rays, secondary_rays, shadow_rays;
    uint64_t slab_tests_any, tri_tests_any; /* the part of slab_tests / tri_tests spent in is_shadowed */
} orc_stats;

typedef struct {
    uint64_t max_depth, nodes, leaves, max_tris_in_leaf;
    float avg_leaf_depth, avg_tris_in_leaf;
    uint64_t leaf_refs; /* sum of triangle references over all leaves */
} orc_tree_stats;

/* ---- scene ---- */
orc_scene* orc_scene_load_obj(const char* obj_path, const char* mtl_basepath, int triangulate);
orc_scene* orc_scene_from_arrays(uint64_t n_tri, const float* pos9, const float* norm9, const int32_t* matid,
                                 uint32_t n_mat, const float* materials18);
void orc_scene_free(orc_scene*);
uint64_t orc_scene_num_triangles(const orc_scene*);
uint32_t orc_scene_num_materials(const orc_scene*);
uint32_t orc_scene_num_lights(const orc_scene*);
int orc_scene_has_normals(const orc_scene*);
void orc_scene_export(const orc_scene*, float* pos9, float* norm9, int32_t* matid, float* materials18,
                      float* face_normal3, int32_t* light_tri);
const char* orc_last_error(void);

/* ---- octree ---- */
orc_octree* orc_octree_build(const orc_scene*, uint32_t max_triangles_per_bb, uint32_t max_depth);
void orc_octree_free(orc_octree*);
void orc_octree_stats(const orc_octree*, orc_tree_stats*);
uint64_t orc_octree_num_nodes(const orc_octree*);
/* Nodes in creation (= breadth-first) order.  Per node: is_leaf, depth, box min/max, the 8 child
 * node indices (-1 = null) and the [begin,count) range into the leaf/list triangle index pool. */
void orc_octree_export(const orc_octree*, uint8_t* is_leaf, uint32_t* depth, float* bmin3, float* bmax3,
                       int32_t* child8, uint64_t* tri_begin, uint32_t* tri_count);
uint64_t orc_octree_pool_size(const orc_octree*);
void orc_octree_export_pool(const orc_octree*, uint32_t* pool);

/* ---- queries ---- */
/* Ads::Search restated: candidate triangle indices in reference order.  Returns the full count
 * (which may exceed cap; only cap entries are written). */
uint64_t orc_search(const orc_octree*, const float* orig3, const float* dir3, uint32_t* out_tri, uint64_t cap,
                    orc_stats* stats);
/* closest-hit loop of ray_color (renderer.cc:116-142) over n rays; out_tri = global triangle index
 * or -1, out_t = t or +inf. */
void orc_trace_closest(const orc_scene*, const orc_octree*, uint64_t n, const float* orig, const float* dir,
                       int cull, float* out_t, int32_t* out_tri, orc_stats* stats);
/* any-hit loop of is_shadowed (renderer.cc:288-313): out_mask[i] = 1 if shadowed. */
void orc_trace_any(const orc_scene*, const orc_octree*, uint64_t n, const float* point, const float* light,
                   int cull, uint8_t* out_mask, orc_stats* stats);

/* ---- primitives (for unit KATs) ---- */
float orc_ray_triangle(const float* orig3, const float* dir3, const float* tri9, int cull);
int orc_ray_aabb(const float* orig3, const float* dir3, const float* bmin3, const float* bmax3);
int orc_tri_aabb(const float* tri9, const float* normal3, const float* bmin3, const float* bmax3);
/* primary ray directions, row-major, 3 floats per pixel (renderer.cc:82-112) */
void orc_generate_rays(const orc_render_cfg*, float* out_dir);
/* camera frame constants handed to the device ray generator: p00[3], qw[3], qh[3] */
void orc_camera_frame(const orc_render_cfg*, float* p00, float* qw, float* qh);
/* the shared counter-based generator: 31-bit integer */
uint32_t orc_rng_u31(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id, uint32_t light,
                     uint32_t sample, uint32_t dim);

/* ---- renderer ---- */
/* Renders into out_rgb (width*height*3 floats, clamped colours as stored in the reference's img
 * vector, renderer.cc:49-51).  Optional out_hit_tri / out_hit_t receive the primary (sample 0)
 * closest hit per pixel.  Only rows [row_begin,row_end) are touched when a subset is set. */
int orc_render(const orc_scene*, const orc_octree*, const orc_render_cfg*, float* out_rgb, int32_t* out_hit_tri,
               float* out_hit_t, orc_stats* stats);
/* P3 text exactly as save_image_to_pmm + write_color (renderer.cc:381-390, color.h:12-15).
 * Returns bytes needed (excluding NUL); writes at most cap bytes. */
uint64_t orc_ppm_text(int32_t width, int32_t height, const float* rgb, char* out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
