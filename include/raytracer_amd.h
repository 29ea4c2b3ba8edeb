/*
 * raytracer_amd.h -- C ABI of the MI355X (gfx950) ray-scene query backend.
 *
 * This is the boundary a host renderer binds to: plain pointers and sizes, no C++
 * types, no exceptions.  Every entry point cites the interface of the reference
 * ray tracer that it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *  - All functions return 0 on success, a negative rt_status on failure;
 *    rt_last_error() then returns a thread-local message.
 *  - `mem` arguments say where the caller's ray / result buffers live:
 *    RT_MEM_HOST (the library stages them) or RT_MEM_DEVICE (used in place;
 *    must belong to the handle's device).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    rt_ads_trace_closest / rt_ads_trace_any with device buffers and rt_ads_search_noalloc are
 *    asynchronous on that stream; with host buffers they return after the results are in the
 *    buffers.  rt_render and rt_render_primary_hits are SYNCHRONOUS: they return after the frame
 *    is complete on `stream` (the host follows the recursion levels through pinned memory the
 *    kernels write, without synchronising the stream inside the frame, and synchronises once at
 *    the end); their level buffers grow on demand, so the first frames of a size may allocate.
 *    rt_ads_build, rt_ads_search, rt_scene_set_shading and rt_ads_export_tree run on the default
 *    stream and return when done.  While rt_ads_collect_search_stats is on, every query call is
 *    synchronous.
 *  - A handle is bound to one device and must be used from one host thread at a
 *    time, like the reference's Ads object (its counters are plain mutable fields,
 *    src/ADS/Octree/octree.h:55-61).
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ads rt_ads; /* opaque: one acceleration data structure resident on one GPU */

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / unknown name (renderer.cc:355,364,375 throw runtime_error) */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed */
    RT_ERR_BUILD = -3,       /* octree invariant violated (octree.cc:124-152 prints and exit(1)s) */
    RT_ERR_NOT_BUILT = -4,   /* query before rt_ads_build */
    RT_ERR_CAPACITY = -5,    /* caller's output buffer too small */
    RT_ERR_UNSUPPORTED = -6
};

enum rt_mem { RT_MEM_HOST = 0, RT_MEM_DEVICE = 1 };

/* Renderer::ACCELERATION_DATA_STRUCTURE (src/Renderer/renderer.h:24) extended by one value.  The config
 * string for it, under the reference's own (misspelled) key "acceleratied_data_structure"."name"
 * (renderer.cc:368-376), is "octree_gpu". */
enum rt_ads_kind { RT_ADS_NONE = 0, RT_ADS_OCTREE = 1, RT_ADS_OCTREE_PARAMETRIC = 2, RT_ADS_OCTREE_GPU = 3 };
/* Renderer::RENDER_TYPE / SHADING_TYPE (src/Renderer/renderer.h:22-23) */
enum rt_render_type { RT_RENDER_DISTANCE = 0, RT_RENDER_DIFFUSION = 1, RT_RENDER_PHONG = 2, RT_RENDER_BLINN_PHONG = 3 };
enum rt_shading_type { RT_SHADING_FLAT = 0, RT_SHADING_SMOOTH = 1 };
/* Primary-ray frame: RT_ASPECT_COMPAT keeps the reference's integer `height / width`
 * (renderer.cc:93); RT_ASPECT_TRUE uses the real ratio. */
enum rt_aspect_mode { RT_ASPECT_COMPAT = 0, RT_ASPECT_TRUE = 1 };

/* Where Octree::Build runs.  Both builders produce the same tree, bit for bit; RT_BUILDER_AUTO builds scenes of
 * fewer than 16384 triangles on the host and larger ones on the device. */
enum rt_builder { RT_BUILDER_AUTO = 0, RT_BUILDER_HOST = 1, RT_BUILDER_DEVICE = 2 };

/* Maps a config name to rt_ads_kind; returns RT_ERR_INVALID for unknown names
 * (renderer.cc:374-376: "Invalid data structure"). */
int rt_ads_kind_from_name(const char* name, int* out_kind);

/* Octree::config_setup (src/ADS/Octree/octree.cc:268-275): the two tuning keys of the
 * "acceleratied_data_structure" object, plus the device to live on. */
typedef struct rt_ads_config {
    uint32_t struct_size;           /* = sizeof(rt_ads_config) */
    int32_t device;                 /* HIP device ordinal */
    uint32_t max_triangles_per_bb;  /* "max_triangles_per_BB" */
    uint32_t max_depth;             /* "max_depth" */
    uint32_t builder;               /* rt_builder; optional key "builder": "auto" | "host" | "device" */
} rt_ads_config;

/* Ads::Ads + Octree::Octree (src/ADS/ads.h:18, src/ADS/Octree/octree.h:17). */
int rt_ads_create(const rt_ads_config* cfg, rt_ads** out);
/* Ads::~Ads (src/ADS/ads.h:19): frees host and device memory. */
void rt_ads_destroy(rt_ads* ads);

/* Ads::Build / Octree::Build (src/ADS/ads.h:21, src/ADS/Octree/octree.cc:19-171).
 * positions: n_triangles x 3 vertices x 3 floats, host memory, triangle order = the order of the
 * reference's Scene::GetTriangles().  The data is copied; the caller may free it afterwards.
 * Builds the octree with the reference's exact splitting rule -- on the host (then flattened in breadth-first
 * order and uploaded) or directly on the device, one pass per depth (rt_builder) -- into the same resident
 * layout.  n_triangles == 0 is accepted (octree.cc:24-27); queries then report no candidates. */
int rt_ads_build(rt_ads* ads, const float* positions, uint64_t n_triangles);

/* Copies the resident tree back to the host: internal-node records (128 bytes each) and leaf slots (48 bytes
 * each) in the layout of rt_host_tree_export_flat, for inspection and for comparing the two builders.  Any
 * pointer may be NULL; the counts are always reported.  No reference counterpart beyond Octree::PrintStats. */
int rt_ads_export_tree(const rt_ads* ads, void* node_records, void* leaf_slots, uint32_t* root_info2, uint64_t* n_records,
                       uint64_t* n_slots, uint32_t* max_stack, int* built_on_device);

/* Which optional forms serve the wave-together kernels on this tree (none of them changes the candidate sets of
 * octree.cc:186-191 or a result bit): small_tree_triangles > 0 = the small-tree form is resident (<= 32 leaves,
 * <= 1024 triangles), clusters > 0 = the clustered form for any-hit rays is resident, fast_any_hit = 1 if every
 * triangle has |e1| * |e2| < 2^30, so that the any-hit triangle test of shadow rays (unit directions) never needs the
 * full division and runs its hand-scheduled form.  Any pointer may be NULL. */
int rt_ads_tree_forms(const rt_ads* ads, uint64_t* small_tree_triangles, uint32_t* clusters, int* fast_any_hit);

/* Octree::PrintStats / calculate_stats, structural part (src/ADS/Octree/octree.cc:209-266). */
typedef struct rt_tree_stats {
    uint64_t max_depth, nodes, leaves, max_triangles_in_leaf;
    float avg_leaf_depth, avg_triangles_in_leaf;
    uint64_t leaf_refs;        /* triangle references summed over all leaves */
    uint64_t internal_nodes;
    double build_seconds;      /* octree.cc:168-170 */
    uint64_t device_bytes;     /* size of the flattened structure in HBM */
} rt_tree_stats;
int rt_ads_tree_stats(const rt_ads* ads, rt_tree_stats* out);

/* Query counters, as printed by Octree::PrintStats (octree.cc:218-223) and by Renderer::RenderScene
 * ("Ray-triangle collision count", renderer.cc:60); they accumulate over every query call on the handle
 * until reset.  search_calls, closest_rays and any_rays are always counted.  The other counters follow the
 * reference's definitions exactly -- nodes and leaves taken from Search's queue (octree.cc:185-187), candidates
 * returned (octree.cc:204), ray/triangle tests of the loops over the candidates in list order up to the any-hit
 * loop's early return (renderer.cc:137,305,311), slab tests (collision_ray_aabb calls, octree.cc:195) -- which the
 * fast kernels do not visit in that order: rt_ads_search always counts them (it walks breadth-first like the
 * reference); the trace and render calls count them only while rt_ads_collect_search_stats is on, through an
 * additional instrumented pass over the same rays.  counted_rays = the queries those counters cover. */
typedef struct rt_query_stats {
    uint64_t search_calls;     /* "Search method call count": one per ray */
    uint64_t closest_rays, any_rays;
    uint64_t search_node_count;      /* "Search node count" */
    uint64_t search_return_count;    /* "Search return count" */
    uint64_t search_leaves_visited;  /* "Search leaves visited" */
    uint64_t ray_triangle_tests;     /* "Ray-triangle collision count" */
    uint64_t slab_tests;
    uint64_t counted_rays;
    double search_seconds;           /* "Search time": device time of the traversal kernels (rt_render) or wall time of the call */
    /* the part of slab_tests / ray_triangle_tests / counted_rays that belongs to any-hit (shadow) rays */
    uint64_t slab_tests_any, ray_triangle_tests_any, counted_rays_any;
} rt_query_stats;
int rt_ads_query_stats(const rt_ads* ads, rt_query_stats* out);
int rt_ads_reset_query_stats(rt_ads* ads);
/* on != 0: rt_ads_trace_closest, rt_ads_trace_any, rt_render and rt_render_primary_hits also run the instrumented
 * pass that fills the reference-defined counters of rt_query_stats (slower; results are unchanged). */
int rt_ads_collect_search_stats(rt_ads* ads, int on);

/* Ads::Search (src/ADS/ads.h:22, src/ADS/Octree/octree.cc:173-207) for a batch of rays.
 * For ray i the candidate triangle indices, in exactly the reference's order (leaves in breadth-first
 * visiting order, duplicates kept), are written to out_tri[out_offsets[i] .. out_offsets[i+1]).
 * rays: n x 6 floats (origin xyz, direction xyz), host memory.  out_offsets: n+1 entries.
 * If the total exceeds cap, RT_ERR_CAPACITY is returned and *out_total holds the size needed. */
int rt_ads_search(rt_ads* ads, const float* rays, uint64_t n, uint64_t* out_offsets, uint32_t* out_tri, uint64_t cap,
                  uint64_t* out_total);

/* rt_ads_search for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  rays (n x 6 floats), out_offsets
 * (n + 1 entries), out_tri (cap entries) and out_total (1 entry) are DEVICE memory of the handle's device; the call
 * is asynchronous on `stream` and allocates nothing: the traversal queues and per-ray counts live in a fixed
 * scratch array inside the library (RT_SEARCH_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends).
 * Limits: n <= RT_SEARCH_NOALLOC_MAX_RAYS and n * (octree nodes) * 8 bytes <= RT_SEARCH_NOALLOC_SCRATCH_BYTES,
 * otherwise RT_ERR_CAPACITY and nothing is launched.  *out_total receives the total list length; if it exceeds cap,
 * the entries at positions >= cap are not written (the caller checks *out_total after synchronising).  All calls on
 * one GPU share the scratch array: calls that overlap on a GPU (different streams or handles) race.  The call is not
 * counted in rt_query_stats (it brings nothing back to the host; the counters of Octree::PrintStats are rt_ads_search's). */
#define RT_SEARCH_NOALLOC_MAX_RAYS 4096u
#define RT_SEARCH_NOALLOC_SCRATCH_BYTES (16u << 20)
int rt_ads_search_noalloc(rt_ads* ads, const float* rays, uint64_t n, uint64_t* out_offsets, uint32_t* out_tri, uint64_t cap,
                          uint64_t* out_total, void* stream);

/* Ads::Search + the closest-hit loop of Renderer::ray_color (src/Renderer/renderer.cc:116-142) for n rays.
 * origins, dirs: n x 3 floats.  out_t[i] = hit distance or +inf; out_tri[i] = index of the winning
 * triangle in Build order, or -1.  Ties in t go to the triangle that comes first in the reference's
 * candidate list. */
int rt_ads_trace_closest(rt_ads* ads, const float* origins, const float* dirs, uint64_t n, int cull_backfaces, int mem,
                         float* out_t, int32_t* out_tri, void* stream);

/* Ads::Search + the any-hit loop of Renderer::is_shadowed (src/Renderer/renderer.cc:288-313) for n
 * (surface point, light point) pairs: out_shadowed[i] = 1 if some candidate is hit at
 * epsilon <= t < |light - point|. */
int rt_ads_trace_any(rt_ads* ads, const float* points, const float* lights, uint64_t n, int cull_backfaces, int mem,
                     uint8_t* out_shadowed, void* stream);

/* Per-triangle shading inputs and the material table (Triangle::vertices[].norm, Triangle::material_id,
 * src/triangle.h:10-14; Material, src/material.h:7-16 as 18 floats: ambient, diffuse, specular,
 * transmittance, emission, shininess, ior, dissolve).  normals may be NULL (flat shading only,
 * obj_loader.cc:161-163).  Triangles whose material emits become the light list in Build order
 * (obj_loader.cc:188-191).  Call after rt_ads_build. */
int rt_scene_set_shading(rt_ads* ads, const float* normals, const int32_t* material_ids, const float* materials,
                         uint32_t n_materials);

/* Renderer::config_setup + Camera (src/Renderer/renderer.cc:337-379, src/camera.h:15-21). */
typedef struct rt_render_config {
    uint32_t struct_size;        /* = sizeof(rt_render_config) */
    float cam_pos[3], cam_up[3], cam_dir[3];
    float fov;
    int32_t width, height, samples_per_pixel;
    int32_t max_depth;            /* "max_depth" of the renderer: recursion limit */
    int32_t samples_per_triangle;
    int32_t render_type;          /* rt_render_type */
    int32_t shading_type;         /* rt_shading_type */
    int32_t cull_backfaces;
    float max_distance;
    float background[3];
    int32_t aspect_mode;          /* rt_aspect_mode */
    uint32_t seed;                /* counter-based generator key ("seed" of the config) */
    /* Image partition for multi-GPU: the image is cut into bands of 8 pixel rows (one row of 8x8
     * wavefront tiles); this call renders the bands b with b % band_world == band_rank. */
    int32_t band_rank, band_world;
} rt_render_config;

typedef struct rt_render_stats {
    uint64_t primary_rays, secondary_rays, shadow_rays; /* each is one Search-equivalent query */
    uint64_t rays;                /* sum of the three */
    uint32_t levels;              /* recursion levels that had work */
    float ms_closest, ms_shadow, ms_shade, ms_total;  /* device time by kernel class (HIP events) */
    uint32_t launches_closest, launches_shadow, launches_shade;
    /* The events cover at most the first 64 launches of each class per call; these are the rays of exactly the timed
     * launches, so that rays / ms is a kernel rate also for frames with more launches (many samples per pixel). */
    uint64_t timed_closest_rays, timed_shadow_rays;
    /* Scenes whose materials spawn at most one secondary ray per hit: once a recursion level is small, the rest of the
     * recursion (closest hits, shadow rays, shading of every deeper level) runs as ONE launch; its time and count. */
    float ms_tail;
    uint32_t launches_tail;
    /* Shadow rays (counted in shadow_rays, as the reference casts them) that were answered without a walk: the light
     * sample they test adds a zero to the hit's colour whether it is visible or not -- the light faces away from the
     * hit, or the surface faces away from the light and has no specular colour -- so it is reported as shadowed and the
     * image has the same bits (the second case with flat shading only).  Finite lights and materials, big trees
     * (per-lane walks) only; RT_SKIP_UNLIT=0 at rt_ads_create traces every ray. */
    uint64_t shadow_rays_not_traced;
} rt_render_stats;

/* Renderer::RenderScene's hot loop (src/Renderer/renderer.cc:37-52) on the device: primary rays
 * (renderer.cc:82-112), closest hit, local illumination with area-light shadow samples
 * (renderer.cc:202-247, 288-313), reflection / refraction recursion (renderer.cc:170-192,315-335),
 * fuzzysampling (renderer.cc:43-49).  Random numbers come from a counter-based generator keyed by
 * (seed, pixel, sample, path node, light, light sample, dimension), so the image does not depend on
 * how the work is partitioned.  out_rgb: height x width x 3 floats (clamped colours as the reference
 * stores them before write_color); only the pixels of this call's bands are written. */
int rt_render(rt_ads* ads, const rt_render_config* cfg, float* out_rgb, int mem, rt_render_stats* stats, void* stream);

/* How many sub-frames a rt_render call renders concurrently (1..8, default 1, or the value of the environment variable
 * RT_RENDER_STREAMS at rt_ads_create).  The call's bands are dealt round-robin to n sub-frames; each has its own level
 * buffers and runs on a HIP stream and a host thread of the library's own (the first one on the caller's stream and
 * thread), ordered after what the caller's stream holds at the call; the call still returns when the whole frame is
 * complete, and the image does not depend on n.  A frame of many small recursion levels is a chain of launches that
 * each last as long as their slowest wavefront; sub-frames fill the gaps of one another.  No reference counterpart
 * (Renderer::RenderScene, renderer.cc:37-52, is one sequential loop). */
int rt_ads_set_render_streams(rt_ads* ads, int n);

/* Per-light forms of the wave-together any-hit walks (no reference counterpart; the visibility bits are the ones
 * Renderer::is_shadowed, renderer.cc:288-313, computes).  With back-face culling on, a triangle whose front side holds a
 * whole light -- every vertex of the light triangle at least a rounding margin in front of its plane -- can never be an
 * accepted occluder of a shadow ray that ends on that light (collision_detection.h:31-35 rejects it from behind; from the
 * front the ray ends before it reaches the plane, renderer.cc:302), so rt_render's any-hit passes towards that light
 * walk lists without it.  Built by rt_scene_set_shading for trees in the small-tree or clustered form and at most 64
 * lights; RT_LIGHT_CULL=0 in the environment at rt_ads_create switches it off.  Reports the lights that have such lists,
 * the (light, triangle) pairs of the scene and how many of them are left out. */
int rt_ads_light_forms(const rt_ads* ads, uint32_t* lights_with_forms, uint64_t* pairs, uint64_t* pairs_left_out);

/* No counterpart in the reference: inspection of a second set of lists of the small-tree form.  A shadow ray starts on
 * the triangle its hit lies on; with back-face culling on, a triangle that has ALL of that hit triangle behind its plane
 * (by a rounding margin) cannot be an accepted occluder of the ray either -- running against the normal it moves away
 * from the plane, running with it it meets a back face.  rt_scene_set_shading builds, per light and hit triangle, the
 * light's list without those triangles, and a wavefront of rt_render whose hits lie on one triangle walks that list
 * (frames whose camera is within 4 scene diagonals of the scene only); same visibility bits as Renderer::is_shadowed
 * (renderer.cc:288-313).  RT_HIT_CULL=0 in the environment at rt_ads_create switches it off.  Reports the (hit triangle,
 * listed triangle) pairs of those lists and how many of them are left out; zeros: no such lists. */
int rt_ads_hit_lists(const rt_ads* ads, uint64_t* pairs, uint64_t* pairs_left_out);

/* Frames in flight.  rt_render returns when its frame is complete, so a caller that renders a sequence of frames (or,
 * with several GPUs, its bands of each) leaves the GPU to one frame at a time -- and a short frame is a chain of
 * launches that each last as long as their slowest wavefront.  rt_render_submit queues a frame and returns at once with a
 * ticket; up to 8 frames may be in flight per handle, each with its own level buffers, stream and host thread of the
 * library (the host side of a frame sizes every recursion level from totals the kernels publish).  The frame starts
 * behind what `stream` holds at the call.  rt_render_wait blocks until that frame is complete in out_rgb_device (device
 * memory, height x width x 3 floats, only this call's bands written), returns its statistics and frees the ticket.
 * Frames in flight must write different buffers or different bands; the image of a frame is the one rt_render gives.
 * RT_ERR_CAPACITY when 8 frames are in flight already; RT_ERR_UNSUPPORTED while rt_ads_collect_search_stats is on.
 * Calls on one handle come from one thread at a time.  No reference counterpart (Renderer::RenderScene,
 * renderer.cc:37-52, renders one frame per process). */
int rt_render_submit(rt_ads* ads, const rt_render_config* cfg, float* out_rgb_device, void* stream, int* ticket);
int rt_render_wait(rt_ads* ads, int ticket, rt_render_stats* stats);

/* Primary-ray closest hits only (sample 0, no shading): per pixel t and triangle index. */
int rt_render_primary_hits(rt_ads* ads, const rt_render_config* cfg, float* out_t, int32_t* out_tri, int mem, void* stream);

/* The shared counter-based generator, exposed so that host code can reproduce a sample:
 * returns the 31-bit integer k; the float is (float)k / (float)2147483647. */
uint32_t rt_rng_u31(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id, uint32_t light, uint32_t sample,
                    uint32_t dim);

/* ---- host-only view of Build, no GPU needed ------------------------------------------------------------
 * The octree exactly as rt_ads_build constructs it before the upload (Octree::Build,
 * src/ADS/Octree/octree.cc:19-171): nodes in the reference's queue (breadth-first) order with their
 * boxes, child links (octant 0..7, -1 = null) and triangle index lists, plus the flattened device
 * layout (128-byte internal-node records, 48-byte leaf-slot records).  For inspection and tests. */
typedef struct rt_host_tree rt_host_tree;
int rt_host_tree_build(const float* positions, uint64_t n_triangles, uint32_t max_triangles_per_bb, uint32_t max_depth,
                       rt_host_tree** out);
void rt_host_tree_destroy(rt_host_tree* t);
int rt_host_tree_sizes(const rt_host_tree* t, uint64_t* n_nodes, uint64_t* n_pool, uint64_t* n_records, uint64_t* n_slots,
                       uint32_t* max_stack);
int rt_host_tree_export(const rt_host_tree* t, uint8_t* is_leaf, uint32_t* depth, float* bmin, float* bmax, int32_t* child8,
                        uint64_t* tri_begin, uint32_t* tri_count, uint32_t* pool);
int rt_host_tree_export_flat(const rt_host_tree* t, void* node_records, void* leaf_slots, uint32_t* root_info2);
/* The clustered form of the same tree, which the wave-together any-hit kernels walk on trees with more than 32 leaves
 * (no reference counterpart: the candidates of Octree::Search, octree.cc:186-191, are unchanged -- it only lets a wave
 * test a triangle once where several reached leaves list it).  Leaves are grouped into clusters: the leaves of each
 * maximal subtree with <= 32 leaves, and above those subtrees the leaf children of each node.
 * node_records (as many as rt_host_tree_export_flat writes): info[o][1] of an internal child carries 0x40000000 if
 * that child roots such a subtree; info[o][0] of a leaf = first slot | ordinal << 24, the ordinal numbering the leaves
 * of its cluster in slot order; the first two padding words of the record that owns a cluster (the subtree's root, or
 * the parent of the leaf children) = first record and record count of the cluster's list in cluster_triangles.
 * cluster_triangles (48-byte leaf-slot records, *n_cluster_triangles of them): per cluster its distinct triangles in
 * ascending index, first padding word = the set of the cluster's leaves (bit = ordinal) that list the triangle; every
 * list is padded to a multiple of 4 records with all-zero records.  Call once with NULL arrays for the count.
 * *n_clusters = 0 and nothing written if the form does not apply (root leaf, no triangle listed twice in any cluster,
 * >= 2^24 slots). */
int rt_host_tree_export_clustered(const rt_host_tree* t, void* node_records, void* cluster_triangles, uint64_t* n_cluster_triangles,
                                  uint32_t* root_info1, uint32_t* n_clusters);
int rt_host_tree_stats(const rt_host_tree* t, rt_tree_stats* out);

/* ---- host-side scene input and image output (no GPU needed) ---------------------------------------------
 * LoadScene (src/ObjLoader/obj_loader.cc:130-212): Wavefront OBJ/MTL -> triangles in file order with per-vertex
 * normals and material ids + the material table (18 floats per material as in rt_scene_set_shading).  Quads are
 * split like the reference's loader does; a face without a valid material is an error. */
typedef struct rt_scene_data rt_scene_data;
int rt_scene_load_obj(const char* obj_path, const char* basepath, int triangulate, rt_scene_data** out);
void rt_scene_data_free(rt_scene_data* s);
int rt_scene_data_sizes(const rt_scene_data* s, uint64_t* n_triangles, uint32_t* n_materials, int* has_normals);
int rt_scene_data_export(const rt_scene_data* s, float* positions, float* normals, int32_t* material_ids, float* materials);
/* Renderer::save_image_to_pmm + write_color (src/Renderer/renderer.cc:381-390, src/color.h:12-15): P3 text. */
int rt_write_ppm(const char* path, int32_t width, int32_t height, const float* rgb);

/* Diagnostics for tests: every (ray, triangle) pair through the three forms in which the kernels evaluate
 * collision_ray_triangle (src/collision_detection.h:12-68) -- the compiled form and the two hand-scheduled any-hit forms
 * (triangle in scalar registers / per lane) -- so that a test can compare the distances they compute bit for bit.
 * origins, dirs: n_rays x 3 floats (unit-length or zero directions: the hand-scheduled forms' precondition);
 * positions: n_triangles x 9 floats; out: 3 x n_rays x n_triangles floats, form-major, +inf = no usable hit
 * (t == infinity || t < epsilon, renderer.cc:126). */
int rt_debug_triangle_tests(int device, const float* origins, const float* dirs, uint32_t n_rays, const float* positions,
                            uint32_t n_triangles, int cull_backfaces, float* out);

const char* rt_last_error(void);
/* Library / device identification, for logs. */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
