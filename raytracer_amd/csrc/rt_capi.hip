// rt_capi.hip -- the C ABI declared in include/raytracer_amd.h: handle management, host-side
// octree build + upload, kernel launches.  Everything below the ABI is C++/HIP for gfx950.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "host_octree.h"
#include "rt_build.h"
#include "rt_kernels.h"
#include "scene_io.h"

using namespace rt;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T>
struct DevBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        size_t want = std::max<size_t>(n, 64);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    // like reserve, but the first `keep` elements survive a reallocation (capacity at least doubles)
    hipError_t grow_keep(size_t n, size_t keep) {
        if (n <= cap) return hipSuccess;
        const size_t want = std::max<size_t>({n, 2 * cap, 64});
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Level {
    DevBuf<RayRec> rays;
    DevBuf<NodeOut> out;
    DevBuf<float> child;  // kNodeChildFloats per node: the colours its children return
    DevBuf<float> hit_t;
    DevBuf<int32_t> hit_tri;
    DevBuf<uint32_t> work;
    DevBuf<uint32_t> vis;  // visibility bits of the level's shadow samples
    uint32_t n = 0;
    hipError_t reserve(uint32_t count, bool need_rays) {
        hipError_t e;
        if (need_rays && (e = rays.reserve(count)) != hipSuccess) return e;
        if ((e = out.reserve(count)) != hipSuccess) return e;
        if ((e = child.reserve((size_t)count * kNodeChildFloats)) != hipSuccess) return e;
        if ((e = hit_t.reserve(count)) != hipSuccess) return e;
        if ((e = hit_tri.reserve(count)) != hipSuccess) return e;
        return work.reserve(count);
    }
    void release() {
        rays.release(); out.release(); child.release(); hit_t.release(); hit_tri.release(); work.release(); vis.release();
    }
};

// Workspace of rt_ads_search: owned by the handle, grow-only.
struct SearchWork {
    DevBuf<float> rays;
    DevBuf<uint32_t> queue, out;
    DevBuf<uint64_t> counts, offs;
    DevBuf<unsigned long long> stats;
    unsigned long long* h_stats = nullptr;  // pinned
    std::vector<uint32_t> host;
    void release() {
        rays.release(); queue.release(); out.release(); counts.release(); offs.release(); stats.release();
        if (h_stats) (void)hipHostFree(h_stats);
        h_stats = nullptr;
    }
};

constexpr int kMaxTimed = 64;
// Everything one frame in flight owns: the level buffers of the recursion, the counters, the mailboxes through which
// the kernels tell the host the level sizes, the timing events.  A handle has one per concurrent sub-frame.
constexpr int kMaxRenderStreams = 8;
#ifndef RT_MAX_FRAMES_IN_FLIGHT
#define RT_MAX_FRAMES_IN_FLIGHT 8
#endif
constexpr int kMaxFramesInFlight = RT_MAX_FRAMES_IN_FLIGHT;  // slots of rt_render_submit
struct FrameCtx {
    std::vector<Level> levels;
    DevBuf<uint32_t> d_counters;  // one CNT_WORDS block per level
    // what the device tells the host while a frame is in flight, in pinned host memory the kernels write directly
    LevelSlot* h_slots = nullptr;      // one per recursion level
    FrameTotals* h_totals = nullptr;
    DevBuf<FrameTotals> d_totals;
    uint32_t slot_seq = 0;
    DevBuf<float> d_accum;
    DevBuf<float> d_tail;  // k_tail: (colour, factor) per fused level and path
    hipEvent_t ev[2 + 6 * kMaxTimed] = {};
    bool events_ready = false;
    hipStream_t own_stream = nullptr;  // sub-frames 1..: a stream of the library's own
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void release() {
        for (auto& l : levels) l.release();
        d_counters.release(); d_totals.release(); d_accum.release(); d_tail.release();
        if (h_slots) (void)hipHostFree(h_slots);
        if (h_totals) (void)hipHostFree(h_totals);
        if (events_ready)
            for (auto& e : ev) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        h_slots = nullptr; h_totals = nullptr; events_ready = false; own_stream = nullptr; ev_fork = ev_join = nullptr;
    }
};

// Host threads that run the sub-frames 1.. of a rt_render call (sub-frame 0 runs on the caller's thread).  The threads
// are started once and sleep between frames.
class SubFramePool {
  public:
    explicit SubFramePool(int workers) {
        for (int k = 0; k < workers; ++k) threads_.emplace_back([this, k] { loop(k); });
    }
    ~SubFramePool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            quit_ = true;
        }
        cv_job_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int workers() const { return (int)threads_.size(); }
    // Runs f(1) .. f(n - 1) on the workers and f(0) on the caller; returns when all have returned.
    void run(int n, const std::function<void(int)>& f) {
        {
            std::lock_guard<std::mutex> lock(m_);
            job_ = &f;
            n_jobs_ = n;
            pending_ = std::min(n - 1, (int)threads_.size());
            ++generation_;
        }
        cv_job_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lock(m_);
        cv_done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(int k) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f = nullptr;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_job_.wait(lock, [&] { return quit_ || (generation_ != seen && k + 1 < n_jobs_); });
                if (quit_) return;
                seen = generation_;
                f = job_;
            }
            (*f)(k + 1);
            {
                std::lock_guard<std::mutex> lock(m_);
                --pending_;
            }
            cv_done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    const std::function<void(int)>* job_ = nullptr;
    int n_jobs_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

// One frame in flight of rt_render_submit / rt_render_wait: a frame context, a stream and a host thread of its own (the
// host side of a frame polls the level totals the kernels publish, so every frame in flight needs a thread).
struct AsyncSlot {
    FrameCtx ctx;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    enum { IDLE, QUEUED, DONE } state = IDLE;
    bool quit = false;
    rt_render_config cfg{};
    float* d_out = nullptr;
    int n_in_flight = 1;  // frames in flight on the GPU when this one was submitted (itself included)
    rt_render_stats st{};
    int rc = 0;
    char err[512] = "";
};

constexpr uint64_t kDeviceBuildMinTriangles = 16384;
constexpr uint32_t kTailMaxRays = 1u << 18;
constexpr bool kClipDefault = true;  // RT_CLIP=0: every ray walks the whole line like the reference (same results)
constexpr uint64_t kWaveFormMaxSlots = 32768;  // trees with fewer leaf slots are walked by the wave together, whatever the rays

// Launches KERNEL<cull, mode>(args...) for run-time cull / traversal mode.
// lds: bytes of dynamic LDS (the per-lane form's stacks; 0 for the others).
#define RT_LAUNCH_MODE(KERNEL, CULL, mode, grid, block, lds, stream, ...)                                           \
    do {                                                                                                            \
        if ((mode) == MODE_POOL) hipLaunchKernelGGL((KERNEL<CULL, MODE_POOL>), grid, block, 0, stream, __VA_ARGS__);          \
        else if ((mode) == MODE_LANE) hipLaunchKernelGGL((KERNEL<CULL, MODE_LANE>), grid, block, lds, stream, __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<CULL, MODE_WAVE>), grid, block, 0, stream, __VA_ARGS__);                    \
    } while (0)
#define RT_LAUNCH2(KERNEL, cull, mode, grid, block, lds, stream, ...)                                               \
    do {                                                                                                            \
        if (cull) RT_LAUNCH_MODE(KERNEL, true, mode, grid, block, lds, stream, __VA_ARGS__);                        \
        else RT_LAUNCH_MODE(KERNEL, false, mode, grid, block, lds, stream, __VA_ARGS__);                            \
    } while (0)

template <class... Args>
void launch_shadow(bool cull, int mode, bool fast_tri, bool side, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (mode == MODE_LANE && side && cull) {  // (per-light trees exist for culled tests only)
        if (fast_tri) hipLaunchKernelGGL((k_shadow<true, MODE_LANE, 1, true, true>), grid, block, lds, stream, args...);
        else hipLaunchKernelGGL((k_shadow<true, MODE_LANE, 1, false, true>), grid, block, lds, stream, args...);
        return;
    }
#define RT_SHADOW_CASE(MODE, WAVES)                                                                                          \
    if (mode == MODE) {                                                                                                      \
        if (fast_tri && cull) hipLaunchKernelGGL((k_shadow<true, MODE, WAVES, true>), grid, block, lds, stream, args...);     \
        else if (fast_tri) hipLaunchKernelGGL((k_shadow<false, MODE, WAVES, true>), grid, block, lds, stream, args...);       \
        else if (cull) hipLaunchKernelGGL((k_shadow<true, MODE, WAVES, false>), grid, block, lds, stream, args...);           \
        else hipLaunchKernelGGL((k_shadow<false, MODE, WAVES, false>), grid, block, lds, stream, args...);                    \
        return;                                                                                                              \
    }
    RT_SHADOW_CASE(MODE_POOL, 1)   // one-wave blocks
    RT_SHADOW_CASE(MODE_LANE, 1)   // one-wave blocks
    RT_SHADOW_CASE(MODE_CLUSTER, 4)
    RT_SHADOW_CASE(MODE_WAVE, 4)
#undef RT_SHADOW_CASE
}

}  // namespace

// ---- fixed device scratch of rt_ads_search_noalloc ---------------------------------------------------------------------
// Layout: RT_SEARCH_NOALLOC_MAX_RAYS 64-bit candidate counts, then the rays' breadth-first queues (two words per
// entry, one queue of (octree nodes) entries per ray).  Every GPU the library runs on holds its own copy of the array
// at its own address; `scratch` below is only the host-side stand-in of the symbol, never a pointer to pass on: the
// device address is fetched once per device with hipGetSymbolAddress (scratch_for_device) and that pointer is what
// kernels, memsets and copies get.
constexpr size_t kNoallocCountBytes = (size_t)RT_SEARCH_NOALLOC_MAX_RAYS * sizeof(uint64_t);
static __device__ __align__(256) unsigned char scratch[kNoallocCountBytes + RT_SEARCH_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

// The device address of `scratch` on `device` (which must be the current device).
int scratch_for_device(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_UNSUPPORTED, "device ordinal %d beyond %d", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned a null address for the search scratch");
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}
}  // namespace

struct rt_ads {
    int device = 0;
    uint32_t max_per_bb = 16, max_depth = 10;
    bool built = false, shading_set = false, has_normals = false, branching = false, branching_any = false, use_flat = true;
    int traversal = 0;  // 0 = choose per launch, 1 = always wave-together, 2 = always per-lane, 3 = always the task pool
    // walk only the part of a ray that can contribute a result (Span in rt_traverse.h); false: the reference's whole line
    bool clip = false;
    uint64_t prune_min_slots = kWaveFormMaxSlots;  // see closest_opts()
    uint32_t pool_target_waves = 8192, lane_target_waves = 2048, pool_min_rays = 1;
    uint32_t tail_target_waves = 1024;
    uint32_t tail_max_rays = kTailMaxRays;  // levels with at most this many rays end the level-by-level loop (k_tail); 0 = never
    int builder = 0;    // rt_builder requested at creation
    bool device_built = false;
    uint64_t n_tri = 0;
    std::vector<float> positions;
    // summary of the resident tree (the arrays themselves live in d_nodes / d_slots / d_flat_*)
    bool empty = true;
    uint32_t root_info[2] = {0, 0};
    uint32_t max_stack = 0;
    uint64_t n_rec = 0, n_slots = 0, n_tree_nodes = 0, n_flat_tris = 0;
    double build_seconds = 0.0;
    TreeShape shape;
    DevBuf<NodeRec> d_nodes;
    DevBuf<LeafTri> d_slots, d_tri_geom, d_flat_tris;
    DevBuf<float> d_slot_geom;  // 9 floats per leaf slot (DevOctree::slot_geom)
    DevBuf<NodeRec> d_flat_nodes;
    // clustered form (ClusterForm in host_octree.h): any-hit walks of trees that stay in the wave-together form
    DevBuf<NodeRec> d_cl_nodes;
    DevBuf<LeafTri> d_cl_tris;
    uint32_t n_clusters = 0, cl_root_info1 = 0;
    bool use_clusters = true;
    // every triangle has |e1| * |e2| far below the range limit of the refined reciprocal: shadow rays (unit directions)
    // may use the any-hit triangle test that never needs the full division (any_hit_exec)
    bool fast_tri = false;
    // Per-light forms (make_light_forms): host copies of the small-tree triangle list and of the clustered form, kept
    // from Build so that rt_scene_set_shading can derive each light's lists once it knows the lights
    std::vector<LeafTri> h_flat_tris, h_cl_tris;
    std::vector<NodeRec> h_cl_nodes;
    bool light_cull = true;  // RT_LIGHT_CULL=0 at creation: every light's any-hit passes walk the full lists
    bool skip_unlit = true;  // RT_SKIP_UNLIT=0 at creation: shadow rays whose answer cannot change the image are traced too
    bool hit_cull = true;    // RT_HIT_CULL=0 at creation: no per-(light, hit triangle) lists in the small-tree form
    uint64_t hit_pairs = 0, hit_pairs_culled = 0;  // (hit triangle, listed triangle) pairs of those lists, and the ones left out
    double scene_mid[3] = {0, 0, 0}, scene_diag = 0;  // the scene's box, for the camera check of the hit lists
    DevBuf<uint2> d_hit_lists;
    DevBuf<LightForm> d_light_forms;
    DevBuf<LeafTri> d_light_flat, d_light_cl_tris;
    DevBuf<NodeRec> d_light_cl_nodes;
    DevBuf<NodeRec> d_light_lane_nodes;   // per-lane form: per light a copy of the node records ...
    DevBuf<float> d_light_lane_geom;      // ... and the geometry of the slots its leaves keep
    bool light_forms_ready = false, light_forms_lane = false;
    bool lights_merged = false;  // every light has the same forms (one copy of the arrays)
    bool light_merge = true;     // RT_LIGHT_MERGE=0 at creation: never merge
    // per-lane any-hit passes with per-light trees: launches of more rays than this serve one light per pass, smaller
    // ones keep all lights of a hit side by side (RT_LANE_SIDE_MAX_RAYS)
    uint64_t lane_side_max_rays = 1ull << 22;
    uint64_t light_pairs = 0, light_pairs_culled = 0;  // (light, triangle) pairs of the scene / those left out
    DevBuf<TriShade> d_tri_shade;
    DevBuf<float> d_vnorm;
    DevBuf<MatDev> d_mats;
    DevBuf<LightDev> d_lights;
    uint32_t n_lights = 0;
    // rt_render: the frame is rendered as `render_streams` sub-frames (bands dealt round-robin) that run concurrently,
    // each with its own level buffers on its own stream and host thread (frame[0]: the caller's stream and thread)
    FrameCtx frame[kMaxRenderStreams];
    int render_streams = 1;
    SubFramePool* pool = nullptr;
    // rt_render_submit / rt_render_wait: whole frames in flight, one slot each (created on first use)
    AsyncSlot* async_slots = nullptr;
    int async_in_flight = 0;
    DevBuf<float> d_stage;
    DevBuf<uint8_t> d_io;  // staging for host-memory query calls
    SearchWork search;
    unsigned char* noalloc_scratch = nullptr;  // this device's copy of the fixed scratch array (rt_ads_search_noalloc)
    rt_query_stats q{};
    bool collect_stats = false;  // run the instrumented pass for the reference-defined counters of rt_query_stats
    DevBuf<unsigned long long> d_qstats;      // QS_WORDS totals of the instrumented pass
    unsigned long long* h_qstats = nullptr;   // pinned
};

namespace {

DevOctree dev_tree(const rt_ads* a) {
    DevOctree t;
    t.nodes = a->d_nodes.p;
    t.slots = a->d_slots.p;
    t.root_info0 = a->root_info[0];
    t.root_info1 = a->empty ? 0u : a->root_info[1];
    t.n_nodes = (uint32_t)a->n_rec;
    t.n_slots = (uint32_t)a->n_slots;
    t.flat_nodes = a->d_flat_nodes.p;
    t.flat_tris = a->d_flat_tris.p;
    t.n_flat_tris = a->use_flat ? (uint32_t)a->n_flat_tris : 0u;
    const bool clustered = a->use_clusters && a->n_clusters != 0;
    t.cl_nodes = clustered ? a->d_cl_nodes.p : nullptr;
    t.cl_tris = clustered ? a->d_cl_tris.p : nullptr;
    t.cl_root_info1 = a->cl_root_info1;
    t.slot_geom = a->d_slot_geom.p;
    return t;
}

// Which traversal a launch of n_rays rays uses.  Small trees are always walked by the wave together (their small-tree
// form bounds the work by the triangle count).  Large trees: camera rays of an 8x8 pixel tile are coherent and stay
// wave-together; other rays are walked per lane when the launch is large enough to fill the chip that way.  Smaller
// launches are bound by the latency of one wave's chain of dependent steps and share their descent in a task pool
// (measured on a 1.3 M-triangle scene, 88 K closest-hit rays: 400 us per lane, 120 us pool; break-even near half a
// million closest-hit and a quarter million any-hit rays).
constexpr uint64_t kPoolMaxRaysClosest = 1u << 19, kPoolMaxRaysAny = 1u << 18;
int traversal_mode(const rt_ads* a, bool coherent_primary, bool any_hit, uint64_t n_rays) {
    const bool small_tree = a->use_flat && a->n_flat_tris != 0;
    // any-hit rays that the wave walks together take the clustered form when the tree has one and no small-tree form
    const int wave = (any_hit && !small_tree && a->use_clusters && a->n_clusters != 0) ? MODE_CLUSTER : MODE_WAVE;
    if (a->max_stack == 0) return MODE_WAVE;  // (a root-leaf tree has nothing to descend)
    if (a->traversal == 1) return wave;
    const bool pool_ok = a->shape.max_tris_in_leaf < (1u << kPoolCountBits);
    if (a->traversal == 2) return MODE_LANE;
    if (a->traversal == 3) return pool_ok ? MODE_POOL : MODE_LANE;
    if (small_tree || a->n_slots < kWaveFormMaxSlots) return wave;
    // Clipped walks are short (an ordered closest-hit walk ends soon after its first hit, an any-hit walk covers the
    // segment only), and the per-lane form then beats both the union walk of a tile of camera rays (1.31 M triangles, 4K:
    // 1.05 against 1.32 ms with the union walk ordered and pruned as well) and the task pool for small launches (88 K rays: 31 against 104 us).
    if (a->clip) return MODE_LANE;
    if (coherent_primary) return wave;
    return (pool_ok && n_rays < (any_hit ? kPoolMaxRaysAny : kPoolMaxRaysClosest)) ? MODE_POOL : MODE_LANE;
}

// Pool form: how many lanes of a wave carry rays (64 >> result).  A launch too small to give every SIMD a few waves
// is spread over more waves; at least max_group rays (a power of two) stay together.
uint32_t pool_sparse(const rt_ads* a, int mode, uint64_t n_rays, uint32_t max_group) {
    // (the per-lane form as well: a small launch lasts as long as one wave's chain of dependent steps, which is shorter
    // with fewer rays per wave -- fewer pair chunks, and the lanes without a ray take over work of the others)
    if (mode != MODE_POOL && mode != MODE_LANE) return 0;
    // waves the launch is spread over at least, if it has the rays: measured best on a 1.3 M-triangle scene
    const uint32_t target = mode == MODE_POOL ? a->pool_target_waves : a->lane_target_waves;
    uint32_t s = 0;
    while ((64u >> (s + 1)) >= std::max(max_group, a->pool_min_rays) && (n_rays >> (6 - s)) < target) ++s;
    return s;
}

// Threads per block of k_closest.  Its hit compaction costs one atomic per block and the whole block waits for its
// slowest wave there: big blocks pay off where waves take similar, short times (small trees), not on big trees where
// a wave over a finely tessellated region runs many times longer than its neighbours.
// Closest-hit launches: the wave-together walk of a tree below prune_min_slots leaf slots is not pruned by the best hit
// (see closest_opts).
bool closest_prunes(const rt_ads* a, int mode) { return a->clip && !(mode == MODE_WAVE && a->n_slots < a->prune_min_slots); }
uint32_t closest_block(const rt_ads* a, int mode) {
    return (mode == MODE_WAVE && a->n_slots < kWaveFormMaxSlots && !closest_prunes(a, mode)) ? 1024u : 256u;
}

template <class... Args>
void launch_closest(bool cull, int mode, bool prune, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    const bool big = block.x > 256;
    if (mode == MODE_WAVE && prune) {  // (never with the big blocks of the small trees)
        if (cull) hipLaunchKernelGGL((k_closest<true, MODE_WAVE_PRUNED, false>), grid, block, 0, stream, args...);
        else hipLaunchKernelGGL((k_closest<false, MODE_WAVE_PRUNED, false>), grid, block, 0, stream, args...);
    } else if (mode == MODE_POOL) {
        if (cull) hipLaunchKernelGGL((k_closest<true, MODE_POOL, false>), grid, block, 0, stream, args...);
        else hipLaunchKernelGGL((k_closest<false, MODE_POOL, false>), grid, block, 0, stream, args...);
    } else if (mode == MODE_LANE) {
        if (cull) hipLaunchKernelGGL((k_closest<true, MODE_LANE, false>), grid, block, lds, stream, args...);
        else hipLaunchKernelGGL((k_closest<false, MODE_LANE, false>), grid, block, lds, stream, args...);
    } else if (big) {
        if (cull) hipLaunchKernelGGL((k_closest<true, MODE_WAVE, true>), grid, block, 0, stream, args...);
        else hipLaunchKernelGGL((k_closest<false, MODE_WAVE, true>), grid, block, 0, stream, args...);
    } else {
        if (cull) hipLaunchKernelGGL((k_closest<true, MODE_WAVE, false>), grid, block, 0, stream, args...);
        else hipLaunchKernelGGL((k_closest<false, MODE_WAVE, false>), grid, block, 0, stream, args...);
    }
}

// Threads a launch needs for n rays: whole waves of 64 >> sparse rays.
uint64_t mode_threads(int, uint64_t n_rays, uint32_t sparse = 0) {
    const uint64_t per_wave = 64u >> sparse;
    return (n_rays + per_wave - 1) / per_wave * 64u;
}

// Levels of internal nodes of the resident tree (the depth of the per-lane stacks), the packed launch options of a
// traversal kernel, and the dynamic LDS a launch of the per-lane form needs.
uint32_t tree_levels(const rt_ads* a) { return std::max<uint32_t>(a->max_stack / 8u, 1u); }
uint32_t launch_opts(const rt_ads* a, uint32_t sparse) { return trace_opts(sparse, a->clip, tree_levels(a)); }
// Closest-hit launches: the wave-together walk of a tree below prune_min_slots leaf slots is not pruned by the best hit.
// Such a tree is a few levels deep, the union walk of a bundle opens most of it anyway, and the bookkeeping of the
// pruned walk (ordered pushes, entry distances) then costs more than it saves: Cornell-Sphere's closest-hit passes take
// 0.84 ms unpruned against 0.91 ms pruned.  The result is the same either way.
uint32_t closest_opts(const rt_ads* a, int mode, uint32_t sparse) { return trace_opts(sparse, closest_prunes(a, mode), tree_levels(a)); }
size_t lane_lds(const rt_ads* a, int mode, uint32_t block_threads, bool closest) {
    return mode == MODE_LANE ? (size_t)(block_threads / 64u) * lane_stack_words(tree_levels(a), closest) * sizeof(uint32_t) : 0;
}

int use_device(const rt_ads* a) {
    HIP_TRY(hipSetDevice(a->device));
    return RT_OK;
}

// The instrumented pass (rt_stats.h): zero the totals before the kernels of a call, fetch and accumulate them after.
int stats_begin(rt_ads* a, hipStream_t stream) {
    HIP_TRY(a->d_qstats.reserve(QS_WORDS));
    if (!a->h_qstats) HIP_TRY(hipHostMalloc((void**)&a->h_qstats, QS_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(a->d_qstats.p, 0, QS_WORDS * sizeof(unsigned long long), stream));
    return RT_OK;
}
int stats_end(rt_ads* a, hipStream_t stream) {  // synchronises the stream
    HIP_TRY(hipMemcpyAsync(a->h_qstats, a->d_qstats.p, QS_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    a->q.search_node_count += a->h_qstats[QS_NODES];
    a->q.search_leaves_visited += a->h_qstats[QS_LEAVES];
    a->q.search_return_count += a->h_qstats[QS_RETURNED];
    a->q.slab_tests += a->h_qstats[QS_SLAB];
    a->q.ray_triangle_tests += a->h_qstats[QS_TRI];
    a->q.counted_rays += a->h_qstats[QS_RAYS];
    a->q.slab_tests_any += a->h_qstats[QS_ANY + QS_SLAB];
    a->q.ray_triangle_tests_any += a->h_qstats[QS_ANY + QS_TRI];
    a->q.counted_rays_any += a->h_qstats[QS_ANY + QS_RAYS];
    return RT_OK;
}

bool tri_equal(const float* pa, const float* na, const float* pb, const float* nb) {
    // src/triangle.h:15-19 with src/vertex.h:12 and src/vec3.h:32
    auto veq = [&](int i, int j) {
        const f3 p0 = mk(pa[i * 3], pa[i * 3 + 1], pa[i * 3 + 2]), p1 = mk(pb[j * 3], pb[j * 3 + 1], pb[j * 3 + 2]);
        f3 n0 = mk(0, 0, 0), n1 = mk(0, 0, 0);
        if (na) n0 = mk(na[i * 3], na[i * 3 + 1], na[i * 3 + 2]);
        if (nb) n1 = mk(nb[j * 3], nb[j * 3 + 1], nb[j * 3 + 2]);
        return eq_eps(p0, p1) && eq_eps(n0, n1);
    };
    return (veq(0, 0) && veq(1, 1) && veq(2, 2)) || (veq(0, 1) && veq(1, 2) && veq(2, 0)) || (veq(0, 2) && veq(1, 0) && veq(2, 1));
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err; }
const char* rt_version(void) { return "raytracer_amd 0.1 (gfx950, wave64 packet octree traversal)"; }

uint32_t rt_rng_u31(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id, uint32_t light, uint32_t sample,
                    uint32_t dim) {
    return rng_finish(rng_prefix(seed, pixel, spp_sample, node_id), light, sample, dim);
}

int rt_ads_kind_from_name(const char* name, int* out_kind) {
    if (!name || !out_kind) return fail(RT_ERR_INVALID, "null argument");
    if (!strcmp(name, "none")) *out_kind = RT_ADS_NONE;
    else if (!strcmp(name, "octree")) *out_kind = RT_ADS_OCTREE;
    else if (!strcmp(name, "octee_parametric")) *out_kind = RT_ADS_OCTREE_PARAMETRIC;  // the reference's spelling
    else if (!strcmp(name, "octree_gpu")) *out_kind = RT_ADS_OCTREE_GPU;
    else return fail(RT_ERR_INVALID, "Invalid data structure");
    return RT_OK;
}

}  // extern "C"

struct rt_host_tree {
    HostOctree host;
    FlatOctree flat;
};

extern "C" {

}  // extern "C"

struct rt_scene_data {
    SceneData d;
};

extern "C" {

int rt_scene_load_obj(const char* obj_path, const char* basepath, int triangulate, rt_scene_data** out) {
    if (!obj_path || !out) return fail(RT_ERR_INVALID, "null argument");
    rt_scene_data* s = new rt_scene_data();
    std::string err;
    if (!load_obj(obj_path, basepath ? basepath : "", triangulate != 0, s->d, err)) {
        delete s;
        return fail(RT_ERR_INVALID, "%s", err.c_str());
    }
    *out = s;
    return RT_OK;
}
void rt_scene_data_free(rt_scene_data* s) { delete s; }
int rt_scene_data_sizes(const rt_scene_data* s, uint64_t* n_triangles, uint32_t* n_materials, int* has_normals) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    if (n_triangles) *n_triangles = s->d.material_ids.size();
    if (n_materials) *n_materials = (uint32_t)(s->d.materials.size() / 18);
    if (has_normals) *has_normals = s->d.has_normals ? 1 : 0;
    return RT_OK;
}
int rt_scene_data_export(const rt_scene_data* s, float* positions, float* normals, int32_t* material_ids, float* materials) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    if (positions && !s->d.positions.empty()) memcpy(positions, s->d.positions.data(), s->d.positions.size() * sizeof(float));
    if (normals && !s->d.normals.empty()) memcpy(normals, s->d.normals.data(), s->d.normals.size() * sizeof(float));
    if (material_ids && !s->d.material_ids.empty()) memcpy(material_ids, s->d.material_ids.data(), s->d.material_ids.size() * sizeof(int32_t));
    if (materials && !s->d.materials.empty()) memcpy(materials, s->d.materials.data(), s->d.materials.size() * sizeof(float));
    return RT_OK;
}
int rt_write_ppm(const char* path, int32_t width, int32_t height, const float* rgb) {
    if (!path || !rgb || width < 1 || height < 1) return fail(RT_ERR_INVALID, "bad argument");
    std::string err;
    if (!write_ppm(path, width, height, rgb, err)) return fail(RT_ERR_INVALID, "%s", err.c_str());
    return RT_OK;
}

int rt_host_tree_build(const float* positions, uint64_t n, uint32_t max_per_bb, uint32_t max_depth, rt_host_tree** out) {
    if ((!positions && n) || !out) return fail(RT_ERR_INVALID, "null argument");
    rt_host_tree* t = new rt_host_tree();
    std::string err;
    if (!build_octree(positions, n, max_per_bb, max_depth, t->host, err)) {
        delete t;
        return fail(RT_ERR_BUILD, "%s", err.c_str());
    }
    if (!flatten_octree(t->host, positions, t->flat, err)) {
        delete t;
        return fail(RT_ERR_BUILD, "%s", err.c_str());
    }
    *out = t;
    return RT_OK;
}
void rt_host_tree_destroy(rt_host_tree* t) { delete t; }
int rt_host_tree_sizes(const rt_host_tree* t, uint64_t* n_nodes, uint64_t* n_pool, uint64_t* n_records, uint64_t* n_slots,
                       uint32_t* max_stack) {
    if (!t) return fail(RT_ERR_INVALID, "null argument");
    if (n_nodes) *n_nodes = t->host.nodes.size();
    if (n_pool) *n_pool = t->host.pool.size();
    if (n_records) *n_records = t->flat.nodes.size();
    if (n_slots) *n_slots = t->flat.slots.size();
    if (max_stack) *max_stack = t->flat.max_stack;
    return RT_OK;
}
int rt_host_tree_export(const rt_host_tree* t, uint8_t* is_leaf, uint32_t* depth, float* bmin, float* bmax, int32_t* child8,
                        uint64_t* tri_begin, uint32_t* tri_count, uint32_t* pool) {
    if (!t) return fail(RT_ERR_INVALID, "null argument");
    for (size_t i = 0; i < t->host.nodes.size(); ++i) {
        const HostNode& n = t->host.nodes[i];
        if (is_leaf) is_leaf[i] = n.leaf;
        if (depth) depth[i] = n.depth;
        if (bmin) { bmin[i * 3] = n.bmin.x; bmin[i * 3 + 1] = n.bmin.y; bmin[i * 3 + 2] = n.bmin.z; }
        if (bmax) { bmax[i * 3] = n.bmax.x; bmax[i * 3 + 1] = n.bmax.y; bmax[i * 3 + 2] = n.bmax.z; }
        if (child8) for (int o = 0; o < 8; ++o) child8[i * 8 + o] = n.child[o];
        if (tri_begin) tri_begin[i] = n.tri_begin;
        if (tri_count) tri_count[i] = n.tri_count;
    }
    if (pool && !t->host.pool.empty()) memcpy(pool, t->host.pool.data(), t->host.pool.size() * sizeof(uint32_t));
    return RT_OK;
}
int rt_host_tree_export_flat(const rt_host_tree* t, void* node_records, void* leaf_slots, uint32_t* root_info2) {
    if (!t) return fail(RT_ERR_INVALID, "null argument");
    if (node_records && !t->flat.nodes.empty()) memcpy(node_records, t->flat.nodes.data(), t->flat.nodes.size() * sizeof(NodeRec));
    if (leaf_slots && !t->flat.slots.empty()) memcpy(leaf_slots, t->flat.slots.data(), t->flat.slots.size() * sizeof(LeafTri));
    if (root_info2) {
        root_info2[0] = t->flat.root_info[0];
        root_info2[1] = t->host.empty ? 0u : t->flat.root_info[1];
    }
    return RT_OK;
}
int rt_host_tree_export_clustered(const rt_host_tree* t, void* node_records, void* cluster_triangles, uint64_t* n_cluster_triangles,
                                  uint32_t* root_info1, uint32_t* n_clusters) {
    if (!t) return fail(RT_ERR_INVALID, "null argument");
    ClusterForm cf;
    if (!t->host.empty) cluster_octree(t->flat.nodes.data(), t->flat.nodes.size(), t->flat.slots.data(), t->flat.slots.size(), t->flat.root_info, cf);
    if (n_clusters) *n_clusters = cf.n_clusters;
    if (root_info1) *root_info1 = cf.root_info1;
    if (n_cluster_triangles) *n_cluster_triangles = cf.tris.size();
    if (cf.n_clusters == 0) return RT_OK;
    if (node_records) memcpy(node_records, cf.nodes.data(), cf.nodes.size() * sizeof(NodeRec));
    if (cluster_triangles) memcpy(cluster_triangles, cf.tris.data(), cf.tris.size() * sizeof(LeafTri));
    return RT_OK;
}
int rt_host_tree_stats(const rt_host_tree* t, rt_tree_stats* out) {
    if (!t || !out) return fail(RT_ERR_INVALID, "null argument");
    const TreeShape s = tree_shape(t->host);
    memset(out, 0, sizeof *out);
    out->max_depth = s.max_depth; out->nodes = s.nodes; out->leaves = s.leaves; out->max_triangles_in_leaf = s.max_tris_in_leaf;
    out->avg_leaf_depth = s.avg_leaf_depth; out->avg_triangles_in_leaf = s.avg_tris_in_leaf; out->leaf_refs = s.leaf_refs;
    out->internal_nodes = s.internal; out->build_seconds = t->host.build_seconds;
    out->device_bytes = t->flat.nodes.size() * sizeof(NodeRec) + t->flat.slots.size() * sizeof(LeafTri);
    return RT_OK;
}

int rt_ads_create(const rt_ads_config* cfg, rt_ads** out) {
    if (!cfg || !out) return fail(RT_ERR_INVALID, "null argument");
    if (cfg->struct_size != sizeof(rt_ads_config)) return fail(RT_ERR_INVALID, "rt_ads_config.struct_size mismatch");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (cfg->device < 0 || cfg->device >= count) return fail(RT_ERR_INVALID, "device %d out of range (%d visible)", cfg->device, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (prop.warpSize != 64) return fail(RT_ERR_UNSUPPORTED, "device %d is not a wave64 device", cfg->device);
    rt_ads* a = new rt_ads();
    a->device = cfg->device;
    a->max_per_bb = cfg->max_triangles_per_bb;
    a->max_depth = cfg->max_depth;
    if (cfg->builder > RT_BUILDER_DEVICE) {
        delete a;
        return fail(RT_ERR_INVALID, "unknown builder %u", cfg->builder);
    }
    a->builder = (int)cfg->builder;
    if (const char* e = getenv("RT_LIGHT_CULL")) a->light_cull = !(e[0] == '0');
    if (const char* e = getenv("RT_SKIP_UNLIT")) a->skip_unlit = !(e[0] == '0');
    if (const char* e = getenv("RT_HIT_CULL")) a->hit_cull = !(e[0] == '0');
    if (const char* e = getenv("RT_LIGHT_MERGE")) a->light_merge = !(e[0] == '0');
    if (const char* e = getenv("RT_LANE_SIDE_MAX_RAYS")) a->lane_side_max_rays = (uint64_t)atoll(e);
    if (const char* e = getenv("RT_RENDER_STREAMS")) a->render_streams = std::min(std::max(atoi(e), 1), kMaxRenderStreams);
    // the fixed scratch of rt_ads_search_noalloc: its address on this device, looked up with the device current
    {
        hipError_t e = hipSetDevice(cfg->device);
        int rc = e == hipSuccess ? scratch_for_device(cfg->device, &a->noalloc_scratch) : fail(RT_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
        if (rc != RT_OK) {
            delete a;
            return rc;
        }
    }
    *out = a;
    return RT_OK;
}

void rt_ads_destroy(rt_ads* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    delete a->pool;  // joins the sub-frame threads
    if (a->async_slots) {
        for (int k = 0; k < kMaxFramesInFlight; ++k) {
            AsyncSlot& s = a->async_slots[k];
            {
                std::lock_guard<std::mutex> lock(s.m);
                s.quit = true;
            }
            s.cv.notify_all();
            if (s.th.joinable()) s.th.join();  // (a frame still in flight is completed first)
            s.ctx.release();
        }
        delete[] a->async_slots;
    }
    a->d_nodes.release(); a->d_slots.release(); a->d_slot_geom.release(); a->d_tri_geom.release(); a->d_tri_shade.release(); a->d_vnorm.release();
    a->d_mats.release(); a->d_lights.release(); a->d_flat_nodes.release(); a->d_flat_tris.release(); a->d_cl_nodes.release(); a->d_cl_tris.release(); a->d_stage.release(); a->d_io.release();
    for (auto& x : a->frame) x.release();
    a->d_light_forms.release(); a->d_light_flat.release(); a->d_light_cl_tris.release(); a->d_light_cl_nodes.release();
    a->d_hit_lists.release();
    a->d_light_lane_nodes.release(); a->d_light_lane_geom.release();
    a->search.release();
    a->d_qstats.release();
    if (a->h_qstats) (void)hipHostFree(a->h_qstats);
    delete a;
}

}  // extern "C"

namespace {

int check_tree_limits(const rt_ads* a) {
    if (a->max_stack > 128)
        return fail(RT_ERR_UNSUPPORTED, "octree too deep for the register-resident traversal stack (%u > 128 entries)", a->max_stack);
    if (a->n_slots > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "more than 2^31-1 leaf triangle references");
    if (a->n_rec >= (1u << 24)) return fail(RT_ERR_UNSUPPORTED, "more than 2^24 internal nodes");
    if (a->shape.max_depth > (uint64_t)kLaneMaxLevels) return fail(RT_ERR_UNSUPPORTED, "octree deeper than %d levels", kLaneMaxLevels);
    return RT_OK;
}

// Octree::Build on the host (host_octree.cpp), flattened and uploaded.
int host_build(rt_ads* a) {
    const uint64_t n = a->n_tri;
    HostOctree host;
    FlatOctree flat;
    std::string err;
    if (!build_octree(a->positions.data(), n, a->max_per_bb, a->max_depth, host, err)) return fail(RT_ERR_BUILD, "%s", err.c_str());
    if (!flatten_octree(host, a->positions.data(), flat, err)) return fail(RT_ERR_BUILD, "%s", err.c_str());
    a->empty = host.empty;
    a->root_info[0] = flat.root_info[0];
    a->root_info[1] = flat.root_info[1];
    a->max_stack = flat.max_stack;
    a->n_rec = flat.nodes.size();
    a->n_slots = flat.slots.size();
    a->n_tree_nodes = host.nodes.size();
    a->n_flat_tris = flat.flat_tris.size();
    a->h_flat_tris = flat.flat_tris;
    a->build_seconds = host.build_seconds;
    a->shape = tree_shape(host);
    if (int rc = check_tree_limits(a)) return rc;
    HIP_TRY(a->d_nodes.reserve(flat.nodes.size()));
    HIP_TRY(a->d_slots.reserve(flat.slots.size() + 4));  // +4: triangle records are fetched in groups of 4
    HIP_TRY(a->d_tri_geom.reserve(n));
    if (!flat.nodes.empty()) HIP_TRY(hipMemcpy(a->d_nodes.p, flat.nodes.data(), flat.nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
    if (!flat.slots.empty()) HIP_TRY(hipMemcpy(a->d_slots.p, flat.slots.data(), flat.slots.size() * sizeof(LeafTri), hipMemcpyHostToDevice));
    if (!flat.flat_tris.empty()) {
        HIP_TRY(a->d_flat_nodes.reserve(flat.flat_nodes.size()));
        HIP_TRY(a->d_flat_tris.reserve(flat.flat_tris.size() + 4));
        HIP_TRY(hipMemcpy(a->d_flat_nodes.p, flat.flat_nodes.data(), flat.flat_nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(a->d_flat_tris.p, flat.flat_tris.data(), flat.flat_tris.size() * sizeof(LeafTri), hipMemcpyHostToDevice));
        // the pass over this array runs in blocks of 4 records; the padding records list no leaf (mask 0)
        HIP_TRY(hipMemset(a->d_flat_tris.p + flat.flat_tris.size(), 0, 4 * sizeof(LeafTri)));
    }
    if (n) {
        std::vector<LeafTri> g(n);
        for (uint64_t i = 0; i < n; ++i) {
            const float* p = a->positions.data() + i * 9;
            LeafTri& s = g[i];
            s.ax = p[0]; s.ay = p[1]; s.az = p[2];
            s.e1x = p[3] - p[0]; s.e1y = p[4] - p[1]; s.e1z = p[5] - p[2];
            s.e2x = p[6] - p[0]; s.e2y = p[7] - p[1]; s.e2z = p[8] - p[2];
            s.tri = (uint32_t)i;
            s.pad0 = s.pad1 = 0;
        }
        HIP_TRY(hipMemcpy(a->d_tri_geom.p, g.data(), n * sizeof(LeafTri), hipMemcpyHostToDevice));
    }
    return RT_OK;
}

// The clustered form of the resident tree (either builder's), for trees whose any-hit rays are walked by the wave
// together: the records are read back, annotated on the host (cluster_octree) and uploaded beside the plain arrays.
int make_cluster_form(rt_ads* a) {
    a->n_clusters = 0;
    a->cl_root_info1 = 0;
    a->h_cl_nodes.clear();
    a->h_cl_tris.clear();
    if (a->empty || a->n_rec == 0 || !(a->root_info[1] & kChildInternal)) return RT_OK;
    if (a->traversal != 1 && a->n_slots >= kWaveFormMaxSlots) return RT_OK;  // its any-hit rays use the per-lane / pool forms
    if (a->n_slots > kClusterSlotMask) return RT_OK;
    std::vector<NodeRec> nodes(a->n_rec);
    std::vector<LeafTri> slots(a->n_slots);
    HIP_TRY(hipMemcpy(nodes.data(), a->d_nodes.p, nodes.size() * sizeof(NodeRec), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(slots.data(), a->d_slots.p, slots.size() * sizeof(LeafTri), hipMemcpyDeviceToHost));
    ClusterForm cf;
    cluster_octree(nodes.data(), nodes.size(), slots.data(), slots.size(), a->root_info, cf);
    if (cf.n_clusters == 0) return RT_OK;
    HIP_TRY(a->d_cl_nodes.reserve(cf.nodes.size()));
    HIP_TRY(a->d_cl_tris.reserve(cf.tris.size()));  // (every list is a whole number of 4-record blocks)
    HIP_TRY(hipMemcpy(a->d_cl_nodes.p, cf.nodes.data(), cf.nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(a->d_cl_tris.p, cf.tris.data(), cf.tris.size() * sizeof(LeafTri), hipMemcpyHostToDevice));
    a->n_clusters = cf.n_clusters;
    a->cl_root_info1 = cf.root_info1;
    a->h_cl_nodes = std::move(cf.nodes);
    a->h_cl_tris = std::move(cf.tris);
    return RT_OK;
}

// What the builders of the per-light forms share (make_light_forms below states the rule): the handle, its triangle
// count, the sets of forms to build, culled[l * n + j] (triangle j cannot occlude the lights of set l) and the records
// that are filled in.
struct LightFormJob {
    rt_ads* a;
    uint64_t n;
    size_t nb;
    const std::vector<uint8_t>& culled;
    std::vector<LightForm>& forms;
    double D, margin_scale;  // the diagonal of the scene's box; the test switch that shrinks the margins
    const double* lo;        // the scene's box
    const double* hi;
};
#define LIGHT_FORM_JOB_LOCALS                           \
    rt_ads* a = J.a;                                    \
    const uint64_t n = J.n;                             \
    const size_t nb = J.nb;                             \
    const std::vector<uint8_t>& culled = J.culled;      \
    std::vector<LightForm>& forms = J.forms;            \
    LeafTri zero;                                       \
    memset(&zero, 0, sizeof zero)

// Small-tree form, per light: the triangles in their order, padded to whole blocks with records that list no leaf; and
// the per-hit-triangle lists.
int flat_light_lists(const LightFormJob& J) {
    LIGHT_FORM_JOB_LOCALS;
    const double D = J.D, margin_scale = J.margin_scale;
    const double* lo = J.lo;
    const double* hi = J.hi;
    std::vector<LeafTri> all;
    std::vector<size_t> first(nb);
    std::vector<uint32_t> count(nb);
    for (size_t l = 0; l < nb; ++l) {
        first[l] = all.size();
        for (const LeafTri& t : a->h_flat_tris)
            if (!culled[l * n + t.tri]) all.push_back(t);
        do all.push_back(zero); while ((all.size() - first[l]) % kTriChunk != 0);  // (at least one record: a count of 0 means "no small-tree form")
        count[l] = (uint32_t)(all.size() - first[l]);
    }
    // Per (light, hit triangle h): the light's list without the triangles j that have ALL of h behind their plane.
    // A shadow ray that starts on h starts behind such a plane: running against N_j it moves away from the plane and
    // its crossing lies at t < 0, running with N_j it meets a back face -- never an accepted occluder, by the same
    // argument and the same margin m_j as above with the hit point in the light's place.  The hit point is not
    // exactly on h: the reference's t of the hit is off by at most 4 * 2^-24 * (k_h (|q| + t) + t), and |q| + t stays
    // below 10 D for a camera within 4 D of the scene (render_bands checks that per frame), so h must lie behind
    // the plane by m_j + m_hit(h), m_hit(h) = 64 * 2^-24 * (10 k_h + 5) * D -- sixteen times that bound again.
    // A wave whose hits lie on one triangle walks that triangle's list (k_shadow).
    std::vector<uint2> hit_lists;
    a->hit_pairs = a->hit_pairs_culled = 0;
    // (at most 2^21 list records, 96 MB: a list per light form and triangle, each as long as the light's own at most)
    if (a->hit_cull && n >= 2 && n <= kFlatMaxTris && (uint64_t)nb * n * (a->h_flat_tris.size() + kTriChunk) <= (1ull << 21)) {
        struct Plane { double N[3], nn, m, m_hit; bool ok; };
        std::vector<Plane> pl(n);
        for (uint64_t j = 0; j < n; ++j) {
            const float* p = a->positions.data() + j * 9;
            double e1[3], e2[3];
            for (int k = 0; k < 3; ++k) {
                e1[k] = (double)(float)(p[3 + k] - p[k]);
                e2[k] = (double)(float)(p[6 + k] - p[k]);
            }
            Plane& P = pl[j];
            P.N[0] = e1[1] * e2[2] - e1[2] * e2[1];
            P.N[1] = e1[2] * e2[0] - e1[0] * e2[2];
            P.N[2] = e1[0] * e2[1] - e1[1] * e2[0];
            P.nn = std::sqrt(P.N[0] * P.N[0] + P.N[1] * P.N[1] + P.N[2] * P.N[2]);
            const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
            const double kappa = l1 * l2 / P.nn;
            P.m = margin_scale * 64.0 * std::ldexp(1.0, -24) * (2.0 * kappa + 1.0) * D;
            P.m_hit = margin_scale * 64.0 * std::ldexp(1.0, -24) * (10.0 * kappa + 5.0) * D;
            P.ok = P.nn > 0.0 && std::isfinite(P.nn) && std::isfinite(P.m) && std::isfinite(P.m_hit);
        }
        std::vector<uint8_t> behind(n * n, 0);  // behind[h * n + j]: all of triangle h lies behind the plane of j
        for (uint64_t h = 0; h < n; ++h) {
            if (!pl[h].ok) continue;
            const float* q = a->positions.data() + h * 9;
            for (uint64_t j = 0; j < n; ++j) {
                if (j == h || !pl[j].ok) continue;
                const float* p = a->positions.data() + j * 9;
                bool all_behind = true;
                for (int v = 0; v < 3 && all_behind; ++v) {
                    double side = 0.0;
                    for (int k = 0; k < 3; ++k) side += pl[j].N[k] * ((double)q[v * 3 + k] - (double)p[k]);
                    all_behind = side / pl[j].nn <= -(pl[j].m + pl[h].m_hit);
                }
                behind[h * n + j] = all_behind ? 1 : 0;
            }
        }
        hit_lists.resize(nb * n);
        for (size_t l = 0; l < nb; ++l)
            for (uint64_t h = 0; h < n; ++h) {
                const size_t begin = all.size();
                for (const LeafTri& t : a->h_flat_tris) {
                    if (culled[l * n + t.tri]) continue;
                    ++a->hit_pairs;
                    if (behind[h * n + t.tri]) ++a->hit_pairs_culled;
                    else all.push_back(t);
                }
                do all.push_back(zero); while ((all.size() - begin) % kTriChunk != 0);
                hit_lists[l * n + h] = make_uint2((uint32_t)(begin - first[l]), (uint32_t)(all.size() - begin));
            }
    }
    for (int k = 0; k < kTriChunk; ++k) all.push_back(zero);  // the pass reads whole blocks
    HIP_TRY(a->d_light_flat.reserve(all.size()));
    HIP_TRY(hipMemcpy(a->d_light_flat.p, all.data(), all.size() * sizeof(LeafTri), hipMemcpyHostToDevice));
    if (!hit_lists.empty()) {
        HIP_TRY(a->d_hit_lists.reserve(hit_lists.size()));
        HIP_TRY(hipMemcpy(a->d_hit_lists.p, hit_lists.data(), hit_lists.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
    for (size_t l = 0; l < nb; ++l) {
        forms[l].flat_tris = a->d_light_flat.p + first[l];
        forms[l].n_flat_tris = count[l];
        forms[l].hit_lists = hit_lists.empty() ? nullptr : a->d_hit_lists.p + l * n;
    }
    // the frames whose camera is near enough for the hit lists' margin: within 4 D of the middle of the scene's box
    for (int k = 0; k < 3; ++k) a->scene_mid[k] = 0.5 * (lo[k] + hi[k]);
    a->scene_diag = D;
    return RT_OK;
}

// Clustered form, per light: every cluster's list without the culled triangles, and node records that point at them.
int cluster_light_forms(const LightFormJob& J) {
    LIGHT_FORM_JOB_LOCALS;
    const size_t n_rec = a->h_cl_nodes.size();
    std::vector<NodeRec> nodes;
    std::vector<LeafTri> tris;
    std::vector<size_t> tri_first(nb);
    nodes.reserve(nb * n_rec);
    for (size_t l = 0; l < nb; ++l) {
        tri_first[l] = tris.size();
        for (size_t r = 0; r < n_rec; ++r) {
            NodeRec rec = a->h_cl_nodes[r];
            const uint32_t begin = rec.pad[0], cnt = rec.pad[1];
            if (cnt != 0) {
                const size_t at = tris.size();
                for (uint32_t k = 0; k < cnt; ++k) {
                    const LeafTri& t = a->h_cl_tris[begin + k];
                    if (t.pad0 != 0 && !culled[l * n + t.tri]) tris.push_back(t);  // (pad0 == 0: the old list's padding)
                }
                while ((tris.size() - at) % kTriChunk != 0) tris.push_back(zero);
                rec.pad[0] = (uint32_t)(at - tri_first[l]);
                rec.pad[1] = (uint32_t)(tris.size() - at);
            }
            nodes.push_back(rec);
        }
    }
    for (int k = 0; k < kTriChunk; ++k) tris.push_back(zero);
    HIP_TRY(a->d_light_cl_nodes.reserve(nodes.size()));
    HIP_TRY(a->d_light_cl_tris.reserve(tris.size()));
    HIP_TRY(hipMemcpy(a->d_light_cl_nodes.p, nodes.data(), nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(a->d_light_cl_tris.p, tris.data(), tris.size() * sizeof(LeafTri), hipMemcpyHostToDevice));
    for (size_t l = 0; l < nb; ++l) {
        forms[l].cl_nodes = a->d_light_cl_nodes.p + l * n_rec;
        forms[l].cl_tris = a->d_light_cl_tris.p + tri_first[l];
    }
    return RT_OK;
}

// Per-lane form (big trees), per light: the node records with the light's leaf ranges -- octants and subtrees left without
// a triangle become empty -- and the geometry of the slots that remain.
int lane_light_forms(const LightFormJob& J) {
    LIGHT_FORM_JOB_LOCALS;
    (void)zero;
    const size_t n_rec = a->n_rec, n_slots = a->n_slots;
    std::vector<NodeRec> nodes(n_rec);
    std::vector<LeafTri> slots(n_slots);
    HIP_TRY(hipMemcpy(nodes.data(), a->d_nodes.p, n_rec * sizeof(NodeRec), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(slots.data(), a->d_slots.p, n_slots * sizeof(LeafTri), hipMemcpyDeviceToHost));
    std::vector<NodeRec> all_nodes(nb * n_rec);
    std::vector<std::vector<float>> geom_of(nb);
    std::vector<size_t> geom_first(nb);
    std::vector<uint32_t> root1(nb);
    std::vector<int> bad(nb, 0);
    // one host thread per light (at most kMaxLaneLightForms)
    auto one_light = [&](size_t l) {
        const uint8_t* cl = culled.data() + l * n;
        std::vector<float>& geom = geom_of[l];
        std::vector<uint32_t> pref(n_slots + 1);
        // the slots that stay, in their order; pref[s] = how many of the slots before s stay
        uint32_t kept = 0;
        for (size_t sidx = 0; sidx < n_slots; ++sidx) {
            pref[sidx] = kept;
            kept += cl[slots[sidx].tri] ? 0u : 1u;
        }
        pref[n_slots] = kept;
        geom.resize(((size_t)kept + kTriChunk) * 9, 0.0f);  // (a lane without a pair reads record 0 of its list)
        for (size_t sidx = 0; sidx < n_slots; ++sidx) {
            const LeafTri& t = slots[sidx];
            if (cl[t.tri]) continue;
            float* g = geom.data() + (size_t)pref[sidx] * 9;
            g[0] = t.ax; g[1] = t.ay; g[2] = t.az; g[3] = t.e1x; g[4] = t.e1y; g[5] = t.e1z; g[6] = t.e2x; g[7] = t.e2y; g[8] = t.e2z;
        }
        // records bottom-up (children follow their parents in the array): leaf ranges renumbered, leaves and
        // subtrees without a triangle become empty octants
        NodeRec* out = all_nodes.data() + l * n_rec;
        for (size_t r = n_rec; r-- > 0;) {
            NodeRec rec = nodes[r];
            uint32_t nonempty = 0;
            for (int c = 0; c < 8; ++c) {
                const uint32_t i0 = rec.info[c][0], i1 = rec.info[c][1];
                if (i1 == 0) continue;
                if (i1 & kChildInternal) {
                    if (i0 <= r || i0 >= n_rec) { bad[l] = 1; return; }  // (records not in breadth-first order)
                    if (out[i0].nonempty == 0) rec.info[c][0] = rec.info[c][1] = 0;
                } else {
                    if ((uint64_t)i0 + i1 > n_slots) { bad[l] = 2; return; }  // (leaf range beyond the slot array)
                    const uint32_t nb = pref[i0], nc = pref[i0 + i1] - pref[i0];
                    rec.info[c][0] = nc ? nb : 0u;
                    rec.info[c][1] = nc;
                }
                if (rec.info[c][1] != 0) nonempty |= 1u << c;
            }
            rec.nonempty = nonempty;
            node_ordered_nonempty(nonempty, &rec.pad[1]);
            out[r] = rec;
        }
        root1[l] = out[a->root_info[0]].nonempty != 0 ? a->root_info[1] : 0u;  // (0: nothing can occlude this light)
    };
    {
        std::vector<std::thread> workers;
        for (size_t l = 1; l < nb; ++l) workers.emplace_back(one_light, l);
        one_light(0);
        for (auto& w : workers) w.join();
    }
    for (size_t l = 0; l < nb; ++l)
        if (bad[l]) return fail(RT_ERR_BUILD, "internal: the resident tree is not in the layout the per-light copies assume (%d)", bad[l]);
    size_t geom_total = 0;
    for (size_t l = 0; l < nb; ++l) {
        geom_first[l] = geom_total;
        geom_total += geom_of[l].size();
    }
    HIP_TRY(a->d_light_lane_nodes.reserve(all_nodes.size()));
    HIP_TRY(a->d_light_lane_geom.reserve(geom_total));
    HIP_TRY(hipMemcpy(a->d_light_lane_nodes.p, all_nodes.data(), all_nodes.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
    for (size_t l = 0; l < nb; ++l)
        HIP_TRY(hipMemcpy(a->d_light_lane_geom.p + geom_first[l], geom_of[l].data(), geom_of[l].size() * sizeof(float), hipMemcpyHostToDevice));
    for (size_t l = 0; l < nb; ++l) {
        forms[l].lane_nodes = a->d_light_lane_nodes.p + l * n_rec;
        forms[l].lane_slot_geom = a->d_light_lane_geom.p + geom_first[l];
        forms[l].lane_root_info1 = root1[l];
        forms[l].lane_rec_base = (uint32_t)(l * n_rec);
        forms[l].lane_slot_base = (uint32_t)(geom_first[l] / 9);
    }
    a->light_forms_lane = true;
    return RT_OK;
}
#undef LIGHT_FORM_JOB_LOCALS

// Per-light forms of the small-tree triangle list and of the clustered form.
//
// With back faces culled (collision_detection.h:31-35: d = dot(e1, cross(s, e2)) < epsilon is a miss) a triangle j can
// occlude a shadow ray only where the ray runs AGAINST its geometric normal N = e1 x e2, i.e. from the side N points to
// towards the other side.  A shadow ray ends at a point of the light (renderer.cc:302 keeps t < dist only).  If the whole
// light triangle lies on the side N points to, at least m away from j's plane, a ray that ends on the light and runs
// against N has not reached the plane yet when it ends: its plane crossing lies beyond the light by at least m along the
// ray, whatever its origin.  The reference's own arithmetic cannot turn that into an accepted hit as long as m exceeds
// what rounding can move the computed t: t = dot(e2, w) / d with relative errors of a few 2^-24 in each factor moves t by
// less than 4 * 2^-24 * (k * (|q| + dist) + dist), k = |e1||e2| / |N| (1 for a right angle at a), also for grazing rays,
// where numerator and d shrink together; a ray that runs parallel to the plane keeps its distance h >= m from it and
// gets t = h |N| / (rounding noise of d) > 31 D.  With every point of the scene inside a box of diagonal D,
//     m_j = 64 * 2^-24 * (2 k_j + 1) * D
// is sixteen times that bound (0.011 for the Cornell box: the light quad hangs 0.1 below the ceiling).  Such a triangle
// is left out of light l's lists: the candidates it would have contributed are never accepted, so every visibility bit
// is the one the reference computes.  Nothing changes for two-sided tests (cull off), for caller-supplied segments
// (rt_ads_trace_any) or for closest-hit rays: those walk the full lists.
constexpr uint32_t kMaxLightForms = 64, kMaxLaneLightForms = 8;
int make_light_forms(rt_ads* a, const std::vector<uint32_t>& light_tri) {
    a->light_forms_ready = a->light_forms_lane = a->lights_merged = false;
    a->light_pairs = a->light_pairs_culled = 0;
    a->hit_pairs = a->hit_pairs_culled = 0;
    a->scene_diag = 0.0;
    if (!a->light_cull) return RT_OK;
    const size_t nl = light_tri.size();
    const bool have_flat = !a->h_flat_tris.empty(), have_cl = !a->h_cl_nodes.empty();
    // a tree whose any-hit rays are walked per lane: per-light node records and slot geometry (a few lights only: every
    // light costs a copy of the node records and up to a copy of the slot geometry)
    const bool have_lane = !a->empty && (a->root_info[1] & kChildInternal) && a->n_rec != 0 && nl <= kMaxLaneLightForms &&
                           (uint64_t)a->n_rec * nl < (1ull << 24) && (uint64_t)(a->n_slots + kTriChunk) * nl < (1ull << 31) &&
                           (a->traversal == 2 || (a->traversal == 0 && a->n_slots >= kWaveFormMaxSlots && !(a->use_flat && a->n_flat_tris != 0)));
    if (nl == 0 || nl > kMaxLightForms || (!have_flat && !have_cl && !have_lane)) return RT_OK;
    const uint64_t n = a->n_tri;
    // the box of all vertices and its diagonal
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint64_t i = 0; i < n * 3; ++i)
        for (int k = 0; k < 3; ++k) {
            const double x = a->positions[i * 3 + k];
            lo[k] = std::min(lo[k], x);
            hi[k] = std::max(hi[k], x);
        }
    double D = 0.0;
    for (int k = 0; k < 3; ++k) D += (hi[k] - lo[k]) * (hi[k] - lo[k]);
    D = std::sqrt(D);
    if (!std::isfinite(D)) return RT_OK;
    // (RT_LIGHT_CULL_MARGIN_SCALE: a test switch that shrinks the margin, to show where the rule starts to fail)
    double margin_scale = 1.0;
    if (const char* e = getenv("RT_LIGHT_CULL_MARGIN_SCALE")) margin_scale = atof(e);
    // culled[l * n + j]: triangle j cannot occlude light l
    std::vector<uint8_t> culled(nl * n, 0);
    auto classify = [&](uint64_t j_begin, uint64_t j_end) {
    for (uint64_t j = j_begin; j < j_end; ++j) {
        const float* p = a->positions.data() + j * 9;
        // the edges as the kernels hold them (rounded to float), everything after that in double
        double e1[3], e2[3], N[3];
        for (int k = 0; k < 3; ++k) {
            e1[k] = (double)(float)(p[3 + k] - p[k]);
            e2[k] = (double)(float)(p[6 + k] - p[k]);
        }
        N[0] = e1[1] * e2[2] - e1[2] * e2[1];
        N[1] = e1[2] * e2[0] - e1[0] * e2[2];
        N[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double nn = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        if (!(nn > 0.0) || !std::isfinite(nn)) continue;  // a degenerate triangle stays in every list
        const double kappa = l1 * l2 / nn;
        const double m = margin_scale * 64.0 * std::ldexp(1.0, -24) * (2.0 * kappa + 1.0) * D;
        if (!std::isfinite(m)) continue;
        for (size_t l = 0; l < nl; ++l) {
            const float* q = a->positions.data() + (size_t)light_tri[l] * 9;
            bool in_front = true;
            for (int v = 0; v < 3 && in_front; ++v) {
                double side = 0.0;
                for (int k = 0; k < 3; ++k) side += N[k] * ((double)q[v * 3 + k] - (double)p[k]);
                in_front = side / nn >= m;
            }
            if (in_front) culled[l * n + j] = 1;
        }
    }
    };
    {
        const unsigned n_threads = n >= (1u << 16) ? 8u : 1u;
        std::vector<std::thread> workers;
        for (unsigned k = 1; k < n_threads; ++k) workers.emplace_back(classify, n * k / n_threads, n * (k + 1) / n_threads);
        classify(0, n / n_threads);
        for (auto& w : workers) w.join();
    }
    // Lights that leave out nearly the same triangles -- the two halves of a light quad -- share ONE set of forms that
    // leaves out what ALL of them leave out (a subset of each light's own set, so the rule holds for every one): half the
    // memory and cache footprint, and the lights of a hit can be traced side by side through the same arrays.
    const size_t nl_all = nl;
    size_t nl_build = nl;
    if (a->light_merge && nl >= 2) {
        std::vector<uint8_t> all(culled.begin(), culled.begin() + n);
        uint64_t most = 0;
        for (size_t l = 0; l < nl; ++l) {
            uint64_t cnt = 0;
            for (uint64_t j = 0; j < n; ++j) {
                cnt += culled[l * n + j];
                all[j] &= culled[l * n + j];
            }
            most = std::max(most, cnt);
        }
        uint64_t common = 0;
        for (uint8_t c : all) common += c;
        if (common * 10 >= most * 9) {
            std::copy(all.begin(), all.end(), culled.begin());
            culled.resize(n);
            nl_build = 1;
            a->lights_merged = true;
        }
    }
    for (uint8_t c : culled) a->light_pairs_culled += c;
    a->light_pairs_culled *= nl_all / nl_build;
    a->light_pairs = (uint64_t)nl_all * n;
    const size_t nb = nl_build;  // sets of forms to build
    std::vector<LightForm> forms(nl_all);
    for (auto& f : forms) memset(&f, 0, sizeof f);
    const LightFormJob job{a, n, nb, culled, forms, D, margin_scale, lo, hi};
    if (have_flat)
        if (int rc = flat_light_lists(job)) return rc;
    if (have_cl)
        if (int rc = cluster_light_forms(job)) return rc;
    if (have_lane)
        if (int rc = lane_light_forms(job)) return rc;
    for (size_t l = nb; l < nl_all; ++l) forms[l] = forms[0];  // (merged: every light points at the one set)
    HIP_TRY(a->d_light_forms.reserve(nl_all));
    HIP_TRY(hipMemcpy(a->d_light_forms.p, forms.data(), nl_all * sizeof(LightForm), hipMemcpyHostToDevice));
    a->light_forms_ready = true;
    return RT_OK;
}

// 8-channel exclusive scan of n elements (rt_build.h); out receives n + 1 elements.
template <bool MASK>
int scan8(const void* in, uint32_t n, DevBuf<uint32_t>& bsum, uint32_t* out) {
    const uint32_t nb = (n + kScanTile - 1) / kScanTile;
    HIP_TRY(bsum.reserve(((size_t)nb + 1) * 8));
    if (nb) hipLaunchKernelGGL((k_scan_reduce<MASK>), dim3(nb), dim3(kScanThreads), 0, 0, in, n, bsum.p);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kScanThreads), 0, 0, bsum.p, nb);
    hipLaunchKernelGGL((k_scan_write<MASK>), dim3(std::max(nb, 1u)), dim3(kScanThreads), 0, 0, in, n, bsum.p, nb, out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Octree::Build on the device (rt_build.h): one pass per depth, the tree is written directly in its resident layout.
int device_build(rt_ads* a) {
    const auto t_begin = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)a->n_tri;
    a->empty = n == 0;
    a->root_info[0] = a->root_info[1] = 0;
    a->max_stack = 0;
    a->n_rec = a->n_slots = a->n_tree_nodes = a->n_flat_tris = 0;
    a->h_flat_tris.clear();
    a->shape = TreeShape();
    a->build_seconds = 0.0;
    if (n == 0) return RT_OK;  // octree.cc:24-27

    // root box on the host (octree.cc:33-48): a min/max sweep over the caller's array
    const float* pos = a->positions.data();
    f3 mn = mk(pos[0], pos[1], pos[2]), mx = mn;
    for (uint64_t i = 0; i < (uint64_t)n * 3; ++i) {
        const float* p = pos + i * 3;
        mn.x = p[0] < mn.x ? p[0] : mn.x;
        mn.y = p[1] < mn.y ? p[1] : mn.y;
        mn.z = p[2] < mn.z ? p[2] : mn.z;
        mx.x = p[0] > mx.x ? p[0] : mx.x;
        mx.y = p[1] > mx.y ? p[1] : mx.y;
        mx.z = p[2] > mx.z ? p[2] : mx.z;
    }
    const f3 rmin = mn - vabs(mn * mk(kAabbEps, kAabbEps, kAabbEps));
    const f3 rmax = mx + vabs(mx * mk(kAabbEps, kAabbEps, kAabbEps));

    DevBuf<float> d_pos;
    DevBuf<TriBuild> d_tb;
    DevBuf<BuildNode> d_bn[2];
    DevBuf<uint32_t> d_list[2], d_owner[2], d_S, d_V, d_P, d_cinfo, d_bsum, d_flags;
    DevBuf<uint8_t> d_mask;
    uint32_t* h_tot = nullptr;
    auto cleanup = [&] {
        d_pos.release(); d_tb.release(); d_bn[0].release(); d_bn[1].release(); d_list[0].release(); d_list[1].release();
        d_owner[0].release(); d_owner[1].release(); d_S.release(); d_V.release(); d_P.release(); d_cinfo.release();
        d_bsum.release(); d_flags.release(); d_mask.release();
        if (h_tot) (void)hipHostFree(h_tot);
    };
#define BUILD_TRY(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            cleanup();                                                                                         \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                                      \
    } while (0)
#define BUILD_RC(expr)         \
    do {                       \
        if (int rc_ = (expr)) { \
            cleanup();         \
            return rc_;        \
        }                      \
    } while (0)

    const dim3 blk(256);
    auto grid = [](uint32_t count) { return dim3((count + 255) / 256); };
    BUILD_TRY(hipHostMalloc((void**)&h_tot, 16 * sizeof(uint32_t)));
    BUILD_TRY(d_pos.reserve((size_t)n * 9));
    BUILD_TRY(d_tb.reserve(n));
    BUILD_TRY(a->d_tri_geom.reserve(n));
    BUILD_TRY(d_flags.reserve(2));  // [0] = triangles of subdivided nodes that reached no octant, [1] = largest leaf
    BUILD_TRY(hipMemcpy(d_pos.p, pos, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    BUILD_TRY(hipMemsetAsync(d_flags.p, 0, 2 * sizeof(uint32_t), 0));
    hipLaunchKernelGGL(k_build_prep, grid(n), blk, 0, 0, d_pos.p, n, d_tb.p, a->d_tri_geom.p);
    // depth 0: the root and the identity list
    BUILD_TRY(d_bn[0].reserve(1));
    BUILD_TRY(d_list[0].reserve(n));
    BUILD_TRY(d_owner[0].reserve(n));
    {
        BuildNode root{};
        root.bmin[0] = rmin.x; root.bmin[1] = rmin.y; root.bmin[2] = rmin.z;
        root.bmax[0] = rmax.x; root.bmax[1] = rmax.y; root.bmax[2] = rmax.z;
        root.begin = 0;
        root.count = n;
        root.parent = kNoParent;
        BUILD_TRY(hipMemcpy(d_bn[0].p, &root, sizeof root, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_build_iota, grid(n), blk, 0, 0, d_list[0].p, n);
    BUILD_TRY(hipMemsetAsync(d_owner[0].p, 0, (size_t)n * sizeof(uint32_t), 0));
    // first guesses for the outputs; both grow (keeping their contents) when a depth needs more
    BUILD_TRY(a->d_nodes.reserve(std::max<size_t>(1024, n / 8)));
    BUILD_TRY(a->d_slots.reserve((size_t)n * 3 + 4));

    uint32_t n_nodes = 1, n_list = n;
    uint64_t rec_base = 0, slot_base = 0;
    uint64_t total_nodes = 0, leaves = 0, total_depth = 0, total_tris = 0, max_leaf_depth = 0;
    int deepest_internal = -1;
    bool root_internal = false;
    int cur = 0;
    for (uint32_t depth = 0;; ++depth) {
        const int at_max = depth == a->max_depth ? 1 : 0;
        BUILD_TRY(d_mask.reserve(n_list));
        BUILD_TRY(d_S.reserve(((size_t)n_list + 1) * 8));
        BUILD_TRY(d_V.reserve((size_t)n_nodes * 8));
        BUILD_TRY(d_P.reserve(((size_t)n_nodes + 1) * 8));
        BUILD_TRY(d_cinfo.reserve((size_t)n_nodes * 16));
        hipLaunchKernelGGL(k_build_masks, grid(n_list), blk, 0, 0, d_bn[cur].p, d_list[cur].p, d_owner[cur].p, n_list, d_tb.p,
                           a->max_per_bb, at_max, d_mask.p, d_flags.p);
        BUILD_RC(scan8<true>(d_mask.p, n_list, d_bsum, d_S.p));
        hipLaunchKernelGGL(k_build_decide, grid(n_nodes), blk, 0, 0, d_bn[cur].p, n_nodes, d_S.p, a->max_per_bb, at_max, (int)depth,
                           d_V.p, d_flags.p + 1);
        BUILD_RC(scan8<false>(d_V.p, n_nodes, d_bsum, d_P.p));
        BUILD_TRY(hipMemcpyAsync(h_tot, d_P.p + (size_t)n_nodes * 8, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, 0));
        BUILD_TRY(hipMemcpyAsync(h_tot + 8, d_flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, 0));
        BUILD_TRY(hipStreamSynchronize(0));
        const uint32_t n_int = h_tot[CH_INTERNAL], n_slot = h_tot[CH_SLOTS], n_child = h_tot[CH_CHILDREN], n_child_list = h_tot[CH_CHILD_LIST];
        if (h_tot[8] != 0) {  // the reference's own invariant (octree.cc:124-152: it prints and exits)
            const uint32_t lost = h_tot[8];  // (cleanup() frees h_tot)
            cleanup();
            return fail(RT_ERR_BUILD, "octree build: %u triangles of subdivided nodes at depth %u were assigned to no octant", lost, depth);
        }
        if (rec_base + n_int >= (1u << 24)) {
            cleanup();
            return fail(RT_ERR_UNSUPPORTED, "more than 2^24 internal nodes");
        }
        if (slot_base + n_slot > 0x7fffffffull || (uint64_t)n_child_list > 0x7fffffffull) {
            cleanup();
            return fail(RT_ERR_UNSUPPORTED, "more than 2^31-1 leaf triangle references");
        }
        const uint32_t n_leaf_nodes = n_nodes - n_int;
        total_nodes += n_nodes;
        if (depth > 0) {
            leaves += n_leaf_nodes;
            total_depth += (uint64_t)depth * n_leaf_nodes;
            total_tris += n_slot;
            if (n_leaf_nodes) max_leaf_depth = depth;
        } else {
            root_internal = n_int != 0;
        }
        if (n_int) deepest_internal = (int)depth;
        BUILD_TRY(a->d_nodes.grow_keep(rec_base + n_int, rec_base));
        BUILD_TRY(a->d_slots.grow_keep(slot_base + n_slot + 4, slot_base));
        const int nxt = cur ^ 1;
        BUILD_TRY(d_bn[nxt].reserve(n_child));
        BUILD_TRY(d_list[nxt].reserve(n_child_list));
        BUILD_TRY(d_owner[nxt].reserve(n_child_list));
        hipLaunchKernelGGL(k_build_emit, grid(n_nodes), blk, 0, 0, d_bn[cur].p, n_nodes, d_S.p, d_P.p, (uint32_t)rec_base, (uint32_t)slot_base,
                           a->d_nodes.p, d_bn[nxt].p, d_cinfo.p);
        hipLaunchKernelGGL(k_build_scatter, grid(n_list), blk, 0, 0, d_bn[cur].p, d_list[cur].p, d_owner[cur].p, d_mask.p, d_S.p, d_P.p,
                           d_cinfo.p, n_list, (uint32_t)slot_base, a->d_tri_geom.p, a->d_slots.p, d_list[nxt].p, d_owner[nxt].p);
        BUILD_TRY(hipGetLastError());
        rec_base += n_int;
        slot_base += n_slot;
        if (n_child == 0) break;
        n_nodes = n_child;
        n_list = n_child_list;
        cur = nxt;
    }
    BUILD_TRY(hipMemcpy(h_tot, d_flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));  // also waits for the last pass
    const uint32_t max_leaf = h_tot[1];
    cleanup();
#undef BUILD_TRY
#undef BUILD_RC

    a->n_rec = rec_base;
    a->n_slots = slot_base;
    a->n_tree_nodes = total_nodes;
    a->root_info[0] = 0;
    if (root_internal) {
        a->root_info[1] = kChildInternal;
        a->max_stack = 8u * (uint32_t)deepest_internal + 8u;  // see flatten_octree
    } else {
        a->root_info[1] = n;  // root leaf: every triangle is a candidate (octree.cc:56-63, 186-191)
        a->max_stack = 0;
    }
    TreeShape& s = a->shape;  // octree.cc:226-266
    s.nodes = total_nodes;
    s.internal = rec_base;
    s.leaves = leaves;
    s.leaf_refs = slot_base;
    s.max_depth = max_leaf_depth;
    s.max_tris_in_leaf = max_leaf;
    s.avg_leaf_depth = (float)(long long)total_depth / (float)leaves;
    s.avg_tris_in_leaf = (float)(long long)total_tris / (float)leaves;
    a->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return check_tree_limits(a);
}

}  // namespace

extern "C" {

int rt_ads_build(rt_ads* a, const float* positions, uint64_t n) {
    if (!a || (!positions && n)) return fail(RT_ERR_INVALID, "null argument");
    if (n > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "more than 2^31-1 triangles");
    if (a->async_in_flight > 0) return fail(RT_ERR_INVALID, "%d frame(s) still in flight: rt_render_wait them before rebuilding", a->async_in_flight);
    if (int rc = use_device(a)) return rc;
    a->built = false;
    a->shading_set = false;
    a->n_tri = n;
    a->positions.assign(positions, positions + n * 9);
    int builder = a->builder;
    if (const char* e = getenv("RT_BUILDER")) builder = !strcmp(e, "host") ? RT_BUILDER_HOST : (!strcmp(e, "device") ? RT_BUILDER_DEVICE : builder);
    // small scenes build in microseconds on the host and get the small-tree form; large ones are built on the device
    a->device_built = builder == RT_BUILDER_DEVICE || (builder == RT_BUILDER_AUTO && n >= kDeviceBuildMinTriangles);
    if (int rc = a->device_built ? device_build(a) : host_build(a)) return rc;
    if (const char* e = getenv("RT_DISABLE_SMALL_TREE")) a->use_flat = !(e[0] == '1');
    a->traversal = 0;
    if (const char* e = getenv("RT_TRAVERSAL")) a->traversal = !strcmp(e, "wave") ? 1 : (!strcmp(e, "lane") ? 2 : (!strcmp(e, "pool") ? 3 : 0));
    a->pool_target_waves = 8192;
    a->pool_min_rays = 1;
    if (const char* e = getenv("RT_POOL_WAVES")) a->pool_target_waves = (uint32_t)atoi(e);
    a->prune_min_slots = kWaveFormMaxSlots;
    if (const char* e = getenv("RT_PRUNE_MIN_SLOTS")) a->prune_min_slots = (uint64_t)atoll(e);
    a->tail_max_rays = kTailMaxRays;
    if (const char* e = getenv("RT_TAIL_MAX_RAYS")) a->tail_max_rays = (uint32_t)atoi(e);
    a->tail_target_waves = 1024;
    if (const char* e = getenv("RT_TAIL_WAVES")) a->tail_target_waves = (uint32_t)atoi(e);
    a->lane_target_waves = 2048;
    if (const char* e = getenv("RT_LANE_WAVES")) a->lane_target_waves = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_POOL_MIN_RAYS")) a->pool_min_rays = (uint32_t)atoi(e);
    {
        double m1 = 0.0, m2 = 0.0;
        bool finite = true;
        for (uint64_t i = 0; i < n; ++i) {
            const float* p = positions + i * 9;
            double a1 = 0.0, a2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double x1 = (double)p[3 + k] - (double)p[k], x2 = (double)p[6 + k] - (double)p[k];
                a1 += x1 * x1;
                a2 += x2 * x2;
            }
            finite = finite && std::isfinite(a1) && std::isfinite(a2);
            m1 = std::max(m1, a1);
            m2 = std::max(m2, a2);
        }
        // |d| = |dot(e1, cross(s, e2))| <= |e1| |e2| |s|; a factor 2^10 covers |s| = 1 + a few ulp and every rounding on the way
        a->fast_tri = finite && std::sqrt(m1) * std::sqrt(m2) * 1024.0 < (double)kRcpFastMax;
        if (const char* e = getenv("RT_DISABLE_FAST_TRI")) a->fast_tri = a->fast_tri && !(e[0] == '1');
    }
    a->use_clusters = true;
    if (const char* e = getenv("RT_DISABLE_CLUSTERS")) a->use_clusters = !(e[0] == '1');
    a->clip = kClipDefault;
    if (const char* e = getenv("RT_CLIP")) a->clip = e[0] == '1';
    if (int rc = make_cluster_form(a)) return rc;
    // the slots' geometry alone, for the forms that read it per lane
    HIP_TRY(a->d_slot_geom.reserve((size_t)std::max<uint64_t>(a->n_slots, 1) * 9));
    if (a->n_slots) {
        hipLaunchKernelGGL(k_pack_slot_geometry, dim3((unsigned)((a->n_slots + 255) / 256)), dim3(256), 0, 0, a->d_slots.p, (uint32_t)a->n_slots,
                           a->d_slot_geom.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(0));
    }
    a->built = true;
    return RT_OK;
}

int rt_ads_tree_forms(const rt_ads* a, uint64_t* small_tree_triangles, uint32_t* clusters, int* fast_any_hit) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (small_tree_triangles) *small_tree_triangles = a->use_flat ? a->n_flat_tris : 0;
    if (clusters) *clusters = a->use_clusters ? a->n_clusters : 0u;
    if (fast_any_hit) *fast_any_hit = a->fast_tri ? 1 : 0;
    return RT_OK;
}

int rt_ads_light_forms(const rt_ads* a, uint32_t* lights_with_forms, uint64_t* pairs, uint64_t* pairs_left_out) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (!a->shading_set) return fail(RT_ERR_NOT_BUILT, "rt_scene_set_shading has not been called");
    const bool on = a->light_cull && a->light_forms_ready;
    if (lights_with_forms) *lights_with_forms = on ? a->n_lights : 0u;
    if (pairs) *pairs = on ? a->light_pairs : 0u;
    if (pairs_left_out) *pairs_left_out = on ? a->light_pairs_culled : 0u;
    return RT_OK;
}

int rt_ads_hit_lists(const rt_ads* a, uint64_t* pairs, uint64_t* pairs_left_out) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (!a->shading_set) return fail(RT_ERR_NOT_BUILT, "rt_scene_set_shading has not been called");
    const bool on = a->light_cull && a->light_forms_ready && a->hit_cull;
    if (pairs) *pairs = on ? a->hit_pairs : 0u;
    if (pairs_left_out) *pairs_left_out = on ? a->hit_pairs_culled : 0u;
    return RT_OK;
}

int rt_ads_export_tree(const rt_ads* a, void* node_records, void* leaf_slots, uint32_t* root_info2, uint64_t* n_records,
                       uint64_t* n_slots, uint32_t* max_stack, int* built_on_device) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (int rc = use_device(a)) return rc;
    if (n_records) *n_records = a->n_rec;
    if (n_slots) *n_slots = a->n_slots;
    if (max_stack) *max_stack = a->max_stack;
    if (built_on_device) *built_on_device = a->device_built ? 1 : 0;
    if (root_info2) {
        root_info2[0] = a->root_info[0];
        root_info2[1] = a->empty ? 0u : a->root_info[1];
    }
    if (node_records && a->n_rec) HIP_TRY(hipMemcpy(node_records, a->d_nodes.p, a->n_rec * sizeof(NodeRec), hipMemcpyDeviceToHost));
    if (leaf_slots && a->n_slots) HIP_TRY(hipMemcpy(leaf_slots, a->d_slots.p, a->n_slots * sizeof(LeafTri), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_ads_tree_stats(const rt_ads* a, rt_tree_stats* out) {
    if (!a || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    memset(out, 0, sizeof *out);
    out->max_depth = a->shape.max_depth;
    out->nodes = a->shape.nodes;
    out->leaves = a->shape.leaves;
    out->max_triangles_in_leaf = a->shape.max_tris_in_leaf;
    out->avg_leaf_depth = a->shape.avg_leaf_depth;
    out->avg_triangles_in_leaf = a->shape.avg_tris_in_leaf;
    out->leaf_refs = a->shape.leaf_refs;
    out->internal_nodes = a->shape.internal;
    out->build_seconds = a->build_seconds;
    out->device_bytes = a->n_rec * sizeof(NodeRec) + a->n_slots * (sizeof(LeafTri) + 9 * sizeof(float)) + a->n_tri * sizeof(LeafTri);
    return RT_OK;
}

int rt_ads_query_stats(const rt_ads* a, rt_query_stats* out) {
    if (!a || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = a->q;
    return RT_OK;
}
int rt_ads_reset_query_stats(rt_ads* a) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    a->q = rt_query_stats{};
    return RT_OK;
}
int rt_ads_collect_search_stats(rt_ads* a, int on) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    a->collect_stats = on != 0;
    return RT_OK;
}

int rt_ads_search(rt_ads* a, const float* rays, uint64_t n, uint64_t* out_offsets, uint32_t* out_tri, uint64_t cap, uint64_t* out_total) {
    if (!a || (!rays && n) || !out_offsets || !out_total) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (int rc = use_device(a)) return rc;
    out_offsets[0] = 0;
    *out_total = 0;
    if (n == 0) return RT_OK;
    if (n > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "too many rays in one call");
    // A ray can enqueue every node once; chunk the batch so that the queues fit a bounded workspace.  The workspace
    // belongs to the handle and only grows: a renderer that calls Search ray by ray allocates on its first call only.
    const uint64_t queue_cap = std::max<uint64_t>(a->n_tree_nodes, 1);
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / (queue_cap * 8)));
    SearchWork& ws = a->search;
    const DevOctree T = dev_tree(a);
    const auto t_begin = std::chrono::steady_clock::now();
    HIP_TRY(ws.rays.reserve(chunk * 6));
    HIP_TRY(ws.queue.reserve(chunk * queue_cap * 2));
    HIP_TRY(ws.stats.reserve(2));
    if (!ws.h_stats) HIP_TRY(hipHostMalloc((void**)&ws.h_stats, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ws.stats.p, 0, 2 * sizeof(unsigned long long), 0));
    uint64_t returned = 0;

    // Small batches (the drop-in Ads::Search sends one ray per call): one launch writes every ray's list into a region
    // of its own together with the list's length, one copy brings both back.  A list longer than its region falls
    // through to the exact two-pass form below.
    constexpr uint64_t kRegion = 1024, kSmallBatch = 64;
    bool done = false;
    if (n <= std::min<uint64_t>(kSmallBatch, chunk)) {  // (the workspace holds the rays and queues of `chunk` rays)
        const uint32_t m = (uint32_t)n;
        const size_t words = (size_t)m * 2 + (size_t)m * kRegion;  // m 64-bit counts, then m regions
        HIP_TRY(ws.out.reserve(words));
        ws.host.resize(words);
        HIP_TRY(hipMemcpyAsync(ws.rays.p, rays, (size_t)m * 6 * sizeof(float), hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(k_search<true>, dim3((m + 63) / 64), dim3(64), 0, 0, T, ws.rays.p, m, ws.queue.p, (uint32_t)queue_cap,
                           (uint64_t*)ws.out.p, (const uint64_t*)nullptr, ws.out.p + (size_t)m * 2, (uint64_t)0, kRegion, ws.stats.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ws.host.data(), ws.out.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, 0));
        HIP_TRY(hipMemcpyAsync(ws.h_stats, ws.stats.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, 0));
        HIP_TRY(hipStreamSynchronize(0));
        const uint64_t* counts = reinterpret_cast<const uint64_t*>(ws.host.data());
        bool fits = true;
        uint64_t total = 0;
        for (uint32_t i = 0; i < m; ++i) {
            if (counts[i] == ~0ull) return fail(RT_ERR_CAPACITY, "search queue overflow");
            fits = fits && counts[i] <= kRegion;
            out_offsets[i] = total;
            total += counts[i];
        }
        out_offsets[m] = total;
        *out_total = total;
        if (total > cap || (!out_tri && total)) return fail(RT_ERR_CAPACITY, "candidate buffer too small: need %llu entries", (unsigned long long)total);
        if (fits) {
            for (uint32_t i = 0; i < m; ++i)
                if (counts[i]) memcpy(out_tri + out_offsets[i], ws.host.data() + (size_t)m * 2 + (size_t)i * kRegion, counts[i] * sizeof(uint32_t));
            returned = total;
            done = true;
        } else {
            HIP_TRY(hipMemsetAsync(ws.stats.p, 0, 2 * sizeof(unsigned long long), 0));  // the two-pass form counts again
        }
    }
    if (!done) {
        std::vector<uint64_t> counts(n);
        HIP_TRY(ws.counts.reserve(chunk));
        HIP_TRY(ws.offs.reserve(chunk));
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                uint64_t total = 0;
                for (uint64_t i = 0; i < n; ++i) {
                    out_offsets[i] = total;
                    total += counts[i];
                }
                out_offsets[n] = total;
                *out_total = total;
                if (total > cap || (!out_tri && total)) return fail(RT_ERR_CAPACITY, "candidate buffer too small: need %llu entries", (unsigned long long)total);
                returned = total;
            }
            for (uint64_t b = 0; b < n; b += chunk) {
                const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - b);
                HIP_TRY(hipMemcpy(ws.rays.p, rays + b * 6, (size_t)m * 6 * sizeof(float), hipMemcpyHostToDevice));
                const dim3 grid((m + 63) / 64), block(64);
                if (pass == 0) {
                    hipLaunchKernelGGL(k_search<false>, grid, block, 0, 0, T, ws.rays.p, m, ws.queue.p, (uint32_t)queue_cap, ws.counts.p,
                                       (const uint64_t*)nullptr, (uint32_t*)nullptr, (uint64_t)0, (uint64_t)0, ws.stats.p);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipMemcpy(counts.data() + b, ws.counts.p, (size_t)m * sizeof(uint64_t), hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < m; ++i)
                        if (counts[b + i] == ~0ull) return fail(RT_ERR_CAPACITY, "search queue overflow");
                } else {
                    std::vector<uint64_t> rel(m);
                    const uint64_t base = out_offsets[b];
                    for (uint32_t i = 0; i < m; ++i) rel[i] = out_offsets[b + i] - base;
                    const uint64_t chunk_total = out_offsets[b + m] - base;
                    if (chunk_total == 0) continue;
                    HIP_TRY(ws.out.reserve(chunk_total));
                    HIP_TRY(hipMemcpy(ws.offs.p, rel.data(), (size_t)m * sizeof(uint64_t), hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(k_search<true>, grid, block, 0, 0, T, ws.rays.p, m, ws.queue.p, (uint32_t)queue_cap, (uint64_t*)nullptr,
                                       ws.offs.p, ws.out.p, chunk_total, (uint64_t)0, (unsigned long long*)nullptr);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipMemcpy(out_tri + base, ws.out.p, chunk_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
                }
            }
        }
        HIP_TRY(hipMemcpy(ws.h_stats, ws.stats.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    // octree.cc:174,185-187,204
    a->q.search_calls += n;
    a->q.search_node_count += ws.h_stats[0];
    a->q.search_leaves_visited += ws.h_stats[1];
    a->q.search_return_count += returned;
    a->q.counted_rays += n;
    a->q.search_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return RT_OK;
}

int rt_ads_search_noalloc(rt_ads* a, const float* rays, uint64_t n, uint64_t* out_offsets, uint32_t* out_tri, uint64_t cap,
                          uint64_t* out_total, void* stream_) {
    if (!a || (!rays && n) || !out_offsets || !out_total || (!out_tri && cap)) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (n > RT_SEARCH_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_CAPACITY, "rt_ads_search_noalloc takes at most %u rays per call (%llu given)", RT_SEARCH_NOALLOC_MAX_RAYS, (unsigned long long)n);
    const uint64_t queue_cap = std::max<uint64_t>(a->n_tree_nodes, 1);
    if (n * queue_cap * 2 * sizeof(uint32_t) > (uint64_t)RT_SEARCH_NOALLOC_SCRATCH_BYTES)
        return fail(RT_ERR_CAPACITY, "rt_ads_search_noalloc: %llu rays x %llu octree nodes need %llu bytes of queue, the fixed scratch holds %u",
                    (unsigned long long)n, (unsigned long long)queue_cap, (unsigned long long)(n * queue_cap * 8), RT_SEARCH_NOALLOC_SCRATCH_BYTES);
    static_assert(RT_SEARCH_NOALLOC_MAX_RAYS <= kSearchScanMax, "one block scans the counts");
    if (int rc = use_device(a)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* ptr = a->noalloc_scratch;  // this device's scratch array
    uint64_t* d_counts = reinterpret_cast<uint64_t*>(ptr);
    uint32_t* d_queue = reinterpret_cast<uint32_t*>(ptr + kNoallocCountBytes);
    const DevOctree T = dev_tree(a);
    const uint32_t m = (uint32_t)n;
    // Every word of the scratch that is read has been written by this call before: the counts by the first kernel,
    // a queue entry by the ray that later reads it.  Nothing of an earlier call's contents is used.
    if (m) {
        hipLaunchKernelGGL(k_search<false>, dim3((m + 63) / 64), dim3(64), 0, stream, T, rays, m, d_queue, (uint32_t)queue_cap, d_counts,
                           (const uint64_t*)nullptr, (uint32_t*)nullptr, (uint64_t)0, (uint64_t)0, (unsigned long long*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_search_offsets, dim3(1), dim3(256), 0, stream, (const uint64_t*)d_counts, m, out_offsets, out_total);
    HIP_TRY(hipGetLastError());
    if (m) {
        hipLaunchKernelGGL(k_search<true>, dim3((m + 63) / 64), dim3(64), 0, stream, T, rays, m, d_queue, (uint32_t)queue_cap, (uint64_t*)nullptr,
                           (const uint64_t*)out_offsets, out_tri, cap, (uint64_t)0, (unsigned long long*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    // (no query counter is touched: the call is asynchronous and brings no totals back to the host)
    return RT_OK;
}

int rt_ads_trace_closest(rt_ads* a, const float* origins, const float* dirs, uint64_t n, int cull, int mem, float* out_t,
                         int32_t* out_tri, void* stream_) {
    if (!a || ((!origins || !dirs || !out_t || !out_tri) && n)) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (n > 0xffffffffull * 64) return fail(RT_ERR_UNSUPPORTED, "too many rays in one call");
    if (int rc = use_device(a)) return rc;
    if (n == 0) return RT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const float *d_o = origins, *d_d = dirs;
    float* d_t = out_t;
    int32_t* d_tri = out_tri;
    if (mem == RT_MEM_HOST) {
        const size_t b3 = n * 3 * sizeof(float), b1 = n * sizeof(float);
        HIP_TRY(a->d_io.reserve(2 * b3 + 2 * b1));
        uint8_t* p = a->d_io.p;
        HIP_TRY(hipMemcpyAsync(p, origins, b3, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(p + b3, dirs, b3, hipMemcpyHostToDevice, stream));
        d_o = (const float*)p;
        d_d = (const float*)(p + b3);
        d_t = (float*)(p + 2 * b3);
        d_tri = (int32_t*)(p + 2 * b3 + b1);
    } else if (mem != RT_MEM_DEVICE) {
        return fail(RT_ERR_INVALID, "bad mem argument");
    }
    const int mode = traversal_mode(a, false, false, n);
    const uint32_t sparse = pool_sparse(a, mode, n, 1);
    const dim3 block(256), grid((unsigned)((mode_threads(mode, n, sparse) + 255) / 256));
    if (mode == MODE_WAVE && closest_prunes(a, mode)) {
        if (cull) hipLaunchKernelGGL((k_trace_closest<true, MODE_WAVE_PRUNED>), grid, block, 0, stream, dev_tree(a), d_o, d_d, n, d_t, d_tri, closest_opts(a, mode, sparse));
        else hipLaunchKernelGGL((k_trace_closest<false, MODE_WAVE_PRUNED>), grid, block, 0, stream, dev_tree(a), d_o, d_d, n, d_t, d_tri, closest_opts(a, mode, sparse));
    } else {
        RT_LAUNCH2(k_trace_closest, cull != 0, mode, grid, block, lane_lds(a, mode, block.x, true), stream, dev_tree(a), d_o, d_d, n, d_t, d_tri,
                   closest_opts(a, mode, sparse));
    }
    HIP_TRY(hipGetLastError());
    if (a->collect_stats) {
        if (int rc = stats_begin(a, stream)) return rc;
        const dim3 sgrid((unsigned)((n + 255) / 256));
        if (cull) hipLaunchKernelGGL((k_stats_rays<false, true>), sgrid, block, 0, stream, dev_tree(a), d_o, d_d, n, a->d_qstats.p);
        else hipLaunchKernelGGL((k_stats_rays<false, false>), sgrid, block, 0, stream, dev_tree(a), d_o, d_d, n, a->d_qstats.p);
        HIP_TRY(hipGetLastError());
        if (int rc = stats_end(a, stream)) return rc;
    }
    if (mem == RT_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out_t, d_t, n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(out_tri, d_tri, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    a->q.search_calls += n;
    a->q.closest_rays += n;
    return RT_OK;
}

int rt_ads_trace_any(rt_ads* a, const float* points, const float* lights, uint64_t n, int cull, int mem, uint8_t* out, void* stream_) {
    if (!a || ((!points || !lights || !out) && n)) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (n > 0xffffffffull * 64) return fail(RT_ERR_UNSUPPORTED, "too many rays in one call");
    if (int rc = use_device(a)) return rc;
    if (n == 0) return RT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const float *d_p = points, *d_l = lights;
    uint8_t* d_out = out;
    if (mem == RT_MEM_HOST) {
        const size_t b3 = n * 3 * sizeof(float);
        HIP_TRY(a->d_io.reserve(2 * b3 + n));
        uint8_t* p = a->d_io.p;
        HIP_TRY(hipMemcpyAsync(p, points, b3, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(p + b3, lights, b3, hipMemcpyHostToDevice, stream));
        d_p = (const float*)p;
        d_l = (const float*)(p + b3);
        d_out = p + 2 * b3;
    } else if (mem != RT_MEM_DEVICE) {
        return fail(RT_ERR_INVALID, "bad mem argument");
    }
    const int mode = traversal_mode(a, false, true, n);
    const uint32_t sparse = pool_sparse(a, mode, n, 1);
    const uint32_t opts = launch_opts(a, sparse);
    const dim3 block(256), grid((unsigned)((mode_threads(mode, n, sparse) + 255) / 256));
    if (mode == MODE_CLUSTER || (mode == MODE_WAVE && a->fast_tri)) {
        const DevOctree T = dev_tree(a);
        if (mode == MODE_CLUSTER && a->fast_tri) {
            if (cull) hipLaunchKernelGGL((k_trace_any<true, MODE_CLUSTER, true>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
            else hipLaunchKernelGGL((k_trace_any<false, MODE_CLUSTER, true>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
        } else if (mode == MODE_CLUSTER) {
            if (cull) hipLaunchKernelGGL((k_trace_any<true, MODE_CLUSTER>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
            else hipLaunchKernelGGL((k_trace_any<false, MODE_CLUSTER>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
        } else {
            if (cull) hipLaunchKernelGGL((k_trace_any<true, MODE_WAVE, true>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
            else hipLaunchKernelGGL((k_trace_any<false, MODE_WAVE, true>), grid, block, 0, stream, T, d_p, d_l, n, d_out, opts);
        }
    } else {
        RT_LAUNCH2(k_trace_any, cull != 0, mode, grid, block, lane_lds(a, mode, block.x, false), stream, dev_tree(a), d_p, d_l, n, d_out, opts);
    }
    HIP_TRY(hipGetLastError());
    if (a->collect_stats) {
        if (int rc = stats_begin(a, stream)) return rc;
        const dim3 sgrid((unsigned)((n + 255) / 256));
        if (cull) hipLaunchKernelGGL((k_stats_rays<true, true>), sgrid, block, 0, stream, dev_tree(a), d_p, d_l, n, a->d_qstats.p);
        else hipLaunchKernelGGL((k_stats_rays<true, false>), sgrid, block, 0, stream, dev_tree(a), d_p, d_l, n, a->d_qstats.p);
        HIP_TRY(hipGetLastError());
        if (int rc = stats_end(a, stream)) return rc;
    }
    if (mem == RT_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    a->q.search_calls += n;
    a->q.any_rays += n;
    return RT_OK;
}

int rt_debug_triangle_tests(int device, const float* origins, const float* dirs, uint32_t n_rays, const float* positions, uint32_t n_tris, int cull,
                            float* out) {
    if (!origins || !dirs || !positions || !out || !n_rays || !n_tris) return fail(RT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    std::vector<LeafTri> g(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float* p = positions + (size_t)i * 9;
        LeafTri& s = g[i];
        s.ax = p[0]; s.ay = p[1]; s.az = p[2];
        s.e1x = p[3] - p[0]; s.e1y = p[4] - p[1]; s.e1z = p[5] - p[2];
        s.e2x = p[6] - p[0]; s.e2y = p[7] - p[1]; s.e2z = p[8] - p[2];
        s.tri = i;
        s.pad0 = s.pad1 = 0;
    }
    DevBuf<float> d_o, d_d, d_out;
    DevBuf<LeafTri> d_t;
    const size_t n_out = (size_t)3 * n_rays * n_tris;
    auto done = [&](int rc) {
        d_o.release(); d_d.release(); d_out.release(); d_t.release();
        return rc;
    };
    if (d_o.reserve((size_t)n_rays * 3) != hipSuccess || d_d.reserve((size_t)n_rays * 3) != hipSuccess || d_out.reserve(n_out) != hipSuccess ||
        d_t.reserve(n_tris) != hipSuccess)
        return done(fail(RT_ERR_HIP, "allocation failed"));
    if (hipMemcpy(d_o.p, origins, (size_t)n_rays * 12, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_d.p, dirs, (size_t)n_rays * 12, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_t.p, g.data(), (size_t)n_tris * sizeof(LeafTri), hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(RT_ERR_HIP, "copy failed"));
    const dim3 grid((n_rays + 255) / 256), block(256);
    if (cull) hipLaunchKernelGGL(k_debug_triangle_tests<true>, grid, block, 0, 0, d_o.p, d_d.p, n_rays, d_t.p, n_tris, d_out.p);
    else hipLaunchKernelGGL(k_debug_triangle_tests<false>, grid, block, 0, 0, d_o.p, d_d.p, n_rays, d_t.p, n_tris, d_out.p);
    if (hipGetLastError() != hipSuccess || hipMemcpy(out, d_out.p, n_out * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(RT_ERR_HIP, "kernel or copy failed"));
    return done(RT_OK);
}

int rt_scene_set_shading(rt_ads* a, const float* normals, const int32_t* material_ids, const float* materials, uint32_t n_mat) {
    if (!a || !material_ids || !materials) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (a->async_in_flight > 0) return fail(RT_ERR_INVALID, "%d frame(s) still in flight: rt_render_wait them first", a->async_in_flight);
    if (int rc = use_device(a)) return rc;
    const uint64_t n = a->n_tri;
    for (uint64_t i = 0; i < n; ++i)
        if (material_ids[i] < 0 || (uint32_t)material_ids[i] >= n_mat)
            return fail(RT_ERR_INVALID, "triangle %llu has material id %d outside [0,%u)", (unsigned long long)i, material_ids[i], n_mat);
    std::vector<MatDev> mats(std::max<uint32_t>(n_mat, 1));
    a->branching = false;      // some material spawns two secondary rays per hit
    a->branching_any = false;  // some material spawns a secondary ray at all
    for (uint32_t m = 0; m < n_mat; ++m) {
        const float* f = materials + (size_t)m * 18;
        MatDev& d = mats[m];
        for (int k = 0; k < 3; ++k) {
            d.diffuse[k] = f[3 + k];
            d.specular[k] = f[6 + k];
            d.transmittance[k] = f[9 + k];
            d.emission[k] = f[12 + k];
        }
        d.shininess = f[15];
        d.ior = f[16];
        d.has_specular = !eq_eps(mk(f[6], f[7], f[8]), mk(0, 0, 0));        // renderer.cc:179
        d.has_transmittance = !eq_eps(mk(f[9], f[10], f[11]), mk(0, 0, 0)); // renderer.cc:185
        if (d.has_specular && d.has_transmittance) a->branching = true;
        if (d.has_specular || d.has_transmittance) a->branching_any = true;
    }
    // light list: emissive triangles in Build order (obj_loader.cc:188-191)
    std::vector<uint32_t> light_tri;
    for (uint64_t i = 0; i < n; ++i) {
        const float* f = materials + (size_t)material_ids[i] * 18;
        if (f[12] != 0.0 || f[13] != 0.0 || f[14] != 0.0) light_tri.push_back((uint32_t)i);
    }
    std::vector<LightDev> lights(std::max<size_t>(light_tri.size(), 1));
    for (size_t l = 0; l < light_tri.size(); ++l) {
        const float* p = a->positions.data() + (size_t)light_tri[l] * 9;
        const f3 v[3] = {mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), mk(p[6], p[7], p[8])};
        LightDev& L = lights[l];
        const f3 e1 = v[1] - v[0], e2 = v[2] - v[0], nn = face_normal(v);
        L.a[0] = v[0].x; L.a[1] = v[0].y; L.a[2] = v[0].z;
        L.e1[0] = e1.x; L.e1[1] = e1.y; L.e1[2] = e1.z;
        L.e2[0] = e2.x; L.e2[1] = e2.y; L.e2[2] = e2.z;
        L.n[0] = nn.x; L.n[1] = nn.y; L.n[2] = nn.z;
        L.area = 0.5f * length(cross(v[1] - v[0], v[2] - v[1]));  // src/triangle.h:34-40
        const float* f = materials + (size_t)material_ids[light_tri[l]] * 18;
        L.emission[0] = f[12]; L.emission[1] = f[13]; L.emission[2] = f[14];
    }
    // Which shadow rays need no answer (shadow_ray_idle in rt_kernels.h): the rule multiplies zeros with the lights'
    // and materials' numbers, so those must be finite, and moderate enough that no product of them overflows --
    // I_l = emission * area * cos / (spt * dist^2 + 1e-9) times a diffuse colour stays below 1e30.
    std::vector<uint32_t> mat_dark(mats.size(), 0);
    {
        bool lights_fine = a->skip_unlit;
        double max_e = 0.0, max_area = 0.0, max_diffuse = 1.0;
        for (size_t l = 0; l < light_tri.size(); ++l) {
            const LightDev& L = lights[l];
            for (int k = 0; k < 3; ++k) {
                lights_fine = lights_fine && std::isfinite(L.n[k]) && std::isfinite(L.emission[k]) && std::isfinite(L.a[k]) &&
                              std::isfinite(L.e1[k]) && std::isfinite(L.e2[k]);
                max_e = std::max(max_e, (double)std::fabs(L.emission[k]));
            }
            lights_fine = lights_fine && std::isfinite(L.area);
            max_area = std::max(max_area, (double)L.area);
        }
        for (uint32_t m = 0; m < n_mat; ++m)
            for (int k = 0; k < 3; ++k)
                if (std::isfinite(mats[m].diffuse[k])) max_diffuse = std::max(max_diffuse, (double)std::fabs(mats[m].diffuse[k]));
        lights_fine = lights_fine && max_e * max_area * 1e9 * max_diffuse < 1e30;
        for (uint32_t m = 0; m < n_mat && lights_fine; ++m) {
            const MatDev& d = mats[m];
            bool fine = std::isfinite(d.shininess) && d.shininess >= 0.0f && d.shininess <= 1e6f;
            for (int k = 0; k < 3; ++k) fine = fine && std::isfinite(d.diffuse[k]) && std::isfinite(d.specular[k]) && std::isfinite(d.emission[k]);
            if (!fine) continue;
            mat_dark[m] = kDarkFinite;
            if (d.emission[0] == 0.0f && d.emission[1] == 0.0f && d.emission[2] == 0.0f) mat_dark[m] |= kDarkEmissionZero;
            if (d.specular[0] == 0.0f && d.specular[1] == 0.0f && d.specular[2] == 0.0f) mat_dark[m] |= kDarkSpecZero;
        }
    }
    std::vector<TriShade> ts(std::max<uint64_t>(n, 1));
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = a->positions.data() + i * 9;
        const f3 v[3] = {mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), mk(p[6], p[7], p[8])};
        const f3 nn = face_normal(v);
        TriShade& t = ts[i];
        t.nx = nn.x; t.ny = nn.y; t.nz = nn.z;
        t.mat = material_ids[i];
        t.light_eq = -1;
        t.spawn = (mats[material_ids[i]].has_specular ? 1u : 0u) + (mats[material_ids[i]].has_transmittance ? 1u : 0u);
        t.dark = (std::isfinite(nn.x) && std::isfinite(nn.y) && std::isfinite(nn.z)) ? mat_dark[material_ids[i]] : 0u;
        t.pad = 0;
        for (size_t l = 0; l < light_tri.size(); ++l) {
            const uint32_t lt = light_tri[l];
            if (tri_equal(a->positions.data() + (size_t)lt * 9, normals ? normals + (size_t)lt * 9 : nullptr, p, normals ? normals + i * 9 : nullptr)) {
                t.light_eq = (int32_t)l;
                break;
            }
        }
    }
    HIP_TRY(a->d_mats.reserve(mats.size()));
    HIP_TRY(hipMemcpy(a->d_mats.p, mats.data(), mats.size() * sizeof(MatDev), hipMemcpyHostToDevice));
    HIP_TRY(a->d_lights.reserve(lights.size()));
    HIP_TRY(hipMemcpy(a->d_lights.p, lights.data(), lights.size() * sizeof(LightDev), hipMemcpyHostToDevice));
    HIP_TRY(a->d_tri_shade.reserve(ts.size()));
    HIP_TRY(hipMemcpy(a->d_tri_shade.p, ts.data(), ts.size() * sizeof(TriShade), hipMemcpyHostToDevice));
    a->has_normals = normals != nullptr;
    if (normals && n) {
        HIP_TRY(a->d_vnorm.reserve(n * 9));
        HIP_TRY(hipMemcpy(a->d_vnorm.p, normals, n * 9 * sizeof(float), hipMemcpyHostToDevice));
    }
    a->n_lights = (uint32_t)light_tri.size();
    if (int rc = make_light_forms(a, light_tri)) return rc;
    a->shading_set = true;
    return RT_OK;
}

}  // extern "C"

namespace {

// Camera frame, src/Renderer/renderer.cc:82-96.  tanf runs on the host (same libm call as the reference).
int frame_params(const rt_ads* a, const rt_render_config* c, FrameParams& F, uint32_t& n0) {
    if (c->struct_size != sizeof(rt_render_config)) return fail(RT_ERR_INVALID, "rt_render_config.struct_size mismatch");
    if (c->width < 1 || c->height < 1) return fail(RT_ERR_INVALID, "image size must be positive");
    if (c->samples_per_pixel < 1) return fail(RT_ERR_INVALID, "samples_per_pixel must be >= 1 (renderer.cc:43 would wrap)");
    if (c->render_type < 0 || c->render_type > 3) return fail(RT_ERR_INVALID, "Invalid render type");
    if (c->shading_type < 0 || c->shading_type > 1) return fail(RT_ERR_INVALID, "Invalid shading type");
    if (c->shading_type == RT_SHADING_SMOOTH && !a->has_normals) return fail(RT_ERR_INVALID, "smooth shading needs vertex normals");
    if (c->band_world < 1 || c->band_rank < 0 || c->band_rank >= c->band_world) return fail(RT_ERR_INVALID, "bad band_rank/band_world");
    if (c->max_depth < 0 || c->max_depth > 30) return fail(RT_ERR_INVALID, "renderer max_depth must be in [0,30]");
    if ((uint64_t)c->width * (uint64_t)c->height > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "image too large");
    const f3 dir = mk(c->cam_dir[0], c->cam_dir[1], c->cam_dir[2]), up = mk(c->cam_up[0], c->cam_up[1], c->cam_up[2]);
    const float t = 1.0f;
    const f3 b = cross(dir, up);
    const float gw = 2 * t * tanf(c->fov / 2.0f);
    const float gh = c->aspect_mode == RT_ASPECT_TRUE ? gw * ((float)c->height / (float)c->width) : gw * (float)(c->height / c->width);
    F.qw = b * (gw / (float)(c->width - 1));
    F.qh = up * (gh / (float)(c->height - 1));
    F.p00 = ((dir * t) - (b * (gw / 2))) + (up * (gh / 2));
    F.cam_pos = mk(c->cam_pos[0], c->cam_pos[1], c->cam_pos[2]);
    F.bg = mk(c->background[0], c->background[1], c->background[2]);
    F.width = c->width;
    F.height = c->height;
    F.tiles_x = (c->width + 7) / 8;
    F.band_rank = c->band_rank;
    F.band_world = c->band_world;
    F.render_type = c->render_type;
    F.shading_type = c->shading_type;
    F.cull = c->cull_backfaces;
    F.max_depth = c->max_depth;
    F.spt = c->samples_per_triangle;
    F.max_distance = c->max_distance;
    F.seed = c->seed;
    F.spp_sample = 0;
    F.spp = c->samples_per_pixel;
    const int bands = (c->height + 7) / 8;
    const int local_bands = bands > c->band_rank ? (bands - c->band_rank + c->band_world - 1) / c->band_world : 0;
    const uint64_t nodes = (uint64_t)local_bands * (uint64_t)F.tiles_x * 64ull;
    if (nodes > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "image too large");
    n0 = (uint32_t)nodes;
    return RT_OK;
}

SceneDev scene_dev(const rt_ads* a) {
    SceneDev S;
    S.tree = dev_tree(a);
    S.tri_geom = a->d_tri_geom.p;
    S.tri_shade = a->d_tri_shade.p;
    S.tri_vnorm = a->d_vnorm.p;
    S.mats = a->d_mats.p;
    S.lights = a->d_lights.p;
    S.n_lights = a->n_lights;
    S.light_forms = (a->light_cull && a->light_forms_ready) ? a->d_light_forms.p : nullptr;
    S.lights_merged = a->lights_merged ? 1u : 0u;
    return S;
}

int ensure_events(FrameCtx& x) {
    if (x.events_ready) return RT_OK;
    for (auto& e : x.ev) HIP_TRY(hipEventCreate(&e));
    x.events_ready = true;
    return RT_OK;
}

constexpr int kMaxLevels = 32;  // renderer max_depth <= 30, plus the camera level

// Waits until the closest-hit pass launched with sequence number `seq` has published its level's totals in `slot`.
// The host polls pinned memory; the stream is not synchronised, so the device keeps running what was queued behind
// that pass.  A stream that has drained (or failed) without the flag appearing is an error, not a reason to spin on.
int wait_slot(const LevelSlot* slot, uint32_t seq, hipStream_t stream) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; ++spins) {
        if (__atomic_load_n(&slot->flag, __ATOMIC_ACQUIRE) == seq) return RT_OK;
        __builtin_ia32_pause();
        if ((spins & 0xfff) != 0) continue;
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (spins == 0x1000 || waited > 0.05) {
            const hipError_t e = hipStreamQuery(stream);  // also makes sure the queued work has been submitted
            if (e == hipSuccess) {
                if (__atomic_load_n(&slot->flag, __ATOMIC_ACQUIRE) == seq) return RT_OK;
                return fail(RT_ERR_HIP, "internal: the stream drained but the level totals were never published");
            }
            if (e != hipErrorNotReady) return fail(RT_ERR_HIP, "stream failed while a level was in flight: %s", hipGetErrorString(e));
        }
        if (waited > 300.0) return fail(RT_ERR_HIP, "timed out waiting for a level's totals");
    }
}

int ensure_frame_mailboxes(FrameCtx& x) {
    if (!x.h_slots) {
        HIP_TRY(hipHostMalloc((void**)&x.h_slots, kMaxLevels * sizeof(LevelSlot), hipHostMallocMapped | hipHostMallocCoherent));
        memset(x.h_slots, 0, kMaxLevels * sizeof(LevelSlot));
    }
    if (!x.h_totals) {
        HIP_TRY(hipHostMalloc((void**)&x.h_totals, sizeof(FrameTotals), hipHostMallocMapped | hipHostMallocCoherent));
        memset(x.h_totals, 0, sizeof(FrameTotals));
    }
    HIP_TRY(x.d_totals.reserve(1));
    return RT_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// One sub-frame of rt_render: the bands c->band_rank of c->band_world, rendered into the device frame buffer d_out with
// the level buffers of `x`, on `stream`; returns after the stream has drained.  The instrumented pass (collect_stats) is
// begun and ended by the caller.
// n_sub: the sub-frames in flight on this GPU (this one included).
int render_bands(rt_ads* a, FrameCtx& x, const rt_render_config* c, float* d_out, hipStream_t stream, int n_sub, rt_render_stats& st_out) {
    if (int rc = use_device(a)) return rc;  // (a worker thread starts on device 0)
    FrameParams F;
    uint32_t n0 = 0;
    if (int rc = frame_params(a, c, F, n0)) return rc;
    // the per-(light, hit triangle) lists of the small-tree form assume a camera within 4 D of the scene (make_light_forms)
    bool hit_lists_ok = false;
    if (a->scene_diag > 0.0) {
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) d2 += ((double)c->cam_pos[k] - a->scene_mid[k]) * ((double)c->cam_pos[k] - a->scene_mid[k]);
        hit_lists_ok = d2 <= 16.0 * a->scene_diag * a->scene_diag;  // (false for a NaN)
    }
    if (int rc = ensure_events(x)) return rc;
    const SceneDev S = scene_dev(a);
    const uint32_t max_levels = (uint32_t)c->max_depth + 1;
    if (x.levels.size() < max_levels) x.levels.resize(max_levels);
    HIP_TRY(x.d_counters.reserve((size_t)max_levels * CNT_WORDS));
    if (c->samples_per_pixel > 1) { HIP_TRY(x.d_accum.reserve((size_t)n0 * 3)); }

    rt_render_stats st{};
    uint64_t valid_pixels = 0;  // of this call's bands, per sample
    for (int b = c->band_rank; b < (c->height + 7) / 8; b += c->band_world) valid_pixels += (uint64_t)std::min(8, c->height - b * 8) * (uint64_t)c->width;
    // Per-class kernel times: ONE event per boundary between timed kernels -- the event that ends a kernel's interval
    // starts the next one's (an event costs the stream about 5 us; two per boundary were 2-3 % of a 2.5 ms frame).
    // mark(cls) closes the open interval and opens one of class cls; CLS_OTHER only closes.  At most kMaxTimed
    // intervals per class are taken per call.
    enum { CLS_OTHER = 0, CLS_CLOSEST = 1, CLS_SHADOW = 2, CLS_SHADE = 3, CLS_TAIL = 4 };
    constexpr int kMaxMarks = 6 * kMaxTimed;
    hipEvent_t* ev_total = x.ev;
    hipEvent_t* ev_mark = x.ev + 2;
    int n_marks = 0, n_cls[5] = {0, 0, 0, 0, 0};
    uint8_t mark_cls[kMaxMarks];
    bool last_timed = false;
    auto mark = [&](int cls) -> int {
        const bool want = cls != CLS_OTHER && n_cls[cls] < kMaxTimed;
        last_timed = want;
        if (!want && (n_marks == 0 || mark_cls[n_marks - 1] == CLS_OTHER)) return RT_OK;  // no interval is open
        if (n_marks >= kMaxMarks) {
            last_timed = false;
            return RT_OK;
        }
        HIP_TRY(hipEventRecord(ev_mark[n_marks], stream));
        mark_cls[n_marks++] = (uint8_t)(want ? cls : CLS_OTHER);
        if (want) ++n_cls[cls];
        return RT_OK;
    };
    if (int rc = ensure_frame_mailboxes(x)) return rc;
    HIP_TRY(hipEventRecord(ev_total[0], stream));
    const dim3 block(256);
    // The frame runs without a host synchronisation.  Level sizes stay on the device: the closest-hit pass of level L
    // reads its ray count from level L-1's counters and its last block publishes (hits, upper bound of the next level's
    // rays) into pinned host memory; the host waits for that flag only after it has queued the level's any-hit pass, so
    // the device never idles, and sizes the shading pass and the next level's buffers with it.  Launch grids of a level
    // come from the published upper bound; surplus threads leave at once.
    const uint32_t seq_base = x.slot_seq;
    x.slot_seq = seq_base + (uint32_t)c->samples_per_pixel + 1u;  // the next frame starts beyond every sequence number of this one
    hipLaunchKernelGGL(k_frame_begin, dim3(1), block, 0, stream, x.d_counters.p, max_levels, x.h_slots, seq_base, x.d_totals.p);
    HIP_TRY(hipGetLastError());
    x.h_totals->secondary_rays = x.h_totals->shadow_rays = x.h_totals->deepest_level = x.h_totals->idle_shadow_rays = 0;
    uint32_t level_seq[kMaxLevels];
    for (auto& q : level_seq) q = seq_base;
    const bool lit = (c->render_type == RT_RENDER_PHONG || c->render_type == RT_RENDER_BLINN_PHONG) && a->n_lights > 0 && c->samples_per_triangle > 0;
    const uint32_t words = (uint32_t)((std::max(c->samples_per_triangle, 1) + 31) / 32);
    uint32_t level_n[kMaxLevels] = {};  // launch bound of each level of the current sample (resolve pass)
    // the fused tail: chains only (no material reflects AND refracts), the per-lane form, all shadow rays of a hit within
    // one wave; the instrumented pass counts level by level
    uint32_t pshift_all = 0;
    while ((1u << pshift_all) < (uint32_t)std::max(c->samples_per_triangle, 1) && pshift_all < 30) ++pshift_all;
    const bool tail_ok = !a->collect_stats && !a->branching && a->tail_max_rays > 0 && traversal_mode(a, false, false, 1) == MODE_LANE &&
                         (!lit || ((uint64_t)a->n_lights << pshift_all) <= 64);
    for (int s = 0; s < c->samples_per_pixel && n0 > 0; ++s) {
        F.spp_sample = (uint32_t)s;
        uint32_t cap = n0, levels = 0;  // cap: upper bound of the level's ray count (exact for level 0)
        for (uint32_t L = 0; L < max_levels && cap > 0; ++L) {
            Level& lv = x.levels[L];
            uint32_t* cnt = x.d_counters.p + (size_t)L * CNT_WORDS;
            // A small level of a scene whose paths are chains: the rest of the recursion runs as one launch (k_tail).
            // (what decides is the load of the whole GPU: concurrent sub-frames share the limit)
            if (L > 0 && tail_ok && (uint64_t)cap * (uint64_t)n_sub <= a->tail_max_rays) {
                uint32_t pshift_t = 0, gshift_t = 0;
                while ((1u << pshift_t) < (uint32_t)std::max(c->samples_per_triangle, 1)) ++pshift_t;
                while (lit && (1u << gshift_t) < (a->n_lights << pshift_t)) ++gshift_t;
                uint32_t sparse_t = 0;  // (fewer paths per wave while the launch stays below tail_target_waves waves)
                while ((64u >> (sparse_t + 1)) >= (1u << gshift_t) && ((((uint64_t)cap << gshift_t) >> (6 - sparse_t)) < a->tail_target_waves)) ++sparse_t;
                const dim3 grid_t((uint32_t)(mode_threads(MODE_LANE, (uint64_t)cap << gshift_t, sparse_t) / 64u));
                HIP_TRY(x.d_tail.reserve((size_t)(max_levels - L) * cap * 6));
                if (int rc = mark(CLS_TAIL)) return rc;
                const size_t lds_t = lane_lds(a, MODE_LANE, 64, true);
                const uint32_t opts_t = launch_opts(a, sparse_t);
#define RT_TAIL(CULL, FAST) hipLaunchKernelGGL((k_tail<CULL, FAST>), grid_t, dim3(64), lds_t, stream, S, F, L, lv.rays.p, cap, x.levels[L - 1].child.p, \
                                               x.d_tail.p, pshift_t, gshift_t, cnt, opts_t)
                if (F.cull && a->fast_tri) RT_TAIL(true, true);
                else if (F.cull) RT_TAIL(true, false);
                else if (a->fast_tri) RT_TAIL(false, true);
                else RT_TAIL(false, false);
#undef RT_TAIL
                HIP_TRY(hipGetLastError());
                st.launches_tail++;
                break;
            }
            HIP_TRY(lv.reserve(cap, L > 0));
            lv.n = cap;
            level_n[L] = cap;
            const uint32_t* n_dev = L > 0 ? cnt - CNT_WORDS + CNT_NEXT : nullptr;
            const LevelSlot* slot = x.h_slots + L;
            const uint32_t seq = ++level_seq[L];  // CNT_SEQ of the level after this launch
            const int mode_c = traversal_mode(a, L == 0, false, cap);
            const uint32_t sparse_c = pool_sparse(a, mode_c, cap, 1);
            const dim3 block_c(closest_block(a, mode_c)), grid_c((uint32_t)((mode_threads(mode_c, cap, sparse_c) + block_c.x - 1) / block_c.x));
            if (int rc = mark(CLS_CLOSEST)) return rc;
            const bool timed_c = last_timed;
            launch_closest(F.cull != 0, mode_c, closest_prunes(a, mode_c), grid_c, block_c, lane_lds(a, mode_c, block_c.x, true), stream, S, F, L, lv.rays.p, cap, lv.hit_t.p,
                           lv.hit_tri.p, lv.out.p, lv.work.p, cnt, closest_opts(a, mode_c, sparse_c));
            HIP_TRY(hipGetLastError());
            st.launches_closest++;
            if (a->collect_stats) {
                if (int rc = mark(CLS_OTHER)) return rc;
                const dim3 sgrid((cap + 255) / 256);
                if (F.cull) hipLaunchKernelGGL(k_stats_closest<true>, sgrid, block, 0, stream, S, F, L, lv.rays.p, cap, n_dev, a->d_qstats.p);
                else hipLaunchKernelGGL(k_stats_closest<false>, sgrid, block, 0, stream, S, F, L, lv.rays.p, cap, n_dev, a->d_qstats.p);
                HIP_TRY(hipGetLastError());
            }
            // The any-hit pass.  Its launch is sized by the level's hits when the host knows them, else by the level's
            // bound (surplus threads leave at once, at about 2.3 us per million of them).  Level 0 (nearly every camera
            // ray hits) and small levels are launched at once with the bound, so that the device never waits for the
            // host; for a large deeper level the host first waits for the hit count -- a bubble of a few microseconds
            // instead of an over-sized launch.
            bool timed_a = false;
            uint32_t pshift = 0, gshift = 0;
            while ((1u << pshift) < (uint32_t)std::max(c->samples_per_triangle, 1) && pshift < 30) ++pshift;
            auto launch_any_hit = [&](uint32_t n_hits) -> int {
                HIP_TRY(lv.vis.reserve((size_t)n_hits * a->n_lights * words));
                if (int rc = mark(CLS_SHADOW)) return rc;
                timed_a = last_timed;
                // lanes per hit in the any-hit pass: the (light, sample) pairs of a hit, samples padded to a power of
                // two per light, side by side in at most 64 lanes.  Per-lane traversal gains from rays of one origin
                // side by side (all lights of a hit in one wave); the wave-together forms measured faster with one
                // light per pass (a narrower bundle of rays).
                const int mode_a = traversal_mode(a, false, true, (uint64_t)n_hits * a->n_lights * (uint64_t)c->samples_per_triangle);
                gshift = 0;
                const bool wave_a = mode_a == MODE_WAVE || mode_a == MODE_CLUSTER;
                // (with per-light forms of the tree a per-lane pass, too, serves one light at a time)
                // ... unless the launch is small: then the lights of a hit stay side by side (a shorter chain of walks per
                // wave) and every lane walks its own light's tree
                // (lights with merged forms walk one tree: side by side, like without the forms)
                const bool light_trees = mode_a == MODE_LANE && F.cull && a->light_cull && a->light_forms_lane && !a->lights_merged;
                const bool light_passes = wave_a || (light_trees && (uint64_t)n_hits * a->n_lights * (uint64_t)c->samples_per_triangle > a->lane_side_max_rays);
                while (!light_passes && (1ull << gshift) < ((uint64_t)a->n_lights << pshift) && gshift < 6) ++gshift;
                if (gshift < pshift) gshift = pshift < 6 ? pshift : 6;  // a group never exceeds the wave
                const dim3 block_a(wave_a ? 256 : 64);
                const uint32_t sparse_a = pool_sparse(a, mode_a, (uint64_t)n_hits << gshift, 1u << gshift);
                const dim3 grid_a((uint32_t)((mode_threads(mode_a, (uint64_t)n_hits << gshift, sparse_a) + block_a.x - 1) / block_a.x));
                launch_shadow(F.cull != 0, mode_a, a->fast_tri, light_trees && gshift > pshift, grid_a, block_a, lane_lds(a, mode_a, block_a.x, false), stream, S, F, L, lv.rays.p, cap,
                              lv.hit_t.p, lv.hit_tri.p, lv.work.p, lv.vis.p, words, pshift, gshift, cnt, launch_opts(a, sparse_a) | (hit_lists_ok ? kOptHitLists : 0u));
                HIP_TRY(hipGetLastError());
                st.launches_shadow++;
                if (a->collect_stats) {
                    if (int rc = mark(CLS_OTHER)) return rc;
                    const uint64_t threads = (uint64_t)n_hits * a->n_lights * (uint64_t)c->samples_per_triangle;
                    if (threads > 0x7fffffffull * 256ull) return fail(RT_ERR_UNSUPPORTED, "level too large for the instrumented pass");
                    const dim3 sgrid((uint32_t)((threads + 255) / 256));
                    if (F.cull) hipLaunchKernelGGL(k_stats_shadow<true>, sgrid, block, 0, stream, S, F, L, lv.rays.p, cap, lv.hit_t.p, lv.hit_tri.p, lv.work.p, cnt, a->d_qstats.p);
                    else hipLaunchKernelGGL(k_stats_shadow<false>, sgrid, block, 0, stream, S, F, L, lv.rays.p, cap, lv.hit_t.p, lv.hit_tri.p, lv.work.p, cnt, a->d_qstats.p);
                    HIP_TRY(hipGetLastError());
                }
                return RT_OK;
            };
            const bool any_hit_first = lit && (L == 0 || ((uint64_t)cap << std::min(pshift, 6u)) * a->n_lights < (3ull << 20));
            if (any_hit_first) {
                if (int rc = launch_any_hit(cap)) return rc;
            } else {
                if (int rc = mark(CLS_OTHER)) return rc;  // the host's wait below is not the closest-hit pass's time
            }
            // the level's totals, published by the closest-hit pass
            if (int rc = wait_slot(slot, seq, stream)) return rc;
            const uint32_t n_level = slot->n, n_work = slot->n_work, spawn = slot->spawn;
            const uint64_t level_shadow = (uint64_t)slot->shadow_lo | ((uint64_t)slot->shadow_hi << 32);
            if (lit && !any_hit_first && n_work > 0) {
                if (int rc = launch_any_hit(n_work)) return rc;
            }
            if (n_level > cap) return fail(RT_ERR_HIP, "internal: recursion queue overflow (%u > %u)", n_level, cap);
            if (timed_c) st.timed_closest_rays += L == 0 ? valid_pixels : n_level;
            if (timed_a) st.timed_shadow_rays += level_shadow;
            if (n_level > 0) levels = L + 1;
            if (n_work == 0) {  // nothing to shade, nothing spawned
                if (int rc = mark(CLS_OTHER)) return rc;
                break;
            }
            RayRec* next_rays = nullptr;
            uint32_t next_cap = 0;
            if (L + 1 < max_levels && spawn > 0) {
                HIP_TRY(x.levels[L + 1].reserve(spawn, true));
                next_rays = x.levels[L + 1].rays.p;
                next_cap = spawn;
            }
            const dim3 grid_s((n_work + 255) / 256);
            if (int rc = mark(CLS_SHADE)) return rc;
            hipLaunchKernelGGL(k_shade, grid_s, block, 0, stream, S, F, L, lv.rays.p, cap, lv.hit_t.p, lv.hit_tri.p, lv.work.p, lv.vis.p, words,
                               lv.out.p, next_rays, next_cap, cnt);
            HIP_TRY(hipGetLastError());
            st.launches_shade++;
            cap = next_cap;
        }
        if (int rc = mark(CLS_OTHER)) return rc;  // the resolve passes are not timed
        st.levels = std::max(st.levels, levels);
        for (uint32_t L = levels; L-- > 1;) {
            Level& lv = x.levels[L];
            const uint32_t* n_dev = x.d_counters.p + (size_t)(L - 1) * CNT_WORDS + CNT_NEXT;
            hipLaunchKernelGGL(k_resolve, dim3((level_n[L] + 255) / 256), block, 0, stream, lv.out.p, lv.child.p, lv.rays.p, n_dev, S.mats,
                               x.levels[L - 1].child.p);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_resolve_pixels, dim3((n0 + 255) / 256), block, 0, stream, F, x.levels[0].out.p, x.levels[0].child.p, S.mats, n0, x.d_accum.p,
                           d_out);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_roll_counters, dim3(1), block, 0, stream, x.d_counters.p, max_levels, x.d_totals.p, x.h_totals);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev_total[1], stream));
    // valid pixels of this call's bands
    {
        const int bands = (c->height + 7) / 8;
        uint64_t rows = 0;
        for (int b = c->band_rank; b < bands; b += c->band_world) rows += (uint64_t)std::min(8, c->height - b * 8);
        st.primary_rays = rows * (uint64_t)c->width * (uint64_t)c->samples_per_pixel;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev_total[0], ev_total[1]));
    st.ms_total = ms;
    for (int k = 0; k + 1 < n_marks; ++k) {
        if (mark_cls[k] == CLS_OTHER) continue;
        HIP_TRY(hipEventElapsedTime(&ms, ev_mark[k], ev_mark[k + 1]));
        if (mark_cls[k] == CLS_CLOSEST) st.ms_closest += ms;
        else if (mark_cls[k] == CLS_SHADOW) st.ms_shadow += ms;
        else if (mark_cls[k] == CLS_TAIL) st.ms_tail += ms;
        else st.ms_shade += ms;
    }
    st.secondary_rays = x.h_totals->secondary_rays;  // written by the last k_roll_counters, complete after the synchronisation above
    st.shadow_rays = x.h_totals->shadow_rays;
    st.shadow_rays_not_traced = x.h_totals->idle_shadow_rays;
    st.levels = std::max(st.levels, (uint32_t)x.h_totals->deepest_level);
    st.rays = st.primary_rays + st.secondary_rays + st.shadow_rays;
    st_out = st;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_ads_set_render_streams(rt_ads* a, int n) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (n < 1 || n > kMaxRenderStreams) return fail(RT_ERR_INVALID, "render streams must be in [1,%d]", kMaxRenderStreams);
    a->render_streams = n;
    return RT_OK;
}

int rt_render(rt_ads* a, const rt_render_config* c, float* out_rgb, int mem, rt_render_stats* stats, void* stream_) {
    if (!a || !c || !out_rgb) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (!a->shading_set) return fail(RT_ERR_NOT_BUILT, "rt_scene_set_shading has not been called");
    if (mem != RT_MEM_HOST && mem != RT_MEM_DEVICE) return fail(RT_ERR_INVALID, "bad mem argument");
    if (int rc = use_device(a)) return rc;
    {
        FrameParams F;
        uint32_t n0 = 0;
        if (int rc = frame_params(a, c, F, n0)) return rc;  // argument checks
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t frame_floats = (size_t)c->width * c->height * 3;
    float* d_out = out_rgb;
    if (mem == RT_MEM_HOST) {
        HIP_TRY(a->d_stage.reserve(frame_floats));
        d_out = a->d_stage.p;
    }
    // The sub-frames: this call's bands dealt round-robin once more.  Short frames are chains of small launches, each
    // as long as its slowest wave; independent sub-frames on streams of their own fill the gaps of one another.  The
    // instrumented pass keeps one set of totals, so it runs the frame as one piece.
    const int bands_mine = std::max(0, ((c->height + 7) / 8 - c->band_rank + c->band_world - 1) / c->band_world);
    int n_sub = a->collect_stats ? 1 : std::min(a->render_streams, std::max(bands_mine, 1));
    if ((uint64_t)c->band_world * (uint64_t)n_sub > 0x7fffffffull) n_sub = 1;
    if (a->collect_stats) {
        if (int rc = stats_begin(a, stream)) return rc;
    }
    rt_render_stats sub_stats[kMaxRenderStreams] = {};
    int sub_rc[kMaxRenderStreams] = {};
    char sub_err[kMaxRenderStreams][sizeof g_err] = {};
    if (n_sub == 1) {
        sub_rc[0] = render_bands(a, a->frame[0], c, d_out, stream, 1, sub_stats[0]);
        if (sub_rc[0]) return sub_rc[0];
    } else {
        if (!a->pool || a->pool->workers() < n_sub - 1) {
            delete a->pool;
            a->pool = new SubFramePool(kMaxRenderStreams - 1);
        }
        // sub-frames 1.. start after what the caller has queued on its stream, and the caller's stream continues after them
        if (!a->frame[0].ev_fork) HIP_TRY(hipEventCreateWithFlags(&a->frame[0].ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(a->frame[0].ev_fork, stream));
        for (int k = 1; k < n_sub; ++k) {
            FrameCtx& x = a->frame[k];
            if (!x.own_stream) HIP_TRY(hipStreamCreateWithFlags(&x.own_stream, hipStreamNonBlocking));
            if (!x.ev_join) HIP_TRY(hipEventCreateWithFlags(&x.ev_join, hipEventDisableTiming));
            HIP_TRY(hipStreamWaitEvent(x.own_stream, a->frame[0].ev_fork, 0));
        }
        const std::function<void(int)> job = [&](int k) {
            rt_render_config sub = *c;
            sub.band_world = c->band_world * n_sub;
            sub.band_rank = c->band_rank + k * c->band_world;
            sub_rc[k] = render_bands(a, a->frame[k], &sub, d_out, k == 0 ? stream : a->frame[k].own_stream, n_sub, sub_stats[k]);
            if (sub_rc[k]) snprintf(sub_err[k], sizeof sub_err[k], "%s", g_err);  // (the message is the worker thread's)
        };
        a->pool->run(n_sub, job);
        for (int k = 0; k < n_sub; ++k)
            if (sub_rc[k]) return fail(sub_rc[k], "%s", sub_err[k]);
    }
    rt_render_stats st = sub_stats[0];
    for (int k = 1; k < n_sub; ++k) {
        const rt_render_stats& s = sub_stats[k];
        st.primary_rays += s.primary_rays; st.secondary_rays += s.secondary_rays; st.shadow_rays += s.shadow_rays; st.rays += s.rays;
        st.levels = std::max(st.levels, s.levels);
        st.ms_closest += s.ms_closest; st.ms_shadow += s.ms_shadow; st.ms_shade += s.ms_shade; st.ms_tail += s.ms_tail;
        st.ms_total = std::max(st.ms_total, s.ms_total);
        st.launches_closest += s.launches_closest; st.launches_shadow += s.launches_shadow; st.launches_shade += s.launches_shade;
        st.launches_tail += s.launches_tail;
        st.shadow_rays_not_traced += s.shadow_rays_not_traced;
        st.timed_closest_rays += s.timed_closest_rays; st.timed_shadow_rays += s.timed_shadow_rays;
    }
    if (mem == RT_MEM_HOST) {
        const int bands = (c->height + 7) / 8;
        if (c->band_world == 1) {  // the whole frame in one copy
            HIP_TRY(hipMemcpyAsync(out_rgb, d_out, frame_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
        } else
        for (int b = c->band_rank; b < bands; b += c->band_world) {
            const size_t row0 = (size_t)b * 8, rows = (size_t)std::min(8, c->height - b * 8);
            const size_t off = row0 * c->width * 3, cnt = rows * c->width * 3;
            HIP_TRY(hipMemcpyAsync(out_rgb + off, d_out + off, cnt * sizeof(float), hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (a->collect_stats) {
        if (int rc = stats_end(a, stream)) return rc;
    }
    a->q.search_calls += st.rays;
    a->q.closest_rays += st.primary_rays + st.secondary_rays;
    a->q.any_rays += st.shadow_rays;
    a->q.search_seconds += ((double)st.ms_closest + (double)st.ms_shadow) * 1e-3;
    if (stats) *stats = st;
    return RT_OK;
}

}  // extern "C"

namespace {

// The host thread of one frame slot: renders the frames submitted to it, one at a time.
void async_slot_loop(rt_ads* a, AsyncSlot* s) {
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(s->m);
            s->cv.wait(lock, [&] { return s->quit || s->state == AsyncSlot::QUEUED; });
            if (s->state != AsyncSlot::QUEUED) return;  // quit with nothing queued
        }
        rt_render_stats st{};
        // (n_sub = 1: with whole frames in flight a small level need not fill the chip on its own, and the fused tail
        // is the cheapest way through it -- measured with 2..8 frames in flight)
        const int rc = render_bands(a, s->ctx, &s->cfg, s->d_out, s->ctx.own_stream, 1, st);
        {
            std::lock_guard<std::mutex> lock(s->m);
            s->st = st;
            s->rc = rc;
            if (rc) snprintf(s->err, sizeof s->err, "%s", g_err);
            s->state = AsyncSlot::DONE;
        }
        s->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int rt_render_submit(rt_ads* a, const rt_render_config* c, float* out_rgb_device, void* stream_, int* ticket) {
    if (!a || !c || !out_rgb_device || !ticket) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (!a->shading_set) return fail(RT_ERR_NOT_BUILT, "rt_scene_set_shading has not been called");
    if (a->collect_stats) return fail(RT_ERR_UNSUPPORTED, "the instrumented pass keeps one set of totals: use rt_render while rt_ads_collect_search_stats is on");
    if (int rc = use_device(a)) return rc;
    {
        FrameParams F;
        uint32_t n0 = 0;
        if (int rc = frame_params(a, c, F, n0)) return rc;  // argument checks
    }
    if (!a->async_slots) a->async_slots = new AsyncSlot[kMaxFramesInFlight];
    int k = 0;
    for (; k < kMaxFramesInFlight; ++k) {
        std::lock_guard<std::mutex> lock(a->async_slots[k].m);
        if (a->async_slots[k].state == AsyncSlot::IDLE) break;
    }
    if (k == kMaxFramesInFlight) return fail(RT_ERR_CAPACITY, "%d frames are in flight already: rt_render_wait one first", kMaxFramesInFlight);
    AsyncSlot& s = a->async_slots[k];
    FrameCtx& x = s.ctx;
    if (!x.own_stream) HIP_TRY(hipStreamCreateWithFlags(&x.own_stream, hipStreamNonBlocking));
    if (!x.ev_fork) HIP_TRY(hipEventCreateWithFlags(&x.ev_fork, hipEventDisableTiming));
    // the frame starts behind what the caller's stream holds now
    HIP_TRY(hipEventRecord(x.ev_fork, (hipStream_t)stream_));
    HIP_TRY(hipStreamWaitEvent(x.own_stream, x.ev_fork, 0));
    if (!s.th.joinable()) s.th = std::thread(async_slot_loop, a, &s);
    {
        std::lock_guard<std::mutex> lock(s.m);
        s.cfg = *c;
        s.d_out = out_rgb_device;
        s.n_in_flight = ++a->async_in_flight;
        s.rc = 0;
        s.state = AsyncSlot::QUEUED;
    }
    s.cv.notify_all();
    *ticket = k;
    return RT_OK;
}

int rt_render_wait(rt_ads* a, int ticket, rt_render_stats* stats) {
    if (!a) return fail(RT_ERR_INVALID, "null argument");
    if (!a->async_slots || ticket < 0 || ticket >= kMaxFramesInFlight) return fail(RT_ERR_INVALID, "no such frame in flight (ticket %d)", ticket);
    AsyncSlot& s = a->async_slots[ticket];
    rt_render_stats st{};
    int rc = 0;
    {
        std::unique_lock<std::mutex> lock(s.m);
        if (s.state == AsyncSlot::IDLE) return fail(RT_ERR_INVALID, "no such frame in flight (ticket %d)", ticket);
        s.cv.wait(lock, [&] { return s.state == AsyncSlot::DONE; });
        st = s.st;
        rc = s.rc;
        if (rc) snprintf(g_err, sizeof g_err, "%s", s.err);
        s.state = AsyncSlot::IDLE;
    }
    --a->async_in_flight;
    if (rc) return rc;
    a->q.search_calls += st.rays;
    a->q.closest_rays += st.primary_rays + st.secondary_rays;
    a->q.any_rays += st.shadow_rays;
    a->q.search_seconds += ((double)st.ms_closest + (double)st.ms_shadow) * 1e-3;
    if (stats) *stats = st;
    return RT_OK;
}

int rt_render_primary_hits(rt_ads* a, const rt_render_config* c, float* out_t, int32_t* out_tri, int mem, void* stream_) {
    if (!a || !c || !out_t || !out_tri) return fail(RT_ERR_INVALID, "null argument");
    if (!a->built) return fail(RT_ERR_NOT_BUILT, "rt_ads_build has not been called");
    if (!a->shading_set) return fail(RT_ERR_NOT_BUILT, "rt_scene_set_shading has not been called");
    if (mem != RT_MEM_HOST && mem != RT_MEM_DEVICE) return fail(RT_ERR_INVALID, "bad mem argument");
    if (int rc = use_device(a)) return rc;
    FrameParams F;
    uint32_t n0 = 0;
    if (int rc = frame_params(a, c, F, n0)) return rc;
    if (c->band_world != 1) return fail(RT_ERR_UNSUPPORTED, "rt_render_primary_hits renders the whole image");
    hipStream_t stream = (hipStream_t)stream_;
    FrameCtx& x = a->frame[0];
    if (x.levels.empty()) x.levels.resize(1);
    Level& lv = x.levels[0];
    HIP_TRY(lv.reserve(n0, false));
    HIP_TRY(x.d_counters.reserve(CNT_WORDS));
    hipLaunchKernelGGL(k_frame_begin, dim3(1), dim3(256), 0, stream, x.d_counters.p, 1u, (LevelSlot*)nullptr, 0u, (FrameTotals*)nullptr);
    HIP_TRY(hipGetLastError());
    const size_t npix = (size_t)c->width * c->height;
    float* d_t = out_t;
    int32_t* d_tri = out_tri;
    if (mem == RT_MEM_HOST) {
        HIP_TRY(a->d_io.reserve(npix * 8));
        d_t = (float*)a->d_io.p;
        d_tri = (int32_t*)(a->d_io.p + npix * 4);
    }
    const dim3 block(256), grid((n0 + 255) / 256);
    const int mode_c = traversal_mode(a, true, false, n0);
    const uint32_t sparse_c = pool_sparse(a, mode_c, n0, 1);
    const dim3 block_c(closest_block(a, mode_c)), grid_c((uint32_t)((mode_threads(mode_c, n0, sparse_c) + block_c.x - 1) / block_c.x));
    launch_closest(F.cull != 0, mode_c, closest_prunes(a, mode_c), grid_c, block_c, lane_lds(a, mode_c, block_c.x, true), stream, scene_dev(a), F, 0u, (const RayRec*)nullptr, n0,
                   lv.hit_t.p, lv.hit_tri.p, lv.out.p, lv.work.p, x.d_counters.p, closest_opts(a, mode_c, sparse_c));
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_primary_hits, grid, block, 0, stream, F, lv.hit_t.p, lv.hit_tri.p, n0, d_t, d_tri);
    HIP_TRY(hipGetLastError());
    if (a->collect_stats) {
        const SceneDev S = scene_dev(a);
        if (int rc = stats_begin(a, stream)) return rc;
        if (F.cull) hipLaunchKernelGGL(k_stats_closest<true>, grid, block, 0, stream, S, F, 0u, (const RayRec*)nullptr, n0, (const uint32_t*)nullptr, a->d_qstats.p);
        else hipLaunchKernelGGL(k_stats_closest<false>, grid, block, 0, stream, S, F, 0u, (const RayRec*)nullptr, n0, (const uint32_t*)nullptr, a->d_qstats.p);
        HIP_TRY(hipGetLastError());
        if (int rc = stats_end(a, stream)) return rc;
    }
    if (mem == RT_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out_t, d_t, npix * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(out_tri, d_tri, npix * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    const uint64_t rays = npix;
    a->q.search_calls += rays;
    a->q.closest_rays += rays;
    return RT_OK;
}

}  // extern "C"

#ifdef RT_LANE_PROFILE
// 64 values: [0,32) per-lane form (closest, any-hit: 16 each), [32,48) wave-together forms, [48,64) task pool (closest, any-hit: 8 each);
// the counters are cleared.
extern "C" int rt_debug_lane_profile(unsigned long long* out64) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(rt::g_lane_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out64 + 32, HIP_SYMBOL(rt::g_wave_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out64 + 48, HIP_SYMBOL(rt::g_pool_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_lane_prof), z, sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_wave_prof), z, sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(rt::g_pool_prof), z, sizeof(unsigned long long) * 16) != hipSuccess;
}
#endif
