// rt_build.h -- device-side octree construction (gfx950 kernels).
//
// Builds the same tree as the host builder (host_octree.cpp), which restates Octree::Build of the reference
// (src/ADS/Octree/octree.cc:19-171), and writes it straight into the device layout (NodeRec / LeafTri arrays of
// host_octree.h) without a host round trip of the tree.
//
// The reference pops a FIFO queue, so all nodes of one depth are processed before any node of the next and
// children are numbered in (parent order, octant order).  The device builder therefore works one depth at a time
// over two flat arrays: the nodes of the depth (creation order) and the concatenation of their triangle lists
// (each ascending, as in the reference).  Per depth:
//   k_build_masks    one thread per (node, triangle) list entry: the 8 triangle/octant-box overlap tests -> 8 bits
//   scan (3 kernels) exclusive prefix sums of each of the 8 bits over the whole list array: the difference between
//                    an entry and its node's first entry is the entry's position in the octant's list (stable,
//                    so the lists stay ascending); the difference across a node is the octant's count
//   k_build_decide   one thread per node: leaf or internal (octree.cc:71-74, 107-122, 154-164)
//   scan             over the nodes: record index of internal nodes, first slot of leaves, first child node and
//                    first child list entry of internal nodes -- all in creation (= breadth-first) order
//   k_build_emit     one thread per node: the parent's link to it, its own NodeRec (octant planes), its children
//   k_build_scatter  one thread per list entry: leaf -> LeafTri slot; internal -> copies into the octants' lists
// The triangle/box test and the octant boxes are the shared host/device functions of rt_sat.h.
#pragma once

#include <hip/hip_runtime.h>

#include "host_octree.h"
#include "rt_sat.h"

namespace rt {

struct TriBuild {  // what the overlap test reads per triangle
    float v[9];
    float n[3];
};
static_assert(sizeof(TriBuild) == 48, "TriBuild must be 48 bytes");

struct BuildNode {
    float bmin[3], bmax[3];
    uint32_t begin, count;  // its triangle list inside the depth's list array
    uint32_t parent;        // (NodeRec index of the parent << 3) | octant; kNoParent for the root
    uint32_t internal;      // set by k_build_decide
    uint32_t pad[2];
};
static_assert(sizeof(BuildNode) == 48, "BuildNode must be 48 bytes");
constexpr uint32_t kNoParent = 0xffffffffu;

// node-scan channels
enum { CH_INTERNAL = 0, CH_SLOTS = 1, CH_CHILDREN = 2, CH_CHILD_LIST = 3 };

constexpr int kScanThreads = 256, kScanItems = 8, kScanTile = kScanThreads * kScanItems;

__global__ void __launch_bounds__(256) k_build_prep(const float* __restrict__ positions, uint32_t n, TriBuild* __restrict__ tb,
                                                     LeafTri* __restrict__ geom) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = positions + (size_t)i * 9;
    const f3 a = mk(p[0], p[1], p[2]), b = mk(p[3], p[4], p[5]), c = mk(p[6], p[7], p[8]);
    const f3 nrm = face_normal3(a, b, c);
    TriBuild t;
    for (int k = 0; k < 9; ++k) t.v[k] = p[k];
    t.n[0] = nrm.x; t.n[1] = nrm.y; t.n[2] = nrm.z;
    tb[i] = t;
    LeafTri g;
    g.ax = a.x; g.ay = a.y; g.az = a.z;
    g.e1x = b.x - a.x; g.e1y = b.y - a.y; g.e1z = b.z - a.z;
    g.e2x = c.x - a.x; g.e2y = c.y - a.y; g.e2z = c.z - a.z;
    g.tri = i;
    g.pad0 = g.pad1 = 0;
    geom[i] = g;
}

__global__ void __launch_bounds__(256) k_build_iota(uint32_t* __restrict__ list, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[i] = i;
}

// A node is subdivided (its octant lists are computed) unless it was created as a leaf (count <= max_per_bb,
// octree.cc:154-164) or sits at the depth limit (octree.cc:71-74).
__device__ __forceinline__ bool build_candidate(const BuildNode& nd, uint32_t max_per_bb, bool at_max_depth) {
    return !at_max_depth && nd.count > max_per_bb;
}

__global__ void __launch_bounds__(256) k_build_masks(const BuildNode* __restrict__ nodes, const uint32_t* __restrict__ list,
                                                      const uint32_t* __restrict__ owner, uint32_t n_list,
                                                      const TriBuild* __restrict__ tb, uint32_t max_per_bb, int at_max_depth,
                                                      uint8_t* __restrict__ mask, uint32_t* __restrict__ uncovered) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_list) return;
    const BuildNode nd = nodes[owner[j]];
    if (!build_candidate(nd, max_per_bb, at_max_depth != 0)) {
        mask[j] = 0;
        return;
    }
    const TriBuild t = tb[list[j]];
    const f3 a = mk(t.v[0], t.v[1], t.v[2]), b = mk(t.v[3], t.v[4], t.v[5]), c = mk(t.v[6], t.v[7], t.v[8]);
    const f3 nrm = mk(t.n[0], t.n[1], t.n[2]);
    const OctantFrame fr = octant_frame(mk(nd.bmin[0], nd.bmin[1], nd.bmin[2]), mk(nd.bmax[0], nd.bmax[1], nd.bmax[2]));
    uint32_t m = 0;
#pragma unroll 1
    for (unsigned o = 0; o < 8; ++o) {
        f3 cmin, cmax;
        octant_box(fr, o, cmin, cmax);
        if (sat_tri_box(a, b, c, nrm, cmin, cmax)) m |= 1u << o;
    }
    mask[j] = (uint8_t)m;
    // octree.cc:124-152: a triangle of a subdivided node that reaches no octant is a build error
    if (m == 0) atomicAdd(uncovered, 1u);
}

// ---- 8-channel exclusive scan over N elements; out has N + 1 elements (the last one holds the totals) ----------
// MASK = true: element j contributes bit o of mask[j] to channel o; false: element j is 8 uint32 values.
template <bool MASK>
__device__ __forceinline__ void scan_load(const void* __restrict__ in, uint32_t j, uint32_t v[8]) {
    if (MASK) {
        const uint32_t m = ((const uint8_t*)in)[j];
#pragma unroll
        for (int o = 0; o < 8; ++o) v[o] = (m >> o) & 1u;
    } else {
        const uint4* p = (const uint4*)in + (size_t)j * 2;
        const uint4 q0 = p[0], q1 = p[1];
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
}

// Exclusive prefix of `mine` over the threads of the block, per channel; total[] = the block's sums.
__device__ __forceinline__ void block_exclusive8(const uint32_t mine[8], uint32_t excl[8], uint32_t total[8]) {
    __shared__ uint32_t s_wave[kScanThreads / 64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        uint32_t x = mine[o];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        inc[o] = x;
    }
    if (lane == 63) {
#pragma unroll
        for (int o = 0; o < 8; ++o) s_wave[wave][o] = inc[o];
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) {
            const uint32_t s = s_wave[w][o];
            if (w < wave) base += s;
            tot += s;
        }
        excl[o] = base + inc[o] - mine[o];
        total[o] = tot;
    }
    __syncthreads();
}

template <bool MASK>
__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const void* __restrict__ in, uint32_t n, uint32_t* __restrict__ bsum) {
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = first + k;
        if (j >= n) break;
        uint32_t v[8];
        scan_load<MASK>(in, j, v);
#pragma unroll
        for (int o = 0; o < 8; ++o) mine[o] += v[o];
    }
    uint32_t excl[8], total[8];
    block_exclusive8(mine, excl, total);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) bsum[(size_t)blockIdx.x * 8 + o] = total[o];
    }
}

// One block: turns the nb block sums into exclusive prefixes in place; bsum[nb] receives the grand totals.
__global__ void __launch_bounds__(kScanThreads) k_scan_blocks(uint32_t* __restrict__ bsum, uint32_t nb) {
    uint32_t carry[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < nb; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (i < nb) {
#pragma unroll
            for (int o = 0; o < 8; ++o) mine[o] = bsum[(size_t)i * 8 + o];
        }
        uint32_t excl[8], total[8];
        block_exclusive8(mine, excl, total);
        if (i < nb) {
#pragma unroll
            for (int o = 0; o < 8; ++o) bsum[(size_t)i * 8 + o] = carry[o] + excl[o];
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) carry[o] += total[o];
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) bsum[(size_t)nb * 8 + o] = carry[o];
    }
}

template <bool MASK>
__global__ void __launch_bounds__(kScanThreads) k_scan_write(const void* __restrict__ in, uint32_t n, const uint32_t* __restrict__ bsum,
                                                              uint32_t nb, uint32_t* __restrict__ out) {
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t vals[kScanItems][8];
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = first + k;
        if (j < n) {
            scan_load<MASK>(in, j, vals[k]);
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o) vals[k][o] = 0;
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) mine[o] += vals[k][o];
    }
    uint32_t excl[8], total[8];
    block_exclusive8(mine, excl, total);
    uint32_t run[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) run[o] = bsum[(size_t)blockIdx.x * 8 + o] + excl[o];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t j = first + k;
        if (j < n) {
            uint4* p = (uint4*)out + (size_t)j * 2;
            p[0] = make_uint4(run[0], run[1], run[2], run[3]);
            p[1] = make_uint4(run[4], run[5], run[6], run[7]);
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) run[o] += vals[k][o];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) out[(size_t)n * 8 + o] = bsum[(size_t)nb * 8 + o];
    }
}

// ---- per-depth decisions and output -------------------------------------------------------------------------------
__device__ __forceinline__ void octant_counts(const uint32_t* __restrict__ S, const BuildNode& nd, uint32_t cnt[8]) {
    const uint4* lo = (const uint4*)S + (size_t)nd.begin * 2;
    const uint4* hi = (const uint4*)S + (size_t)(nd.begin + nd.count) * 2;
    const uint4 a0 = lo[0], a1 = lo[1], b0 = hi[0], b1 = hi[1];
    cnt[0] = b0.x - a0.x; cnt[1] = b0.y - a0.y; cnt[2] = b0.z - a0.z; cnt[3] = b0.w - a0.w;
    cnt[4] = b1.x - a1.x; cnt[5] = b1.y - a1.y; cnt[6] = b1.z - a1.z; cnt[7] = b1.w - a1.w;
}

__global__ void __launch_bounds__(256) k_build_decide(BuildNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ S,
                                                       uint32_t max_per_bb, int at_max_depth, int depth, uint32_t* __restrict__ V,
                                                       uint32_t* __restrict__ max_leaf) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) return;
    const BuildNode nd = nodes[k];
    bool internal = false;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (build_candidate(nd, max_per_bb, at_max_depth != 0)) {
        octant_counts(S, nd, cnt);
        // octree.cc:107-122: a node whose eight octants all list every one of its triangles stays a leaf
        for (int o = 0; o < 8; ++o) internal = internal || cnt[o] != nd.count;
    }
    uint32_t children = 0, child_list = 0;
    if (internal) {
        for (int o = 0; o < 8; ++o) {
            children += cnt[o] ? 1u : 0u;
            child_list += cnt[o];
        }
    }
    nodes[k].internal = internal ? 1u : 0u;
    uint4* v = (uint4*)V + (size_t)k * 2;
    v[0] = make_uint4(internal ? 1u : 0u, internal ? 0u : nd.count, children, child_list);
    v[1] = make_uint4(0, 0, 0, 0);
    // the reference's statistics only see leaves that hang below an internal node (octree.cc:244-252)
    if (!internal && depth > 0) atomicMax(max_leaf, nd.count);
}

__global__ void __launch_bounds__(256) k_build_emit(const BuildNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ S,
                                                     const uint32_t* __restrict__ P, uint32_t rec_base, uint32_t slot_base,
                                                     NodeRec* __restrict__ recs, BuildNode* __restrict__ next_nodes,
                                                     uint32_t* __restrict__ cinfo) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) return;
    const BuildNode nd = nodes[k];
    const uint32_t* pk = P + (size_t)k * 8;
    const uint32_t rec = rec_base + pk[CH_INTERNAL];
    if (nd.parent != kNoParent) {
        uint32_t* info = recs[nd.parent >> 3].info[nd.parent & 7u];
        if (nd.internal) {
            info[0] = rec;
            info[1] = kChildInternal;
        } else {
            info[0] = slot_base + pk[CH_SLOTS];
            info[1] = nd.count;
        }
    }
    if (!nd.internal) return;
    uint32_t cnt[8];
    octant_counts(S, nd, cnt);
    const OctantFrame fr = octant_frame(mk(nd.bmin[0], nd.bmin[1], nd.bmin[2]), mk(nd.bmax[0], nd.bmax[1], nd.bmax[2]));
    uint32_t nonempty = 0;
    for (int o = 0; o < 8; ++o) nonempty |= cnt[o] ? (1u << o) : 0u;
    NodeRec r;
    // a plane is recorded only if a listed octant uses it (the others are never read); bit of axis x = o>>2, y = o>>1, z = o
    const uint32_t uses[3][2] = {{0x0fu, 0xf0u}, {0x33u, 0xccu}, {0x55u, 0xaau}};
    const unsigned rep[3][2] = {{0u, 4u}, {0u, 2u}, {0u, 1u}};
    for (int ax = 0; ax < 3; ++ax)
        for (int b = 0; b < 2; ++b) {
            f3 cmin, cmax;
            octant_box(fr, rep[ax][b], cmin, cmax);
            const bool used = (nonempty & uses[ax][b]) != 0;
            r.lo[ax][b] = used ? comp(cmin, ax) : 0.0f;
            r.hi[ax][b] = used ? comp(cmax, ax) : 0.0f;
        }
    r.nonempty = nonempty;
    r.pad[0] = 0;
    node_ordered_nonempty(nonempty, &r.pad[1]);
    for (int o = 0; o < 8; ++o) r.info[o][0] = r.info[o][1] = 0;  // the children fill theirs in the next depth's pass
    recs[rec] = r;
    uint32_t ci = pk[CH_CHILDREN], cl = pk[CH_CHILD_LIST];
    for (unsigned o = 0; o < 8; ++o) {
        cinfo[(size_t)k * 16 + o] = cl;
        cinfo[(size_t)k * 16 + 8 + o] = ci;
        if (!cnt[o]) continue;
        f3 cmin, cmax;
        octant_box(fr, o, cmin, cmax);
        BuildNode c;
        c.bmin[0] = cmin.x; c.bmin[1] = cmin.y; c.bmin[2] = cmin.z;
        c.bmax[0] = cmax.x; c.bmax[1] = cmax.y; c.bmax[2] = cmax.z;
        c.begin = cl;
        c.count = cnt[o];
        c.parent = (rec << 3) | o;
        c.internal = 0;
        c.pad[0] = c.pad[1] = 0;
        next_nodes[ci] = c;
        ++ci;
        cl += cnt[o];
    }
}

__global__ void __launch_bounds__(256) k_build_scatter(const BuildNode* __restrict__ nodes, const uint32_t* __restrict__ list,
                                                        const uint32_t* __restrict__ owner, const uint8_t* __restrict__ mask,
                                                        const uint32_t* __restrict__ S, const uint32_t* __restrict__ P,
                                                        const uint32_t* __restrict__ cinfo, uint32_t n_list, uint32_t slot_base,
                                                        const LeafTri* __restrict__ geom, LeafTri* __restrict__ slots,
                                                        uint32_t* __restrict__ next_list, uint32_t* __restrict__ next_owner) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_list) return;
    const uint32_t k = owner[j];
    const BuildNode nd = nodes[k];
    const uint32_t tri = list[j];
    if (!nd.internal) {
        // leaf: its list becomes consecutive slots, in list (ascending triangle) order
        const float4* src = (const float4*)(geom + tri);
        float4* dst = (float4*)(slots + (size_t)slot_base + P[(size_t)k * 8 + CH_SLOTS] + (j - nd.begin));
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
        return;
    }
    const uint32_t m = mask[j];
    const uint32_t* sj = S + (size_t)j * 8;
    const uint32_t* sb = S + (size_t)nd.begin * 8;
    for (unsigned o = 0; o < 8; ++o) {
        if (!((m >> o) & 1u)) continue;
        const uint32_t dest = cinfo[(size_t)k * 16 + o] + (sj[o] - sb[o]);
        next_list[dest] = tri;
        next_owner[dest] = cinfo[(size_t)k * 16 + 8 + o];
    }
}

}  // namespace rt
