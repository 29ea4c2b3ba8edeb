// host_octree.h -- host-side octree construction and its flattening into the device layout.
//
// The tree is built with exactly the splitting rule of the reference's Octree::Build
// (src/ADS/Octree/octree.cc:19-171 of the reference): same boxes, same triangle-to-octant
// assignment (13-axis SAT, src/collision_detection.h:89-138), same leaf criteria, nodes
// numbered in the reference's queue (breadth-first) order.  It is then flattened into two
// arrays that the gfx950 kernels read with scalar loads:
//
//   NodeRec[ internal node ]  128 B: the 12 planes of its 8 octant boxes + what each octant is
//   LeafTri[ leaf slot ]       48 B: one triangle reference of one leaf, geometry inlined
//
// Leaf slots are laid out leaf after leaf in breadth-first leaf order, each leaf's triangles in
// ascending triangle index.  The reference's candidate list for a ray is the concatenation of the
// visited leaves' lists in breadth-first order (octree.cc:186-191), so the global slot index of a
// reference is its rank in every candidate list that contains it: the smallest slot among equal
// hit distances is the reference's winner (src/Renderer/renderer.cc:130-132).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "rt_math.h"

namespace rt {

struct HostNode {
    bool leaf = false;
    uint32_t depth = 0;
    f3 bmin{0, 0, 0}, bmax{0, 0, 0};
    uint64_t tri_begin = 0;  // into HostOctree::pool
    uint32_t tri_count = 0;
    int32_t child[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
};

struct HostOctree {
    bool empty = true;            // no triangles: the reference leaves its root null (octree.cc:24-27)
    std::vector<HostNode> nodes;  // breadth-first (creation) order, nodes[0] = root
    std::vector<uint32_t> pool;   // triangle index lists of all nodes
    double build_seconds = 0.0;
};

// Geometry of one triangle as the kernels want it: a, e1 = b - a, e2 = c - a
// (src/collision_detection.h:21-22 computes the same two subtractions per test).
struct alignas(16) LeafTri {
    float ax, ay, az;
    float e1x, e1y, e1z;
    float e2x, e2y, e2z;
    uint32_t tri;  // triangle index in Build order
    uint32_t pad0, pad1;
};
static_assert(sizeof(LeafTri) == 48, "LeafTri must be 48 bytes");

constexpr uint32_t kChildInternal = 0x80000000u;
constexpr uint32_t kFlatMaxLeaves = 32, kFlatMaxTris = 1024;
// Clustered form (see ClusterForm): flag of an internal child that is the root of a cluster, and where a cluster's leaf
// keeps its ordinal inside info[0] (slot index below, ordinal above).
constexpr uint32_t kChildCluster = 0x40000000u;
constexpr uint32_t kClusterMaxLeaves = 32, kClusterOrdShift = 24, kClusterSlotMask = (1u << kClusterOrdShift) - 1u;

// One internal node.  The boxes of its 8 octants share their planes: octant o = (bx<<2)|(by<<1)|bz has
// min.x = lo[0][bx], max.x = hi[0][bx], min.y = lo[1][by], ... (octree.cc:93-97 computes every octant's corner from
// the same per-axis terms, so the two candidates per axis and side are bit-identical across octants; the flattener
// checks it).  12 plane values replace 48 box floats, and the slab products (min - o) * inv are computed once per
// plane instead of once per octant -- the same products the reference computes, just not repeated.
// info[o][1] == 0: octant o is null/empty (skipped before any box test, octree.cc:195);
// info[o][1] & kChildInternal: internal child whose NodeRec index is info[o][0]; otherwise a leaf with info[o][1]
// triangles whose slots start at info[o][0].
// pad[1..2] of a record of the plain tree: `nonempty` in the octant order of each of the 8 sign patterns a ray direction
// can have (byte f of the 64 bits pad[1] | pad[2] << 32: bit k = bit (k ^ f) of nonempty, see node_ordered_nonempty);
// the per-lane walks read the byte of their ray.  The clustered form's copy of the records uses pad[0..1] for its lists.
struct alignas(128) NodeRec {
    float lo[3][2];     // [axis][bit]: min plane of the octants whose axis bit is 0 / 1
    float hi[3][2];     // [axis][bit]: max plane
    uint32_t nonempty;  // bit o set: octant o exists and lists triangles (info[o][1] != 0)
    uint32_t pad[3];
    uint32_t info[8][2];
};
static_assert(sizeof(NodeRec) == 128, "NodeRec must be 128 bytes");

// A ray that runs backwards along an axis (1 / dir < 0) meets the octants with that axis bit set first.  With
// flip = (x backwards) << 2 | (y backwards) << 1 | (z backwards), the octants in the order k = 0..7 -> octant k ^ flip
// are front to back for the ray.  word 0: the masks for flip 0..3 (one byte each), word 1: for flip 4..7.
RT_HD void node_ordered_nonempty(uint32_t nonempty, uint32_t out[2]) {
    out[0] = out[1] = 0;
    for (uint32_t f = 0; f < 8; ++f) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < 8; ++k) m |= ((nonempty >> (k ^ f)) & 1u) << k;
        out[f >> 2] |= m << ((f & 3u) * 8u);
    }
}

struct FlatOctree {
    std::vector<NodeRec> nodes;  // internal nodes, breadth-first
    std::vector<LeafTri> slots;  // leaf slots, breadth-first leaf order
    uint32_t root_info[2] = {0, 0};  // like NodeRec::info for the root (never box-tested, octree.cc:177)
    uint32_t max_stack = 0;          // deepest traversal stack any ray can need
    // Small-tree form, filled when the tree has at most kFlatMaxLeaves leaves and kFlatMaxTris triangles:
    // flat_nodes = nodes with leaf octants' info[0] = leaf ordinal; flat_tris[j] = triangle j with pad0 = the
    // bit mask of the leaves that list it.
    std::vector<NodeRec> flat_nodes;
    std::vector<LeafTri> flat_tris;
};

// Clustered form of a tree that is too big for the small-tree form.  The leaves are grouped into CLUSTERS: the leaves
// of each maximal subtree with at most kClusterMaxLeaves leaves, and -- above those subtrees -- the leaf children of
// each node.  For a cluster the wave-together any-hit walk first only records, per lane, which of the cluster's leaves
// the lane reaches (one bit per leaf ordinal; ordinals follow the slot order) and then walks the cluster's list of
// DISTINCT triangles once: a triangle is tested for all lanes whose reached leaves list it -- instead of once per
// listing leaf (a wall triangle of a Cornell box is listed by most of the root's octants).
//   nodes = the tree's records with (a) kChildCluster set in info[1] of every internal child that is the root of such
//           a subtree, (b) info[0] of every leaf = slot index | ordinal << kClusterOrdShift, and (c) in the record that
//           owns a cluster (the subtree's root, or the node whose leaf children form it) pad[0] = index of the
//           cluster's first record in `tris`, pad[1] = number of its records (a multiple of 4);
//   tris  = per cluster its distinct triangles in ascending index, pad0 = the set of the cluster's leaves (by ordinal)
//           that list the triangle; lists are padded to whole blocks of 4 with all-zero records (listed by no leaf).
struct ClusterForm {
    std::vector<NodeRec> nodes;
    std::vector<LeafTri> tris;
    uint32_t root_info1 = 0;  // root_info[1], with kChildCluster if the whole tree is one cluster
    uint32_t n_clusters = 0;  // subtree clusters (at least 1 when the form applies)
};
// Fills `out` from the flattened tree; leaves it empty (n_clusters = 0) when the form does not apply: root leaf, no
// cluster with a triangle listed twice, or slot indices that need more than kClusterOrdShift bits.
void cluster_octree(const NodeRec* nodes, size_t n_rec, const LeafTri* slots, size_t n_slots, const uint32_t root_info[2], ClusterForm& out);

struct TreeShape {
    uint64_t max_depth = 0, nodes = 0, leaves = 0, max_tris_in_leaf = 0, leaf_refs = 0, internal = 0;
    float avg_leaf_depth = 0, avg_tris_in_leaf = 0;
};

// positions: n x 9 floats.  Returns false and sets err if the reference's build invariant (every parent
// triangle lands in at least one octant, octree.cc:124-152) is violated.
bool build_octree(const float* positions, uint64_t n, uint32_t max_tris_per_bb, uint32_t max_depth, HostOctree& out,
                  std::string& err);
bool flatten_octree(const HostOctree& t, const float* positions, FlatOctree& out, std::string& err);
TreeShape tree_shape(const HostOctree& t);  // octree.cc:226-266

// Exact host versions of the reference's primitives (used by the builder; exported for unit tests).
bool tri_box_overlap(const f3 v[3], f3 normal, f3 bmin, f3 bmax);  // src/collision_detection.h:89-138
f3 face_normal(const f3 v[3]);                                      // src/triangle.h:28-32

}  // namespace rt
