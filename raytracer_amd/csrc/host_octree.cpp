// host_octree.cpp -- see host_octree.h.
#include "host_octree.h"
#include "rt_sat.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rt {

f3 face_normal(const f3 v[3]) { return face_normal3(v[0], v[1], v[2]); }

bool tri_box_overlap(const f3 v[3], f3 normal, f3 bmin, f3 bmax) { return sat_tri_box(v[0], v[1], v[2], normal, bmin, bmax); }

namespace {

struct TriGeom {
    f3 v[3];
    f3 n;
};

struct Split {  // work of one frontier node
    f3 cmin[8], cmax[8];
    std::vector<uint32_t> list[8];
};

unsigned worker_count() {
    if (const char* e = std::getenv("RT_BUILD_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) return (unsigned)v;
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    return std::min(hw, 16u);
}

template <class F>
void parallel_for(size_t n, unsigned workers, F&& f) {
    if (workers <= 1 || n <= 1) {
        for (size_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    const unsigned k = (unsigned)std::min<size_t>(workers, n);
    pool.reserve(k);
    for (unsigned w = 0; w < k; ++w)
        pool.emplace_back([&] {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto& t : pool) t.join();
}

}  // namespace

bool build_octree(const float* positions, uint64_t n, uint32_t max_per_bb, uint32_t max_depth, HostOctree& out,
                  std::string& err) {
    const auto t_begin = std::chrono::steady_clock::now();
    out = HostOctree();
    if (n == 0) return true;  // octree.cc:24-27
    out.empty = false;

    std::vector<TriGeom> tris(n);
    for (uint64_t i = 0; i < n; ++i) {
        for (int v = 0; v < 3; ++v) tris[i].v[v] = mk(positions[i * 9 + v * 3], positions[i * 9 + v * 3 + 1], positions[i * 9 + v * 3 + 2]);
        tris[i].n = face_normal(tris[i].v);
    }

    const float aabb_eps = kAabbEps;
    f3 mn = tris[0].v[0], mx = tris[0].v[0];  // octree.cc:33-44
    for (const TriGeom& t : tris)
        for (const f3& p : t.v) {
            mn.x = p.x < mn.x ? p.x : mn.x;
            mn.y = p.y < mn.y ? p.y : mn.y;
            mn.z = p.z < mn.z ? p.z : mn.z;
            mx.x = p.x > mx.x ? p.x : mx.x;
            mx.y = p.y > mx.y ? p.y : mx.y;
            mx.z = p.z > mx.z ? p.z : mx.z;
        }

    HostNode root;
    root.bmin = mn - vabs(mn * mk(aabb_eps, aabb_eps, aabb_eps));  // octree.cc:48
    root.bmax = mx + vabs(mx * mk(aabb_eps, aabb_eps, aabb_eps));
    root.depth = 0;
    root.tri_begin = 0;
    root.tri_count = (uint32_t)n;
    out.pool.resize(n);
    for (uint64_t i = 0; i < n; ++i) out.pool[i] = (uint32_t)i;
    out.nodes.push_back(root);
    if (n <= max_per_bb) {  // octree.cc:56-63
        out.nodes[0].leaf = true;
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        return true;
    }

    // The reference pops a FIFO queue (octree.cc:65-69).  All nodes of one depth leave the queue before any
    // node of the next depth, and children are appended in (parent order, octant order), so processing one
    // depth at a time and committing the children in that order numbers the nodes identically -- and lets
    // the (node, octant) assignment work of a whole depth run on several host threads.
    const unsigned workers = worker_count();
    std::vector<uint32_t> frontier{0};
    std::vector<Split> work;
    while (!frontier.empty()) {
        std::vector<uint32_t> todo;  // frontier nodes that are actually subdivided (not at max depth)
        for (uint32_t ni : frontier) {
            if (out.nodes[ni].depth == max_depth)
                out.nodes[ni].leaf = true;  // octree.cc:71-74
            else
                todo.push_back(ni);
        }
        work.clear();
        work.resize(todo.size());
        for (size_t k = 0; k < todo.size(); ++k) {  // octree.cc:81-97
            const HostNode& cur = out.nodes[todo[k]];
            const OctantFrame fr = octant_frame(cur.bmin, cur.bmax);
            for (unsigned o = 0; o < 8; ++o) octant_box(fr, o, work[k].cmin[o], work[k].cmax[o]);
        }
        parallel_for(todo.size() * 8, workers, [&](size_t task) {  // octree.cc:100-105
            const size_t k = task >> 3;
            const unsigned o = (unsigned)(task & 7);
            const HostNode& cur = out.nodes[todo[k]];
            std::vector<uint32_t>& dst = work[k].list[o];
            const uint32_t* src = out.pool.data() + cur.tri_begin;
            for (uint32_t j = 0; j < cur.tri_count; ++j) {
                const TriGeom& t = tris[src[j]];
                if (tri_box_overlap(t.v, t.n, work[k].cmin[o], work[k].cmax[o])) dst.push_back(src[j]);
            }
        });
        std::vector<uint32_t> next;
        std::vector<uint8_t> seen;
        for (size_t k = 0; k < todo.size(); ++k) {
            const uint32_t ni = todo[k];
            const uint32_t parent_count = out.nodes[ni].tri_count;
            bool split = false;
            for (unsigned o = 0; o < 8; ++o)
                if (work[k].list[o].size() != parent_count) split = true;  // octree.cc:107-109
            if (!split) {  // octree.cc:119-122
                out.nodes[ni].leaf = true;
                continue;
            }
            // octree.cc:124-152: every parent triangle must be in at least one octant.  The octant lists are
            // ascending subsequences of the parent's list, so a merge-style count of distinct entries suffices.
            {
                const uint32_t* src = out.pool.data() + out.nodes[ni].tri_begin;
                size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t covered = 0;
                for (uint32_t j = 0; j < parent_count; ++j) {
                    bool hit = false;
                    for (unsigned o = 0; o < 8; ++o) {
                        const auto& l = work[k].list[o];
                        if (pos[o] < l.size() && l[pos[o]] == src[j]) {
                            ++pos[o];
                            hit = true;
                        }
                    }
                    covered += hit;
                }
                if (covered != parent_count) {
                    err = "octree build: " + std::to_string(parent_count - covered) + " of " + std::to_string(parent_count) +
                          " parent triangles were assigned to no octant (node " + std::to_string(ni) + ")";
                    return false;
                }
            }
            for (unsigned o = 0; o < 8; ++o) {  // octree.cc:112-116, 154-164
                const auto& l = work[k].list[o];
                if (l.empty()) continue;
                HostNode c;
                c.depth = out.nodes[ni].depth + 1;
                c.bmin = work[k].cmin[o];
                c.bmax = work[k].cmax[o];
                c.tri_begin = out.pool.size();
                c.tri_count = (uint32_t)l.size();
                out.pool.insert(out.pool.end(), l.begin(), l.end());
                const uint32_t ci = (uint32_t)out.nodes.size();
                if (l.size() > max_per_bb) {
                    c.leaf = false;
                    next.push_back(ci);
                } else {
                    c.leaf = true;
                }
                out.nodes.push_back(c);
                out.nodes[ni].child[o] = (int32_t)ci;
            }
            for (auto& l : work[k].list) std::vector<uint32_t>().swap(l);
        }
        frontier.swap(next);
    }
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return true;
}

bool flatten_octree(const HostOctree& t, const float* positions, FlatOctree& out, std::string& err) {
    out = FlatOctree();
    if (t.empty) return true;
    const size_t nn = t.nodes.size();
    std::vector<uint32_t> rec_of(nn, 0), slot_of(nn, 0);
    uint32_t n_rec = 0;
    uint64_t n_slot = 0;
    for (size_t i = 0; i < nn; ++i) {
        if (t.nodes[i].leaf) {
            slot_of[i] = (uint32_t)n_slot;
            n_slot += t.nodes[i].tri_count;
        } else {
            rec_of[i] = n_rec++;
        }
    }
    out.nodes.resize(n_rec);
    out.slots.resize(n_slot);
    uint32_t deepest_internal = 0;
    for (size_t i = 0; i < nn; ++i) {
        const HostNode& n = t.nodes[i];
        if (n.leaf) {
            for (uint32_t k = 0; k < n.tri_count; ++k) {
                const uint32_t ti = t.pool[n.tri_begin + k];
                const float* p = positions + (size_t)ti * 9;
                LeafTri& s = out.slots[slot_of[i] + k];
                s.ax = p[0]; s.ay = p[1]; s.az = p[2];
                s.e1x = p[3] - p[0]; s.e1y = p[4] - p[1]; s.e1z = p[5] - p[2];
                s.e2x = p[6] - p[0]; s.e2y = p[7] - p[1]; s.e2z = p[8] - p[2];
                s.tri = ti;
                s.pad0 = s.pad1 = 0;
            }
            continue;
        }
        deepest_internal = std::max(deepest_internal, n.depth);
        NodeRec& r = out.nodes[rec_of[i]];
        std::memset(&r, 0, sizeof r);
        bool have_lo[3][2] = {{false, false}, {false, false}, {false, false}};
        for (int o = 0; o < 8; ++o) {
            const int32_t ci = n.child[o];
            if (ci < 0 || t.nodes[ci].tri_count == 0) continue;  // octree.cc:195
            const HostNode& c = t.nodes[ci];
            const int bit[3] = {(o >> 2) & 1, (o >> 1) & 1, o & 1};
            const float mn[3] = {c.bmin.x, c.bmin.y, c.bmin.z}, mx[3] = {c.bmax.x, c.bmax.y, c.bmax.z};
            for (int ax = 0; ax < 3; ++ax) {
                if (have_lo[ax][bit[ax]]) {
                    if (std::memcmp(&r.lo[ax][bit[ax]], &mn[ax], 4) != 0 || std::memcmp(&r.hi[ax][bit[ax]], &mx[ax], 4) != 0) {
                        err = "flatten: octant boxes of node " + std::to_string(i) + " do not share their planes";
                        return false;
                    }
                } else {
                    r.lo[ax][bit[ax]] = mn[ax];
                    r.hi[ax][bit[ax]] = mx[ax];
                    have_lo[ax][bit[ax]] = true;
                }
            }
            if (c.leaf) {
                r.info[o][0] = slot_of[ci];
                r.info[o][1] = c.tri_count;
            } else {
                r.info[o][0] = rec_of[ci];
                r.info[o][1] = kChildInternal;
            }
            r.nonempty |= 1u << o;
        }
        node_ordered_nonempty(r.nonempty, &r.pad[1]);
    }
    // small-tree form
    {
        uint32_t n_leaves = 0;
        for (const HostNode& n : t.nodes) n_leaves += n.leaf ? 1u : 0u;
        const size_t n_tri = t.nodes[0].tri_count;
        if (!t.nodes[0].leaf && n_leaves <= kFlatMaxLeaves && n_tri <= kFlatMaxTris) {
            std::vector<uint32_t> ordinal(nn, 0);
            uint32_t next = 0;
            for (size_t i = 0; i < nn; ++i)
                if (t.nodes[i].leaf) ordinal[i] = next++;
            out.flat_tris.resize(n_tri);
            for (size_t j = 0; j < n_tri; ++j) {
                const float* p = positions + j * 9;
                LeafTri& s = out.flat_tris[j];
                s.ax = p[0]; s.ay = p[1]; s.az = p[2];
                s.e1x = p[3] - p[0]; s.e1y = p[4] - p[1]; s.e1z = p[5] - p[2];
                s.e2x = p[6] - p[0]; s.e2y = p[7] - p[1]; s.e2z = p[8] - p[2];
                s.tri = (uint32_t)j;
                s.pad0 = s.pad1 = 0;
            }
            for (size_t i = 0; i < nn; ++i) {
                const HostNode& n = t.nodes[i];
                if (!n.leaf) continue;
                for (uint32_t k = 0; k < n.tri_count; ++k) out.flat_tris[t.pool[n.tri_begin + k]].pad0 |= 1u << ordinal[i];
            }
            out.flat_nodes = out.nodes;
            for (size_t i = 0; i < nn; ++i) {
                const HostNode& n = t.nodes[i];
                if (n.leaf) continue;
                NodeRec& r = out.flat_nodes[rec_of[i]];
                for (int o = 0; o < 8; ++o) {
                    const int32_t ci = n.child[o];
                    if (ci >= 0 && t.nodes[ci].tri_count != 0 && t.nodes[ci].leaf) r.info[o][0] = ordinal[ci];
                }
            }
        }
    }
    if (t.nodes[0].leaf) {
        out.root_info[0] = 0;
        out.root_info[1] = t.nodes[0].tri_count;
        out.max_stack = 0;
    } else {
        out.root_info[0] = 0;
        out.root_info[1] = kChildInternal;
        // Depth-first traversal that pushes every reached octant (leaves too): after popping a node at depth d the
        // stack holds at most 7 pending siblings per level above plus up to 8 fresh children.
        out.max_stack = 8 * deepest_internal + 8;
    }
    return true;
}

void cluster_octree(const NodeRec* nodes, size_t n_rec, const LeafTri* slots, size_t n_slots, const uint32_t root_info[2], ClusterForm& out) {
    out = ClusterForm();
    if (n_rec == 0 || !(root_info[1] & kChildInternal) || n_slots > kClusterSlotMask) return;
    // leaves below every internal node; records are in breadth-first order, so children follow their parents
    std::vector<uint32_t> below(n_rec, 0);
    for (size_t r = n_rec; r-- > 0;) {
        uint64_t sum = 0;
        for (int o = 0; o < 8; ++o) {
            const uint32_t i0 = nodes[r].info[o][0], i1 = nodes[r].info[o][1];
            if (i1 == 0) continue;
            sum += (i1 & kChildInternal) ? below[i0] : 1u;
        }
        below[r] = (uint32_t)std::min<uint64_t>(sum, 0xffffffffu);
    }
    out.nodes.assign(nodes, nodes + n_rec);
    for (NodeRec& r : out.nodes) r.pad[0] = r.pad[1] = r.pad[2] = 0;  // a record that owns no cluster: an empty list
    out.root_info1 = root_info[1];
    std::vector<uint32_t> roots;
    if (below[root_info[0]] <= kClusterMaxLeaves) {
        out.root_info1 |= kChildCluster;
        roots.push_back(root_info[0]);
    }
    for (size_t r = 0; r < n_rec; ++r) {
        if (below[r] <= kClusterMaxLeaves) continue;  // inside a cluster (or its root): its children are not roots
        for (int o = 0; o < 8; ++o) {
            const uint32_t i0 = nodes[r].info[o][0], i1 = nodes[r].info[o][1];
            if ((i1 & kChildInternal) && below[i0] <= kClusterMaxLeaves) {
                out.nodes[r].info[o][1] = i1 | kChildCluster;
                roots.push_back(i0);
            }
        }
    }
    bool any_repeat = false;
    std::vector<uint32_t> stack;
    std::vector<std::pair<uint32_t, uint32_t>> leaves;  // (first slot, position of its info pair = record * 8 + octant)
    std::vector<std::pair<uint32_t, uint32_t>> listed;  // (triangle, slot)
    // One cluster: numbers its leaves in slot order (= breadth-first leaf order), appends its distinct triangles in
    // ascending index, each with its leaf set, padded to whole blocks of 4 with records that no leaf lists, and writes
    // where that list is into the padding words of record `owner`.
    auto finish_cluster = [&](uint32_t owner) {
        std::sort(leaves.begin(), leaves.end());
        listed.clear();
        for (size_t k = 0; k < leaves.size(); ++k) {
            const uint32_t r = leaves[k].second >> 3, o = leaves[k].second & 7u;
            out.nodes[r].info[o][0] = leaves[k].first | ((uint32_t)k << kClusterOrdShift);
            for (uint32_t j = 0; j < nodes[r].info[o][1]; ++j)
                listed.emplace_back(slots[leaves[k].first + j].tri, leaves[k].first + j);
        }
        // the ordinal of a slot's leaf: the last leaf whose first slot is <= the slot
        auto leaf_of = [&](uint32_t slot) {
            size_t lo = 0, hi = leaves.size();
            while (hi - lo > 1) {
                const size_t mid = (lo + hi) / 2;
                if (leaves[mid].first <= slot) lo = mid;
                else hi = mid;
            }
            return (uint32_t)lo;
        };
        std::sort(listed.begin(), listed.end());
        const size_t first = out.tris.size();
        size_t a = 0;
        while (a < listed.size()) {
            size_t b = a;
            uint32_t set = 0;
            while (b < listed.size() && listed[b].first == listed[a].first) set |= 1u << leaf_of(listed[b++].second);
            any_repeat = any_repeat || b - a > 1;
            LeafTri t = slots[listed[a].second];
            t.pad0 = set;
            t.pad1 = 0;
            out.tris.push_back(t);
            a = b;
        }
        while ((out.tris.size() - first) % 4 != 0) {
            LeafTri t;
            std::memset(&t, 0, sizeof t);
            out.tris.push_back(t);
        }
        out.nodes[owner].pad[0] = (uint32_t)first;
        out.nodes[owner].pad[1] = (uint32_t)(out.tris.size() - first);
    };
    for (const uint32_t root : roots) {  // a subtree
        leaves.clear();
        stack.assign(1, root);
        while (!stack.empty()) {
            const uint32_t r = stack.back();
            stack.pop_back();
            for (int o = 0; o < 8; ++o) {
                const uint32_t i0 = nodes[r].info[o][0], i1 = nodes[r].info[o][1];
                if (i1 == 0) continue;
                if (i1 & kChildInternal) stack.push_back(i0);
                else leaves.emplace_back(i0, r * 8u + (uint32_t)o);
            }
        }
        finish_cluster(root);
    }
    for (size_t r = 0; r < n_rec; ++r) {  // the leaf children of a node above the subtrees
        if (below[r] <= kClusterMaxLeaves) continue;
        leaves.clear();
        for (int o = 0; o < 8; ++o) {
            const uint32_t i0 = nodes[r].info[o][0], i1 = nodes[r].info[o][1];
            if (i1 != 0 && !(i1 & kChildInternal)) leaves.emplace_back(i0, (uint32_t)r * 8u + (uint32_t)o);
        }
        if (!leaves.empty()) finish_cluster((uint32_t)r);
    }
    out.n_clusters = any_repeat ? (uint32_t)std::max<size_t>(roots.size(), 1) : 0u;
    if (out.n_clusters == 0) out = ClusterForm();
}

TreeShape tree_shape(const HostOctree& t) {
    TreeShape s;
    if (t.empty) return s;
    long long total_depth = 0, total_tris = 0;
    for (const HostNode& n : t.nodes) {
        s.nodes++;
        if (n.leaf) {
            s.leaf_refs += n.tri_count;
            continue;
        }
        s.internal++;
        for (int o = 0; o < 8; ++o) {
            if (n.child[o] < 0) continue;
            const HostNode& c = t.nodes[n.child[o]];
            if (!c.leaf) continue;
            s.leaves++;
            s.max_depth = std::max<uint64_t>(s.max_depth, c.depth);
            s.max_tris_in_leaf = std::max<uint64_t>(s.max_tris_in_leaf, c.tri_count);
            total_depth += c.depth;
            total_tris += c.tri_count;
        }
    }
    s.avg_leaf_depth = (float)total_depth / (float)s.leaves;
    s.avg_tris_in_leaf = (float)total_tris / (float)s.leaves;
    return s;
}

}  // namespace rt
