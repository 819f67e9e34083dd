// Internal layout of the opaque qf_bvh handle (shared by bvh_build.cpp and exact.hip).
#pragma once
#include <stdint.h>

#include <vector>

#define QF_BVH_MAX_HITS 64
// Triangles per leaf of the binary (build) tree.  The wide tree merges sibling subtrees of up to QF_BVH8_LEAF_MAX
// triangles into one leaf: its 8 lanes test 8 triangles at once.
#define QF_BVH_LEAF_MAX 4
#define QF_BVH8_LEAF_MAX 8
// The builder switches from SAH to halving the index range below QF_BVH_SAH_DEPTH, which bounds the depth of the
// binary tree by QF_BVH_SAH_DEPTH + ceil(log2(n_tri)) <= 32 + 30 for any input, however lopsided its SAH splits are.
#define QF_BVH_MAX_DEPTH 64
#define QF_BVH_SAH_DEPTH 32
// Largest per-ray traversal stack (entries) the wide traversal accepts; the builder computes the exact bound of a
// tree (sum over a root-to-leaf path of children - 1) and qf_bvh_create fails above this (never for a real mesh).
#define QF_BVH8_MAX_STACK 384

// Binary build tree, host only.  Node = 16 floats (64 B), both children's boxes inline:
//   [0..2] child0 lo   [3..5] child0 hi   [6..8] child1 lo   [9..11] child1 hi
//   [12] child0  [13] child1  (int bits: >= 0 inner node index, < 0 leaf with first triangle = ~child)
//   [14] count0  [15] count1  (int bits: triangles in the leaf, 0 for inner children)
//
// Wide tree (device + host mirror), the one that is traversed.  Node = 8 children x 8 floats (256 B):
//   child j at floats [8j .. 8j+7] = lo.xyz, hi.xyz, token (int bits), 0
//   token >= 0: inner child = node index;  token < 0 and != QF_BVH8_EMPTY: leaf = ~(first * 8 + count - 1), 1 <= count
//   <= 8, first < 2^28;  empty slot: token = QF_BVH8_EMPTY with an inverted box (lo = +inf, hi = -inf).
// Nodes are stored level by level (breadth first): level_start[l] .. level_start[l+1] are the nodes of level l, so a
// refit is one launch per level, bottom-up.
// Triangle = 3 x float4 in leaf order: (v0.xyz, original id bits), (v1.xyz, 0), (v2.xyz, 0).
#define QF_BVH8_EMPTY ((int32_t)0x80000000)

struct qf_bvh {
    float *d_nodes8 = nullptr;
    float *d_tris = nullptr;
    // triangle-chunk culling of the camera-coherent pass (exact.hip: chunk_boxes_kernel / cull_chunks_kernel), allocated
    // on first use: boxes of 64 consecutive leaf-order triangles; [2 counters | visible chunk ids]
    float *d_chunk_box = nullptr;
    int32_t *d_visible = nullptr;
    bool chunk_dirty = true;         // the triangles changed (build / refit): boxes are recomputed before the next use
    int cull_parity = 0;
    // depth-slab pass (qf_raster_intersect_slabs): per-chunk distance range, per-slab chunk lists, control block
    float *d_slab_range = nullptr;
    int32_t *d_slab_lists = nullptr;
    void *d_slab_ctl = nullptr;
    int32_t *d_slab_snapshot = nullptr;   // [rays] 16-byte records (direction | the pixel's count at the start of a slab pass)
    int64_t slab_snapshot_rays = 0;
    int64_t n_tri = 0;
    int64_t n_nodes = 0;             // binary build tree
    int64_t n_nodes8 = 0;            // wide tree
    int32_t max_depth = 0;           // deepest inner node of the binary tree (root = 1)
    int32_t max_stack8 = 1;          // exact bound of the wide traversal's per-ray stack for this tree
    float eps = 0.f;                 // box inflation (absolute), fixed at build time
    float min_sep = 0.f;             // > 0: the trimesh/Embree re-origin rule (qf_bvh_set_min_separation)
    // mean crossings of a random line through the mesh's bounding box, 2 area(mesh) / area(box) (Cauchy-Crofton), from
    // the build-time vertices: with fewer than max_hits / 2 the K-lists (almost) never fill and the general traversal
    // visits the hit children in ANY order (exact.hip, bvh_launch)
    float depth_complexity = 0.f;
    std::vector<float> h_nodes;      // binary tree (build, host refit, inspection)
    std::vector<float> h_nodes8;     // wide tree mirror
    std::vector<int32_t> level_start8;   // [levels + 1]
    std::vector<int32_t> h_tri_ids;  // leaf order -> original triangle id
};
