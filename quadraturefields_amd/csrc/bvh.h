// Internal layout of the opaque qf_bvh handle (shared by bvh_build.cpp and exact.hip).
#pragma once
#include <stdint.h>

#include <vector>

#define QF_BVH_MAX_HITS 64
#define QF_BVH_LEAF_MAX 4
// The traversal keeps one deferred sibling per level: its stack (exact.hip, kStack) holds QF_BVH_MAX_DEPTH entries.  The
// builder switches from SAH to halving the index range below QF_BVH_SAH_DEPTH, which bounds the depth by
// QF_BVH_SAH_DEPTH + ceil(log2(n_tri)) <= 32 + 30 for any input, however lopsided its SAH splits are.
#define QF_BVH_MAX_DEPTH 64
#define QF_BVH_SAH_DEPTH 32

// Node = 16 floats (64 B), both children's boxes inline (Aila-Laine style):
//   [0..2] child0 lo   [3..5] child0 hi   [6..8] child1 lo   [9..11] child1 hi
//   [12] child0  [13] child1  (int bits: >= 0 inner node index, < 0 leaf with first triangle = ~child)
//   [14] count0  [15] count1  (int bits: triangles in the leaf, 0 for inner children)
// Triangle = 3 x float4 in leaf order: (v0.xyz, original id bits), (v1.xyz, 0), (v2.xyz, 0).
struct qf_bvh {
    float *d_nodes = nullptr;
    float *d_tris = nullptr;
    int64_t n_tri = 0;
    int64_t n_nodes = 0;
    int32_t max_depth = 0;           // deepest inner node (root = 1)
    std::vector<float> h_nodes;      // host mirror (refit + inspection)
    std::vector<int32_t> h_tri_ids;  // leaf order -> original triangle id
    std::vector<int32_t> h_parent;   // parent node of each node (-1 for the root)
};
