// Host-side BVH construction (binned SAH, binary, children's boxes stored in the parent) + upload.
//
// Replaces the acceleration-structure build hidden in trimesh.ray.ray_pyembree.RayMeshIntersector
// (examples/mesh_utils.py:223) and the OptiX `Intersector(vertices, max_hits, device)` constructor /
// `update_vertices` (examples/mesh_utils.py:77-84, examples/train_finetune.py:716-718).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "bvh.h"
#include "qf_common.h"

namespace {

constexpr int kBins = 16;

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
    }
    void grow(const float *p)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

Box tri_box(const float *v)
{
    Box b;
    b.reset();
    b.grow(v);
    b.grow(v + 3);
    b.grow(v + 6);
    return b;
}

inline int32_t as_int(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

void store_child(float *node, int side, const Box &b, float eps, int32_t child, int32_t count)
{
    float *lo = node + 6 * side, *hi = node + 6 * side + 3;
    for (int k = 0; k < 3; ++k) {
        // inflate so the fp32 slab test never culls a triangle the exact-order hit test accepts
        lo[k] = b.lo[k] - eps;
        hi[k] = b.hi[k] + eps;
    }
    node[12 + side] = as_float(child);
    node[14 + side] = as_float(count);
}

struct Task { int32_t node, begin, end, depth; };

// Chooses a partition of ids[begin,end) and returns mid.
int32_t partition(std::vector<int32_t> &ids, const std::vector<Box> &boxes, const std::vector<float> &cent,
                  int32_t begin, int32_t end)
{
    Box cb;
    cb.reset();
    for (int32_t i = begin; i < end; ++i) cb.grow(&cent[3 * (size_t)ids[i]]);
    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        const float ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.f)) continue;
        const float scale = kBins / ext;
        Box bin_box[kBins];
        int bin_cnt[kBins];
        for (int b = 0; b < kBins; ++b) { bin_box[b].reset(); bin_cnt[b] = 0; }
        for (int32_t i = begin; i < end; ++i) {
            const int32_t t = ids[i];
            int b = (int)((cent[3 * (size_t)t + axis] - cb.lo[axis]) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            bin_box[b].grow(boxes[t]);
            ++bin_cnt[b];
        }
        float right_area[kBins];
        int right_cnt[kBins];
        Box acc;
        acc.reset();
        int cnt = 0;
        for (int b = kBins - 1; b > 0; --b) {
            acc.grow(bin_box[b]);
            cnt += bin_cnt[b];
            right_area[b] = acc.half_area();
            right_cnt[b] = cnt;
        }
        acc.reset();
        cnt = 0;
        for (int b = 0; b < kBins - 1; ++b) {
            acc.grow(bin_box[b]);
            cnt += bin_cnt[b];
            if (cnt == 0 || right_cnt[b + 1] == 0) continue;
            const float cost = acc.half_area() * cnt + right_area[b + 1] * right_cnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
    }
    int32_t mid = -1;
    if (best_axis >= 0) {
        const float ext = cb.hi[best_axis] - cb.lo[best_axis];
        const float scale = kBins / ext;
        const float lo = cb.lo[best_axis];
        auto it = std::partition(ids.begin() + begin, ids.begin() + end, [&](int32_t t) {
            int b = (int)((cent[3 * (size_t)t + best_axis] - lo) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            return b <= best_bin;
        });
        mid = (int32_t)(it - ids.begin());
    }
    if (mid <= begin || mid >= end) {   // coincident centroids: split the index range in half
        mid = begin + (end - begin) / 2;
    }
    return mid;
}

float scene_eps(const float *tri_verts, int64_t n_tri)
{
    Box all;
    all.reset();
    for (int64_t i = 0; i < n_tri * 3; ++i) all.grow(tri_verts + 3 * i);
    float ext = 0.f, mag = 0.f;
    for (int k = 0; k < 3; ++k) {
        ext = std::max(ext, all.hi[k] - all.lo[k]);
        mag = std::max(mag, std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])));
    }
    if (!(ext >= 0.f) || !std::isfinite(ext)) ext = 0.f;
    return 2e-5f * std::max(ext, mag) + 1e-30f;
}

int upload(qf_bvh *bvh, const float *tri_verts)
{
    const int64_t n = bvh->n_tri;
    std::vector<float> tris((size_t)n * 12);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t id = bvh->h_tri_ids[(size_t)i];
        const float *v = tri_verts + 9 * (size_t)id;
        float *o = &tris[(size_t)i * 12];
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = as_float(id);
        o[4] = v[3]; o[5] = v[4]; o[6] = v[5]; o[7] = 0.f;
        o[8] = v[6]; o[9] = v[7]; o[10] = v[8]; o[11] = 0.f;
    }
    if (!bvh->d_nodes) QF_HIP_TRY(hipMalloc((void **)&bvh->d_nodes, std::max<size_t>(bvh->h_nodes.size(), 16) * sizeof(float)));
    if (!bvh->d_tris) QF_HIP_TRY(hipMalloc((void **)&bvh->d_tris, std::max<size_t>(tris.size(), 12) * sizeof(float)));
    if (!bvh->h_nodes.empty())
        QF_HIP_TRY(hipMemcpy(bvh->d_nodes, bvh->h_nodes.data(), bvh->h_nodes.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!tris.empty()) QF_HIP_TRY(hipMemcpy(bvh->d_tris, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice));
    return QF_OK;
}

void build_host(qf_bvh *bvh, const float *tri_verts, int64_t n_tri, int sah_depth = QF_BVH_SAH_DEPTH)
{
    bvh->n_tri = n_tri;
    bvh->h_tri_ids.resize((size_t)n_tri);
    std::iota(bvh->h_tri_ids.begin(), bvh->h_tri_ids.end(), 0);
    bvh->h_nodes.clear();
    if (n_tri == 0) { bvh->n_nodes = 0; return; }
    std::vector<Box> boxes((size_t)n_tri);
    std::vector<float> cent((size_t)n_tri * 3);
    for (int64_t i = 0; i < n_tri; ++i) {
        boxes[(size_t)i] = tri_box(tri_verts + 9 * i);
        for (int k = 0; k < 3; ++k) cent[3 * (size_t)i + k] = 0.5f * (boxes[(size_t)i].lo[k] + boxes[(size_t)i].hi[k]);
    }
    const float eps = scene_eps(tri_verts, n_tri);
    std::vector<int32_t> &ids = bvh->h_tri_ids;
    std::vector<float> &nodes = bvh->h_nodes;
    nodes.reserve((size_t)n_tri * 8);
    nodes.resize(16, 0.f);
    std::vector<Task> stack;
    if (n_tri <= QF_BVH_LEAF_MAX) {
        Box b;
        b.reset();
        for (int64_t i = 0; i < n_tri; ++i) b.grow(boxes[(size_t)i]);
        store_child(nodes.data(), 0, b, eps, ~0, (int32_t)n_tri);
        Box empty;
        empty.reset();
        store_child(nodes.data(), 1, empty, 0.f, ~0, 0);
    } else {
        stack.push_back({0, 0, (int32_t)n_tri, 1});
    }
    bvh->max_depth = n_tri > 0 ? 1 : 0;
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        bvh->max_depth = std::max(bvh->max_depth, t.depth);
        // below QF_BVH_SAH_DEPTH: halve the index range, so the remaining depth is at most log2(count)
        const int32_t mid = t.depth >= sah_depth ? t.begin + (t.end - t.begin) / 2
                                                        : partition(ids, boxes, cent, t.begin, t.end);
        const int32_t rng[2][2] = {{t.begin, mid}, {mid, t.end}};
        for (int side = 0; side < 2; ++side) {
            const int32_t b0 = rng[side][0], b1 = rng[side][1];
            Box b;
            b.reset();
            for (int32_t i = b0; i < b1; ++i) b.grow(boxes[(size_t)ids[i]]);
            if (b1 - b0 <= QF_BVH_LEAF_MAX) {
                store_child(&nodes[(size_t)t.node * 16], side, b, eps, ~b0, b1 - b0);
            } else {
                const int32_t child = (int32_t)(nodes.size() / 16);
                nodes.resize(nodes.size() + 16, 0.f);
                store_child(&nodes[(size_t)t.node * 16], side, b, eps, child, 0);
                stack.push_back({child, b0, b1, t.depth + 1});
            }
        }
    }
    bvh->n_nodes = (int64_t)(nodes.size() / 16);
}

// New vertex positions, same topology: recompute boxes bottom-up (children have larger indices).
void refit_host(qf_bvh *bvh, const float *tri_verts)
{
    const float eps = scene_eps(tri_verts, bvh->n_tri);
    std::vector<float> &nodes = bvh->h_nodes;
    for (int64_t n = bvh->n_nodes - 1; n >= 0; --n) {
        float *node = &nodes[(size_t)n * 16];
        for (int side = 0; side < 2; ++side) {
            const int32_t child = as_int(node[12 + side]);
            const int32_t count = as_int(node[14 + side]);
            Box b;
            b.reset();
            if (child < 0) {
                const int32_t first = ~child;
                for (int32_t k = 0; k < count; ++k) b.grow(tri_box(tri_verts + 9 * (size_t)bvh->h_tri_ids[(size_t)(first + k)]));
                store_child(node, side, b, count > 0 ? eps : 0.f, child, count);
            } else {
                const float *c = &nodes[(size_t)child * 16];
                // child boxes are already inflated; take their union as is
                for (int s = 0; s < 2; ++s) {
                    Box cb;
                    for (int k = 0; k < 3; ++k) { cb.lo[k] = c[6 * s + k]; cb.hi[k] = c[6 * s + 3 + k]; }
                    if (cb.lo[0] <= cb.hi[0]) b.grow(cb);
                }
                store_child(node, side, b, 0.f, child, 0);
            }
        }
    }
}

}  // namespace

extern "C" int qf_bvh_create(const float *tri_verts, int64_t n_tri, qf_bvh **out)
{
    return qf_bvh_create_ex(tri_verts, n_tri, QF_BVH_SAH_DEPTH, out);
}

extern "C" int qf_bvh_create_ex(const float *tri_verts, int64_t n_tri, int32_t sah_depth, qf_bvh **out)
{
    if (!out || n_tri < 0 || n_tri > 0x3fffffff || (n_tri > 0 && !tri_verts)) return QF_ERR_INVALID_ARGUMENT;
    if (sah_depth < 1 || sah_depth > QF_BVH_SAH_DEPTH) return QF_ERR_INVALID_ARGUMENT;
    qf_bvh *bvh = new (std::nothrow) qf_bvh();
    if (!bvh) return QF_ERR_INVALID_ARGUMENT;
    build_host(bvh, tri_verts, n_tri, sah_depth);
    if (bvh->max_depth > QF_BVH_MAX_DEPTH) { qf_bvh_destroy(bvh); return QF_ERR_UNSUPPORTED; }   // cannot happen, see bvh.h
    int rc = upload(bvh, tri_verts);
    if (rc != QF_OK) { qf_bvh_destroy(bvh); return rc; }
    *out = bvh;
    return QF_OK;
}

extern "C" int32_t qf_bvh_max_depth(const qf_bvh *bvh) { return bvh ? bvh->max_depth : -1; }

extern "C" int qf_bvh_refit(qf_bvh *bvh, const float *tri_verts, int64_t n_tri)
{
    if (!bvh || n_tri != bvh->n_tri || (n_tri > 0 && !tri_verts)) return QF_ERR_INVALID_ARGUMENT;
    if (n_tri == 0) return QF_OK;
    refit_host(bvh, tri_verts);
    return upload(bvh, tri_verts);
}

extern "C" void qf_bvh_destroy(qf_bvh *bvh)
{
    if (!bvh) return;
    if (bvh->d_nodes) (void)hipFree(bvh->d_nodes);
    if (bvh->d_tris) (void)hipFree(bvh->d_tris);
    delete bvh;
}

extern "C" int64_t qf_bvh_num_triangles(const qf_bvh *bvh) { return bvh ? bvh->n_tri : -1; }
extern "C" int64_t qf_bvh_num_nodes(const qf_bvh *bvh) { return bvh ? bvh->n_nodes : -1; }

extern "C" int qf_bvh_copy_nodes(const qf_bvh *bvh, float *nodes_host, int64_t capacity_nodes)
{
    if (!bvh || !nodes_host || capacity_nodes < bvh->n_nodes) return QF_ERR_INVALID_ARGUMENT;
    std::memcpy(nodes_host, bvh->h_nodes.data(), (size_t)bvh->n_nodes * 16 * sizeof(float));
    return QF_OK;
}

extern "C" int qf_bvh_copy_tri_ids(const qf_bvh *bvh, int32_t *ids_host, int64_t capacity)
{
    if (!bvh || !ids_host || capacity < bvh->n_tri) return QF_ERR_INVALID_ARGUMENT;
    std::memcpy(ids_host, bvh->h_tri_ids.data(), (size_t)bvh->n_tri * sizeof(int32_t));
    return QF_OK;
}
