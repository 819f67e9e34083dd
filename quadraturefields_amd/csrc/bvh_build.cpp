// Host-side BVH construction: binned-SAH binary tree (children's boxes stored in the parent), collapsed into the
// 8-wide tree the device traverses (bvh.h), upload, and refits (host vertices or device vertices).
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
    // the caller may have kernels in flight on non-blocking streams that read the old tree: wait for them
    QF_HIP_TRY(hipDeviceSynchronize());
    if (!bvh->d_nodes8) QF_HIP_TRY(hipMalloc((void **)&bvh->d_nodes8, std::max<size_t>(bvh->h_nodes8.size(), 64) * sizeof(float)));
    if (!bvh->d_tris) QF_HIP_TRY(hipMalloc((void **)&bvh->d_tris, std::max<size_t>(tris.size(), 12) * sizeof(float)));
    if (!bvh->h_nodes8.empty())
        QF_HIP_TRY(hipMemcpy(bvh->d_nodes8, bvh->h_nodes8.data(), bvh->h_nodes8.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!tris.empty()) QF_HIP_TRY(hipMemcpy(bvh->d_tris, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice));
    bvh->chunk_dirty = true;
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
    const float eps = bvh->eps = scene_eps(tri_verts, n_tri);
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
    const float eps = bvh->eps = scene_eps(tri_verts, bvh->n_tri);
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


// ---------------------------------------------------------------------------------------------------------------------
// The 8-wide tree (bvh.h).  Built from the binary tree: a wide node starts from a binary node's two children and keeps
// replacing the largest-area child that still holds more than QF_BVH8_LEAF_MAX triangles by its own two children until
// it has 8; a child subtree of at most 8 triangles becomes ONE leaf (its triangles are contiguous in leaf order).

struct Ref { float lo[3], hi[3]; int32_t child, count; };   // binary encoding: child >= 0 inner, < 0 leaf ~first

inline Ref bin_ref(const float *node, int side)
{
    Ref r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = node[6 * side + k]; r.hi[k] = node[6 * side + 3 + k]; }
    r.child = as_int(node[12 + side]);
    r.count = as_int(node[14 + side]);
    return r;
}

inline float ref_area(const Ref &r)
{
    const float dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

inline int32_t leaf_token(int32_t first, int32_t count) { return ~((first << 3) | ((count - 1) & 7)); }

void set_empty8(float *c)
{
    const float inf = std::numeric_limits<float>::infinity();
    c[0] = c[1] = c[2] = inf;
    c[3] = c[4] = c[5] = -inf;
    c[6] = as_float(QF_BVH8_EMPTY);
    c[7] = 0.f;
}

int collapse8(qf_bvh *bvh)
{
    bvh->h_nodes8.clear();
    bvh->level_start8.clear();
    bvh->n_nodes8 = 0;
    bvh->max_stack8 = 1;
    if (bvh->n_tri == 0) return QF_OK;
    const std::vector<float> &bn = bvh->h_nodes;
    const int64_t nb = bvh->n_nodes;
    // triangle range of every binary inner node (children have larger indices than their parent)
    std::vector<int32_t> first((size_t)nb), total((size_t)nb);
    for (int64_t n = nb - 1; n >= 0; --n) {
        int32_t f = 0x7fffffff, t = 0;
        for (int side = 0; side < 2; ++side) {
            const int32_t c = as_int(bn[(size_t)n * 16 + 12 + side]), k = as_int(bn[(size_t)n * 16 + 14 + side]);
            if (c < 0) { if (k > 0) { f = std::min(f, ~c); t += k; } }
            else { f = std::min(f, first[(size_t)c]); t += total[(size_t)c]; }
        }
        first[(size_t)n] = f;
        total[(size_t)n] = t;
    }
    std::vector<float> &wn = bvh->h_nodes8;
    std::vector<int32_t> queue;          // binary node behind each wide node, in breadth-first order
    queue.push_back(0);
    bvh->level_start8.push_back(0);
    size_t head = 0;
    while (head < queue.size()) {
        const size_t level_end = queue.size();
        for (; head < level_end; ++head) {
            const int32_t b = queue[head];
            Ref refs[8];
            int n_ref = 0;
            refs[n_ref++] = bin_ref(&bn[(size_t)b * 16], 0);
            refs[n_ref++] = bin_ref(&bn[(size_t)b * 16], 1);
            while (n_ref < 8) {
                int best = -1;
                float best_area = -1.f;
                for (int i = 0; i < n_ref; ++i) {
                    if (refs[i].child < 0 || total[(size_t)refs[i].child] <= QF_BVH8_LEAF_MAX) continue;
                    const float a = ref_area(refs[i]);
                    if (a > best_area) { best_area = a; best = i; }
                }
                if (best < 0) break;
                const int32_t c = refs[best].child;
                refs[best] = bin_ref(&bn[(size_t)c * 16], 0);
                refs[n_ref++] = bin_ref(&bn[(size_t)c * 16], 1);
            }
            const size_t at = wn.size();
            wn.resize(at + 64);
            int slot = 0;
            for (int i = 0; i < n_ref; ++i) {
                const Ref &r = refs[i];
                int32_t token;
                if (r.child < 0) {
                    if (r.count <= 0) continue;                          // the empty sibling of a tiny mesh's root
                    token = leaf_token(~r.child, r.count);
                } else if (total[(size_t)r.child] <= QF_BVH8_LEAF_MAX) {
                    token = leaf_token(first[(size_t)r.child], total[(size_t)r.child]);
                } else {
                    token = (int32_t)queue.size();
                    queue.push_back(r.child);
                }
                float *c = &wn[at + 8 * (size_t)slot++];
                for (int k = 0; k < 3; ++k) { c[k] = r.lo[k]; c[3 + k] = r.hi[k]; }
                c[6] = as_float(token);
                c[7] = 0.f;
            }
            for (; slot < 8; ++slot) set_empty8(&wn[at + 8 * (size_t)slot]);
        }
        bvh->level_start8.push_back((int32_t)level_end);     // = start of the next level, or the node count at the end
    }
    bvh->n_nodes8 = (int64_t)queue.size();
    // exact stack bound: need(n) = max(h, h - 1 + max over inner children need(c)), h = children of n
    std::vector<int32_t> need((size_t)bvh->n_nodes8, 0);
    for (int64_t n = bvh->n_nodes8 - 1; n >= 0; --n) {
        int h = 0, deepest = 0;
        for (int j = 0; j < 8; ++j) {
            const int32_t tok = as_int(wn[(size_t)n * 64 + 8 * j + 6]);
            if (tok == QF_BVH8_EMPTY) continue;
            ++h;
            if (tok >= 0) deepest = std::max(deepest, need[(size_t)tok]);
        }
        need[(size_t)n] = std::max(h, h - 1 + deepest);
    }
    bvh->max_stack8 = std::max(1, need[0]) + 1;
    return bvh->max_stack8 <= QF_BVH8_MAX_STACK ? QF_OK : QF_ERR_UNSUPPORTED;
}

// New vertex positions: every wide child box again from its triangles (leaf order), bottom-up level by level.
void refit8_host(qf_bvh *bvh, const float *tri_verts)
{
    std::vector<float> &wn = bvh->h_nodes8;
    const float eps = bvh->eps;
    for (int64_t n = bvh->n_nodes8 - 1; n >= 0; --n) {       // children have larger indices (breadth-first order)
        for (int j = 0; j < 8; ++j) {
            float *c = &wn[(size_t)n * 64 + 8 * (size_t)j];
            const int32_t tok = as_int(c[6]);
            if (tok == QF_BVH8_EMPTY) continue;
            Box b;
            b.reset();
            if (tok < 0) {
                const int32_t packed = ~tok, first = packed >> 3, cnt = (packed & 7) + 1;
                for (int32_t k = 0; k < cnt; ++k) b.grow(tri_box(tri_verts + 9 * (size_t)bvh->h_tri_ids[(size_t)(first + k)]));
                for (int k = 0; k < 3; ++k) { c[k] = b.lo[k] - eps; c[3 + k] = b.hi[k] + eps; }
            } else {
                const float *ch = &wn[(size_t)tok * 64];
                for (int q = 0; q < 8; ++q) {
                    if (as_int(ch[8 * q + 6]) == QF_BVH8_EMPTY) continue;
                    Box cb;
                    for (int k = 0; k < 3; ++k) { cb.lo[k] = ch[8 * q + k]; cb.hi[k] = ch[8 * q + 3 + k]; }
                    b.grow(cb);
                }
                for (int k = 0; k < 3; ++k) { c[k] = b.lo[k]; c[3 + k] = b.hi[k]; }
            }
        }
    }
}

// ---- device-side refit (training: the vertices live on the device, train_finetune.py:708-718) -------------------------
__global__ void refit_tris_kernel(float4 *tris, const float *verts, int64_t n_tri)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_tri; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = tris[i * 3];
        const int64_t id = __float_as_int(a.w);
        const float *v = verts + id * 9;
        tris[i * 3 + 0] = make_float4(v[0], v[1], v[2], a.w);
        tris[i * 3 + 1] = make_float4(v[3], v[4], v[5], 0.f);
        tris[i * 3 + 2] = make_float4(v[6], v[7], v[8], 0.f);
    }
}

// one thread per (node, child) of the level [node0, node1)
__global__ void refit_level_kernel(float *nodes, const float4 *tris, int node0, int node1, float eps)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = node0 + (gid >> 3);
    const int j = (int)(gid & 7);
    if (n >= node1) return;
    float *c = nodes + n * 64 + 8 * j;
    const int tok = __float_as_int(c[6]);
    if (tok == QF_BVH8_EMPTY) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (tok < 0) {
        const int packed = ~tok, first = packed >> 3, cnt = (packed & 7) + 1;
        for (int k = 0; k < cnt; ++k) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float4 p = tris[(int64_t)(first + k) * 3 + v];
                lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
                hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] -= eps; hi[k] += eps; }
    } else {
        const float *ch = nodes + (int64_t)tok * 64;
        for (int q = 0; q < 8; ++q) {
            if (__float_as_int(ch[8 * q + 6]) == QF_BVH8_EMPTY) continue;
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], ch[8 * q + k]); hi[k] = fmaxf(hi[k], ch[8 * q + 3 + k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { c[k] = lo[k]; c[3 + k] = hi[k]; }
}

}  // namespace

extern "C" int qf_bvh_create(const float *tri_verts, int64_t n_tri, qf_bvh **out)
{
    return qf_bvh_create_ex(tri_verts, n_tri, QF_BVH_SAH_DEPTH, out);
}

extern "C" int qf_bvh_create_ex(const float *tri_verts, int64_t n_tri, int32_t sah_depth, qf_bvh **out)
{
    if (!out || n_tri < 0 || n_tri >= (1 << 28) || (n_tri > 0 && !tri_verts)) return QF_ERR_INVALID_ARGUMENT;
    if (sah_depth < 1 || sah_depth > QF_BVH_SAH_DEPTH) return QF_ERR_INVALID_ARGUMENT;
    qf_bvh *bvh = new (std::nothrow) qf_bvh();
    if (!bvh) return QF_ERR_INVALID_ARGUMENT;
    build_host(bvh, tri_verts, n_tri, sah_depth);
    {   // depth complexity (see bvh.h): triangle areas and the bounding box in double
        double area = 0.0, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int64_t t = 0; t < n_tri; ++t) {
            const float *v = tri_verts + 9 * (size_t)t;
            const double e1[3] = {(double)v[3] - v[0], (double)v[4] - v[1], (double)v[5] - v[2]};
            const double e2[3] = {(double)v[6] - v[0], (double)v[7] - v[1], (double)v[8] - v[2]};
            const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            area += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            for (int k = 0; k < 9; ++k) { lo[k % 3] = std::min(lo[k % 3], (double)v[k]); hi[k % 3] = std::max(hi[k % 3], (double)v[k]); }
        }
        const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        const double box = n_tri > 0 ? 2.0 * (ex * ey + ey * ez + ex * ez) : 0.0;
        bvh->depth_complexity = box > 0.0 ? (float)(2.0 * area / box) : 0.f;
    }
    if (bvh->max_depth > QF_BVH_MAX_DEPTH) { qf_bvh_destroy(bvh); return QF_ERR_UNSUPPORTED; }   // cannot happen, see bvh.h
    int rc = collapse8(bvh);
    if (rc == QF_OK) rc = upload(bvh, tri_verts);
    if (rc != QF_OK) { qf_bvh_destroy(bvh); return rc; }
    *out = bvh;
    return QF_OK;
}

extern "C" int32_t qf_bvh_max_depth(const qf_bvh *bvh) { return bvh ? bvh->max_depth : -1; }
extern "C" int32_t qf_bvh_max_stack(const qf_bvh *bvh) { return bvh ? bvh->max_stack8 : -1; }
extern "C" int64_t qf_bvh_num_wide_nodes(const qf_bvh *bvh) { return bvh ? bvh->n_nodes8 : -1; }

extern "C" int qf_bvh_set_min_separation(qf_bvh *bvh, float min_separation)
{
    if (!bvh || min_separation != min_separation) return QF_ERR_INVALID_ARGUMENT;
    bvh->min_sep = min_separation > 0.f ? min_separation : 0.f;
    return QF_OK;
}

extern "C" float qf_bvh_min_separation(const qf_bvh *bvh) { return bvh ? bvh->min_sep : -1.f; }

extern "C" int qf_bvh_refit(qf_bvh *bvh, const float *tri_verts, int64_t n_tri)
{
    if (!bvh || n_tri != bvh->n_tri || (n_tri > 0 && !tri_verts)) return QF_ERR_INVALID_ARGUMENT;
    if (n_tri == 0) return QF_OK;
    refit_host(bvh, tri_verts);
    refit8_host(bvh, tri_verts);
    return upload(bvh, tri_verts);      // waits for the device first: no traversal may still be reading the old tree
}

extern "C" int qf_bvh_refit_device(qf_bvh *bvh, const float *d_tri_verts, int64_t n_tri, void *stream)
{
    if (!bvh || n_tri != bvh->n_tri || (n_tri > 0 && !d_tri_verts)) return QF_ERR_INVALID_ARGUMENT;
    if (n_tri == 0) return QF_OK;
    hipStream_t st = qf_stream(stream);
    hipLaunchKernelGGL(refit_tris_kernel, dim3(qf_grid_1d(n_tri, 256)), dim3(256), 0, st,
                       reinterpret_cast<float4 *>(bvh->d_tris), d_tri_verts, n_tri);
    QF_LAUNCH_CHECK();
    const int levels = (int)bvh->level_start8.size() - 1;
    for (int l = levels - 1; l >= 0; --l) {
        const int n0 = bvh->level_start8[(size_t)l], n1 = bvh->level_start8[(size_t)l + 1];
        if (n1 <= n0) continue;
        const int64_t threads = (int64_t)(n1 - n0) * 8;
        hipLaunchKernelGGL(refit_level_kernel, dim3((unsigned)qf_div_up(threads, 256)), dim3(256), 0, st, bvh->d_nodes8,
                           reinterpret_cast<const float4 *>(bvh->d_tris), n0, n1, bvh->eps * 1.25f);
        QF_LAUNCH_CHECK();
    }
    bvh->chunk_dirty = true;
    // eps was sized from the extent at build / host-refit time; 1.25x covers the bounded vertex motion of training.
    // The host mirrors (h_nodes, h_nodes8) are NOT updated: inspection copies show the build state.
    return QF_OK;
}

extern "C" void qf_bvh_destroy(qf_bvh *bvh)
{
    if (!bvh) return;
    if (bvh->d_nodes8) (void)hipFree(bvh->d_nodes8);
    if (bvh->d_tris) (void)hipFree(bvh->d_tris);
    if (bvh->d_chunk_box) (void)hipFree(bvh->d_chunk_box);
    if (bvh->d_visible) (void)hipFree(bvh->d_visible);
    if (bvh->d_slab_range) (void)hipFree(bvh->d_slab_range);
    if (bvh->d_slab_lists) (void)hipFree(bvh->d_slab_lists);
    if (bvh->d_slab_ctl) (void)hipFree(bvh->d_slab_ctl);
    if (bvh->d_slab_snapshot) (void)hipFree(bvh->d_slab_snapshot);
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

extern "C" int qf_bvh_copy_wide_nodes(const qf_bvh *bvh, float *nodes_host, int64_t capacity_nodes)
{
    if (!bvh || !nodes_host || capacity_nodes < bvh->n_nodes8) return QF_ERR_INVALID_ARGUMENT;
    std::memcpy(nodes_host, bvh->h_nodes8.data(), (size_t)bvh->n_nodes8 * 64 * sizeof(float));
    return QF_OK;
}
