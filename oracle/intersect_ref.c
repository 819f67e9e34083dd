/* Oracle: brute-force K-nearest multi-hit ray/triangle intersection.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never linked into the product.
 *
 * Stands in for trimesh 3.23.5 `ray_pyembree.RayMeshIntersector.intersects_id(...,
 * multiple_hits=True, max_hits=K)` (un-vendored; called at examples/mesh_utils.py:350-354)
 * and for the OptiX module behind `RayIntersector.intersects_id` (mesh_utils.py:86-109).
 * PARITY UNPINNED against Embree: the reference holds no vector for it.  Semantics fixed
 * here (DESIGN.md "intersection arithmetic contract"):
 *   - every triangle is tested against every ray with fp32 Moller-Trumbore in the exact
 *     operation order below, no FMA contraction (build with -ffp-contract=off);
 *   - both faces count (Embree default), a hit needs det != 0, 0<=u<=1, v>=0, u+v<=1, t>0;
 *   - per ray the K smallest hits by (t, triangle id) are kept, ascending.
 * Embree's re-origin step (skip hits closer than 1e-4*scale behind the previous one) is
 * NOT reproduced; on meshes without coincident faces the two agree.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline int mt_hit(const float *tri, const float *o, const float *d, float *t_out)
{
    const float v0x = tri[0], v0y = tri[1], v0z = tri[2];
    const float e1x = tri[3] - v0x, e1y = tri[4] - v0y, e1z = tri[5] - v0z;
    const float e2x = tri[6] - v0x, e2y = tri[7] - v0y, e2z = tri[8] - v0z;
    const float px = d[1] * e2z - d[2] * e2y;
    const float py = d[2] * e2x - d[0] * e2z;
    const float pz = d[0] * e2y - d[1] * e2x;
    const float det = (e1x * px + e1y * py) + e1z * pz;
    if (!(det != 0.0f)) return 0;            /* also rejects NaN */
    const float inv = 1.0f / det;
    const float tx = o[0] - v0x, ty = o[1] - v0y, tz = o[2] - v0z;
    const float u = ((tx * px + ty * py) + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = ((d[0] * qx + d[1] * qy) + d[2] * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    if (!(t > 0.0f)) return 0;
    *t_out = t;
    return 1;
}

/* tri_verts [n_tri][3][3], rays_o/rays_d [n_rays][3];
 * out_tri/out_t [n_rays][max_hits] (unused slots: -1 / +inf), out_count [n_rays]. */
int qf_oracle_multihit(const float *tri_verts, int64_t n_tri,
                       const float *rays_o, const float *rays_d, int64_t n_rays,
                       int max_hits, int n_threads, int32_t *out_tri, float *out_t, int32_t *out_count)
{
    if (max_hits <= 0 || n_tri < 0 || n_rays < 0) return -1;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (int64_t r = 0; r < n_rays; ++r) {
        int32_t *ht = out_tri + r * max_hits;
        float *tt = out_t + r * max_hits;
        int cnt = 0;
        for (int k = 0; k < max_hits; ++k) { ht[k] = -1; tt[k] = INFINITY; }
        for (int64_t f = 0; f < n_tri; ++f) {
            float t;
            if (!mt_hit(tri_verts + 9 * f, rays_o + 3 * r, rays_d + 3 * r, &t)) continue;
            /* insertion into the ascending (t, tri) list, dropping the largest if full */
            int pos = cnt;
            while (pos > 0 && (tt[pos - 1] > t || (tt[pos - 1] == t && ht[pos - 1] > (int32_t)f))) --pos;
            if (pos >= max_hits) continue;
            int last = cnt < max_hits ? cnt : max_hits - 1;
            for (int k = last; k > pos; --k) { tt[k] = tt[k - 1]; ht[k] = ht[k - 1]; }
            tt[pos] = t; ht[pos] = (int32_t)f;
            if (cnt < max_hits) ++cnt;
        }
        out_count[r] = cnt;
    }
    return 0;
}
