/* Oracle: K-nearest multi-hit ray/triangle intersection -- brute force (the checker) and a host BVH walk (the same
 * answer, fast enough to be the CPU baseline of bench.py).
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never linked into the product.
 *
 * Stands in for trimesh 3.23.5 `ray_pyembree.RayMeshIntersector.intersects_id(...,
 * multiple_hits=True, max_hits=K)` (un-vendored; called at examples/mesh_utils.py:350-354)
 * and for the OptiX module behind `RayIntersector.intersects_id` (mesh_utils.py:86-109).
 * PARITY UNPINNED against Embree: the reference holds no vector for it.  Semantics fixed
 * here (DESIGN.md "intersection arithmetic contract"):
 *   - a triangle is tested against a ray with fp32 Moller-Trumbore in the exact
 *     operation order below, no FMA contraction (build with -ffp-contract=off);
 *   - both faces count (Embree default), a hit needs det != 0, 0<=u<=1, v>=0, u+v<=1, t>0;
 *   - a ray's hits are ordered by (t, triangle id);
 *   - trimesh's multi-hit loop: after each returned hit the ray is re-originated `min_sep` past it
 *     and the next Embree closest-hit query starts there, so with min_sep > 0 the first hit is kept and each later
 *     one iff t > t_last_kept + min_sep (one fp32 addition, one fp32 comparison); min_sep <= 0 keeps every hit;
 *   - the first K kept hits are returned, ascending.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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

typedef struct { float t; int32_t tri; } hit_t;

typedef struct { hit_t *h; int64_t n, cap; } hit_vec;

static void hv_push(hit_vec *v, float t, int32_t tri)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->h = (hit_t *)realloc(v->h, (size_t)v->cap * sizeof(hit_t));
    }
    v->h[v->n].t = t;
    v->h[v->n].tri = tri;
    ++v->n;
}

static int hit_cmp(const void *a, const void *b)
{
    const hit_t *x = (const hit_t *)a, *y = (const hit_t *)b;
    if (x->t < y->t) return -1;
    if (x->t > y->t) return 1;
    return (x->tri > y->tri) - (x->tri < y->tri);
}

/* all hits of a ray -> the row of the result (sort, re-origin chain, first K) */
static int finish_ray(hit_vec *v, int max_hits, float min_sep, int32_t *ht, float *tt)
{
    int cnt = 0;
    float last_t = 0.0f;
    qsort(v->h, (size_t)v->n, sizeof(hit_t), hit_cmp);
    for (int64_t i = 0; i < v->n && cnt < max_hits; ++i) {
        const float t = v->h[i].t;
        if (min_sep > 0.0f && cnt > 0) {
            const volatile float bound = last_t + min_sep;      /* rounded to fp32 before the comparison */
            if (!(t > bound)) continue;
        }
        tt[cnt] = t;
        ht[cnt] = v->h[i].tri;
        ++cnt;
        last_t = t;
    }
    for (int k = cnt; k < max_hits; ++k) { ht[k] = -1; tt[k] = INFINITY; }
    return cnt;
}

/* tri_verts [n_tri][3][3], rays_o/rays_d [n_rays][3];
 * out_tri/out_t [n_rays][max_hits] (unused slots: -1 / +inf), out_count [n_rays]. */
int qf_oracle_multihit(const float *tri_verts, int64_t n_tri,
                       const float *rays_o, const float *rays_d, int64_t n_rays,
                       int max_hits, float min_sep, int n_threads, int32_t *out_tri, float *out_t, int32_t *out_count)
{
    if (max_hits <= 0 || n_tri < 0 || n_rays < 0) return -1;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        hit_vec v = {0, 0, 0};
#pragma omp for schedule(dynamic, 4)
        for (int64_t r = 0; r < n_rays; ++r) {
            v.n = 0;
            for (int64_t f = 0; f < n_tri; ++f) {
                float t;
                if (mt_hit(tri_verts + 9 * f, rays_o + 3 * r, rays_d + 3 * r, &t)) hv_push(&v, t, (int32_t)f);
            }
            out_count[r] = finish_ray(&v, max_hits, min_sep, out_tri + r * max_hits, out_t + r * max_hits);
        }
        free(v.h);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Host BVH: median split on the longest axis of the centroid bounds, leaves of <= 4 triangles, boxes inflated by
 * 1e-4 x the scene extent and tested in double precision with a widened exit, so the walk can never cull a triangle
 * the exact test accepts; it collects ALL hits of a ray and finishes it exactly like the brute force.            */
typedef struct {
    float lo[3], hi[3];
    int32_t left, right;      /* inner: child nodes; leaf: left = -1 - first, right = count */
} bnode;

typedef struct {
    bnode *nodes;
    int32_t n_nodes;
    int32_t *order;           /* leaf order -> triangle id */
    float *tri;               /* [n_tri][9], a private copy */
    int64_t n_tri;
} obvh;

static const float *g_cent;
static int g_axis;
static int cent_cmp(const void *a, const void *b)
{
    const float x = g_cent[3 * (size_t)*(const int32_t *)a + g_axis], y = g_cent[3 * (size_t)*(const int32_t *)b + g_axis];
    return (x > y) - (x < y);
}

static void tri_bounds(const float *t, float *lo, float *hi)
{
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(t[k], t[3 + k]), t[6 + k]);
        hi[k] = fmaxf(fmaxf(t[k], t[3 + k]), t[6 + k]);
    }
}

void *qf_oracle_bvh_build(const float *tri_verts, int64_t n_tri)
{
    if (n_tri < 0 || n_tri > 0x3fffffff) return NULL;
    obvh *b = (obvh *)calloc(1, sizeof(obvh));
    b->n_tri = n_tri;
    b->tri = (float *)malloc((size_t)(n_tri > 0 ? n_tri : 1) * 9 * sizeof(float));
    memcpy(b->tri, tri_verts, (size_t)n_tri * 9 * sizeof(float));
    b->order = (int32_t *)malloc((size_t)(n_tri > 0 ? n_tri : 1) * sizeof(int32_t));
    b->nodes = (bnode *)malloc((size_t)(2 * n_tri + 2) * sizeof(bnode));
    float *cent = (float *)malloc((size_t)(n_tri > 0 ? n_tri : 1) * 3 * sizeof(float));
    float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n_tri; ++i) {
        float lo[3], hi[3];
        b->order[i] = (int32_t)i;
        tri_bounds(b->tri + 9 * i, lo, hi);
        for (int k = 0; k < 3; ++k) {
            cent[3 * i + k] = 0.5f * (lo[k] + hi[k]);
            slo[k] = fminf(slo[k], lo[k]);
            shi[k] = fmaxf(shi[k], hi[k]);
        }
    }
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, fmaxf(shi[k] - slo[k], fmaxf(fabsf(slo[k]), fabsf(shi[k]))));
    if (!(ext >= 0.0f) || isinf(ext)) ext = 0.0f;
    const float eps = 1e-4f * ext + 1e-30f;
    /* explicit work stack of (node, begin, end) */
    int32_t *st = (int32_t *)malloc((size_t)(3 * 128) * sizeof(int32_t));
    int sp = 0;
    b->n_nodes = 0;
    if (n_tri > 0) {
        b->n_nodes = 1;
        st[0] = 0; st[1] = 0; st[2] = (int32_t)n_tri;
        sp = 1;
    }
    g_cent = cent;
    while (sp > 0) {
        --sp;
        const int32_t node = st[3 * sp], begin = st[3 * sp + 1], end = st[3 * sp + 2];
        bnode *nd = &b->nodes[node];
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 3; ++k) { nd->lo[k] = INFINITY; nd->hi[k] = -INFINITY; }
        for (int32_t i = begin; i < end; ++i) {
            float lo[3], hi[3];
            tri_bounds(b->tri + 9 * (size_t)b->order[i], lo, hi);
            for (int k = 0; k < 3; ++k) {
                nd->lo[k] = fminf(nd->lo[k], lo[k] - eps);
                nd->hi[k] = fmaxf(nd->hi[k], hi[k] + eps);
                clo[k] = fminf(clo[k], cent[3 * (size_t)b->order[i] + k]);
                chi[k] = fmaxf(chi[k], cent[3 * (size_t)b->order[i] + k]);
            }
        }
        if (end - begin <= 4) {
            nd->left = -1 - begin;
            nd->right = end - begin;
            continue;
        }
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        g_axis = axis;
        qsort(b->order + begin, (size_t)(end - begin), sizeof(int32_t), cent_cmp);
        const int32_t mid = begin + (end - begin) / 2;
        nd->left = b->n_nodes++;
        nd->right = b->n_nodes++;
        /* depth <= log2(n) + 1 <= 31: the 128-entry work stack is ample */
        st[3 * sp] = nd->left; st[3 * sp + 1] = begin; st[3 * sp + 2] = mid; ++sp;
        st[3 * sp] = nd->right; st[3 * sp + 1] = mid; st[3 * sp + 2] = end; ++sp;
    }
    free(st);
    free(cent);
    return b;
}

void qf_oracle_bvh_free(void *p)
{
    obvh *b = (obvh *)p;
    if (!b) return;
    free(b->nodes); free(b->order); free(b->tri); free(b);
}

static inline double dinv(float d)
{
    double x = (double)d;
    if (fabs(x) < 1e-30) x = (x < 0.0 || (x == 0.0 && signbit(d))) ? -1e-30 : 1e-30;
    return 1.0 / x;
}

int qf_oracle_bvh_multihit(const void *p, const float *rays_o, const float *rays_d, int64_t n_rays, int max_hits,
                           float min_sep, int n_threads, int32_t *out_tri, float *out_t, int32_t *out_count)
{
    const obvh *b = (const obvh *)p;
    if (!b || max_hits <= 0 || n_rays < 0) return -1;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        hit_vec v = {0, 0, 0};
        int32_t stack[128];
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n_rays; ++r) {
            const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
            const double inv[3] = {dinv(d[0]), dinv(d[1]), dinv(d[2])};
            int sp = 0;
            v.n = 0;
            if (b->n_nodes > 0) stack[sp++] = 0;
            while (sp > 0) {
                const bnode *nd = &b->nodes[stack[--sp]];
                double tn = 0.0, tf = INFINITY;
                for (int k = 0; k < 3; ++k) {
                    const double a = ((double)nd->lo[k] - (double)o[k]) * inv[k], c = ((double)nd->hi[k] - (double)o[k]) * inv[k];
                    tn = fmax(tn, fmin(a, c));
                    tf = fmin(tf, fmax(a, c));
                }
                if (tn > tf * 1.000001 + 1e-30) continue;
                if (nd->left < 0) {
                    const int32_t first = -1 - nd->left;
                    for (int32_t k = 0; k < nd->right; ++k) {
                        const int32_t id = b->order[first + k];
                        float t;
                        if (mt_hit(b->tri + 9 * (size_t)id, o, d, &t)) hv_push(&v, t, id);
                    }
                } else {
                    stack[sp++] = nd->left;
                    stack[sp++] = nd->right;
                }
            }
            out_count[r] = finish_ray(&v, max_hits, min_sep, out_tri + r * max_hits, out_t + r * max_hits);
        }
        free(v.h);
    }
    return 0;
}
