// Index-exact kernels for gfx950: BVH multi-hit traversal, sample packing / per-ray sorting, texel
// lookup and baked-texture decode.  This translation unit is compiled with -ffp-contract=off: every
// comparison that decides an INTEGER output (triangle id, hit count, sample order, texel index) uses a
// fixed sequence of individually rounded IEEE operations, restated independently by the oracle
// (oracle/intersect_ref.c, oracle/quantize.py), so those outputs are bit-exact against it.
//
// Replaces: trimesh/Embree `intersects_id` and the OptiX
// `Intersector.find_intersections` (examples/mesh_utils.py:77-96,350-354), the numpy argsort/lexsort
// of sampling_raytrace_numpy / sampling_indexing (mesh_utils.py:359-381,394-403),
// trimesh.triangles.points_to_barycentric + UV lookup (examples/utils.py:1055-1063) and
// FeatureCompression.get_features_from_texture_map (examples/texture_utils.py:149-175).
#include "qf_common.h"
#include "bvh.h"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxHits = QF_BVH_MAX_HITS;

// fp32 Moller-Trumbore, operation order shared verbatim (as a contract, not as code) with the oracle.
__device__ __forceinline__ bool mt_hit(const float4 a, const float4 b, const float4 c, const float ox, const float oy,
                                       const float oz, const float dx, const float dy, const float dz, float *t_out)
{
    const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
    const float px = dy * e2z - dz * e2y;
    const float py = dz * e2x - dx * e2z;
    const float pz = dx * e2y - dy * e2x;
    const float det = (e1x * px + e1y * py) + e1z * pz;
    if (!(det != 0.0f)) return false;
    const float inv = 1.0f / det;
    const float tx = ox - a.x, ty = oy - a.y, tz = oz - a.z;
    const float u = ((tx * px + ty * py) + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = ((dx * qx + dy * qy) + dz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    if (!(t > 0.0f)) return false;
    *t_out = t;
    return true;
}

__device__ __forceinline__ float safe_inv(float d)
{
    const float tiny = 1e-30f;
    if (fabsf(d) < tiny) d = (d < 0.0f || (d == 0.0f && signbit(d))) ? -tiny : tiny;
    return 1.0f / d;
}

// (t, tri) lexicographic "a sorts before b"
__device__ __forceinline__ bool hit_less(float ta, int ia, float tb, int ib) { return ta < tb || (ta == tb && ia < ib); }

// Four consecutive list entries as ONE memory request.  A ray's list starts at ray * K entries, 4-byte aligned only
// (K = 25 is the default everywhere), and gfx9+ global memory takes dwordx4 accesses at dword alignment: these types
// make the compiler emit them.  The kernels that walk [ray][K] lists lane = ray are bound by the number of scattered
// per-lane requests (measured: ~4.75 us per million), so a row costs K / 4 of them instead of K.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));

// A hit as one 64-bit key: t > 0, so its bit pattern orders like its value, and the triangle id breaks ties --
// key order IS the (t, tri) order of the contract.
__device__ __forceinline__ uint64_t hit_key(float t, int id) { return ((uint64_t)__float_as_uint(t) << 32) | (uint32_t)id; }
__device__ __forceinline__ float key_t(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ int key_id(uint64_t k) { return (int)(uint32_t)k; }

// A lane's hit list (cnt <= 32 entries, contiguous in LDS) sorted ascending in t (kTri = false) or (t, tri) (kTri = true)
// THROUGH REGISTERS: a 32-key bitonic network, every index a compile-time constant, 240 compare-exchanges with no
// memory in between.  The per-lane insertion sort it replaces walked the row in LDS, one dependent read-modify-write per
// shift: ~18 us of a tile wave's ~40 us, which is what a row band of a frame sharded over 8 GPUs (one wave per SIMD,
// nothing to overlap with) waited for.  Same order: t > 0, so the bit pattern of t orders like its value, +inf pads the
// tail; hits with equal (t, tri) do not exist (a ray meets a triangle once), equal t alone are interchangeable samples.
template <bool kTri>
__device__ __forceinline__ void sort_row_32(float *row_t, int32_t *row_i, int cnt)
{
    if (kTri) {
        uint64_t key[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) key[k] = k < cnt ? hit_key(row_t[k], row_i[k]) : ~0ull;
#pragma unroll
        for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int l = i ^ (k - 1);
                if (l > i) { const uint64_t a = key[i], b = key[l]; key[i] = a < b ? a : b; key[l] = a < b ? b : a; }
            }
#pragma unroll
            for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int l = i ^ j;
                    if (l > i) { const uint64_t a = key[i], b = key[l]; key[i] = a < b ? a : b; key[l] = a < b ? b : a; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < cnt) { row_t[k] = key_t(key[k]); row_i[k] = key_id(key[k]); }
    } else {
        float key[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) key[k] = k < cnt ? row_t[k] : INFINITY;
#pragma unroll
        for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int l = i ^ (k - 1);
                if (l > i) { const float a = key[i], b = key[l]; key[i] = fminf(a, b); key[l] = fmaxf(a, b); }
            }
#pragma unroll
            for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int l = i ^ j;
                    if (l > i) { const float a = key[i], b = key[l]; key[i] = fminf(a, b); key[l] = fmaxf(a, b); }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < cnt) row_t[k] = key[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide-BVH multi-hit traversal: EIGHT LANES PER RAY.  A node of the 8-wide tree (bvh.h) is 8 children x 32 B; lane j of
// a ray's octet loads child j (the octet reads the node's 256 contiguous bytes), tests its box, and the octet's hit
// mask comes out of one ballot.  A leaf holds up to 8 triangles: lane j runs the exact test on triangle j.  So one
// dependent step decides 8 boxes or 8 triangles (the binary one-ray-per-lane walk of round 1 needed ~3 dependent node
// fetches for the same decision and kept a 64-entry stack per LANE in scratch), a wave carries 8 rays instead of 64 --
// eight times the waves for the same batch, which is what a latency-bound walk over a 2^17-ray training batch lacks --
// and the per-ray state (stack, K-list) lives in LDS, shared by the octet:
//   * stack: tokens of the hit children not taken yet; the nearest hit child is taken next (DPP min over the octet);
//     the capacity is the tree's exact bound, computed by the builder (qf_bvh::max_stack8);
//   * K-list: 64-bit (t, tri) keys, unordered while there is room, then K-nearest replacement with the worst entry
//     found by the octet together; t_limit = the worst entry's t prunes boxes;
//   * at the end the octet rank-sorts the list (each lane ranks every 8th entry) and writes the row ascending.
// min_sep > 0 adds the reference's multi-hit rule (trimesh 3.23.5 ray_pyembree.intersects_id, called at
// examples/mesh_utils.py:350-354): after a kept hit at t_prev the ray is re-originated min_sep past it, so the next
// kept hit is the first with t > t_prev + min_sep -- hits closer than that, and the second copy of a duplicated face,
// are never returned.  The chain runs over the sorted list; when the K-list was full and the chain kept fewer than K
// the traversal runs again for the hits beyond the page (lower bound = the page's last key), until K are kept or a
// page comes back not full.
constexpr int kOctRays = 32;                    // rays per workgroup: 256 threads = 4 waves x 8 octets
constexpr int kTravThreads = kOctRays * 8;
constexpr int kDone = (int)0x80000000;          // == QF_BVH8_EMPTY; no leaf token takes this value (n_tri < 2^28)
constexpr int kMaxPages = 4096;

#define QF_DPP_QUAD_1032 0xB1
#define QF_DPP_QUAD_2301 0x4E
#define QF_DPP_HALF_MIRROR 0x141

__device__ __forceinline__ unsigned oct_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_QUAD_1032, 0xf, 0xf, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_QUAD_2301, 0xf, 0xf, true));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_HALF_MIRROR, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ unsigned oct_max_u32(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_QUAD_1032, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_QUAD_2301, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, QF_DPP_HALF_MIRROR, 0xf, 0xf, true));
    return v;
}
template <int kCtrl>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, kCtrl, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), kCtrl, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t oct_max_u64(uint64_t v)
{
    uint64_t o = dpp_u64<QF_DPP_QUAD_1032>(v); v = o > v ? o : v;
    o = dpp_u64<QF_DPP_QUAD_2301>(v); v = o > v ? o : v;
    o = dpp_u64<QF_DPP_HALF_MIRROR>(v); v = o > v ? o : v;
    return v;
}
// value of lane `src` (0..7) of this octet
__device__ __forceinline__ int oct_bcast(int v, int oct_base, int src)
{
    return __builtin_amdgcn_ds_bpermute((oct_base + src) << 2, v);
}
// LDS written by one lane of the octet is read by the others: same wave, LDS operations complete in order; this only
// keeps the compiler from moving them across each other
__device__ __forceinline__ void oct_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct OctList {
    uint64_t *keys;      // [K] in LDS
    int K, j;
    int count;           // entries in keys (octet-uniform)
    uint64_t worst;      // valid when count == K: the largest key, at worst_slot
    int worst_slot;

    __device__ __forceinline__ void find_worst()
    {
        uint64_t m = 0;
        int slot = 0;
        for (int i = j; i < K; i += 8) {
            const uint64_t k = keys[i];
            if (k >= m) { m = k; slot = i; }
        }
        worst = oct_max_u64(m);
        worst_slot = (int)oct_max_u32(m == worst ? (unsigned)slot : 0u);     // keys are unique: one lane holds it
    }
};

struct TravArgs {
    const float4 *nodes, *tris;
    const float *rays_o, *rays_d;
    int64_t n_rays;
    int root_is_valid, max_hits, image_width, image_height, tiles_x, n_blocks, blocks_per_xcd, stack_cap;
    int stripe_blocks;           // > 0: image-shaped launch, blocks per stripe of the XCD comb (see xcd_block)
    int list_cap;                // entries of the LDS K-list: max_hits, or a few more when the re-origin rule is on
    float min_sep;
    int32_t *hit_tri;
    float *hit_t;
    int32_t *hit_count;
    uint64_t *keep_mask;         // repair launch only (see bvh8_repair_kernel)
    int32_t *raw_count;
    int tcol_offset;             // repair launch: float offset of the [K][256] distance columns in the LDS
    const int32_t *all_flag;     // repair launch, or NULL: *all_flag != 0 -> EVERY ray is traversed (camera_rays_check)
};

// LDS of a workgroup: [kOctRays][K] keys | [kOctRays][K] sorted (only when min_sep > 0) | [kOctRays][stack_cap] stack
extern __shared__ uint64_t trav_lds[];

// The re-origin rule for a ray whose complete, unordered list (c <= K entries) is in hit_t / hit_tri, WITHOUT rewriting
// the list: bit i of the result = the i-th hit in (t, tri) order is kept.  Octet-uniform; *kept_out = number kept.
__device__ __forceinline__ uint64_t oct_keep_mask(const TravArgs &a, int64_t ray, int c, int j, int q, int *kept_out)
{
    const int K = a.max_hits, Kc = a.list_cap;
    uint64_t *keys = trav_lds + (size_t)q * Kc;
    uint64_t *sorted = trav_lds + (size_t)kOctRays * Kc + (size_t)q * Kc;
    for (int e = j; e < c; e += 8) keys[e] = hit_key(a.hit_t[ray * K + e], a.hit_tri[ray * K + e]);
    oct_lds_sync();
    for (int e = j; e < c; e += 8) {
        const uint64_t k = keys[e];
        int rank = 0;
        for (int i = 0; i < c; ++i) rank += keys[i] < k ? 1 : 0;
        sorted[rank] = k;
    }
    oct_lds_sync();
    float last = key_t(sorted[0]);
    int kept = 1;
    uint64_t mask = 1ull;
    for (int i = 1; i < c; ++i) {
        const float t = key_t(sorted[i]);
        if (t > last + a.min_sep) { mask |= 1ull << i; ++kept; last = t; }
    }
    oct_lds_sync();
    *kept_out = kept;
    return mask;
}

// One ray, traversed by the 8 lanes of an octet (j = lane in the octet, q = the octet's LDS slot in the workgroup).
// kOrdered: the nearest hit child is taken next (octet-wide DPP minimum), so that a full K-list's t_limit prunes the boxes
// behind it.  false: the first hit child in slot order -- on a scene whose rays meet far fewer than K triangles the lists
// (almost) never fill, nothing is pruned whatever the order, and the minimum is pure cost (round 4: frame 0.961 -> 0.917
// ms, 2^17 random rays 0.549 -> 0.528).  The hits found are the same either way; bvh_launch picks by the mesh's depth
// complexity.
template <bool kOrdered>
__device__ __forceinline__ void oct_traverse_ray(const TravArgs &ta, int64_t ray, int j, int q, int oct_base)
{
    const int K = ta.max_hits;
    // With the re-origin rule on the list holds a few more than K entries: the chain usually drops a hit or two (a
    // grazing ray crosses a shell twice within min_sep), and with exactly K collected every drop would cost another
    // whole traversal for the next page.
    const int Kc = ta.list_cap;
    const float min_sep = ta.min_sep;
    const float4 *__restrict__ nodes = ta.nodes;
    const float4 *__restrict__ tris = ta.tris;
    const int stack_cap = ta.stack_cap;
    uint64_t *keys = trav_lds + (size_t)q * Kc;
    uint64_t *sorted = trav_lds + (size_t)kOctRays * Kc + (size_t)q * Kc;        // only when min_sep > 0
    int *stack = reinterpret_cast<int *>(trav_lds + (size_t)kOctRays * Kc * (min_sep > 0.0f ? 2 : 1)) + (size_t)q * stack_cap;
    const float *rays_o = ta.rays_o, *rays_d = ta.rays_d;
    const int root_is_valid = ta.root_is_valid;
    float *hit_t = ta.hit_t;
    int32_t *hit_tri = ta.hit_tri, *hit_count = ta.hit_count;
    uint64_t *keep_mask = ta.keep_mask;
    int32_t *raw_count = ta.raw_count;
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
    const float nx = -(ox * ix), ny = -(oy * iy), nz = -(oz * iz);
    float *my_t = hit_t + ray * K;
    int32_t *my_tri = hit_tri + ray * K;

    OctList list;
    list.keys = keys; list.K = Kc; list.j = j;
    int kept = 0;                    // hits written so far (min_sep chain)
    float last_t = 0.0f;             // t of the last kept hit
    uint64_t lo_key = 0;             // page lower bound: only keys > lo_key are collected
    float t_lo = 0.0f, t_accept = 0.0f;      // box pruning below the page / hits the chain would drop anyway

    for (int page = 0; page < kMaxPages; ++page) {
        list.count = 0;
        list.worst = ~0ull;
        list.worst_slot = 0;
        float t_limit = INFINITY;
        int sp = 0;
        int cur = root_is_valid ? 0 : kDone;
        while (cur != kDone) {
            while (cur >= 0) {
                const float4 *np = nodes + (size_t)cur * 16 + j * 2;
                const float4 a = np[0];                 // lo.xyz, hi.x
                const float4 b = np[1];                 // hi.yz, token, 0
                const float ax = __builtin_fmaf(a.x, ix, nx), bx = __builtin_fmaf(a.w, ix, nx);
                const float ay = __builtin_fmaf(a.y, iy, ny), by = __builtin_fmaf(b.x, iy, ny);
                const float az = __builtin_fmaf(a.z, iz, nz), bz = __builtin_fmaf(b.y, iz, nz);
                const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000005f;
                const int tok = __float_as_int(b.z);
                // conservative: the boxes were inflated on the host, the exit distance is widened, NaN counts as a hit
                const bool hit = tok != kDone && !(tn > tf) && !(tn * 0.999999f > t_limit) && !(tf < t_lo);
                const unsigned m8 = (unsigned)(__ballot(hit) >> oct_base) & 0xffu;
                if (m8 == 0) {
                    cur = kDone;
                    if (sp > 0) cur = stack[--sp];
                    continue;
                }
                const int n = __popc(m8);
                int nearest;
                if (kOrdered) {
                    const unsigned key = hit ? ((__float_as_uint(tn) & ~7u) | (unsigned)j) : 0xffffffffu;
                    nearest = (int)(oct_min_u32(key) & 7u);
                } else {
                    nearest = __ffs(m8) - 1;
                }
                if (hit && j != nearest) {
                    const unsigned others = m8 & ~(1u << nearest);
                    stack[sp + __popc(others & ((1u << j) - 1u))] = tok;
                }
                sp += n - 1;
                cur = oct_bcast(tok, oct_base, nearest);
                oct_lds_sync();
            }
            if (cur == kDone) break;
            {
                const int packed = ~cur;
                const int first = packed >> 3, cnt = (packed & 7) + 1;
                bool h = false;
                uint64_t key = 0;
                if (j < cnt) {
                    const float4 a = tris[(size_t)(first + j) * 3 + 0];
                    const float4 b = tris[(size_t)(first + j) * 3 + 1];
                    const float4 c = tris[(size_t)(first + j) * 3 + 2];
                    float t;
                    if (mt_hit(a, b, c, ox, oy, oz, dx, dy, dz, &t)) {
                        key = hit_key(t, __float_as_int(a.w));
                        h = key > lo_key && t > t_accept && key < list.worst;
                    }
                }
                const unsigned m8 = (unsigned)(__ballot(h) >> oct_base) & 0xffu;
                if (m8) {
                    const int n = __popc(m8);
                    if (list.count + n <= Kc) {
                        if (h) keys[list.count + __popc(m8 & ((1u << j) - 1u))] = key;
                        list.count += n;
                        oct_lds_sync();
                        if (list.count == Kc) list.find_worst();
                    } else {                                    // the list fills up or is full: one hit at a time
                        for (unsigned mm = m8; mm; mm &= mm - 1u) {
                            const int src = __ffs(mm) - 1;
                            const uint64_t k = ((uint64_t)(unsigned)oct_bcast((int)(unsigned)(key >> 32), oct_base, src) << 32) |
                                               (unsigned)oct_bcast((int)(unsigned)key, oct_base, src);
                            if (list.count < Kc) {
                                if (j == 0) keys[list.count] = k;
                                ++list.count;
                                oct_lds_sync();
                                if (list.count == Kc) list.find_worst();
                            } else if (k < list.worst) {
                                if (j == 0) keys[list.worst_slot] = k;
                                oct_lds_sync();
                                list.find_worst();
                            }
                        }
                    }
                    if (list.count == Kc) t_limit = key_t(list.worst);
                }
            }
            cur = kDone;
            if (sp > 0) cur = stack[--sp];
        }

        // rank sort of the page: lane j ranks entries j, j+8, ... (keys are unique, so the ranks are a permutation)
        const int count = list.count;
        if (!(min_sep > 0.0f)) {
            for (int e = j; e < count; e += 8) {
                const uint64_t k = keys[e];
                int rank = 0;
                for (int i = 0; i < count; ++i) rank += keys[i] < k ? 1 : 0;
                my_t[rank] = key_t(k);
                my_tri[rank] = key_id(k);
            }
            kept = count;
            break;
        }
        for (int e = j; e < count; e += 8) {
            const uint64_t k = keys[e];
            int rank = 0;
            for (int i = 0; i < count; ++i) rank += keys[i] < k ? 1 : 0;
            sorted[rank] = k;
        }
        oct_lds_sync();
        // the re-origin chain, front to back (octet-uniform; lane 0 writes)
        for (int i = 0; i < count && kept < K; ++i) {
            const uint64_t k = sorted[i];
            const float t = key_t(k);
            if (kept == 0 || t > last_t + min_sep) {
                if (j == 0) { my_t[kept] = t; my_tri[kept] = key_id(k); }
                ++kept;
                last_t = t;
            }
        }
        if (count < Kc || kept >= K) break;         // every hit of the ray has been seen, or K are kept
        lo_key = sorted[Kc - 1];
        t_lo = key_t(lo_key) * 0.999999f;
        t_accept = last_t + min_sep;                // anything closer is dropped by the chain whatever follows
        oct_lds_sync();
    }
    for (int i = kept + j; i < K; i += 8) { my_t[i] = INFINITY; my_tri[i] = -1; }
    if (j == 0) {
        hit_count[ray] = kept;
        if (keep_mask) { keep_mask[ray] = kept >= 64 ? ~0ull : ((1ull << kept) - 1ull); raw_count[ray] = kept; }
    }
}

// XCD-aware block order: hardware deals workgroups round-robin to the 8 XCDs (private L2 each).  A plain batch: XCD x
// walks the x-th CONTIGUOUS eighth of the blocks (one slice of the batch), so its L2 only has to hold that slice's part
// of the tree.  An image: contiguous eighths are row bands, and the object sits in the middle ones -- the XCDs of the
// top and bottom bands idle while two XCDs carry the frame (PMC: one resident wave per SIMD on average).  So the image
// is dealt in STRIPES of two tile rows (8 pixel rows): stripe s goes to XCD s % 8, every XCD gets a comb over the whole
// image (balanced), and consecutive blocks of an XCD are still neighbouring tiles of one stripe (L2-friendly).
__device__ __forceinline__ int xcd_block(const TravArgs &a)
{
    const int x = (int)(blockIdx.x & 7), k = (int)(blockIdx.x >> 3);
    if (a.stripe_blocks > 0) {
        const int s = k / a.stripe_blocks, p = k - s * a.stripe_blocks;
        return (s * 8 + x) * a.stripe_blocks + p;
    }
    return x * a.blocks_per_xcd + k;
}

// Every ray of the batch: a workgroup = 32 rays (image-shaped batches: 8x4 pixels, a wave = 4x2 pixels).
template <bool kOrdered>
__global__ __launch_bounds__(kTravThreads) void bvh8_traverse_kernel(TravArgs a)
{
    const int tid = threadIdx.x, j = tid & 7, q = tid >> 3;
    const int oct_base = (tid & 63) & 56;               // first lane of this octet within its wave
    const int block = xcd_block(a);
    if (block >= a.n_blocks) return;
    int64_t ray;
    if (a.image_width > 0) {
        const int w = q >> 3, r = q & 7;
        const int px = (block % a.tiles_x) * 8 + (w & 1) * 4 + (r & 3);
        const int py = (block / a.tiles_x) * 4 + (w >> 1) * 2 + (r >> 2);
        if (px >= a.image_width || py >= a.image_height) return;
        ray = (int64_t)py * a.image_width + px;
    } else {
        ray = (int64_t)block * kOctRays + q;
    }
    if (ray >= a.n_rays) return;
    oct_traverse_ray<kOrdered>(a, ray, j, q, oct_base);
}

// The repair pass after the camera-coherent intersector: only the rays whose candidate list overflowed (count > K) are
// traversed.  With keep_mask (and min_sep > 0) the same launch decides the re-origin rule for every OTHER ray's
// complete, unordered list without rewriting it (oct_keep_mask): keep_mask[ray], raw_count[ray] = the length of the
// stored list, hit_count[ray] = the number kept -- qf_pack_samples sorts the list the same way and drops the masked
// entries.  A workgroup owns 256 consecutive rays: one lane per ray classifies them (coalesced count reads; rays with
// fewer than two hits are finished here), the rays that need an octet are compacted into an LDS list, and the
// workgroup's 32 octets work that list off -- no octet idles on a background ray.
__global__ __launch_bounds__(kTravThreads) void bvh8_repair_kernel(TravArgs a)
{
    __shared__ int s_list[kTravThreads];
    __shared__ int s_n;
    const int tid = threadIdx.x, j = tid & 7, q = tid >> 3;
    const int oct_base = (tid & 63) & 56;
    const int block = (int)blockIdx.x;          // consecutive blocks on different XCDs: the work (object rays) is spread evenly
    if (block >= a.n_blocks) return;
    const int K = a.max_hits;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int64_t ray0 = (int64_t)block * kTravThreads;
    {
        const int64_t ray = ray0 + tid;
        if (ray < a.n_rays) {
            const int c = a.hit_count[ray];
            int need = 0;
            // all_flag raised: the rays were not the camera's pixel grid, the camera-coherent passes returned at once and
            // the lists are empty -- this launch IS the intersection then, exact for any rays
            if (c > K || (a.all_flag && *a.all_flag)) need = 2;
            else if (a.keep_mask) {
                // The rule can only drop a hit if two of the ray's hits lie within min_sep of each other.  One lane
                // tests that on the distances alone -- no sort: with no such pair every hit is kept whatever the order
                // -- and only the (rare) rays with a close pair go to an octet for the sorted chain.  The margin covers
                // the rounding of the chain's fp32 addition, so "no close pair" can never hide a drop.
                bool close_pair = false;
                if (c >= 2) {
                    const float *row = a.hit_t + ray * K;
                    float *col = reinterpret_cast<float *>(trav_lds) + a.tcol_offset + tid;      // [K][kTravThreads]
                    for (int i0 = 0; i0 < c; i0 += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = (i0 + u < c) ? row[i0 + u] : 0.0f;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + u < c) col[(i0 + u) * kTravThreads] = v[u];
                    }
                    const float ms = a.min_sep * 1.0001f;
                    for (int i = 1; i < c; ++i) {
                        const float ti = col[i * kTravThreads];
                        for (int k = 0; k < i; ++k) {
                            const float tk = col[k * kTravThreads];
                            close_pair |= !(fabsf(ti - tk) > ms + 4e-7f * fmaxf(ti, tk));
                        }
                    }
                }
                if (close_pair) need = 1;
                else { a.keep_mask[ray] = c >= 64 ? ~0ull : ((1ull << c) - 1ull); a.raw_count[ray] = c; }
            }
            if (need) s_list[atomicAdd(&s_n, 1)] = tid | (need << 16);
        }
    }
    __syncthreads();
    const int n = s_n;
    for (int e = q; e < n; e += kOctRays) {
        const int entry = s_list[e];
        const int64_t ray = ray0 + (entry & 0xffff);
        if ((entry >> 16) == 2) {
            oct_traverse_ray<true>(a, ray, j, q, oct_base);      // the rays that overflowed K: their lists DO fill
        } else {
            const int c = a.hit_count[ray];
            int kept;
            const uint64_t mask = oct_keep_mask(a, ray, c, j, q, &kept);
            if (j == 0) { a.keep_mask[ray] = mask; a.raw_count[ray] = c; a.hit_count[ray] = kept; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Camera-coherent intersector: when the rays are the pixel grid of one pinhole camera (every eval frame of the
// reference: nerf_synthetic.py:310-373), the set of rays that can hit a triangle is bounded by the triangle's
// projected screen box.  Each triangle is tested only against those pixels -- with the SAME mt_hit() on the SAME
// (o, d) values as the BVH path, so the hits are bit-identical -- and appended to the pixel's list with one
// atomic.  No tree, no stack, coalesced triangle reads; the lists are sorted afterwards (sort_hits_kernel).
struct RasterCam {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;   // c2w[:3,:3]: columns = camera right / up / back
    float cx, cy, cz;                                      // camera centre
    float fx, fy, px0, py0;                                // pixel = f * (x/z) + p0   (p0 = principal - 0.5)
    int w, h;
};

// kRasterLanes lanes cooperate on one triangle and share its screen box round-robin.
// Guard band in pixels around the projected triangle.  A pixel-centre ray that the fp32 Moller-Trumbore test accepts
// lies, in exact arithmetic, within c * eps * f pixels of the projected triangle (numerator rounding over |det|,
// c ~ 10, eps = 2^-24, f = focal length in pixels: ~1e-3 px at f = 1111, ~3e-3 px at f = 2700), and the projected
// vertices carry ~1e-4 px of rounding.  0.25 px leaves two orders of magnitude.
constexpr float kRasterGuard = 0.25f;

// kWide: the lists are [slot][ray] (capacity max_hits = the wide capacity), for select_nearest_kernel's coalesced reads.
// One triangle against the pixels of its screen box, lane ``sub`` of kRasterLanes (the body of both raster kernels).
// kSlab (depth-slab pass, raster_slab_kernel): only hits with slab.t_lo <= t < slab.t_hi are accepted, a pixel that
// already holds slab.stop_at candidates is skipped before its ray is even loaded, and a candidate is ONE 8-byte key
// (t bits << 32 | tri) in slab.keys [capacity][n_rays].
struct SlabArgs {
    float t_lo, t_hi;
    int stop_at;
    uint64_t *keys;
    const float4 *ray_rec;           // per ray (d.xyz, count when this slab's pass started as int bits); NULL for the first slab
};

// A triangle's screen-space set-up, shared by the passes below: the pixel box it can touch and (when its orientation
// is reliable) the three guard-banded edge functions.  false: no pixel can be hit.
struct TriSetup {
    int x0, x1, y0, y1;
    bool use_edges;
    float ea[3], eb[3], ec[3];
};

__device__ __forceinline__ bool tri_setup(const float4 a, const float4 b, const float4 c, const RasterCam &cam, TriSetup &s)
{
    // conservative screen box of the triangle (projection of a convex set is inside the box of its vertices)
    float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    int behind = 0;
    const float4 vs[3] = {a, b, c};
    float sxs[3], sys[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float dx = vs[k].x - cam.cx, dy = vs[k].y - cam.cy, dz = vs[k].z - cam.cz;
        const float xc = cam.r00 * dx + cam.r10 * dy + cam.r20 * dz;    // R^T (v - c)
        const float yc = cam.r01 * dx + cam.r11 * dy + cam.r21 * dz;
        const float zc = cam.r02 * dx + cam.r12 * dy + cam.r22 * dz;
        const float zv = -zc;                                           // depth along the viewing direction
        if (!(zv > 1e-6f)) { ++behind; sxs[k] = sys[k] = 0.0f; continue; }
        const float sx = cam.fx * (xc / zv) + cam.px0;
        const float sy = -cam.fy * (yc / zv) + cam.py0;
        sxs[k] = sx; sys[k] = sy;
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx);
        miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    if (behind == 3) return false;           // entirely behind the camera: t > 0 is impossible
    // Conservative 2-D reject before any memory is touched: a pixel can only be hit if it lies inside the projected
    // triangle grown by the guard band, i.e. on the inner side of every edge line moved outwards by the guard
    // (|edge| is over-estimated by its L1 length).  Skipped for triangles that straddle the camera plane or project
    // (almost) edge-on, where the orientation is not reliable; the exact test below decides in every case.
    s.use_edges = false;
    if (behind > 0) {                        // straddles the camera plane: no finite box, test every pixel
        s.x0 = 0; s.y0 = 0; s.x1 = cam.w - 1; s.y1 = cam.h - 1;
    } else {
        s.x0 = (int)fmaxf(floorf(minx - kRasterGuard), 0.0f);
        s.y0 = (int)fmaxf(floorf(miny - kRasterGuard), 0.0f);
        s.x1 = (int)fminf(ceilf(maxx + kRasterGuard), (float)(cam.w - 1));
        s.y1 = (int)fminf(ceilf(maxy + kRasterGuard), (float)(cam.h - 1));
        if (!(maxx + kRasterGuard >= 0.0f) || !(maxy + kRasterGuard >= 0.0f) ||
            !(minx - kRasterGuard <= (float)(cam.w - 1)) || !(miny - kRasterGuard <= (float)(cam.h - 1)))
            return false;
        const float area2 = (sxs[1] - sxs[0]) * (sys[2] - sys[0]) - (sys[1] - sys[0]) * (sxs[2] - sxs[0]);
        if (fabsf(area2) > 1e-2f) {
            s.use_edges = true;
            const float sgn = area2 > 0.0f ? 1.0f : -1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int k1 = (k + 1) % 3;
                const float ex = sxs[k1] - sxs[k], ey = sys[k1] - sys[k];
                // E(p) = sgn * ((p.y - v.y) * ex - (p.x - v.x) * ey): positive at the opposite vertex.
                // keep p iff E(p) >= -guard * (|ex| + |ey|) - slack
                s.ea[k] = -sgn * ey;
                s.eb[k] = sgn * ex;
                s.ec[k] = -(s.ea[k] * sxs[k] + s.eb[k] * sys[k]) + (kRasterGuard + 0.05f) * (fabsf(ex) + fabsf(ey)) + 1e-3f;
            }
        }
    }
    return true;
}

template <int kRasterLanes, bool kWide, bool kSlab = false>
__device__ __forceinline__ void raster_triangle(const float4 *__restrict__ tris, int64_t tri_i, int sub, const RasterCam &cam,
                                                const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                int max_hits, int32_t *__restrict__ hit_tri, float *__restrict__ hit_t,
                                                int32_t *__restrict__ hit_count, int32_t *__restrict__ overflow,
                                                const bool cam_origin, const SlabArgs slab = SlabArgs())
{
    const float4 a = tris[tri_i * 3 + 0], b = tris[tri_i * 3 + 1], c = tris[tri_i * 3 + 2];
    const int id = __float_as_int(a.w);
    TriSetup ts;
    if (!tri_setup(a, b, c, cam, ts)) return;
    const int x0 = ts.x0, x1 = ts.x1, y0 = ts.y0, y1 = ts.y1;
    const bool use_edges = ts.use_edges;
    const float *ea = ts.ea, *eb = ts.eb, *ec = ts.ec;
    const int bw = x1 - x0 + 1;
    const int total = bw * (y1 - y0 + 1);
    int px = x0 + sub % bw, py = y0 + sub / bw;
    for (int q = sub; q < total; q += kRasterLanes) {
        const int cx_ = px, cy_ = py;
        px += kRasterLanes;
        while (px > x1) { px -= bw; ++py; }
        if (use_edges) {
            const float fx_ = (float)cx_, fy_ = (float)cy_;
            if (ea[0] * fx_ + eb[0] * fy_ + ec[0] < 0.0f || ea[1] * fx_ + eb[1] * fy_ + ec[1] < 0.0f ||
                ea[2] * fx_ + eb[2] * fy_ + ec[2] < 0.0f)
                continue;
        }
        const int64_t ray = (int64_t)cy_ * cam.w + cx_;
        // the pixel's K nearest are all in nearer slabs.  Decided on the count the pixel had when this pass STARTED: the
        // live count also moves with this slab's own hits, and stopping on it would keep an arbitrary subset of them
        // (the count rides in one 16-byte record with the ray's direction: one request and one round trip for both; read
        // separately they were two of each per candidate pixel of the later passes)
        float dx, dy, dz;
        if (kSlab && slab.ray_rec) {
            const float4 rr = slab.ray_rec[ray];
            if (__float_as_int(rr.w) >= slab.stop_at) continue;
            dx = rr.x; dy = rr.y; dz = rr.z;
        } else {
            dx = rays_d[ray * 3]; dy = rays_d[ray * 3 + 1]; dz = rays_d[ray * 3 + 2];
        }
        // the origin: the camera centre when camera_rays_check has verified that every ray's origin IS that value bit for bit
        // (one scattered 12-byte load less per candidate pixel: 17 % of the pass), the ray's own otherwise
        float ox = cam.cx, oy = cam.cy, oz = cam.cz;
        if (!cam_origin) { ox = rays_o[ray * 3]; oy = rays_o[ray * 3 + 1]; oz = rays_o[ray * 3 + 2]; }
        // origin and direction arrive together: without the pin the compiler sinks the origin's load behind mt_hit's
        // det != 0 branch, a second memory round trip per pixel (measured: configs[2] intersection 2.90 -> 2.62 ms)
        asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz), "+v"(dx), "+v"(dy), "+v"(dz));
        float t;
        if (!mt_hit(a, b, c, ox, oy, oz, dx, dy, dz, &t)) continue;
        if (kSlab) {
            if (!(t >= slab.t_lo && t < slab.t_hi)) continue;       // this hit belongs to another slab's pass
            const int slot = atomicAdd(&hit_count[ray], 1);
            if (slot < max_hits) slab.keys[(int64_t)slot * ((int64_t)cam.w * cam.h) + ray] = hit_key(t, id);
            else atomicAdd(overflow, 1);
            continue;
        }
        const int slot = atomicAdd(&hit_count[ray], 1);
        if (slot < max_hits) {
            const int64_t at = kWide ? (int64_t)slot * ((int64_t)cam.w * cam.h) + ray : ray * max_hits + slot;
            hit_t[at] = t;
            if (hit_tri) hit_tri[at] = id;      // (NULL: a render-only frame's tile pack never reads the ids)
        } else {
            atomicAdd(overflow, 1);      // more than max_hits candidates: the caller re-runs the exact K-nearest BVH path
        }
    }
}

template <int kRasterLanes, bool kWide>
__global__ __launch_bounds__(256) void raster_kernel(const float4 *__restrict__ tris, int64_t n_tri, RasterCam cam,
                                                     const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                     int max_hits, int32_t *__restrict__ hit_tri, float *__restrict__ hit_t,
                                                     int32_t *__restrict__ hit_count, int32_t *__restrict__ overflow,
                                                     const int32_t *__restrict__ ray_flag)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tri_i = gid / kRasterLanes;
    const int sub = (int)(gid % kRasterLanes);
    if (tri_i >= n_tri) return;
    if (ray_flag && *ray_flag) return;           // not this camera's pixel grid (camera_rays_check): the BVH answers
    const bool cam_origin = ray_flag != nullptr; // verified: every origin IS the camera centre
    raster_triangle<kRasterLanes, kWide>(tris, tri_i, sub, cam, rays_o, rays_d, max_hits, hit_tri, hit_t, hit_count, overflow,
                                         cam_origin);
}

// The precondition of the whole camera-coherent route, VERIFIED (round 4): ray i of the batch must be pixel
// (i % w, i / w) of this camera -- the reference's consistent ray set (nerf_synthetic.py:341-366: every ray starts at
// c2w[:3,3], its direction is the normalised pixel-centre direction).  Per ray:
//   * the origin equals the camera centre BIT FOR BIT (the passes then take it from the camera struct; the test uses
//     -0.0 for 0.0);
//   * the direction, projected with the very projection the triangle set-up uses, lands within kRayPixelTol of its own
//     pixel centre -- an order of magnitude inside the 0.25 px guard band, two above the rounding of a consistent ray
//     (3e-4 px at f = 1111, 3e-3 px at f = 8900) -- and in front of the camera;
//   * the direction has unit length (| |d|^2 - 1 | <= kRayUnitTol): the depth-slab passes bin by DISTANCE and accept by
//     t (0.1 % margin), the re-origin rule compares t with a world distance.
// Any violation (jittered directions of add_ray_direction_noise, nerf_synthetic.py:335-340; a stale or wrong camera;
// another up_sample; rays in another order; off-centre origins) raises *flag (zeroed by the caller's fill).  A raised
// flag makes every camera-coherent pass return at once and qf_bvh_repair_overflow traverse EVERY ray through the BVH
// -- exact for any rays -- instead of the guard-band reject silently dropping hits.  15.4 MB streamed per 800x800 frame.
constexpr float kRayPixelTol = 0.02f;
constexpr float kRayUnitTol = 1e-4f;

__device__ __forceinline__ void camera_rays_check(const uint32_t *__restrict__ o_bits, const float *__restrict__ rays_d,
                                                  int64_t n_rays, const RasterCam &cam, int32_t *__restrict__ flag,
                                                  int64_t first, int64_t stride)
{
    const uint32_t c0 = __float_as_uint(cam.cx), c1 = __float_as_uint(cam.cy), c2 = __float_as_uint(cam.cz);
    bool bad = false;
    for (int64_t r = first; r < n_rays; r += stride) {
        const uint32_t o0 = o_bits[r * 3], o1 = o_bits[r * 3 + 1], o2 = o_bits[r * 3 + 2];
        const float dx = rays_d[r * 3], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
        bad = bad || o0 != c0 || o1 != c1 || o2 != c2;
        const float xc = cam.r00 * dx + cam.r10 * dy + cam.r20 * dz;    // R^T d, as tri_setup projects R^T (v - c)
        const float yc = cam.r01 * dx + cam.r11 * dy + cam.r21 * dz;
        const float zv = -(cam.r02 * dx + cam.r12 * dy + cam.r22 * dz);
        const float sx = cam.fx * (xc / zv) + cam.px0, sy = -cam.fy * (yc / zv) + cam.py0;
        const float px = (float)(int)(r % cam.w), py = (float)(int)(r / cam.w);
        // (negated comparisons: a NaN anywhere counts as a violation)
        bad = bad || !(zv > 0.0f) || !(fabsf(sx - px) <= kRayPixelTol) || !(fabsf(sy - py) <= kRayPixelTol) ||
              !(fabsf(dx * dx + dy * dy + dz * dz - 1.0f) <= kRayUnitTol);
    }
    if (bad) *flag = 1;
}

__global__ void camera_rays_check_kernel(const uint32_t *__restrict__ o_bits, const float *__restrict__ rays_d, int64_t n_rays,
                                         RasterCam cam, int32_t *__restrict__ flag)
{
    camera_rays_check(o_bits, rays_d, n_rays, cam, flag, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                      (int64_t)gridDim.x * blockDim.x);
}

// ---- triangle culling for cameras that see a PART of the scene (the row bands of parallel.ShardedFrameRenderer: every
// rank used to project all F triangles for its eighth of the rows).  The triangles are stored in BVH leaf order, so a
// chunk of kCullChunk consecutive ones is a compact piece of surface; chunk_boxes_kernel (once per build / refit) keeps
// its bounding box, cull_chunks_kernel (per frame, one lane per chunk) projects the box's corners with the triangle
// projection above and drops the chunk when its screen box misses the image by more than the guard band -- the same
// conservative reject raster_triangle applies per triangle, so no hit can be lost -- and raster_culled_kernel walks the
// compacted list with a resident grid.  The order in which hits arrive at a pixel's list changes; the lists are sorted
// afterwards (and were never in a defined order).
constexpr int kCullChunk = 64;

__global__ __launch_bounds__(256) void chunk_boxes_kernel(const float4 *__restrict__ tris, int64_t n_tri, int n_chunks,
                                                          float4 *__restrict__ boxes)
{
    const int chunk = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (chunk >= n_chunks) return;
    const int64_t t = (int64_t)chunk * kCullChunk + lane;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (t < n_tri) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 v = tris[t * 3 + k];
            lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
            hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    }
    if (lane == 0) {
        boxes[chunk * 2 + 0] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        boxes[chunk * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
}

// counters[2]: this call appends to counters[parity] and zeroes counters[parity ^ 1] for the next call on the handle
// (stream order: the previous call's raster kernel, which read it, is done) -- no memset launch per frame.
__global__ void cull_chunks_kernel(const float4 *__restrict__ boxes, int n_chunks, RasterCam cam, int32_t *__restrict__ visible,
                                   int32_t *__restrict__ counters, int parity, const uint32_t *__restrict__ o_bits,
                                   const float *__restrict__ rays_d, int64_t n_rays, int32_t *__restrict__ ray_flag)
{
    const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk == 0) counters[parity ^ 1] = 0;
    // (this launch precedes the pass anyway: it also carries the pass's ray check, see camera_rays_check_kernel)
    if (ray_flag) camera_rays_check(o_bits, rays_d, n_rays, cam, ray_flag, chunk, (int64_t)gridDim.x * blockDim.x);
    if (chunk >= n_chunks) return;
    const float4 lo = boxes[chunk * 2], hi = boxes[chunk * 2 + 1];
    if (!(lo.x <= hi.x)) return;                                  // empty chunk
    float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    bool keep = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float dx = ((k & 1) ? hi.x : lo.x) - cam.cx, dy = ((k & 2) ? hi.y : lo.y) - cam.cy, dz = ((k & 4) ? hi.z : lo.z) - cam.cz;
        const float xc = cam.r00 * dx + cam.r10 * dy + cam.r20 * dz;
        const float yc = cam.r01 * dx + cam.r11 * dy + cam.r21 * dz;
        const float zc = cam.r02 * dx + cam.r12 * dy + cam.r22 * dz;
        const float zv = -zc;
        if (!(zv > 1e-4f)) { keep = true; continue; }             // a corner at or behind the camera plane: no finite box
        const float sx = cam.fx * (xc / zv) + cam.px0, sy = -cam.fy * (yc / zv) + cam.py0;
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx);
        miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    // the convex hull of the projected corners contains every projected triangle of the chunk; one extra pixel of margin
    // on top of the per-triangle guard band covers the rounding of these eight projections
    const float g = kRasterGuard + 1.0f;
    if (!keep)
        keep = (maxx + g >= 0.0f) && (maxy + g >= 0.0f) && (minx - g <= (float)(cam.w - 1)) && (miny - g <= (float)(cam.h - 1));
    if (keep) visible[atomicAdd(&counters[parity], 1)] = chunk;
}

template <int kRasterLanes, bool kWide>
__global__ __launch_bounds__(256) void raster_culled_kernel(const float4 *__restrict__ tris, int64_t n_tri, RasterCam cam,
                                                            const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                            int max_hits, int32_t *__restrict__ hit_tri, float *__restrict__ hit_t,
                                                            int32_t *__restrict__ hit_count, int32_t *__restrict__ overflow,
                                                            const int32_t *__restrict__ visible, const int32_t *__restrict__ n_visible,
                                                            const int32_t *__restrict__ ray_flag)
{
    if (ray_flag && *ray_flag) return;           // see raster_kernel
    const bool cam_origin = ray_flag != nullptr;
    constexpr int kTrisPerBlock = 256 / kRasterLanes;
    constexpr int kBlocksPerChunk = kCullChunk / kTrisPerBlock;       // 1 / 2 / 4 for 4 / 8 / 16 lanes per triangle
    const int n_items = *n_visible * kBlocksPerChunk;
    const int sub = threadIdx.x % kRasterLanes;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {  // workgroup-uniform
        const int chunk = visible[item / kBlocksPerChunk];
        const int64_t tri_i = (int64_t)chunk * kCullChunk + (item % kBlocksPerChunk) * kTrisPerBlock + threadIdx.x / kRasterLanes;
        if (tri_i < n_tri)
            raster_triangle<kRasterLanes, kWide>(tris, tri_i, sub, cam, rays_o, rays_d, max_hits, hit_tri, hit_t, hit_count, overflow,
                                                 cam_origin);
    }
}

// ---- depth slabs for dense scenes (BASELINE configs[2]: 36 thin shells, up to 72 crossings per ray, K = 25).  The wide
// pass above collects EVERY crossing of a ray (up to 4K slots, one returning atomic + a scattered write each) only for
// the selection to drop three quarters of them.  Here the visible triangle chunks are binned by their distance from the
// camera into n_slabs slabs of equal thickness (a chunk goes to every slab its distance range touches), the slabs are
// rasterised NEAREST FIRST in separate launches, a hit is accepted only by the pass of the slab its t falls into, and a
// pixel that holds stop_at candidates when a pass starts is skipped by it: all its hits nearer than this slab are
// already in its list -- a complete depth prefix -- and they are enough.  Exact: the slab passes partition the hits by t
// (edges computed by the same expression everywhere), a chunk's slabs cover the t of every hit it can produce (unit
// camera rays: t is the distance from the camera centre; 0.1 % margin), and stop_at = selection capacity + 1 keeps
// "the prefix was the whole list" distinguishable in select_nearest_kernel.  A pixel still overshoots by the hits of
// the slab in which it crosses stop_at, so the candidate lists need stop_at + (crossings per slab) slots, not 4K.
constexpr int kMaxSlabs = 16;

struct SlabCtl {                       // device control block of one frame
    uint32_t dist_min_bits, dist_max_bits;     // over the visible chunks; positive floats order like their bit patterns
    int32_t n_visible;
    int32_t slab_count[kMaxSlabs];
};

__device__ __forceinline__ void slab_range(const SlabCtl *ctl, int n_slabs, float *lo, float *width)
{
    const float dmin = __uint_as_float(ctl->dist_min_bits) * 0.999f, dmax = __uint_as_float(ctl->dist_max_bits) * 1.001f;
    *lo = dmin;
    *width = fmaxf(dmax - dmin, 1e-12f) / (float)n_slabs;
}

// [edge(j), edge(j+1)) in t; the first slab has no lower and the last no upper bound
__device__ __forceinline__ void slab_edges(const SlabCtl *ctl, int n_slabs, int j, float *t_lo, float *t_hi)
{
#pragma clang fp contract(off)
    float lo, w;
    slab_range(ctl, n_slabs, &lo, &w);
    *t_lo = j == 0 ? -INFINITY : lo + (float)j * w;
    *t_hi = j == n_slabs - 1 ? INFINITY : lo + (float)(j + 1) * w;
}

// cull_chunks_kernel + every visible chunk's distance range from the camera centre
__global__ void slab_cull_kernel(const float4 *__restrict__ boxes, int n_chunks, RasterCam cam, int32_t *__restrict__ visible,
                                 float2 *__restrict__ range, SlabCtl *ctl)
{
    const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= n_chunks) return;
    const float4 lo = boxes[chunk * 2], hi = boxes[chunk * 2 + 1];
    if (!(lo.x <= hi.x)) return;
    float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    bool keep = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float dx = ((k & 1) ? hi.x : lo.x) - cam.cx, dy = ((k & 2) ? hi.y : lo.y) - cam.cy, dz = ((k & 4) ? hi.z : lo.z) - cam.cz;
        const float xc = cam.r00 * dx + cam.r10 * dy + cam.r20 * dz;
        const float yc = cam.r01 * dx + cam.r11 * dy + cam.r21 * dz;
        const float zc = cam.r02 * dx + cam.r12 * dy + cam.r22 * dz;
        const float zv = -zc;
        if (!(zv > 1e-4f)) { keep = true; continue; }
        const float sx = cam.fx * (xc / zv) + cam.px0, sy = -cam.fy * (yc / zv) + cam.py0;
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx);
        miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    const float g = kRasterGuard + 1.0f;
    if (!keep)
        keep = (maxx + g >= 0.0f) && (maxy + g >= 0.0f) && (minx - g <= (float)(cam.w - 1)) && (miny - g <= (float)(cam.h - 1));
    if (!keep) return;
    // nearest and farthest point of the box from the camera centre
    const float nx = fmaxf(fmaxf(lo.x - cam.cx, cam.cx - hi.x), 0.0f), ny = fmaxf(fmaxf(lo.y - cam.cy, cam.cy - hi.y), 0.0f);
    const float nz = fmaxf(fmaxf(lo.z - cam.cz, cam.cz - hi.z), 0.0f);
    const float fx = fmaxf(fabsf(lo.x - cam.cx), fabsf(hi.x - cam.cx)), fy = fmaxf(fabsf(lo.y - cam.cy), fabsf(hi.y - cam.cy));
    const float fz = fmaxf(fabsf(lo.z - cam.cz), fabsf(hi.z - cam.cz));
    const float dmin = sqrtf(nx * nx + ny * ny + nz * nz) * 0.9999f, dmax = sqrtf(fx * fx + fy * fy + fz * fz) * 1.0001f + 1e-30f;
    visible[atomicAdd(&ctl->n_visible, 1)] = chunk;
    range[chunk] = make_float2(dmin, dmax);
    atomicMin(&ctl->dist_min_bits, __float_as_uint(dmin));
    atomicMax(&ctl->dist_max_bits, __float_as_uint(dmax));
}

// Every visible chunk goes to the slabs its distance range [dmin, dmax] (margins included) touches: j with
// edge(j+1) > dmin and edge(j) <= dmax -- decided by comparing with the EDGES the passes bin their hits by, so the
// assignment is a superset of the slabs a chunk's hits can fall into without a slab of slack either side (a first
// version added one: every chunk was rasterised three times).  The per-slab list positions are allotted per workgroup
// (LDS counters, one global atomic per slab and workgroup): 46 000 chunks on eight global counters took 1.4 ms.
__global__ __launch_bounds__(256) void slab_assign_kernel(const int32_t *__restrict__ visible, const float2 *__restrict__ range,
                                                          SlabCtl *ctl, int n_slabs, int n_chunks, int32_t *__restrict__ lists)
{
    __shared__ int s_cnt[kMaxSlabs], s_base[kMaxSlabs];
    if (threadIdx.x < kMaxSlabs) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    int chunk = -1, j0 = 0, j1 = -1;
    if (e < ctl->n_visible) {
        chunk = visible[e];
        const float2 r = range[chunk];
        const float dlo = r.x * 0.999f, dhi = r.y * 1.001f;
        j0 = n_slabs;
        for (int j = 0; j < n_slabs; ++j) {
            float t_lo, t_hi;
            slab_edges(ctl, n_slabs, j, &t_lo, &t_hi);
            if (t_hi > dlo && t_lo <= dhi) { j0 = j < j0 ? j : j0; j1 = j; }
        }
        for (int j = j0; j <= j1; ++j) atomicAdd(&s_cnt[j], 1);
    }
    __syncthreads();
    if (threadIdx.x < n_slabs) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&ctl->slab_count[threadIdx.x], s_cnt[threadIdx.x]) : 0;
    __syncthreads();
    for (int j = j0; j <= j1; ++j) lists[(int64_t)j * n_chunks + atomicAdd(&s_base[j], 1)] = chunk;
}

// Between two slab passes: every ray's direction and the count its pixel holds now, as one 16-byte record (the stop rule
// of the next pass is decided on THIS count, see raster_triangle).
__global__ void slab_ray_records_kernel(const float *__restrict__ rays_d, const int32_t *__restrict__ hit_count, int64_t n,
                                        float4 *__restrict__ rec)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        rec[i] = make_float4(rays_d[i * 3], rays_d[i * 3 + 1], rays_d[i * 3 + 2], __int_as_float(hit_count[i]));
}

template <int kRasterLanes>
__global__ __launch_bounds__(256) void raster_slab_kernel(const float4 *__restrict__ tris, int64_t n_tri, RasterCam cam,
                                                          const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                          int capacity, uint64_t *__restrict__ keys, int32_t *__restrict__ hit_count,
                                                          int32_t *__restrict__ overflow, const int32_t *__restrict__ list,
                                                          const SlabCtl *__restrict__ ctl, int slab_j, int n_slabs, int stop_at,
                                                          const float4 *__restrict__ ray_rec,
                                                          const int32_t *__restrict__ ray_flag)
{
    if (ray_flag && *ray_flag) return;           // see raster_kernel
    const bool cam_origin = ray_flag != nullptr;
    constexpr int kTrisPerBlock = 256 / kRasterLanes;
    constexpr int kBlocksPerChunk = kCullChunk / kTrisPerBlock;
    const int n_items = ctl->slab_count[slab_j] * kBlocksPerChunk;
    SlabArgs sa;
    slab_edges(ctl, n_slabs, slab_j, &sa.t_lo, &sa.t_hi);
    sa.stop_at = stop_at;
    sa.keys = keys;
    sa.ray_rec = ray_rec;
    const int sub = threadIdx.x % kRasterLanes;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int chunk = list[item / kBlocksPerChunk];
        const int64_t tri_i = (int64_t)chunk * kCullChunk + (item % kBlocksPerChunk) * kTrisPerBlock + threadIdx.x / kRasterLanes;
        if (tri_i < n_tri)
            raster_triangle<kRasterLanes, true, true>(tris, tri_i, sub, cam, rays_o, rays_d, capacity, nullptr, nullptr,
                                                      hit_count, overflow, cam_origin, sa);
    }
}

// Dense scenes (more than K candidates on most rays): the camera-coherent pass collects up to `wide` candidates per
// ray in [slot][ray] lists, and this kernel keeps each ray's K nearest under (t, tri) -- the rule of
// bvh_traverse_kernel -- in the ordinary [ray][K] lists (arrival order; qf_pack_samples sorts).  lane = ray, its K
// running entries in a private LDS column.  Rays that lost candidates even at `wide` keep count > K and go to
// qf_bvh_repair_overflow.
constexpr int kSelectBlock = 64;     // one wave: K = 64 (+ headroom) needs 36 KB of LDS
constexpr int kSelectHeadroom = 8;   // with the re-origin rule: candidates kept beyond K so that dropped hits can be replaced
__host__ __device__ inline int select_capacity(int max_hits, int wide, float min_sep)
{
    const int cap = min_sep > 0.0f ? max_hits + kSelectHeadroom : max_hits;
    return cap < wide ? cap : wide;
}
// the first `count` keys of a lane's LDS column -> its [K] row of the hit lists, four entries per memory request
__device__ __forceinline__ void write_row_from_keys(const uint64_t *lk, int count, float *row_t, int32_t *row_i)
{
    int i = 0;
    for (; i + 4 <= count; i += 4) {
        f32x4u t4;
        i32x4u i4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint64_t k = lk[(i + e) * kSelectBlock];
            t4[e] = key_t(k);
            i4[e] = key_id(k);
        }
        *reinterpret_cast<f32x4u *>(row_t + i) = t4;
        *reinterpret_cast<i32x4u *>(row_i + i) = i4;
    }
    for (; i < count; ++i) {
        const uint64_t k = lk[i * kSelectBlock];
        row_t[i] = key_t(k);
        row_i[i] = key_id(k);
    }
}

// kKeys: the candidates are 8-byte keys (t bits << 32 | tri) in wide_key [wide][n_rays] (the depth-slab pass).
template <bool kKeys>
__global__ __launch_bounds__(kSelectBlock) void select_nearest_kernel(int64_t n_rays, int wide, int max_hits, float min_sep,
                                                                      const int32_t *__restrict__ wide_tri,
                                                                      const float *__restrict__ wide_t,
                                                                      const uint64_t *__restrict__ wide_key,
                                                                      int32_t *__restrict__ hit_tri, float *__restrict__ hit_t,
                                                                      int32_t *__restrict__ hit_count)
{
    // the lane's column: [cap][block] 8-byte keys (t bits << 32 | tri) -- key order IS the (t, tri) order, so the
    // insertion below is one LDS read, one 64-bit compare and one LDS write per shifted entry (it was two of each on
    // separate t / tri columns: selection 0.42 -> 0.3 ms on configs[2])
    extern __shared__ __attribute__((aligned(8))) unsigned char select_lds[];
    const int cap = select_capacity(max_hits, wide, min_sep);
    uint64_t *lk = reinterpret_cast<uint64_t *>(select_lds) + threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * kSelectBlock + threadIdx.x;
    if (r >= n_rays) return;
    const int cnt = hit_count[r];
    if (cnt > wide) return;
    float *row_t = hit_t + r * max_hits;
    int32_t *row_i = hit_tri + r * max_hits;
    if (cnt <= max_hits) {
        // a plain copy, eight slots per memory round trip (one slot per trip made this path -- most rays of a frame --
        // the kernel's duration: a wave per 64 rays, few waves per CU next to the selection's LDS columns)
        constexpr int kCopy = 8;
        for (int i0 = 0; i0 < cnt; i0 += kCopy) {
            float tb[kCopy];
            int ib[kCopy];
#pragma unroll
            for (int u = 0; u < kCopy; ++u) {
                const int i = i0 + u < cnt ? i0 + u : cnt - 1;
                if (kKeys) {
                    const uint64_t k = wide_key[(int64_t)i * n_rays + r];
                    tb[u] = key_t(k);
                    ib[u] = key_id(k);
                } else {
                    tb[u] = wide_t[(int64_t)i * n_rays + r];
                    ib[u] = wide_tri[(int64_t)i * n_rays + r];
                }
            }
#pragma unroll
            for (int q = 0; q < kCopy / 4; ++q) {
                if (i0 + 4 * q + 4 <= cnt) {                    // a whole quartet: one request per array
                    *reinterpret_cast<f32x4u *>(row_t + i0 + 4 * q) = (f32x4u){tb[4 * q], tb[4 * q + 1], tb[4 * q + 2], tb[4 * q + 3]};
                    *reinterpret_cast<i32x4u *>(row_i + i0 + 4 * q) = (i32x4u){ib[4 * q], ib[4 * q + 1], ib[4 * q + 2], ib[4 * q + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i0 + 4 * q + e < cnt) { row_t[i0 + 4 * q + e] = tb[4 * q + e]; row_i[i0 + 4 * q + e] = ib[4 * q + e]; }
                }
            }
        }
        return;
    }
    // the `held` nearest candidates under (t, tri) (all of them when cnt <= cap), kept SORTED in the column as they
    // arrive: an insertion shifts half the column on average, about what re-finding the maximum after a replacement
    // cost, and there is no sort left to do afterwards.  The candidates are read eight slots at a time: one slot per
    // pass is a dependent global load per pass, eight independent loads in flight cost the same latency once.
    const int held = cnt < cap ? cnt : cap;
    constexpr int kBatch = 8;
    int n = 0;
    uint64_t max_key = 0;                    // the column's last (largest) entry, in registers for the common reject
    for (int i0 = 0; i0 < cnt; i0 += kBatch) {
        uint64_t kb[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int i = i0 + u < cnt ? i0 + u : cnt - 1;
            if (kKeys) kb[u] = wide_key[(int64_t)i * n_rays + r];
            else kb[u] = hit_key(wide_t[(int64_t)i * n_rays + r], wide_tri[(int64_t)i * n_rays + r]);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (i0 + u >= cnt) break;
            const uint64_t key = kb[u];
            int m;                           // entries of the column that stay: [0, m)
            if (n < held) {
                m = n;
                ++n;
            } else {
                if (!(key < max_key)) continue;
                m = n - 1;                   // the last entry falls out
            }
            // position by bisection (log2 dependent LDS reads instead of one per shifted entry), then the move with its
            // reads issued four at a time ahead of the writes (independent of each other; LDS executes a wave's
            // operations in order)
            int lo = 0, hi = m;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (lk[mid * kSelectBlock] < key) lo = mid + 1;
                else hi = mid;
            }
            int j = m - 1;
            for (; j - 3 >= lo; j -= 4) {
                const uint64_t a0 = lk[j * kSelectBlock], a1 = lk[(j - 1) * kSelectBlock], a2 = lk[(j - 2) * kSelectBlock],
                               a3 = lk[(j - 3) * kSelectBlock];
                lk[(j + 1) * kSelectBlock] = a0;
                lk[j * kSelectBlock] = a1;
                lk[(j - 1) * kSelectBlock] = a2;
                lk[(j - 2) * kSelectBlock] = a3;
            }
            for (; j >= lo; --j) lk[(j + 1) * kSelectBlock] = lk[j * kSelectBlock];
            lk[lo * kSelectBlock] = key;
            max_key = lk[(n - 1) * kSelectBlock];
        }
    }
    if (min_sep <= 0.0f) {                   // held == K: any order will do, qf_pack_samples sorts (this one is sorted)
        write_row_from_keys(lk, max_hits, row_t, row_i);
        hit_count[r] = max_hits;
        return;
    }
    // The re-origin rule (bvh8_traverse_kernel) runs over a ray's hits in ascending order, so the K nearest alone do
    // not decide it: every hit the chain drops lets a farther one in.  Run the chain over the held prefix of the
    // ray's hits.  K kept hits are the answer whatever lies behind; fewer are the answer only if the prefix was the
    // whole list.  Otherwise the ray goes to the paged BVH traversal (count > K marks it for qf_bvh_repair_overflow).
    float last_t = key_t(lk[0]);
    int kept = 1;                            // compacted in place: slot `kept` never runs ahead of slot i
    for (int i = 1; i < held && kept < max_hits; ++i) {
        const uint64_t k = lk[i * kSelectBlock];
        const float t = key_t(k);
        if (!(t > last_t + min_sep)) continue;
        last_t = t;
        lk[kept * kSelectBlock] = k;
        ++kept;
    }
    if (kept < max_hits && cnt > held) return;      // hit_count[r] stays > K
    write_row_from_keys(lk, kept, row_t, row_i);
    hit_count[r] = kept;
}

// In-place ascending (t, tri) sort of every ray's (unordered) list, the re-origin rule (min_sep > 0: keep a hit iff it
// is the first or lies more than min_sep behind the last kept one -- see bvh8_traverse_kernel), padding and count
// clamp.  Exact for rays whose list holds ALL their hits (count <= K); the camera-coherent path sends every other ray
// through the BVH repair first.  A workgroup stages 128 rays' rows in LDS (coalesced both ways), lane = ray.
constexpr int kFilterRays = 128;
__global__ __launch_bounds__(kFilterRays) void filter_hits_kernel(int64_t n_rays, int max_hits, float min_sep,
                                                                  int32_t *__restrict__ hit_tri, float *__restrict__ hit_t,
                                                                  int32_t *__restrict__ hit_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char filter_smem[];
    const int K = max_hits, Kp = max_hits | 1;
    float *s_t = reinterpret_cast<float *>(filter_smem);
    int32_t *s_tri = reinterpret_cast<int32_t *>(s_t + kFilterRays * Kp);
    const int tid = threadIdx.x;
    const int64_t ray0 = (int64_t)blockIdx.x * kFilterRays;
    const int nr = (int)((n_rays - ray0) < kFilterRays ? (n_rays - ray0) : kFilterRays);
    for (int i = tid; i < nr * K; i += kFilterRays) {
        const int r = i / K, k = i - r * K;
        s_t[r * Kp + k] = hit_t[ray0 * K + i];
        s_tri[r * Kp + k] = hit_tri[ray0 * K + i];
    }
    __syncthreads();
    if (tid < nr) {
        int cnt = hit_count[ray0 + tid];
        if (cnt > K) cnt = K;
        float *row_t = s_t + tid * Kp;
        int32_t *row_i = s_tri + tid * Kp;
        if (K <= 32) {
            if (cnt > 1) sort_row_32<true>(row_t, row_i, cnt);
        } else {
            for (int i = 1; i < cnt; ++i) {
                const float t = row_t[i];
                const int id = row_i[i];
                int j = i - 1;
                while (j >= 0 && hit_less(t, id, row_t[j], row_i[j])) { row_t[j + 1] = row_t[j]; row_i[j + 1] = row_i[j]; --j; }
                row_t[j + 1] = t;
                row_i[j + 1] = id;
            }
        }
        if (min_sep > 0.0f && cnt > 1) {
            int kept = 1;
            float last_t = row_t[0];
            for (int i = 1; i < cnt; ++i) {
                const float t = row_t[i];
                if (t > last_t + min_sep) { row_t[kept] = t; row_i[kept] = row_i[i]; ++kept; last_t = t; }
            }
            cnt = kept;
        }
        for (int i = cnt; i < K; ++i) { row_t[i] = INFINITY; row_i[i] = -1; }
        hit_count[ray0 + tid] = cnt;
    }
    __syncthreads();
    for (int i = tid; i < nr * K; i += kFilterRays) {
        const int r = i / K, k = i - r * K;
        hit_t[ray0 * K + i] = s_t[r * Kp + k];
        hit_tri[ray0 * K + i] = s_tri[r * Kp + k];
    }
}

// The same sort fed straight from global memory (the tile kernel): lane = ray reads ITS OWN list -- rows of K entries,
// four entries per request (f32x4u) -- into the network's registers, sorts, and leaves the sorted list in
// its LDS row for the loops that index it by rank.  Every load of the list is independent of the others, so the wave
// waits for memory once; the staged variant (coalesced loop -> LDS -> registers) waited once per loop iteration, which
// is what a tile wave's time was made of (one wave per tile, nothing to overlap with).  kN = 16 or 32: network size,
// picked per tile from its longest list.  `deepest` (wave-uniform) bounds what is read.
template <int kN, bool kTri>
__device__ __forceinline__ void load_sort_row(const float *__restrict__ g_t, const int32_t *__restrict__ g_i, int K,
                                              int cnt, int deepest, float *row_t, int32_t *row_i)
{
    float t[kN];
    int32_t id[kN];
#pragma unroll
    for (int j = 0; j < kN / 4; ++j) {
        f32x4u v = {INFINITY, INFINITY, INFINITY, INFINITY};
        i32x4u w = {-1, -1, -1, -1};
        if (4 * j < deepest) {                                  // wave-uniform; deepest <= K
            if (4 * j + 3 < K) {                                // the whole quartet lies inside the row: one request
                v = *reinterpret_cast<const f32x4u *>(g_t + 4 * j);
                if (kTri) w = *reinterpret_cast<const i32x4u *>(g_i + 4 * j);
            } else {                                            // the row's last 1..3 entries
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (4 * j + e < K) {
                        v[e] = g_t[4 * j + e];
                        if (kTri) w[e] = g_i[4 * j + e];
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[4 * j + e] = v[e]; id[4 * j + e] = w[e]; }
    }
    if (kTri) {
        uint64_t key[kN];
#pragma unroll
        for (int k = 0; k < kN; ++k) key[k] = k < cnt ? hit_key(t[k], id[k]) : ~0ull;
#pragma unroll
        for (int k = 2; k <= kN; k <<= 1) {
#pragma unroll
            for (int i = 0; i < kN; ++i) {
                const int l = i ^ (k - 1);
                if (l > i) { const uint64_t a = key[i], b = key[l]; key[i] = a < b ? a : b; key[l] = a < b ? b : a; }
            }
#pragma unroll
            for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
                for (int i = 0; i < kN; ++i) {
                    const int l = i ^ j;
                    if (l > i) { const uint64_t a = key[i], b = key[l]; key[i] = a < b ? a : b; key[l] = a < b ? b : a; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kN; ++k)
            if (k < cnt) { row_t[k] = key_t(key[k]); row_i[k] = key_id(key[k]); }
    } else {
#pragma unroll
        for (int k = 0; k < kN; ++k) t[k] = k < cnt ? t[k] : INFINITY;
#pragma unroll
        for (int k = 2; k <= kN; k <<= 1) {
#pragma unroll
            for (int i = 0; i < kN; ++i) {
                const int l = i ^ (k - 1);
                if (l > i) { const float a = t[i], b = t[l]; t[i] = fminf(a, b); t[l] = fmaxf(a, b); }
            }
#pragma unroll
            for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
                for (int i = 0; i < kN; ++i) {
                    const int l = i ^ j;
                    if (l > i) { const float a = t[i], b = t[l]; t[i] = fminf(a, b); t[l] = fmaxf(a, b); }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kN; ++k)
            if (k < cnt) row_t[k] = t[k];
    }
}

// sampling_raytrace_numpy (mesh_utils.py:359-387): per-ray hit lists (in ANY order) -> packed samples sorted by
// (ray, depth).  A workgroup owns 128 consecutive rays, i.e. one contiguous slice of every output array:
//   1. the rays' [K] rows of hit_t / hit_tri are loaded into LDS with coalesced reads;
//   2. lane = ray: in-LDS insertion sort by (t, tri) -- the intersector's pass order -- then the stable sort by the
//      float64 depth |o + t d - o| of mesh_utils.py:371-375 (a no-op unless rounding reverses two near-equal hits);
//      each (ray, rank) drops its id into a slot map of the slice;
//   3. lane = output sample: coalesced writes of the six sample arrays.
constexpr int kPackRays = 128;

__device__ __forceinline__ double sample_depth64(float t, const double o[3], const double d[3], double p[3])
{
    const double td = (double)t;
    p[0] = o[0] + td * d[0];
    p[1] = o[1] + td * d[1];
    p[2] = o[2] + td * d[2];
    const double qx = p[0] - o[0], qy = p[1] - o[1], qz = p[2] - o[2];
    return sqrt((qx * qx + qy * qy) + qz * qz);      // np.linalg.norm(points - origins, axis=1)
}

__global__ __launch_bounds__(kPackRays) void pack_samples_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t n_rays, int max_hits,
    const int32_t *__restrict__ hit_tri, const float *__restrict__ hit_t, const int32_t *__restrict__ hit_count,
    const int64_t *__restrict__ ray_offset, float *__restrict__ xyz, float *__restrict__ dirs,
    int64_t *__restrict__ index_ray, float *__restrict__ depth, int64_t *__restrict__ index_tri,
    float *__restrict__ origins, const int32_t *__restrict__ inverse, float *__restrict__ xyz_c,
    float *__restrict__ dirs_c, float *__restrict__ depth_c, const uint64_t *__restrict__ keep_mask,
    const int32_t *__restrict__ raw_count, float min_sep, int32_t *__restrict__ close_flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = max_hits, Kp = max_hits | 1;            // odd row stride: conflict-free column access
    float *s_t = reinterpret_cast<float *>(smem);
    int32_t *s_tri = reinterpret_cast<int32_t *>(s_t + kPackRays * Kp);
    uint16_t *s_map = reinterpret_cast<uint16_t *>(s_tri + kPackRays * Kp);
    __shared__ int s_region;

    const int tid = threadIdx.x;
    const int64_t ray0 = (int64_t)blockIdx.x * kPackRays;
    const int nr = (int)((n_rays - ray0) < kPackRays ? (n_rays - ray0) : kPackRays);
    int cnt = 0;
    if (tid < nr) {
        cnt = keep_mask ? raw_count[ray0 + tid] : hit_count[ray0 + tid];
        if (cnt > K) cnt = K;
    }
    if (K <= 32) {
        // lane = ray reads its own list into registers, sorts it by (t, tri) and leaves it in its LDS row (load_sort_row):
        // one memory wait per wave instead of one per iteration of a staging loop
        int deepest = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int other = __shfl_xor(deepest, off, 64);
            deepest = other > deepest ? other : deepest;
        }
        const int64_t own = tid < nr ? ray0 + tid : ray0;
        if (deepest > 16) load_sort_row<32, true>(hit_t + own * K, hit_tri + own * K, K, cnt, deepest, s_t + tid * Kp, s_tri + tid * Kp);
        else if (deepest > 0) load_sort_row<16, true>(hit_t + own * K, hit_tri + own * K, K, cnt, deepest, s_t + tid * Kp, s_tri + tid * Kp);
    } else {
        for (int i = tid; i < nr * K; i += kPackRays) {
            const int r = i / K, k = i - r * K;
            s_t[r * Kp + k] = hit_t[ray0 * K + i];
            s_tri[r * Kp + k] = hit_tri[ray0 * K + i];
        }
    }
    if (tid == 0) s_region = 0;
    __syncthreads();

    const int64_t block_base = ray_offset[ray0];
    if (tid < nr) {
        const int64_t ray = ray0 + tid;
        float *row_t = s_t + tid * Kp;
        int32_t *row_i = s_tri + tid * Kp;
        if (K > 32) {                                         // (t, tri) ascending; K <= 32 is sorted already
            for (int i = 1; i < cnt; ++i) {
                const float t = row_t[i];
                const int id = row_i[i];
                int j = i - 1;
                while (j >= 0 && hit_less(t, id, row_t[j], row_i[j])) { row_t[j + 1] = row_t[j]; row_i[j + 1] = row_i[j]; --j; }
                row_t[j + 1] = t;
                row_i[j + 1] = id;
            }
        }
        if (keep_mask) {                                      // the re-origin rule, decided by qf_bvh_repair_overflow
            const uint64_t mask = keep_mask[ray];
            int kept = 0;
            for (int i = 0; i < cnt; ++i)
                if ((mask >> i) & 1ull) { row_t[kept] = row_t[i]; row_i[kept] = row_i[i]; ++kept; }
            cnt = kept;
        } else if (close_flag && min_sep > 0.0f) {
            // optimistic route: the lists were packed as if the rule dropped nothing; here, with the list sorted, that
            // is checked exactly (the chain drops a hit iff some hit is not more than min_sep behind its predecessor).
            // A violation raises the frame's flag: the host then decides the rule per ray (keep_mask) and packs again.
            bool drop = false;
            for (int i = 1; i < cnt; ++i) drop |= !(row_t[i] > row_t[i - 1] + min_sep);
            if (drop) *close_flag = 1;
        }
        if (cnt > 1) {
            const double o64[3] = {(double)rays_o[ray * 3], (double)rays_o[ray * 3 + 1], (double)rays_o[ray * 3 + 2]};
            const double d64[3] = {(double)rays_d[ray * 3], (double)rays_d[ray * 3 + 1], (double)rays_d[ray * 3 + 2]};
            double p[3];
            double prev = sample_depth64(row_t[0], o64, d64, p);
            bool sorted = true;
            for (int k = 1; k < cnt; ++k) {
                const double dk = sample_depth64(row_t[k], o64, d64, p);
                sorted = sorted && !(prev > dk);
                prev = dk;
            }
            if (!sorted) {                                     // rare: stable insertion by depth, depths recomputed
                for (int i = 1; i < cnt; ++i) {
                    const float t = row_t[i];
                    const int id = row_i[i];
                    const double di = sample_depth64(t, o64, d64, p);
                    int j = i - 1;
                    while (j >= 0 && sample_depth64(row_t[j], o64, d64, p) > di) { row_t[j + 1] = row_t[j]; row_i[j + 1] = row_i[j]; --j; }
                    row_t[j + 1] = t;
                    row_i[j + 1] = id;
                }
            }
        }
        const int local = (int)(ray_offset[ray] - block_base);
        for (int k = 0; k < cnt; ++k) s_map[local + k] = (uint16_t)((tid << 8) | k);
        if (tid == nr - 1) s_region = local + cnt;
    }
    __syncthreads();

    const int region = s_region;
    for (int j = tid; j < region; j += kPackRays) {
        const int m = s_map[j];
        const int rl = m >> 8, k = m & 255;
        const int64_t ray = ray0 + rl;
        const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
        const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
        const double o64[3] = {(double)ox, (double)oy, (double)oz};
        const double d64[3] = {(double)dx, (double)dy, (double)dz};
        double p[3];
        const double dep = sample_depth64(s_t[rl * Kp + k], o64, d64, p);
        // vectors / (|vectors| + 1e-7) in float32 (mesh_utils.py:369-370)
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-7f;
        const int64_t o = block_base + j;
        if (xyz) {              // the ray-major position / direction / origin arrays are optional (see qf_hip.h)
            xyz[o * 3 + 0] = (float)p[0];
            xyz[o * 3 + 1] = (float)p[1];
            xyz[o * 3 + 2] = (float)p[2];
            dirs[o * 3 + 0] = dx / nrm;
            dirs[o * 3 + 1] = dy / nrm;
            dirs[o * 3 + 2] = dz / nrm;
            origins[o * 3 + 0] = ox;
            origins[o * 3 + 1] = oy;
            origins[o * 3 + 2] = oz;
        }
        index_ray[o] = ray;
        depth[o] = (float)dep;
        index_tri[o] = (int64_t)s_tri[rl * Kp + k];
        if (inverse) {          // a second copy at the sample's place in the field kernel's processing order
            const int64_t c = inverse[o];
            xyz_c[c * 3 + 0] = (float)p[0];
            xyz_c[c * 3 + 1] = (float)p[1];
            xyz_c[c * 3 + 2] = (float)p[2];
            dirs_c[c * 3 + 0] = dx / nrm;
            dirs_c[c * 3 + 1] = dy / nrm;
            dirs_c[c * 3 + 2] = dz / nrm;
            if (depth_c) depth_c[c] = (float)dep;
        }
    }
}

// The same packing for a frame that is only going to be RENDERED (row-major width x height image, no ray-major
// arrays wanted): one wave per 8x8 pixel tile, lane = pixel.  The tile's hit lists are staged and sorted exactly as in
// pack_samples_kernel (same comparisons, same re-origin handling, same float64 depth order); then step k writes the
// rank-k samples of the tile's pixels -- position, direction, depth -- at tile_base[tile] + (slots of the ranks
// before) + (pixels before this one that also have a rank-k hit): the coherent order of qf_coherent_layout, produced
// by the ballots directly, so neither the order, nor its inverse, nor index_ray / index_tri / ray-major depths exist
// for such a frame.  Values are pack_samples_kernel's bit for bit (tests).
template <bool kTri>
__global__ __launch_bounds__(64) void pack_tiles_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, int w, int h, int tiles_x, int n_tiles, int max_hits,
    const int32_t *__restrict__ hit_tri, const float *__restrict__ hit_t, const int32_t *__restrict__ hit_count,
    const int64_t *__restrict__ tile_base, const int64_t *__restrict__ total, float *__restrict__ xyz_c,
    float *__restrict__ dirs_c, float *__restrict__ depth_c, int32_t *__restrict__ tri_c, const uint64_t *__restrict__ keep_mask,
    const int32_t *__restrict__ raw_count, float min_sep, int32_t *__restrict__ final_count, int32_t *__restrict__ dropped)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = max_hits, Kp = max_hits | 1;            // odd row stride: conflict-free column access
    // kTri = false: distances only -- the triangle ids are not part of the output then, and hits with equal t are the
    // same sample in either order.  kTri = true (tri_c wanted: baked-texture frames look their texels up by triangle):
    // ids staged too and the lists sorted by (t, tri) like everywhere else, so that ties resolve to the same triangle.
    float *s_t = reinterpret_cast<float *>(smem);
    int32_t *s_tri = reinterpret_cast<int32_t *>(s_t + 64 * Kp);
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int px0 = (tile % tiles_x) * 8, py0 = (tile / tiles_x) * 8;
    const int cols = (w - px0) < 8 ? (w - px0) : 8, rows = (h - py0) < 8 ? (h - py0) : 8;
    const int px = px0 + (lane & 7), py = py0 + (lane >> 3);
    const bool inside = px < w && py < h;
    const int64_t ray = inside ? (int64_t)py * w + px : 0;
    int cnt = 0;
    if (inside) {
        cnt = keep_mask ? raw_count[ray] : hit_count[ray];
        if (cnt > K) cnt = K;
    }
    // the longest list of the tile bounds what is staged: background tiles (40 % of an orbit frame) leave at once, the
    // others read their first `deepest` slots instead of all K
    int deepest = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int other = __shfl_xor(deepest, off, 64);
        deepest = other > deepest ? other : deepest;
    }
    if (deepest == 0) {                                       // wave-uniform
        if (final_count && inside) final_count[ray] = 0;
        return;
    }
    float *row_t = s_t + lane * Kp;
    int32_t *row_i = s_tri + lane * Kp;                       // only touched when kTri
    // everything the wave reads from memory is asked for here, before anything waits: the ray, the tile's first slot and
    // (below) the lists
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    int64_t base = tile_base[tile];
    if (K <= 32) {
        // lists of up to 32 hits: global memory -> registers -> sorted -> the lane's own LDS row (load_sort_row); no lane
        // reads another lane's row below, so no barrier
        if (deepest <= 16) load_sort_row<16, kTri>(hit_t + ray * K, hit_tri + ray * K, K, cnt, deepest, row_t, row_i);
        else load_sort_row<32, kTri>(hit_t + ray * K, hit_tri + ray * K, K, cnt, deepest, row_t, row_i);
    } else {
        // stage: the lists of a tile row's pixels lie K apart
        for (int yy = 0; yy < rows; ++yy) {
            const int64_t row_ray0 = (int64_t)(py0 + yy) * w + px0;
            for (int i = lane; i < cols * deepest; i += 64) {
                const int r = i / deepest, k = i - r * deepest;
                s_t[(yy * 8 + r) * Kp + k] = hit_t[(row_ray0 + r) * K + k];
                if (kTri) s_tri[(yy * 8 + r) * Kp + k] = hit_tri[(row_ray0 + r) * K + k];
            }
        }
        __syncthreads();
    }
    double o64[3] = {0.0, 0.0, 0.0}, d64[3] = {0.0, 0.0, 0.0};
    float dn[3] = {0.0f, 0.0f, 0.0f};
    int n_dropped = 0;
    if (inside) {
        if (K > 32) {                                         // t (or (t, tri)) ascending; K <= 32 is sorted already
            for (int i = 1; i < cnt; ++i) {
                const float t = row_t[i];
                const int id = kTri ? row_i[i] : 0;
                int j = i - 1;
                while (j >= 0 && (kTri ? hit_less(t, id, row_t[j], row_i[j]) : t < row_t[j])) {
                    row_t[j + 1] = row_t[j];
                    if (kTri) row_i[j + 1] = row_i[j];
                    --j;
                }
                row_t[j + 1] = t;
                if (kTri) row_i[j + 1] = id;
            }
        }
        if (keep_mask) {                                      // the re-origin rule, decided by qf_bvh_repair_overflow
            const uint64_t mask = keep_mask[ray];
            int kept = 0;
            for (int i = 0; i < cnt; ++i)
                if ((mask >> i) & 1ull) { row_t[kept] = row_t[i]; if (kTri) row_i[kept] = row_i[i]; ++kept; }
            cnt = kept;
        } else if (min_sep > 0.0f && cnt > 1) {
            // the re-origin rule on the sorted list, as filter_hits_kernel applies it: a hit is kept iff it is the first
            // or lies more than min_sep behind the last kept one.  The tile's slots were allotted from the counts before
            // the rule; what it drops leaves a gap at the end of the tile (filled below).
            float last_t = row_t[0];
            int kept = 1;
            for (int i = 1; i < cnt; ++i) {
                const float t = row_t[i];
                if (t > last_t + min_sep) { row_t[kept] = t; if (kTri) row_i[kept] = row_i[i]; ++kept; last_t = t; }
            }
            n_dropped = cnt - kept;
            cnt = kept;
        }
        if (cnt > 0) {
            o64[0] = (double)ox; o64[1] = (double)oy; o64[2] = (double)oz;
            d64[0] = (double)dx; d64[1] = (double)dy; d64[2] = (double)dz;
            // vectors / (|vectors| + 1e-7) in float32 (mesh_utils.py:369-370)
            const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-7f;
            dn[0] = dx / nrm; dn[1] = dy / nrm; dn[2] = dz / nrm;
        }
        if (cnt > 1) {
            double p[3];
            double prev = sample_depth64(row_t[0], o64, d64, p);
            bool sorted = true;
            for (int k = 1; k < cnt; ++k) {
                const double dk = sample_depth64(row_t[k], o64, d64, p);
                sorted = sorted && !(prev > dk);
                prev = dk;
            }
            if (!sorted) {                                     // rare: stable insertion by depth, depths recomputed
                for (int i = 1; i < cnt; ++i) {
                    const float t = row_t[i];
                    const int id = kTri ? row_i[i] : 0;
                    const double di = sample_depth64(t, o64, d64, p);
                    int j = i - 1;
                    while (j >= 0 && sample_depth64(row_t[j], o64, d64, p) > di) {
                        row_t[j + 1] = row_t[j];
                        if (kTri) row_i[j + 1] = row_i[j];
                        --j;
                    }
                    row_t[j + 1] = t;
                    if (kTri) row_i[j + 1] = id;
                }
            }
        }
    }
    if (final_count && inside) final_count[ray] = cnt;
    const unsigned long long below = (1ull << lane) - 1ull;
    float first_xyz[3] = {0.0f, 0.0f, 0.0f};                  // this lane's nearest sample, for the gap fill
    for (int k = 0;; ++k) {
        const unsigned long long mask = __ballot(cnt > k);
        if (mask == 0ull) break;                              // wave-uniform exit
        if (cnt > k) {
            const int64_t c = base + __popcll(mask & below);
            double p[3];
            const double dep = sample_depth64(row_t[k], o64, d64, p);
            xyz_c[c * 3 + 0] = (float)p[0];
            xyz_c[c * 3 + 1] = (float)p[1];
            xyz_c[c * 3 + 2] = (float)p[2];
            dirs_c[c * 3 + 0] = dn[0];
            dirs_c[c * 3 + 1] = dn[1];
            dirs_c[c * 3 + 2] = dn[2];
            depth_c[c] = (float)dep;
            if (kTri) tri_c[c] = (int32_t)row_i[k];
            if (k == 0) { first_xyz[0] = (float)p[0]; first_xyz[1] = (float)p[1]; first_xyz[2] = (float)p[2]; }
        }
        base += __popcll(mask);
    }
    // Slots the rule emptied: the field kernel streams [0, total) and must find finite points everywhere, nobody reads
    // what it computes there (qf_composite_tiles walks the kept counts).  They get a copy of the tile's first sample.
    const unsigned long long any_drop = __ballot(n_dropped > 0);
    if (any_drop) {
        const int64_t end = tile + 1 < n_tiles ? tile_base[tile + 1] : *total;
        const int src = __ffsll((long long)__ballot(cnt > 0)) - 1;       // a tile that dropped something kept something
        const float gx = __shfl(first_xyz[0], src, 64), gy = __shfl(first_xyz[1], src, 64), gz = __shfl(first_xyz[2], src, 64);
        const float hx = __shfl(dn[0], src, 64), hy = __shfl(dn[1], src, 64), hz = __shfl(dn[2], src, 64);
        const int gtri = kTri ? __shfl((int)row_i[0], src, 64) : 0;
        for (int64_t c = base + lane; c < end; c += 64) {
            xyz_c[c * 3 + 0] = gx; xyz_c[c * 3 + 1] = gy; xyz_c[c * 3 + 2] = gz;
            dirs_c[c * 3 + 0] = hx; dirs_c[c * 3 + 1] = hy; dirs_c[c * 3 + 2] = hz;
            depth_c[c] = 0.0f;
            if (kTri) tri_c[c] = (int32_t)gtri;
        }
        int sum = n_dropped;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0 && dropped) atomicAdd(dropped, sum);
    }
}

// the frame's dropped-hit count -> pinned host memory (host_out[2]), one thread; the counter is left at zero for the
// caller's next frame (which then needs no memset launch of its own)
__global__ void publish_dropped_kernel(int32_t *dropped, int64_t *host_out)
{
    host_out[2] = (int64_t)*dropped;
    *dropped = 0;
}

// Stable per-ray insertion sort of sample indices by fp32 depth (np.lexsort((depth, index_ray)) on grouped rays).
__global__ void resort_kernel(const int64_t *index_ray, const float *depth, int64_t n, int64_t *perm)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = index_ray[i];
        if (i != 0 && index_ray[i - 1] == ray) continue;
        perm[i] = i;
        for (int64_t k = i + 1; k < n && index_ray[k] == ray; ++k) {
            const float dk = depth[k];
            int64_t j = k - 1;
            while (j >= i && depth[perm[j]] > dk) { perm[j + 1] = perm[j]; --j; }
            perm[j + 1] = k;
        }
    }
}

// sampling_indexing (mesh_utils.py:389-412) in one pass: the stable per-ray re-sort by depth AND the gathers of the
// sample arrays through the resulting permutation AND the pack boundaries.  A workgroup owns RS_CHUNK consecutive
// samples (+ halo: a ray of the mesh path has at most QF_BVH_MAX_HITS = 64 samples, so rays starting in the chunk end
// inside the staged window); depths and ray ids are staged in LDS, the first samples of the rays are compacted so that
// consecutive lanes sort different rays (insertion sort of local indices, = np.lexsort((depth, ray)) on grouped
// rays), and the arrays are then written coalesced, reading from (almost always nearly the same) source positions.
// Rays that run past the window take the slow path through global memory.
constexpr int RS_THREADS = 256;
constexpr int RS_CHUNK = 1024;
constexpr int RS_HALO = 64;
constexpr int RS_STAGE = RS_CHUNK + RS_HALO;

__global__ __launch_bounds__(RS_THREADS) void resort_samples_kernel(
    const int64_t *index_ray, const float *depth, int64_t n, const float *points, const float *origins,
    const float *vectors, const int64_t *index_tri, int64_t *perm, float *out_points, float *out_depth,
    float *out_origins, float *out_vectors, int64_t *out_index_tri, uint8_t *boundary, const int32_t *inverse,
    float *out_points_c, float *out_vectors_c)
{
    __shared__ float s_depth[RS_STAGE];
    __shared__ int64_t s_ray[RS_STAGE + 1];
    __shared__ int s_src[RS_STAGE];                  // local source index of the sample that lands at each position
    __shared__ uint8_t s_mine[RS_STAGE];
    __shared__ int s_heads[RS_CHUNK];
    __shared__ int s_nheads;
    const int64_t n_chunks = (n + RS_CHUNK - 1) / RS_CHUNK;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t b0 = chunk * RS_CHUNK;
        const int staged = (int)((n - b0 < RS_STAGE) ? (n - b0) : RS_STAGE);
        const int own = (int)((n - b0 < RS_CHUNK) ? (n - b0) : RS_CHUNK);
        if (threadIdx.x == 0) {
            s_ray[0] = b0 > 0 ? index_ray[b0 - 1] : 0;
            s_nheads = 0;
        }
        for (int k = threadIdx.x; k < staged; k += RS_THREADS) {
            s_depth[k] = depth[b0 + k];
            s_ray[k + 1] = index_ray[b0 + k];
            s_mine[k] = 0;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < own; k += RS_THREADS) {
            const bool head = (b0 + k == 0) || s_ray[k] != s_ray[k + 1];
            if (boundary) boundary[b0 + k] = head ? 1 : 0;
            if (head) s_heads[atomicAdd(&s_nheads, 1)] = k;
        }
        __syncthreads();
        const int n_heads = s_nheads;
        for (int h = threadIdx.x; h < n_heads; h += RS_THREADS) {
            const int k = s_heads[h];
            const int64_t ray = s_ray[k + 1];
            int end = k + 1;
            while (end < staged && s_ray[end + 1] == ray) ++end;
            const bool spills = end == staged && b0 + staged < n && index_ray[b0 + staged] == ray;
            if (!spills) {
                s_src[k] = k;
                s_mine[k] = 1;
                for (int q = k + 1; q < end; ++q) {
                    const float dq = s_depth[q];
                    int j = q - 1;
                    while (j >= k && s_depth[s_src[j]] > dq) { s_src[j + 1] = s_src[j]; --j; }
                    s_src[j + 1] = q;
                    s_mine[q] = 1;
                }
            } else {                                  // longer than the window: sort and gather through global memory
                const int64_t i = b0 + k;
                int64_t e = i + 1;
                while (e < n && index_ray[e] == ray) ++e;
                // perm doubles as the work array; without one, fall back to a selection by rank
                for (int64_t q = i; q < e; ++q) {
                    const float dq = depth[q];
                    int64_t rank = 0;
                    for (int64_t r = i; r < e; ++r) {
                        const float dr = depth[r];
                        rank += (dr < dq) || (dr == dq && r < q);
                    }
                    const int64_t dst = i + rank;
                    if (perm) perm[dst] = q;
                    out_depth[dst] = dq;
                    if (out_index_tri) out_index_tri[dst] = index_tri[q];
                    const int64_t pos_c = inverse ? (int64_t)inverse[dst] : 0;
                    for (int c = 0; c < 3; ++c) {
                        out_points[dst * 3 + c] = points[q * 3 + c];
                        if (out_origins) out_origins[dst * 3 + c] = origins[q * 3 + c];
                        out_vectors[dst * 3 + c] = vectors[q * 3 + c];
                        if (inverse) {
                            out_points_c[pos_c * 3 + c] = points[q * 3 + c];
                            out_vectors_c[pos_c * 3 + c] = vectors[q * 3 + c];
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < staged; k += RS_THREADS) {
            if (!s_mine[k]) continue;
            const int64_t src = b0 + s_src[k], dst = b0 + k;
            if (perm) perm[dst] = src;
            out_depth[dst] = s_depth[s_src[k]];
            if (out_index_tri) out_index_tri[dst] = index_tri[src];
        }
        for (int e = threadIdx.x; e < 3 * staged; e += RS_THREADS) {
            const int k = e / 3, c = e - 3 * k;
            if (!s_mine[k]) continue;
            const int64_t src = (b0 + s_src[k]) * 3 + c, dst = b0 * 3 + e;
            const float pv = points[src], vv = vectors[src];
            out_points[dst] = pv;
            if (out_origins) out_origins[dst] = origins[src];
            out_vectors[dst] = vv;
            if (inverse) {                            // second copy at the sample's place in the coherent order
                const int64_t pc = (int64_t)inverse[b0 + k] * 3 + c;
                out_points_c[pc] = pv;
                out_vectors_c[pc] = vv;
            }
        }
        __syncthreads();
    }
}

// utils.py:1055-1063: float64 Cramer barycentrics -> fp32 clamp / renormalise -> uv -> floor -> clip.
__global__ void texel_indices_kernel(const double *vertices, const int64_t *faces, const float *uv, const float *points,
                                     const int64_t *index_tri, int64_t n, int texture_size, int64_t *texel)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = index_tri[i];
        const int64_t ia = faces[f * 3], ib = faces[f * 3 + 1], ic = faces[f * 3 + 2];
        const double ax = vertices[ia * 3], ay = vertices[ia * 3 + 1], az = vertices[ia * 3 + 2];
        const double e0x = vertices[ib * 3] - ax, e0y = vertices[ib * 3 + 1] - ay, e0z = vertices[ib * 3 + 2] - az;
        const double e1x = vertices[ic * 3] - ax, e1y = vertices[ic * 3 + 1] - ay, e1z = vertices[ic * 3 + 2] - az;
        const double wx = (double)points[i * 3] - ax, wy = (double)points[i * 3 + 1] - ay, wz = (double)points[i * 3 + 2] - az;
        const double d00 = (e0x * e0x + e0y * e0y) + e0z * e0z;
        const double d01 = (e0x * e1x + e0y * e1y) + e0z * e1z;
        const double d11 = (e1x * e1x + e1y * e1y) + e1z * e1z;
        const double d02 = (e0x * wx + e0y * wy) + e0z * wz;
        const double d12 = (e1x * wx + e1y * wy) + e1z * wz;
        const double inv = 1.0 / (d00 * d11 - d01 * d01);
        const double b2d = (d00 * d12 - d01 * d02) * inv;
        const double b1d = (d11 * d02 - d01 * d12) * inv;
        const double b0d = 1.0 - b1d - b2d;
        float b0 = fminf(fmaxf((float)b0d, 0.0f), 1.0f);
        float b1 = fminf(fmaxf((float)b1d, 0.0f), 1.0f);
        float b2 = fminf(fmaxf((float)b2d, 0.0f), 1.0f);
        const float s = (b0 + b1) + b2;
        b0 = b0 / s;
        b1 = b1 / s;
        b2 = b2 / s;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float u = (uv[ia * 2 + k] * b0 + uv[ib * 2 + k] * b1) + uv[ic * 2 + k] * b2;
            float fl = floorf(u);
            // torch: floor -> .long() -> clip(0, T-1); NaN (degenerate triangle) -> int64 min -> 0
            int64_t q = (fl != fl) ? 0 : (fl <= -9.2e18f ? INT64_MIN : (fl >= 9.2e18f ? INT64_MAX : (int64_t)fl));
            if (q < 0) q = 0;
            if (q > texture_size - 1) q = texture_size - 1;
            texel[i * 2 + k] = q;
        }
    }
}

// The same lookup with everything that depends on the triangle alone computed once per mesh: a 128-byte record per
// triangle -- corner a, edges e0 / e1, their three dot products and the reciprocal determinant (13 doubles), the three
// corners' uv (6 floats) -- so that a sample reads ONE line instead of following faces -> 3 vertices -> 3 uv, and
// evaluates two dot products instead of five and no division.  The values are the ones the kernel above computes per
// sample (same operations on the same inputs), so the texels are identical.
struct TexelRecord {
    double ax, ay, az, e0x, e0y, e0z, e1x, e1y, e1z, d00, d01, d11, inv;
    float uv[6];                     // (u, v) of corners a, b, c
};
static_assert(sizeof(TexelRecord) == 128, "one 128-byte line per triangle");

__global__ void texel_records_kernel(const double *vertices, const int64_t *faces, const float *uv, int64_t n_faces,
                                     TexelRecord *records)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_faces; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ia = faces[f * 3], ib = faces[f * 3 + 1], ic = faces[f * 3 + 2];
        TexelRecord r;
        r.ax = vertices[ia * 3]; r.ay = vertices[ia * 3 + 1]; r.az = vertices[ia * 3 + 2];
        r.e0x = vertices[ib * 3] - r.ax; r.e0y = vertices[ib * 3 + 1] - r.ay; r.e0z = vertices[ib * 3 + 2] - r.az;
        r.e1x = vertices[ic * 3] - r.ax; r.e1y = vertices[ic * 3 + 1] - r.ay; r.e1z = vertices[ic * 3 + 2] - r.az;
        r.d00 = (r.e0x * r.e0x + r.e0y * r.e0y) + r.e0z * r.e0z;
        r.d01 = (r.e0x * r.e1x + r.e0y * r.e1y) + r.e0z * r.e1z;
        r.d11 = (r.e1x * r.e1x + r.e1y * r.e1y) + r.e1z * r.e1z;
        r.inv = 1.0 / (r.d00 * r.d11 - r.d01 * r.d01);
        r.uv[0] = uv[ia * 2]; r.uv[1] = uv[ia * 2 + 1];
        r.uv[2] = uv[ib * 2]; r.uv[3] = uv[ib * 2 + 1];
        r.uv[4] = uv[ic * 2]; r.uv[5] = uv[ic * 2 + 1];
        records[f] = r;
    }
}

__device__ __forceinline__ void texel_from_record(const TexelRecord &r, float px, float py, float pz, int texture_size,
                                                  int64_t out[2])
{
    const double wx = (double)px - r.ax, wy = (double)py - r.ay, wz = (double)pz - r.az;
    const double d02 = (r.e0x * wx + r.e0y * wy) + r.e0z * wz;
    const double d12 = (r.e1x * wx + r.e1y * wy) + r.e1z * wz;
    const double b2d = (r.d00 * d12 - r.d01 * d02) * r.inv;
    const double b1d = (r.d11 * d02 - r.d01 * d12) * r.inv;
    const double b0d = 1.0 - b1d - b2d;
    float b0 = fminf(fmaxf((float)b0d, 0.0f), 1.0f);
    float b1 = fminf(fmaxf((float)b1d, 0.0f), 1.0f);
    float b2 = fminf(fmaxf((float)b2d, 0.0f), 1.0f);
    const float s = (b0 + b1) + b2;
    b0 = b0 / s;
    b1 = b1 / s;
    b2 = b2 / s;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = (r.uv[k] * b0 + r.uv[2 + k] * b1) + r.uv[4 + k] * b2;
        float fl = floorf(u);
        int64_t q = (fl != fl) ? 0 : (fl <= -9.2e18f ? INT64_MIN : (fl >= 9.2e18f ? INT64_MAX : (int64_t)fl));
        if (q < 0) q = 0;
        if (q > texture_size - 1) q = texture_size - 1;
        out[k] = q;
    }
}

__global__ void texel_indices_packed_kernel(const TexelRecord *__restrict__ records, const float *__restrict__ points,
                                            const int64_t *__restrict__ index_tri, int64_t n, int texture_size,
                                            int64_t *__restrict__ texel)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const TexelRecord r = records[index_tri[i]];
        int64_t q[2];
        texel_from_record(r, points[i * 3], points[i * 3 + 1], points[i * 3 + 2], texture_size, q);
        texel[i * 2 + 0] = q[0];
        texel[i * 2 + 1] = q[1];
    }
}

struct TexArgs {
    const uint8_t *alpha, *diffuse;
    const uint8_t *colors[QF_MAX_LOBES];
    const uint8_t *lam[QF_MAX_LOBES];
    int size, n_lobes, sigmoid_codec;
    float lambda_thres;
};

__device__ __forceinline__ float decode_color(uint8_t c, int sigmoid_codec)
{
    const float v = (float)c / 255.0f;
    if (sigmoid_codec) return logf(fminf(fmaxf(v / (1.0f - v), 1e-8f), 1e37f));   // ngp.py:277-278
    return v * 2.0f * 12.0f - 12.0f;                                              // ngp.py:280 (B-7)
}

// Device-resident texel record: the 4 + 6L quantised bytes of one texel, contiguous and padded to one 64-byte
// sector -- [alpha | diffuse rgb | (lambda, azimuth, elevation, colour rgb) * L].  The reference keeps 2 + 2L separate
// planes (the PNG set of texture_utils.py:67-124), i.e. 2 + 2L scattered sector reads per sample; a record is ONE.
constexpr int kTexelRecord = QF_TEXEL_RECORD_BYTES;

// Gathers a texel's bytes from the reference's planes into record order.
__device__ __forceinline__ void gather_record(const TexArgs &t, int64_t px, uint8_t *rec)
{
    rec[0] = t.alpha[px];
    rec[1] = t.diffuse[px * 3 + 0];
    rec[2] = t.diffuse[px * 3 + 1];
    rec[3] = t.diffuse[px * 3 + 2];
    for (int l = 0; l < t.n_lobes; ++l) {
        uint8_t *r = rec + 4 + 6 * l;
        r[0] = t.lam[l][px * 3 + 0];
        r[1] = t.lam[l][px * 3 + 1];
        r[2] = t.lam[l][px * 3 + 2];
        r[3] = t.colors[l][px * 3 + 0];
        r[4] = t.colors[l][px * 3 + 1];
        r[5] = t.colors[l][px * 3 + 2];
    }
}

__global__ void texture_pack_kernel(TexArgs t, uint8_t *records)
{
    const int64_t n = (int64_t)t.size * t.size;
    for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < n; px += (int64_t)gridDim.x * blockDim.x) {
        union { uint8_t b[kTexelRecord]; uint4 q[kTexelRecord / 16]; } rec;
#pragma unroll
        for (int k = 0; k < kTexelRecord / 16; ++k) rec.q[k] = make_uint4(0u, 0u, 0u, 0u);
        gather_record(t, px, rec.b);
        uint4 *dst = reinterpret_cast<uint4 *>(records + px * kTexelRecord);
#pragma unroll
        for (int k = 0; k < kTexelRecord / 16; ++k) dst[k] = rec.q[k];
    }
}

// Every quantity a record decodes to is a function of ONE uint8 code, so a workgroup first evaluates the reference's
// dequantisers (the same expressions as decode_record, hence the same bits) for all 256 codes into LDS and then
// decodes by lookup: the 4L sin/cos, L exp and 3+3L colour decodes per sample become LDS reads.
// kLookup: the texel is not read but looked up here, from the sample's position and triangle (texel_from_record): the
// frame path's fusion of qf_texel_indices_packed and this kernel (no int64 [n,2] texel array written and read back).
// The record stays in REGISTERS: sixteen 32-bit words addressed with compile-time indices only -- the lobe loop is fully
// unrolled over QF_MAX_LOBES with a wave-uniform ``l < n_lobes`` guard, so ``word[(4 + 6 l + j) >> 2]`` is a constant
// register and the byte comes out with one shift + mask (v_bfe).  Round 2 indexed a byte array with the run-time lobe
// counter, which put the whole record in scratch: 80 B per lane written and read back per sample.
__device__ __forceinline__ uint32_t rec_byte(const uint32_t (&w)[kTexelRecord / 4], int idx)   // idx: compile-time
{
    return (w[idx >> 2] >> ((idx & 3) * 8)) & 0xffu;
}

// TriT: int64_t (the reference's index_tri) or int32_t (the tile pack's ids).
template <bool kLookup, typename TriT>
__global__ __launch_bounds__(256) void texture_shade_packed_kernel(const uint8_t *__restrict__ records, int size, int n_lobes,
                                                                   int sigmoid_codec, float lambda_thres,
                                                                   const int64_t *__restrict__ texel,
                                                                   const float *__restrict__ dirs, int64_t n,
                                                                   float *__restrict__ rgb, float *__restrict__ sigma,
                                                                   const TexelRecord *__restrict__ tri_records,
                                                                   const float *__restrict__ points,
                                                                   const TriT *__restrict__ index_tri,
                                                                   const int64_t *__restrict__ n_dev)
{
    if (n_dev) { const int64_t nd = *n_dev; n = nd < n ? (nd > 0 ? nd : 0) : n; }    // device-side count (render-only frame)
    __shared__ float s_sigma[256], s_col[256], s_caz[256], s_saz[256], s_sel[256], s_cel[256], s_lam[256];
    {
        const int c = threadIdx.x;
        const float pi = 3.14159274101257324f;   // float32(np.pi)
        const float a = (float)c / 255.0f;
        s_sigma[c] = -logf(fmaxf(1.0f - a, 1e-6f)) / 0.005f;
        s_col[c] = decode_color((uint8_t)c, sigmoid_codec);
        const float az = (float)(uint8_t)(c - 128) / 128.0f * pi;
        const float el = (float)c / 256.0f * pi;
        s_caz[c] = cosf(az);
        s_saz[c] = sinf(az);
        s_sel[c] = sinf(el);
        s_cel[c] = cosf(el);
        s_lam[c] = expf((float)c * lambda_thres / 255.0f - 2.5f);
    }
    __syncthreads();
    const int n16 = (4 + 6 * n_lobes + 15) / 16;       // 16-byte pieces of the record that carry data
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t px;
        if (kLookup) {
            const TexelRecord tr = tri_records[index_tri[i]];
            int64_t rc[2];
            texel_from_record(tr, points[i * 3], points[i * 3 + 1], points[i * 3 + 2], size, rc);
            px = rc[0] * size + rc[1];
        } else {
            px = texel[i * 2] * size + texel[i * 2 + 1];
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(records + px * kTexelRecord);
        uint32_t w[kTexelRecord / 4];
#pragma unroll
        for (int k = 0; k < kTexelRecord / 16; ++k) {
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (k < n16) q = src[k];                   // wave-uniform
            w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
        }
        const float dx = dirs[i * 3], dy = dirs[i * 3 + 1], dz = dirs[i * 3 + 2];
        float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
        for (int l = 0; l < QF_MAX_LOBES; ++l) {
            if (l < n_lobes) {                         // wave-uniform; every index below is a compile-time constant
                const int o = 4 + 6 * l;
                const uint32_t c_lam = rec_byte(w, o), c_az = rec_byte(w, o + 1), c_el = rec_byte(w, o + 2);
                const float se = s_sel[c_el];
                const float x0 = s_caz[c_az] * se, x1 = s_saz[c_az] * se, x2 = s_cel[c_el];
                const float nrm = sqrtf((x0 * x0 + x1 * x1) + x2 * x2);
                const float dotp = ((x0 / nrm) * dx + (x1 / nrm) * dy) + (x2 / nrm) * dz;
                const float e = expf(fabsf(s_lam[c_lam]) * (dotp - 1.0f));
                r += s_col[rec_byte(w, o + 3)] * e;
                g += s_col[rec_byte(w, o + 4)] * e;
                b += s_col[rec_byte(w, o + 5)] * e;
            }
        }
        rgb[i * 3 + 0] = 1.0f / (1.0f + expf(-(s_col[rec_byte(w, 1)] + r)));
        rgb[i * 3 + 1] = 1.0f / (1.0f + expf(-(s_col[rec_byte(w, 2)] + g)));
        rgb[i * 3 + 2] = 1.0f / (1.0f + expf(-(s_col[rec_byte(w, 3)] + b)));
        sigma[i] = s_sigma[rec_byte(w, 0)];
    }
}

// The planar kernels (the reference's 2 + 2L separate planes): the same register discipline -- one texel's bytes are
// read plane by plane inside the unrolled lobe loop, each lobe is decoded and consumed at once (no per-lane feature
// array; round 2 kept float f[60] + a 64-byte record per lane in scratch: 256 B).
__device__ __forceinline__ void decode_lobe(const TexArgs &t, int l, int64_t px, float *o /* [7] */)
{
    const uint8_t lc = t.lam[l][px * 3 + 0], az8 = t.lam[l][px * 3 + 1], el8 = t.lam[l][px * 3 + 2];
    const float pi = 3.14159274101257324f;   // float32(np.pi)
    const float az = (float)(uint8_t)(az8 - 128) / 128.0f * pi;               // uint8 wrap (B-8), ngp.py:246
    const float el = (float)el8 / 256.0f * pi;                                // ngp.py:248
    const float se = sinf(el);
    o[0] = cosf(az) * se;
    o[1] = sinf(az) * se;
    o[2] = cosf(el);
    o[3] = expf((float)lc * t.lambda_thres / 255.0f - 2.5f);                  // ngp.py:261-262
    o[4] = decode_color(t.colors[l][px * 3 + 0], t.sigmoid_codec);
    o[5] = decode_color(t.colors[l][px * 3 + 1], t.sigmoid_codec);
    o[6] = decode_color(t.colors[l][px * 3 + 2], t.sigmoid_codec);
}

__global__ void texture_fetch_kernel(TexArgs t, const int64_t *texel, int64_t n, float *features)
{
    const int width = 3 + 7 * t.n_lobes + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = texel[i * 2] * t.size + texel[i * 2 + 1];
        float *f = features + i * width;
        const float a = (float)t.alpha[px] / 255.0f;
        f[0] = decode_color(t.diffuse[px * 3 + 0], t.sigmoid_codec);
        f[1] = decode_color(t.diffuse[px * 3 + 1], t.sigmoid_codec);
        f[2] = decode_color(t.diffuse[px * 3 + 2], t.sigmoid_codec);
#pragma unroll
        for (int l = 0; l < QF_MAX_LOBES; ++l) {
            if (l < t.n_lobes) {
                float o[7];
                decode_lobe(t, l, px, o);
#pragma unroll
                for (int k = 0; k < 7; ++k) f[3 + 7 * l + k] = o[k];
            }
        }
        f[width - 1] = -logf(fmaxf(1.0f - a, 1e-6f)) / 0.005f;                // texture_utils.py:61-65 (B-9)
    }
}

__global__ void texture_shade_kernel(TexArgs t, const int64_t *texel, const float *dirs, int64_t n, float *rgb,
                                     float *sigma)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = texel[i * 2] * t.size + texel[i * 2 + 1];
        const float dx = dirs[i * 3], dy = dirs[i * 3 + 1], dz = dirs[i * 3 + 2];
        float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
        for (int l = 0; l < QF_MAX_LOBES; ++l) {
            if (l < t.n_lobes) {
                float x[7];
                decode_lobe(t, l, px, x);
                const float nrm = sqrtf((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
                const float dotp = ((x[0] / nrm) * dx + (x[1] / nrm) * dy) + (x[2] / nrm) * dz;
                const float e = expf(fabsf(x[3]) * (dotp - 1.0f));
                r += x[4] * e;
                g += x[5] * e;
                b += x[6] * e;
            }
        }
        rgb[i * 3 + 0] = 1.0f / (1.0f + expf(-(decode_color(t.diffuse[px * 3 + 0], t.sigmoid_codec) + r)));
        rgb[i * 3 + 1] = 1.0f / (1.0f + expf(-(decode_color(t.diffuse[px * 3 + 1], t.sigmoid_codec) + g)));
        rgb[i * 3 + 2] = 1.0f / (1.0f + expf(-(decode_color(t.diffuse[px * 3 + 2], t.sigmoid_codec) + b)));
        const float a = (float)t.alpha[px] / 255.0f;
        sigma[i] = -logf(fmaxf(1.0f - a, 1e-6f)) / 0.005f;
    }
}

int fill_tex_args(const qf_texture_set *tex, TexArgs *t)
{
    if (!tex || !tex->alpha || !tex->diffuse || tex->texture_size < 1) return QF_ERR_INVALID_ARGUMENT;
    if (tex->n_lobes < 1 || tex->n_lobes > QF_MAX_LOBES) return QF_ERR_UNSUPPORTED;
    t->alpha = tex->alpha;
    t->diffuse = tex->diffuse;
    for (int l = 0; l < QF_MAX_LOBES; ++l) {
        t->colors[l] = l < tex->n_lobes ? tex->colors[l] : nullptr;
        t->lam[l] = l < tex->n_lobes ? tex->lambda_axis[l] : nullptr;
        if (l < tex->n_lobes && (!t->colors[l] || !t->lam[l])) return QF_ERR_INVALID_ARGUMENT;
    }
    t->size = tex->texture_size;
    t->n_lobes = tex->n_lobes;
    t->sigmoid_codec = tex->sigmoid_codec;
    t->lambda_thres = tex->lambda_thres;
    return QF_OK;
}

}  // namespace

#define QF_SIMPLE_LAUNCH(kernel, count, ...)                                                                  \
    hipLaunchKernelGGL(kernel, dim3(qf_grid_1d((count), 256)), dim3(256), 0, qf_stream(stream), __VA_ARGS__); \
    QF_LAUNCH_CHECK();

static int bvh_launch(const qf_bvh *bvh, const float *rays_o, const float *rays_d, int64_t n_rays, int32_t max_hits,
                      int32_t image_width, int32_t *hit_tri, float *hit_t, int32_t *hit_count, int only_overflowed,
                      uint64_t *keep_mask, int32_t *raw_count, void *stream, const int32_t *all_flag = nullptr)
{
    if (!bvh || n_rays < 0 || max_hits < 1 || max_hits > kMaxHits || image_width < 0) return QF_ERR_INVALID_ARGUMENT;
    if (bvh->n_tri >= (1 << 28)) return QF_ERR_UNSUPPORTED;      // leaf tokens of the traversal pack (first, count)
    if (n_rays == 0) return QF_OK;
    if (!rays_o || !rays_d || !hit_tri || !hit_t || !hit_count) return QF_ERR_INVALID_ARGUMENT;
    int height = 0, tiles_x = 0;
    int64_t n_blocks = qf_div_up(n_rays, kOctRays);
    if (image_width > 0) {
        if (n_rays % image_width) return QF_ERR_INVALID_ARGUMENT;
        height = (int)(n_rays / image_width);
        tiles_x = (image_width + 7) / 8;
        n_blocks = (int64_t)tiles_x * ((height + 3) / 4);
    }
    const bool sep = bvh->min_sep > 0.0f;
    const int stack_cap = (bvh->max_stack8 < 2 ? 2 : bvh->max_stack8) | 1;       // odd row stride
    // headroom only in the repair launch (dense scenes: every traversed ray fills its list); the all-rays traversal
    // keeps the smaller LDS footprint (measured: +6 % on a frame, +8 % on a training batch with the headroom)
    const int list_cap = (sep && only_overflowed) ? (max_hits + 8 < kMaxHits ? max_hits + 8 : kMaxHits) : max_hits;
    size_t lds = (size_t)kOctRays * ((size_t)list_cap * 8 * (sep ? 2 : 1) + (size_t)stack_cap * 4);
    const size_t tcol_offset = (lds + 3) / 4;
    if (only_overflowed && sep && keep_mask) lds = tcol_offset * 4 + (size_t)kTravThreads * max_hits * 4;   // distance columns
    if (lds > 160 * 1024 - 2048) return QF_ERR_UNSUPPORTED;
    TravArgs a;
    a.nodes = reinterpret_cast<const float4 *>(bvh->d_nodes8);
    a.tris = reinterpret_cast<const float4 *>(bvh->d_tris);
    a.rays_o = rays_o; a.rays_d = rays_d; a.n_rays = n_rays;
    a.root_is_valid = bvh->n_tri > 0 ? 1 : 0;
    a.max_hits = (int)max_hits; a.image_width = (int)image_width; a.image_height = height; a.tiles_x = tiles_x;
    a.stack_cap = stack_cap; a.list_cap = list_cap; a.min_sep = sep ? bvh->min_sep : 0.0f;
    a.hit_tri = hit_tri; a.hit_t = hit_t; a.hit_count = hit_count;
    a.keep_mask = (only_overflowed && sep) ? keep_mask : nullptr;
    a.raw_count = a.keep_mask ? raw_count : nullptr;
    a.tcol_offset = (int)tcol_offset;
    a.all_flag = only_overflowed ? all_flag : nullptr;
    if (only_overflowed) n_blocks = qf_div_up(n_rays, kTravThreads);        // 256 consecutive rays per workgroup
    int64_t per_xcd = qf_div_up(n_blocks, 8);
    a.stripe_blocks = 0;
    if (!only_overflowed) {
        // image: stripes of two tile rows; a large plain batch: stripes of 256 blocks (8 192 consecutive rays) -- it may well
        // be a row-major image handed over without its width, and contiguous eighths would be as lopsided as bands
        // (1.65 -> 1.01 ms for the bench frame); batches under 2^18 rays keep contiguous eighths (a 2^17-ray batch sorted
        // by camera and pixel: 0.42 ms contiguous, 0.63 ms in 64-block stripes)
        a.stripe_blocks = image_width > 0 ? tiles_x * 2 : 256;
        if (n_blocks < (int64_t)a.stripe_blocks * (image_width > 0 ? 16 : 32)) a.stripe_blocks = 0;
    }
    if (a.stripe_blocks > 0) {
        const int64_t stripes = qf_div_up(n_blocks, a.stripe_blocks);
        per_xcd = qf_div_up(stripes, 8) * a.stripe_blocks;
    }
    if (per_xcd * 8 > 0x7fffffff) return QF_ERR_UNSUPPORTED;
    a.n_blocks = (int)n_blocks; a.blocks_per_xcd = (int)per_xcd;
    // front-to-back order only pays when the K-lists fill: a mesh whose rays meet K/2 triangles or more on average
    const bool ordered = bvh->depth_complexity >= 0.5f * (float)max_hits;
    const void *fn = only_overflowed ? reinterpret_cast<const void *>(bvh8_repair_kernel)
                                     : (ordered ? reinterpret_cast<const void *>(bvh8_traverse_kernel<true>)
                                                : reinterpret_cast<const void *>(bvh8_traverse_kernel<false>));
    if (lds > 48 * 1024) QF_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (only_overflowed)
        hipLaunchKernelGGL(bvh8_repair_kernel, dim3((unsigned)(per_xcd * 8)), dim3(kTravThreads), lds, qf_stream(stream), a);
    else if (ordered)
        hipLaunchKernelGGL(bvh8_traverse_kernel<true>, dim3((unsigned)(per_xcd * 8)), dim3(kTravThreads), lds, qf_stream(stream), a);
    else
        hipLaunchKernelGGL(bvh8_traverse_kernel<false>, dim3((unsigned)(per_xcd * 8)), dim3(kTravThreads), lds, qf_stream(stream), a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

static int filter_launch(int64_t n_rays, int32_t max_hits, float min_sep, int32_t *hit_tri, float *hit_t,
                         int32_t *hit_count, hipStream_t st)
{
    const int Kp = max_hits | 1;
    const size_t lds = (size_t)kFilterRays * Kp * 8;
    const int64_t blocks = qf_div_up(n_rays, kFilterRays);
    if (blocks > 0x7fffffff) return QF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(filter_hits_kernel, dim3((unsigned)blocks), dim3(kFilterRays), lds, st, n_rays, (int)max_hits, min_sep,
                       hit_tri, hit_t, hit_count);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_filter_hits(const qf_bvh *bvh, int64_t n_rays, int32_t max_hits, int32_t *hit_tri, float *hit_t,
                              int32_t *hit_count, void *stream)
{
    if (!bvh || n_rays < 0 || max_hits < 1 || max_hits > kMaxHits) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0 || !(bvh->min_sep > 0.0f)) return QF_OK;       // rule off: the lists stay as they are
    if (!hit_tri || !hit_t || !hit_count) return QF_ERR_INVALID_ARGUMENT;
    return filter_launch(n_rays, max_hits, bvh->min_sep, hit_tri, hit_t, hit_count, qf_stream(stream));
}

extern "C" int qf_bvh_intersect(const qf_bvh *bvh, const float *rays_o, const float *rays_d, int64_t n_rays,
                                int32_t max_hits, int32_t image_width, int32_t *hit_tri, float *hit_t,
                                int32_t *hit_count, void *stream)
{
    return bvh_launch(bvh, rays_o, rays_d, n_rays, max_hits, image_width, hit_tri, hit_t, hit_count, 0, nullptr, nullptr,
                      stream);
}

extern "C" int qf_bvh_repair_overflow(const qf_bvh *bvh, const float *rays_o, const float *rays_d, int64_t n_rays,
                                      int32_t max_hits, int32_t image_width, int32_t *hit_tri, float *hit_t,
                                      int32_t *hit_count, uint64_t *keep_mask, int32_t *raw_count,
                                      const int32_t *traverse_all_flag, void *stream)
{
    if ((keep_mask == nullptr) != (raw_count == nullptr)) return QF_ERR_INVALID_ARGUMENT;
    return bvh_launch(bvh, rays_o, rays_d, n_rays, max_hits, image_width, hit_tri, hit_t, hit_count, 1, keep_mask, raw_count,
                      stream, traverse_all_flag);
}

extern "C" int qf_pack_samples(const float *rays_o, const float *rays_d, int64_t n_rays, int32_t max_hits,
                               const int32_t *hit_tri, const float *hit_t, const int32_t *hit_count,
                               const int64_t *ray_offset, float *xyz, float *dirs, int64_t *index_ray, float *depth,
                               int64_t *index_tri, float *origins, const int32_t *inverse, float *xyz_c, float *dirs_c,
                               float *depth_c, const uint64_t *keep_mask, const int32_t *raw_count,
                               float min_separation, int32_t *close_flag, void *stream)
{
    if ((keep_mask == nullptr) != (raw_count == nullptr)) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays < 0 || max_hits < 1 || max_hits > kMaxHits) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if (!rays_o || !rays_d || !hit_tri || !hit_t || !hit_count || !ray_offset) return QF_ERR_INVALID_ARGUMENT;
    if (inverse && (!xyz_c || !dirs_c)) return QF_ERR_INVALID_ARGUMENT;
    if (!index_ray || !depth || !index_tri) return QF_ERR_INVALID_ARGUMENT;
    if ((xyz || dirs || origins) && (!xyz || !dirs || !origins)) return QF_ERR_INVALID_ARGUMENT;
    if (!xyz && !inverse) return QF_ERR_INVALID_ARGUMENT;      // the positions have to go somewhere
    const int Kp = max_hits | 1;
    const size_t lds = (size_t)kPackRays * Kp * 8 + (size_t)kPackRays * max_hits * 2 + 64;
    const int64_t blocks = qf_div_up(n_rays, kPackRays);
    if (blocks > 0x7fffffff) return QF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_samples_kernel, dim3((unsigned)blocks), dim3(kPackRays), lds, qf_stream(stream), rays_o, rays_d,
                       n_rays, (int)max_hits, hit_tri, hit_t, hit_count, ray_offset, xyz, dirs, index_ray, depth, index_tri,
                       origins, inverse, xyz_c, dirs_c, depth_c, keep_mask, raw_count, min_separation, close_flag);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_pack_tiles(const float *rays_o, const float *rays_d, int32_t width, int32_t height, int32_t max_hits,
                             const int32_t *hit_tri, const float *hit_t, const int32_t *hit_count, const int64_t *tile_base,
                             const int64_t *total, float *xyz_c, float *dirs_c, float *depth_c, int32_t *tri_c,
                             const uint64_t *keep_mask, const int32_t *raw_count, float min_separation,
                             int32_t *final_count, int32_t *dropped, int64_t *host_out, int32_t dropped_is_zero,
                             void *stream)
{
    if ((keep_mask == nullptr) != (raw_count == nullptr)) return QF_ERR_INVALID_ARGUMENT;
    if (width < 1 || height < 1 || max_hits < 1 || max_hits > kMaxHits) return QF_ERR_INVALID_ARGUMENT;
    if (!rays_o || !rays_d || !hit_t || !hit_count || !tile_base || !total || !xyz_c || !dirs_c || !depth_c)
        return QF_ERR_INVALID_ARGUMENT;
    if (tri_c && !hit_tri) return QF_ERR_INVALID_ARGUMENT;
    const bool rule_here = !keep_mask && min_separation > 0.0f;
    if (rule_here && (!final_count || !dropped)) return QF_ERR_INVALID_ARGUMENT;     // the counts change: they must go somewhere
    if (host_out && !dropped) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    if (dropped && !host_out && !dropped_is_zero) QF_HIP_TRY(hipMemsetAsync(dropped, 0, sizeof(int32_t), st));
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    const float sep = rule_here ? min_separation : 0.0f;
    if (tri_c) {
        hipLaunchKernelGGL(pack_tiles_kernel<true>, dim3((unsigned)(tiles_x * tiles_y)), dim3(64), (size_t)64 * (max_hits | 1) * 8, st,
                           rays_o, rays_d, (int)width, (int)height, tiles_x, tiles_x * tiles_y, (int)max_hits, hit_tri, hit_t,
                           hit_count, tile_base, total, xyz_c, dirs_c, depth_c, tri_c, keep_mask, raw_count, sep, final_count,
                           dropped);
    } else {
        hipLaunchKernelGGL(pack_tiles_kernel<false>, dim3((unsigned)(tiles_x * tiles_y)), dim3(64), (size_t)64 * (max_hits | 1) * 4, st,
                           rays_o, rays_d, (int)width, (int)height, tiles_x, tiles_x * tiles_y, (int)max_hits, hit_tri, hit_t,
                           hit_count, tile_base, total, xyz_c, dirs_c, depth_c, tri_c, keep_mask, raw_count, sep, final_count,
                           dropped);
    }
    if (host_out) hipLaunchKernelGGL(publish_dropped_kernel, dim3(1), dim3(1), 0, st, dropped, host_out);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_resort_by_depth(const int64_t *index_ray, const float *depth, int64_t n, int64_t *perm, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!index_ray || !depth || !perm) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(resort_kernel, n, index_ray, depth, n, perm);
    return QF_OK;
}

extern "C" int qf_resort_samples(const int64_t *index_ray, const float *depth, int64_t n, const float *points,
                                 const float *origins, const float *vectors, const int64_t *index_tri, int64_t *perm,
                                 float *out_points, float *out_depth, float *out_origins, float *out_vectors,
                                 int64_t *out_index_tri, uint8_t *boundary, const int32_t *inverse, float *out_points_c,
                                 float *out_vectors_c, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!index_ray || !depth || !points || !vectors || !out_points || !out_depth || !out_vectors ||
        (out_origins && !origins) || (out_index_tri && !index_tri) || (inverse && (!out_points_c || !out_vectors_c)))
        return QF_ERR_INVALID_ARGUMENT;
    const int64_t n_chunks = (n + RS_CHUNK - 1) / RS_CHUNK;
    hipLaunchKernelGGL(resort_samples_kernel, dim3((unsigned)(n_chunks < 65536 ? n_chunks : 65536)), dim3(RS_THREADS), 0,
                       qf_stream(stream), index_ray, depth, n, points, origins, vectors, index_tri, perm, out_points,
                       out_depth, out_origins, out_vectors, out_index_tri, boundary, inverse, out_points_c, out_vectors_c);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_texel_indices(const double *vertices, const int64_t *faces, const float *uv, const float *points,
                                const int64_t *index_tri, int64_t n, int32_t texture_size, int64_t *texel, void *stream)
{
    if (n < 0 || texture_size < 1) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!vertices || !faces || !uv || !points || !index_tri || !texel) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texel_indices_kernel, n, vertices, faces, uv, points, index_tri, n, (int)texture_size, texel);
    return QF_OK;
}

extern "C" int qf_texel_records_pack(const double *vertices, const int64_t *faces, const float *uv, int64_t n_faces,
                                     void *records, void *stream)
{
    if (n_faces < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_faces == 0) return QF_OK;
    if (!vertices || !faces || !uv || !records) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texel_records_kernel, n_faces, vertices, faces, uv, n_faces, static_cast<TexelRecord *>(records));
    return QF_OK;
}

extern "C" int qf_texel_indices_packed(const void *records, const float *points, const int64_t *index_tri, int64_t n,
                                       int32_t texture_size, int64_t *texel, void *stream)
{
    if (n < 0 || texture_size < 1) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!records || !points || !index_tri || !texel) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texel_indices_packed_kernel, n, static_cast<const TexelRecord *>(records), points, index_tri, n,
                     (int)texture_size, texel);
    return QF_OK;
}

extern "C" int qf_texture_fetch(const qf_texture_set *tex, const int64_t *texel, int64_t n, float *features, void *stream)
{
    TexArgs t;
    int rc = fill_tex_args(tex, &t);
    if (rc != QF_OK) return rc;
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!texel || !features) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texture_fetch_kernel, n, t, texel, n, features);
    return QF_OK;
}

extern "C" int qf_texture_shade(const qf_texture_set *tex, const int64_t *texel, const float *dirs, int64_t n,
                                float *rgb, float *sigma, void *stream)
{
    TexArgs t;
    int rc = fill_tex_args(tex, &t);
    if (rc != QF_OK) return rc;
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!texel || !dirs || !rgb || !sigma) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texture_shade_kernel, n, t, texel, dirs, n, rgb, sigma);
    return QF_OK;
}

extern "C" int qf_texture_pack(const qf_texture_set *tex, uint8_t *records, void *stream)
{
    TexArgs t;
    int rc = fill_tex_args(tex, &t);
    if (rc != QF_OK) return rc;
    if (!records || 4 + 6 * t.n_lobes > kTexelRecord) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(texture_pack_kernel, (int64_t)t.size * t.size, t, records);
    return QF_OK;
}

extern "C" int qf_texture_shade_packed(const uint8_t *records, int32_t texture_size, int32_t n_lobes,
                                       int32_t sigmoid_codec, float lambda_thres, const int64_t *texel,
                                       const float *dirs, int64_t n, float *rgb, float *sigma, void *stream)
{
    if (!records || texture_size < 1 || n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_lobes < 1 || n_lobes > QF_MAX_LOBES) return QF_ERR_UNSUPPORTED;
    if (n == 0) return QF_OK;
    if (!texel || !dirs || !rgb || !sigma) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH((texture_shade_packed_kernel<false, int64_t>), n, records, (int)texture_size, (int)n_lobes,
                     (int)sigmoid_codec, lambda_thres, texel, dirs, n, rgb, sigma, (const TexelRecord *)nullptr,
                     (const float *)nullptr, (const int64_t *)nullptr, (const int64_t *)nullptr);
    return QF_OK;
}

extern "C" int qf_texture_shade_points(const uint8_t *records, int32_t texture_size, int32_t n_lobes, int32_t sigmoid_codec,
                                       float lambda_thres, const void *triangle_records, const float *points,
                                       const int64_t *index_tri, const int32_t *index_tri32, const float *dirs, int64_t n,
                                       const int64_t *n_device, float *rgb, float *sigma, void *stream)
{
    if (!records || !triangle_records || texture_size < 1 || n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_lobes < 1 || n_lobes > QF_MAX_LOBES) return QF_ERR_UNSUPPORTED;
    if (n == 0) return QF_OK;
    if (!points || (!index_tri == !index_tri32) || !dirs || !rgb || !sigma) return QF_ERR_INVALID_ARGUMENT;   // exactly one id array
    const TexelRecord *tr = static_cast<const TexelRecord *>(triangle_records);
    if (index_tri32) {
        QF_SIMPLE_LAUNCH((texture_shade_packed_kernel<true, int32_t>), n, records, (int)texture_size, (int)n_lobes,
                         (int)sigmoid_codec, lambda_thres, (const int64_t *)nullptr, dirs, n, rgb, sigma, tr, points,
                         index_tri32, n_device);
    } else {
        QF_SIMPLE_LAUNCH((texture_shade_packed_kernel<true, int64_t>), n, records, (int)texture_size, (int)n_lobes,
                         (int)sigmoid_codec, lambda_thres, (const int64_t *)nullptr, dirs, n, rgb, sigma, tr, points,
                         index_tri, n_device);
    }
    return QF_OK;
}

static RasterCam make_raster_cam(const qf_camera *cam)
{
    RasterCam rc;
    const float *m = cam->c2w;     // row-major 3x4
    rc.r00 = m[0]; rc.r01 = m[1]; rc.r02 = m[2]; rc.cx = m[3];
    rc.r10 = m[4]; rc.r11 = m[5]; rc.r12 = m[6]; rc.cy = m[7];
    rc.r20 = m[8]; rc.r21 = m[9]; rc.r22 = m[10]; rc.cz = m[11];
    rc.fx = cam->fx; rc.fy = cam->fy;
    rc.px0 = cam->cx - 0.5f;       // camera_dir.x = (x - cx + 0.5) / fx
    rc.py0 = cam->cy - 0.5f;
    rc.w = cam->width; rc.h = cam->height;
    return rc;
}

// chunk boxes of the handle's triangles (allocated on first use, recomputed after a build / refit)
static int ensure_chunk_boxes(qf_bvh *bvh, hipStream_t st)
{
    const int64_t n_chunks = qf_div_up(bvh->n_tri, kCullChunk);
    if (n_chunks > 0x3fffffff) return QF_ERR_UNSUPPORTED;
    if (!bvh->d_chunk_box) {
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_chunk_box, (size_t)n_chunks * 2 * sizeof(float4)));
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_visible, (size_t)(n_chunks + 2) * sizeof(int32_t)));
        QF_HIP_TRY(hipMemsetAsync(bvh->d_visible, 0, 2 * sizeof(int32_t), st));        // the two counters
        bvh->chunk_dirty = true;
        bvh->cull_parity = 0;
    }
    if (bvh->chunk_dirty) {
        hipLaunchKernelGGL(chunk_boxes_kernel, dim3((unsigned)qf_div_up(n_chunks * 64, 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(bvh->d_tris), bvh->n_tri, (int)n_chunks,
                           reinterpret_cast<float4 *>(bvh->d_chunk_box));
        QF_LAUNCH_CHECK();
        bvh->chunk_dirty = false;
    }
    return QF_OK;
}

// zero the counts, the overflow counter and the origin flag: one fill launch when the caller laid them out back to back
// (hit_count [n_rays] | overflow | origin flag)
static int raster_zero(int64_t n_rays, int32_t *hit_count, int32_t *overflow, int32_t *origin_flag, hipStream_t st)
{
    int64_t words = n_rays;
    bool ovf_done = false, flag_done = origin_flag == nullptr;
    if (overflow == hit_count + words) { ++words; ovf_done = true; }
    if (ovf_done && origin_flag == hit_count + words) { ++words; flag_done = true; }
    QF_HIP_TRY(hipMemsetAsync(hit_count, 0, (size_t)words * sizeof(int32_t), st));
    if (!ovf_done) QF_HIP_TRY(hipMemsetAsync(overflow, 0, sizeof(int32_t), st));
    if (!flag_done) QF_HIP_TRY(hipMemsetAsync(origin_flag, 0, sizeof(int32_t), st));
    return QF_OK;
}

static int raster_launch(qf_bvh *bvh, const qf_camera *cam, const float *rays_o, const float *rays_d, int64_t n_rays,
                         int capacity, bool wide, int32_t *hit_tri, float *hit_t, int32_t *hit_count, int32_t *overflow,
                         int32_t *origin_flag, bool cull, hipStream_t st, bool skip_ids = false)
{
    if (skip_ids) hit_tri = nullptr;         // the pass leaves the id lists alone (qf_raster_intersect sort_lists = 2)
    const int rc_zero = raster_zero(n_rays, hit_count, overflow, origin_flag, st);
    if (rc_zero != QF_OK) return rc_zero;
    const RasterCam rc = make_raster_cam(cam);
    const uint32_t *o_bits = reinterpret_cast<const uint32_t *>(rays_o);
    if (bvh->n_tri > 0) {
        // Lanes per triangle: the per-triangle set-up (projection, edge equations) is replicated in every lane, so few
        // lanes win for pixel-sized triangles (measured on the 983 040-triangle 800x800 frame: 1/2/4/8 lanes ->
        // 0.33/0.25/0.22/0.23 ms); meshes that are coarse relative to the image get more lanes per triangle.
        const int64_t pixels_per_tri = n_rays / bvh->n_tri;
        // (a band's pass is latency-, not throughput-bound, but more lanes per triangle did not help it either: N = 8 band
        // 0.35 / 0.37 / 0.36 ms with 4 / 8 / 16 lanes)
        const int lanes = pixels_per_tri > 64 ? 16 : (pixels_per_tri > 8 ? 8 : 4);
        const int64_t threads = bvh->n_tri * lanes;
        const int64_t blocks = qf_div_up(threads, 256);
        if (blocks > 0x7fffffff) return QF_ERR_UNSUPPORTED;
        if (cull) {
            // chunk boxes (once per build / refit), the visible-chunk list of this camera, then a resident grid over it
            const int64_t n_chunks = qf_div_up(bvh->n_tri, kCullChunk);
            const int rc_boxes = ensure_chunk_boxes(bvh, st);
            if (rc_boxes != QF_OK) return rc_boxes;
            const float4 *tris4 = reinterpret_cast<const float4 *>(bvh->d_tris);
            float4 *boxes = reinterpret_cast<float4 *>(bvh->d_chunk_box);
            int32_t *counters = bvh->d_visible, *visible = bvh->d_visible + 2;
            const int parity = bvh->cull_parity;
            bvh->cull_parity ^= 1;
            hipLaunchKernelGGL(cull_chunks_kernel, dim3((unsigned)qf_div_up(n_chunks, 256)), dim3(256), 0, st, boxes,
                               (int)n_chunks, rc, visible, counters, parity, o_bits, rays_d, n_rays, origin_flag);
            QF_LAUNCH_CHECK();
            const int64_t items = n_chunks * (kCullChunk * lanes / 256);
            const int64_t cap = (int64_t)qf_cu_count_cached() * 8;
            const unsigned grid = (unsigned)(items < cap ? items : cap);
#define QF_RASTER_CULLED(L, WIDE)                                                                                      \
    hipLaunchKernelGGL((raster_culled_kernel<L, WIDE>), dim3(grid), dim3(256), 0, st, tris4, bvh->n_tri, rc, rays_o,  \
                       rays_d, capacity, hit_tri, hit_t, hit_count, overflow, visible, counters + parity, origin_flag)
            if (wide) {
                switch (lanes) {
                case 16: QF_RASTER_CULLED(16, true); break;
                case 8: QF_RASTER_CULLED(8, true); break;
                default: QF_RASTER_CULLED(4, true); break;
                }
            } else {
                switch (lanes) {
                case 16: QF_RASTER_CULLED(16, false); break;
                case 8: QF_RASTER_CULLED(8, false); break;
                default: QF_RASTER_CULLED(4, false); break;
                }
            }
#undef QF_RASTER_CULLED
            QF_LAUNCH_CHECK();
            return QF_OK;
        }
        if (origin_flag) {
            hipLaunchKernelGGL(camera_rays_check_kernel, dim3(qf_grid_1d(n_rays, 256)), dim3(256), 0, st, o_bits, rays_d, n_rays,
                               rc, origin_flag);
            QF_LAUNCH_CHECK();
        }
#define QF_RASTER_LAUNCH(L, WIDE)                                                                                     \
    hipLaunchKernelGGL((raster_kernel<L, WIDE>), dim3((unsigned)blocks), dim3(256), 0, st,                           \
                       reinterpret_cast<const float4 *>(bvh->d_tris), bvh->n_tri, rc, rays_o, rays_d, capacity,      \
                       hit_tri, hit_t, hit_count, overflow, origin_flag)
        if (wide) {
            switch (lanes) {
            case 16: QF_RASTER_LAUNCH(16, true); break;
            case 8: QF_RASTER_LAUNCH(8, true); break;
            default: QF_RASTER_LAUNCH(4, true); break;
            }
        } else {
            switch (lanes) {
            case 16: QF_RASTER_LAUNCH(16, false); break;
            case 8: QF_RASTER_LAUNCH(8, false); break;
            default: QF_RASTER_LAUNCH(4, false); break;
            }
        }
#undef QF_RASTER_LAUNCH
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}

static bool raster_args_ok(const qf_bvh *bvh, const qf_camera *cam, int64_t n_rays, int32_t max_hits)
{
    if (!bvh || !cam || n_rays < 0 || max_hits < 1 || max_hits > kMaxHits) return false;
    if (cam->width < 1 || cam->height < 1 || (int64_t)cam->width * cam->height != n_rays) return false;
    return cam->fx > 0.0f && cam->fy > 0.0f;
}

extern "C" int qf_raster_intersect(qf_bvh *bvh, const qf_camera *cam, const float *rays_o, const float *rays_d,
                                   int64_t n_rays, int32_t max_hits, int32_t *hit_tri, float *hit_t, int32_t *hit_count,
                                   int32_t *overflow, int32_t sort_lists, int32_t cull_chunks, int32_t *origin_flag,
                                   void *stream)
{
    if (!raster_args_ok(bvh, cam, n_rays, max_hits)) return QF_ERR_INVALID_ARGUMENT;
    if (!rays_o || !rays_d || !hit_tri || !hit_t || !hit_count || !overflow) return QF_ERR_INVALID_ARGUMENT;
    if (sort_lists < 0 || sort_lists > 2) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    const int rc = raster_launch(bvh, cam, rays_o, rays_d, n_rays, (int)max_hits, false, hit_tri, hit_t, hit_count, overflow,
                                 origin_flag, cull_chunks != 0, st, sort_lists == 2);
    if (rc != QF_OK) return rc;
    if (sort_lists == 1 && n_rays > 0) return filter_launch(n_rays, max_hits, bvh->min_sep, hit_tri, hit_t, hit_count, st);
    return QF_OK;
}

extern "C" int qf_raster_intersect_wide(qf_bvh *bvh, const qf_camera *cam, const float *rays_o, const float *rays_d,
                                        int64_t n_rays, int32_t max_hits, int32_t wide_hits, int32_t *wide_tri,
                                        float *wide_t, int32_t *hit_tri, float *hit_t, int32_t *hit_count,
                                        int32_t *overflow, int32_t cull_chunks, int32_t *origin_flag, void *stream)
{
    if (!raster_args_ok(bvh, cam, n_rays, max_hits)) return QF_ERR_INVALID_ARGUMENT;
    if (wide_hits < max_hits || wide_hits > 4096) return QF_ERR_INVALID_ARGUMENT;
    if (!rays_o || !rays_d || !wide_tri || !wide_t || !hit_tri || !hit_t || !hit_count || !overflow)
        return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    const int rc = raster_launch(bvh, cam, rays_o, rays_d, n_rays, (int)wide_hits, true, wide_tri, wide_t, hit_count, overflow,
                                 origin_flag, cull_chunks != 0, st);
    if (rc != QF_OK) return rc;
    if (n_rays == 0) return QF_OK;
    const size_t lds = (size_t)select_capacity(max_hits, wide_hits, bvh->min_sep) * kSelectBlock * 2 * sizeof(float);
    hipLaunchKernelGGL(select_nearest_kernel<false>, dim3((unsigned)qf_div_up(n_rays, kSelectBlock)), dim3(kSelectBlock), lds, st,
                       n_rays, (int)wide_hits, (int)max_hits, bvh->min_sep, wide_tri, wide_t, (const uint64_t *)nullptr, hit_tri,
                       hit_t, hit_count);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

namespace {
__global__ void slab_init_kernel(SlabCtl *ctl)
{
    ctl->dist_min_bits = 0x7f800000u;      // +inf
    ctl->dist_max_bits = 0u;
    ctl->n_visible = 0;
    for (int j = 0; j < kMaxSlabs; ++j) ctl->slab_count[j] = 0;
}
}  // namespace

extern "C" int qf_raster_intersect_slabs(qf_bvh *bvh, const qf_camera *cam, const float *rays_o, const float *rays_d,
                                         int64_t n_rays, int32_t max_hits, int32_t wide_hits, int32_t n_slabs,
                                         uint64_t *wide_keys, int32_t *hit_tri, float *hit_t, int32_t *hit_count,
                                         int32_t *overflow, int32_t *origin_flag, void *stream)
{
    if (!raster_args_ok(bvh, cam, n_rays, max_hits)) return QF_ERR_INVALID_ARGUMENT;
    if (n_slabs < 2 || n_slabs > kMaxSlabs) return QF_ERR_INVALID_ARGUMENT;
    const int sel_cap = select_capacity(max_hits, wide_hits, bvh->min_sep);
    if (wide_hits <= sel_cap + 1 || wide_hits > 4096) return QF_ERR_INVALID_ARGUMENT;     // room beyond stop_at for one slab's hits
    if (!rays_o || !rays_d || !wide_keys || !hit_tri || !hit_t || !hit_count || !overflow) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    const int rc_zero = raster_zero(n_rays, hit_count, overflow, origin_flag, st);
    if (rc_zero != QF_OK) return rc_zero;
    if (n_rays == 0 || bvh->n_tri == 0) return QF_OK;
    const int rc_boxes = ensure_chunk_boxes(bvh, st);
    if (rc_boxes != QF_OK) return rc_boxes;
    const int64_t n_chunks = qf_div_up(bvh->n_tri, kCullChunk);
    if (!bvh->d_slab_range) {
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_slab_range, (size_t)n_chunks * sizeof(float2)));
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_slab_lists, (size_t)n_chunks * kMaxSlabs * sizeof(int32_t)));
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_slab_ctl, sizeof(SlabCtl)));
    }
    const RasterCam rc = make_raster_cam(cam);
    const float4 *tris4 = reinterpret_cast<const float4 *>(bvh->d_tris);
    const float4 *boxes = reinterpret_cast<const float4 *>(bvh->d_chunk_box);
    int32_t *visible = bvh->d_visible + 2;
    float2 *range = reinterpret_cast<float2 *>(bvh->d_slab_range);
    SlabCtl *ctl = reinterpret_cast<SlabCtl *>(bvh->d_slab_ctl);
    if (origin_flag) {
        hipLaunchKernelGGL(camera_rays_check_kernel, dim3(qf_grid_1d(n_rays, 256)), dim3(256), 0, st,
                           reinterpret_cast<const uint32_t *>(rays_o), rays_d, n_rays, rc, origin_flag);
    }
    hipLaunchKernelGGL(slab_init_kernel, dim3(1), dim3(1), 0, st, ctl);
    hipLaunchKernelGGL(slab_cull_kernel, dim3((unsigned)qf_div_up(n_chunks, 256)), dim3(256), 0, st, boxes, (int)n_chunks, rc,
                       visible, range, ctl);
    hipLaunchKernelGGL(slab_assign_kernel, dim3((unsigned)qf_div_up(n_chunks, 256)), dim3(256), 0, st, visible, range, ctl,
                       (int)n_slabs, (int)n_chunks, bvh->d_slab_lists);
    QF_LAUNCH_CHECK();
    const int64_t pixels_per_tri = n_rays / bvh->n_tri;
    const int lanes = pixels_per_tri > 64 ? 16 : (pixels_per_tri > 8 ? 8 : 4);
    const int64_t items = n_chunks * (kCullChunk * lanes / 256);
    const int64_t cap = (int64_t)qf_cu_count_cached() * 8;
    const unsigned grid = (unsigned)(items < cap ? items : cap);
    const int stop_at = sel_cap + 1;
    // the counts at the start of each later pass, in one record with the ray's direction: rewritten between the passes
    if (bvh->slab_snapshot_rays < n_rays) {
        // hipFree synchronises the device: grow geometrically, so that a renderer whose ray count creeps up (row bands
        // that move with the cost profile) frees O(log n) times in its life and a steady-state frame never does
        const int64_t grown = bvh->slab_snapshot_rays * 2 > n_rays ? bvh->slab_snapshot_rays * 2 : n_rays;
        if (bvh->d_slab_snapshot) (void)hipFree(bvh->d_slab_snapshot);
        bvh->d_slab_snapshot = nullptr;
        bvh->slab_snapshot_rays = 0;
        QF_HIP_TRY(hipMalloc((void **)&bvh->d_slab_snapshot, (size_t)grown * sizeof(float4)));
        bvh->slab_snapshot_rays = grown;
    }
    float4 *snapshot = reinterpret_cast<float4 *>(bvh->d_slab_snapshot);
    for (int j = 0; j < n_slabs; ++j) {
        if (j > 0) {
            hipLaunchKernelGGL(slab_ray_records_kernel, dim3(qf_grid_1d(n_rays, 256)), dim3(256), 0, st, rays_d, hit_count,
                               n_rays, snapshot);
            QF_LAUNCH_CHECK();
        }
        const int32_t *list = bvh->d_slab_lists + (int64_t)j * n_chunks;
#define QF_RASTER_SLAB(L)                                                                                              \
    hipLaunchKernelGGL((raster_slab_kernel<L>), dim3(grid), dim3(256), 0, st, tris4, bvh->n_tri, rc, rays_o, rays_d,    \
                       (int)wide_hits, wide_keys, hit_count, overflow, list, ctl, j, (int)n_slabs, stop_at,               \
                       j == 0 ? (const float4 *)nullptr : snapshot, origin_flag)
        switch (lanes) {
        case 16: QF_RASTER_SLAB(16); break;
        case 8: QF_RASTER_SLAB(8); break;
        default: QF_RASTER_SLAB(4); break;
        }
#undef QF_RASTER_SLAB
    }
    QF_LAUNCH_CHECK();
    const size_t lds = (size_t)sel_cap * kSelectBlock * 2 * sizeof(float);
    hipLaunchKernelGGL(select_nearest_kernel<true>, dim3((unsigned)qf_div_up(n_rays, kSelectBlock)), dim3(kSelectBlock), lds, st,
                       n_rays, (int)wide_hits, (int)max_hits, bvh->min_sep, (const int32_t *)nullptr, (const float *)nullptr,
                       wide_keys, hit_tri, hit_t, hit_count);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Occupancy-grid ray marching (nerfacc 0.5.3 OccGridEstimator.sampling -> traverse_grids, one
// level, cone_angle = 0).  nerfacc walks the cells with a DDA but keeps the step phase through empty cells, so its
// samples are [t0 + k dt, t0 + (k+1) dt] with t0 the clipped aabb entry, kept iff the sample's MIDPOINT lies before
// the aabb exit and in an occupied cell.  That rule is evaluated directly here (one byte of grid per step, the
// 2 MiB grid is L2 resident), with every operation individually rounded so the oracle reproduces counts exactly.
namespace {

struct MarchArgs {
    float lo[3], hi[3];
    int res[3];
    float near_plane, far_plane, step;
};

__device__ __forceinline__ bool march_range(const MarchArgs &m, const float *o, const float *d, float t_near_ray,
                                            float t_far_ray, float *t0, float *t1)
{
    float tn = -INFINITY, tf = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float inv = safe_inv(d[k]);
        const float a = (m.lo[k] - o[k]) * inv, b = (m.hi[k] - o[k]) * inv;
        tn = fmaxf(tn, fminf(a, b));
        tf = fminf(tf, fmaxf(a, b));
    }
    *t0 = fmaxf(tn, t_near_ray);
    *t1 = fminf(tf, t_far_ray);
    return tn <= tf && *t0 < *t1;
}

__device__ __forceinline__ bool march_occupied(const MarchArgs &m, const uint8_t *binaries, const float *o,
                                               const float *d, float tm)
{
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p = o[k] + d[k] * tm;
        const float u = (p - m.lo[k]) / (m.hi[k] - m.lo[k]) * (float)m.res[k];
        const float f = floorf(u);
        if (!(f >= 0.0f && f < (float)m.res[k])) return false;
        c[k] = (int)f;
    }
    return binaries[((int64_t)c[0] * m.res[1] + c[1]) * m.res[2] + c[2]] != 0;
}

// count != nullptr: pass 1 (count per ray); else pass 2 (write at offsets)
__global__ void grid_march_kernel(MarchArgs m, const uint8_t *binaries, const float *rays_o, const float *rays_d,
                                  const float *t_min, const float *t_max, int64_t n_rays, int32_t *count,
                                  const int64_t *offsets, float *t_starts, float *t_ends, int64_t *ray_indices)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const float o[3] = {rays_o[r * 3], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
        const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
        const float near_r = t_min ? fmaxf(m.near_plane, t_min[r]) : m.near_plane;
        const float far_r = t_max ? fminf(m.far_plane, t_max[r]) : m.far_plane;
        float t0, t1;
        int n = 0;
        int64_t w = count ? 0 : offsets[r];
        if (march_range(m, o, d, near_r, far_r, &t0, &t1)) {
            for (int k = 0; k < (1 << 22); ++k) {      // hard cap: a degenerate ray can never spin
                const float ts = t0 + (float)k * m.step;
                const float te = t0 + (float)(k + 1) * m.step;
                const float tm = (ts + te) * 0.5f;
                if (!(tm < t1)) break;
                if (!march_occupied(m, binaries, o, d, tm)) continue;
                if (!count) { t_starts[w] = ts; t_ends[w] = te; ray_indices[w] = r; ++w; }
                ++n;
            }
        }
        if (count) count[r] = n;
    }
}

int fill_march_args(const float *aabb, const int32_t *res, float near_plane, float far_plane, float step, MarchArgs *m)
{
    if (!aabb || !res || !(step > 0.0f) || !(far_plane > near_plane)) return QF_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < 3; ++k) {
        m->lo[k] = aabb[k];
        m->hi[k] = aabb[3 + k];
        m->res[k] = res[k];
        if (!(aabb[3 + k] > aabb[k]) || res[k] < 1) return QF_ERR_INVALID_ARGUMENT;
        // bound the per-ray step count so the march loop always terminates quickly
        if ((aabb[3 + k] - aabb[k]) / step > 1.0e7f) return QF_ERR_UNSUPPORTED;
    }
    m->near_plane = near_plane;
    m->far_plane = far_plane;
    m->step = step;
    return QF_OK;
}

}  // namespace

extern "C" int qf_grid_march_count(const float *aabb, const int32_t *resolution, const uint8_t *binaries,
                                   const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                                   int64_t n_rays, float near_plane, float far_plane, float step, int32_t *count,
                                   void *stream)
{
    MarchArgs m;
    int rc = fill_march_args(aabb, resolution, near_plane, far_plane, step, &m);
    if (rc != QF_OK) return rc;
    if (n_rays < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if (!binaries || !rays_o || !rays_d || !count) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(grid_march_kernel, dim3(qf_grid_1d(n_rays, 64, 64)), dim3(64), 0, qf_stream(stream), m, binaries,
                       rays_o, rays_d, t_min, t_max, n_rays, count, (const int64_t *)nullptr, (float *)nullptr,
                       (float *)nullptr, (int64_t *)nullptr);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_grid_march_write(const float *aabb, const int32_t *resolution, const uint8_t *binaries,
                                   const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                                   int64_t n_rays, float near_plane, float far_plane, float step, const int64_t *offsets,
                                   float *t_starts, float *t_ends, int64_t *ray_indices, void *stream)
{
    MarchArgs m;
    int rc = fill_march_args(aabb, resolution, near_plane, far_plane, step, &m);
    if (rc != QF_OK) return rc;
    if (n_rays < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if (!binaries || !rays_o || !rays_d || !offsets) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(grid_march_kernel, dim3(qf_grid_1d(n_rays, 64, 64)), dim3(64), 0, qf_stream(stream), m, binaries,
                       rays_o, rays_d, t_min, t_max, n_rays, (int32_t *)nullptr, offsets, t_starts, t_ends, ray_indices);
    QF_LAUNCH_CHECK();
    return QF_OK;
}
