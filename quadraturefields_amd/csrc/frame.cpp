// One camera frame of the mesh-quadrature render as ONE host call: the fixed launch sequence of a render-only frame
// (FrameRenderer.render_async: examples/utils.py:510-620 behind nerf_synthetic.py:310-373's camera rays), composed
// from the library's own entry points -- nothing here launches a kernel of its own.
#include "qf_common.h"

extern "C" int qf_frame_render(qf_bvh *bvh, const qf_frame_job *job, void *stream)
{
    if (!bvh || !job || !job->camera) return QF_ERR_INVALID_ARGUMENT;
    const qf_frame_job &j = *job;
    const int32_t w = j.camera->width, h = j.camera->height;
    if (w < 1 || h < 1 || j.n_rays != (int64_t)w * h || j.max_hits < 1) return QF_ERR_INVALID_ARGUMENT;
    if (!j.rays_o || !j.rays_d || !j.hit_tri || !j.hit_t || !j.hit_count || !j.final_count || !j.tile_base || !j.total ||
        !j.dropped || !j.xyz_c || !j.dirs_c || !j.depth_c)
        return QF_ERR_INVALID_ARGUMENT;
    // The whole job is checked BEFORE the first launch: the entry points below repeat these tests, but by the
    // time qf_field_forward or qf_composite_tiles would refuse, four kernels have run, the handle's cull parity has
    // flipped and the pinned block has been written -- a C caller would get INVALID_ARGUMENT for a half-executed frame.
    if (j.field) {
        const qf_field_desc &f = *j.field;
        if (!j.rgb_c || !j.sigma_c || !j.table || !j.base_w) return QF_ERR_INVALID_ARGUMENT;
        // the frame composites colours: the head must produce them (qf_field_forward's own rules per head)
        if (f.head == QF_HEAD_NGP) {
            if (!j.head_ngp_w) return QF_ERR_INVALID_ARGUMENT;
        } else if (f.head == QF_HEAD_SG) {
            const qf_sg_head *s = j.head_sg;
            if (!s || !s->w1 || !s->b1 || !s->w2 || !s->b2 || !s->wout || !s->bout) return QF_ERR_INVALID_ARGUMENT;
            if (f.n_lobes < 1 || f.n_lobes > QF_MAX_LOBES) return QF_ERR_UNSUPPORTED;
        } else {
            return QF_ERR_INVALID_ARGUMENT;       // QF_HEAD_NONE / QF_HEAD_SG_FEATURES give the compositor no rgb
        }
        for (int k = 0; k < 3; ++k)
            if (!(f.aabb[3 + k] > f.aabb[k])) return QF_ERR_INVALID_ARGUMENT;
        // ... and the compositor's (qf_composite_tiles): an image to write, a known background, its colour when custom
        if (j.bg_mode < 0 || j.bg_mode > 3) return QF_ERR_INVALID_ARGUMENT;
        if (!j.out_packed && (!j.out_rgb || !j.out_alpha || !j.out_depth)) return QF_ERR_INVALID_ARGUMENT;
        if (j.bg_mode == QF_BG_CUSTOM && !j.bkgd) return QF_ERR_INVALID_ARGUMENT;
    }
    const int64_t cap = j.n_rays * (int64_t)j.max_hits;
    int32_t *overflow = j.hit_count + j.n_rays;
    // 1. camera-coherent intersection (lists in arrival order), 2. exact K nearest for the pixels that overflowed
    int rc = qf_raster_intersect(bvh, j.camera, j.rays_o, j.rays_d, j.n_rays, j.max_hits, j.hit_tri, j.hit_t, j.hit_count,
                                 overflow, j.tri_c ? 0 : 2 /* no ids wanted: the pass skips their stores */, j.cull_chunks,
                                 overflow + 1, stream);
    if (rc != QF_OK) return rc;
    // (... and for EVERY pixel when the pass's ray check found that the rays are not this camera's pixel grid: the pass
    // wrote nothing then, and this launch is the intersection)
    rc = qf_bvh_repair_overflow(bvh, j.rays_o, j.rays_d, j.n_rays, j.max_hits, w, j.hit_tri, j.hit_t, j.hit_count, nullptr,
                                nullptr, overflow + 1, stream);
    if (rc != QF_OK) return rc;
    // 3. tile bases + slot total (device; a copy on its way to the pinned block), 4. the tile pack with the re-origin rule
    rc = qf_tile_offsets(j.hit_count, j.max_hits, w, h, j.tile_base, j.total, overflow, overflow + 1, j.host_block, j.dropped,
                         stream);
    if (rc != QF_OK) return rc;
    rc = qf_pack_tiles(j.rays_o, j.rays_d, w, h, j.max_hits, j.hit_tri, j.hit_t, j.hit_count, j.tile_base, j.total, j.xyz_c,
                       j.dirs_c, j.depth_c, j.tri_c, nullptr, nullptr, j.min_separation, j.final_count, j.dropped, nullptr, 1,
                       stream);
    if (rc != QF_OK) return rc;
    if (!j.field) return QF_OK;                 // sampling only
    // 5. the field over min(*total, cap) points, 6. the tile compositor
    rc = qf_field_forward(j.field, j.table, j.base_w, j.head_ngp_w, j.head_sg, j.xyz_c, j.dirs_c, cap, j.total, nullptr,
                          j.rgb_c, j.sigma_c, nullptr, nullptr, nullptr, stream);
    if (rc != QF_OK) return rc;
    return qf_composite_tiles(j.rgb_c, j.sigma_c, j.depth_c, j.delta_const, j.final_count, j.max_hits, j.tile_base, w, h,
                              j.bg_mode, j.bkgd, j.out_rgb, j.out_alpha, j.out_depth, nullptr, j.out_packed, stream);
}
