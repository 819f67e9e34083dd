/* qf_hip.h -- C ABI of libqf_hip.so, the MI355X (gfx950) implementation of the
 * quadrature-field render hot path of ubc-vision/quadraturefields.
 *
 * The reference is pure Python and has no C ABI of its own: every entry
 * point below names the Python call site(s), in /root/reference, whose arithmetic it replaces.
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer owned by the caller
 *     unless the parameter is documented "host";
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream),
 *     with no implicit synchronisation and no hidden allocation except opaque handles;
 *   - the return value is a status code (QF_OK == 0, negative on error); nothing throws across
 *     the boundary; qf_status_string() maps a code to text;
 *   - int64 sizes; per-sample ray / triangle ids are int64 at the Python surface (as in the
 *     reference's LongTensors) and int32 inside the traversal.
 */
#ifndef QF_HIP_H
#define QF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QF_OK 0
#define QF_ERR_INVALID_ARGUMENT (-1)
#define QF_ERR_HIP (-2)
#define QF_ERR_UNSUPPORTED (-3)
#define QF_ERR_NO_DEVICE (-4)

#define QF_ABI_VERSION 5
#define QF_MAX_LEVELS 16
#define QF_MAX_LOBES 8

const char *qf_status_string(int status);
int qf_abi_version(void);
/* Number of compute units of the current device (host query); <0 on error. */
int qf_device_cu_count(void);

/* ------------------------------------------------------------------------------------------
 * Multi-resolution hash grid (tiny-cuda-nn "HashGrid", 3-D, F = 2, linear interpolation).
 * Replaces tcnn.Encoding / the encoding half of tcnn.NetworkWithInputEncoding as configured at
 * examples/radiance_fields/ngp.py:340-358,709-727 and examples/field.py:157-171.
 * qf_grid_desc_init is a HOST computation of the level table.       */
typedef struct qf_grid_desc {
    uint32_t n_levels;             /* must be 16 for the field kernels */
    uint32_t n_features;           /* must be 2 */
    uint32_t log2_hashmap_size;
    uint32_t base_resolution;
    float per_level_scale;
    uint32_t hashed_mask;          /* bit l set: level l uses the spatial hash */
    uint32_t offset[QF_MAX_LEVELS + 1]; /* first table row of each level; offset[n_levels] = rows */
    uint32_t resolution[QF_MAX_LEVELS];
    float scale[QF_MAX_LEVELS];
} qf_grid_desc;

int qf_grid_desc_init(qf_grid_desc *desc /* host, out */, uint32_t n_levels,
                      uint32_t log2_hashmap_size, uint32_t base_resolution,
                      double per_level_scale);

/* [n,3] positions already in grid space (nominally [0,1]) -> [n,32] features, level-major.
 * Replaces tcnn.Encoding.forward (field.py:197-199).                                          */
int qf_grid_encode(const qf_grid_desc *desc /* host */, const float *table /* [rows,2] */,
                   const float *x01, int64_t n, float *out /* [n,32] */, void *stream);

/* Backward of qf_grid_encode (training side).  Replaces tcnn GridEncoding's backward that torch autograd
 * reaches when the reference trains the field (examples/train_finetune.py:465-533, examples/field.py:229-238).
 * dfeat [n,32] = dL/d(encoding).  grad_table [rows,2] is ACCUMULATED into (zero it first) with fp32 atomics; pass
 * NULL to skip.  grad_x01 [n,3] is overwritten; pass NULL to skip.                               */
int qf_grid_encode_backward(const qf_grid_desc *desc /* host */, const float *table, const float *x01,
                            const float *dfeat, int64_t n, float *grad_table, float *grad_x01, void *stream);
/* The same with a caller-provided device workspace (qf_grid_backward_workspace_bytes(n) bytes; torch memory in the
 * Python layer): from 2^15 points on the table gradient is computed by the LDS-partitioned scatter -- every
 * (level, 20 000-row partition, point chunk) accumulates in a workgroup's LDS and is added to grad_table with
 * contiguous atomics, instead of one scattered memory-side atomic request per corner pair.  Same result up to fp32
 * summation order.  workspace NULL / too small or a small batch: falls back to qf_grid_encode_backward.
 * The walk costs O(partitions x points) per level, so a level with more than 64 partitions (1.28 M rows: every
 * hashed level of the deformation field's T = 2^24 table) keeps the quad atomics -- decided per level.            */
int64_t qf_grid_backward_workspace_bytes(int64_t n);
int qf_grid_encode_backward_ws(const qf_grid_desc *desc, const float *table, const float *x01,
                               const float *dfeat, int64_t n, float *grad_table, float *grad_x01,
                               void *workspace, int64_t workspace_bytes, void *stream);

/* Second order: the backward of the INPUT gradient above, grad_x01 = J(x; table)^T dfeat, which the reference
 * reaches through Field.field_grad(create_graph=True) (examples/field.py:206-238) and the losses on it
 * (field.py:253-270).  v [n,3] = dL/d(grad_x01).  Outputs, each optional (NULL to skip): g_dfeat [n,32] =
 * dL/d(dfeat); g_x01 [n,3] = dL/d(x01) (mixed second derivatives of the trilinear blend); grad_table [rows,2],
 * ACCUMULATED into with fp32 atomics.                                                            */
int qf_grid_encode_double_backward(const qf_grid_desc *desc /* host */, const float *table, const float *x01,
                                   const float *dfeat, const float *v, int64_t n, float *g_dfeat,
                                   float *g_x01, float *grad_table,
                                   void *workspace /* or NULL; qf_grid_backward_workspace_bytes(n): large batches take
                                                      the partitioned table scatter of qf_grid_encode_backward_ws */,
                                   int64_t workspace_bytes, void *stream);

/* Grid + 1-hidden-layer 64-wide MLP: x01 [n,3] -> raw [n,16].  Replaces
 * tcnn.NetworkWithInputEncoding.forward for mlp_base (ngp.py:764-768); base_w as below.       */
int qf_grid_mlp_forward(const qf_grid_desc *grid /* host */, const float *table, const float *base_w,
                        const float *x01, int64_t n, float *out16, void *stream);

/* ------------------------------------------------------------------------------------------
 * Radiance fields.  Replaces NGPRadianceField.query_density/_query_rgb/forward
 * (ngp.py:757-809) and NGPRadianceFieldSGNew.query_density/_query_rgb/features/forward
 * (ngp.py:404-470) including the tcnn FullyFusedMLP / SphericalHarmonics / BasicDecoder they
 * call.  Weight pointers use the reference's state-dict layouts:
 *   base_w : [64*32 | 16*64] row-major [out,in], no bias (tcnn mlp_base network params)
 *   head NGP : [64*32 | 64*64 | 16*64] row-major, no bias; input = [SH16 | geo15 | 1.0]
 *   head SG  : BasicDecoder 15->64->64->(3+7L): w1 [64,15], b1 [64], w2 [64,64], b2 [64],
 *              wout [3+7L,64], bout [3+7L]                                                    */
#define QF_HEAD_NONE 0   /* density (+ optional 15 geo features) only */
#define QF_HEAD_NGP 1    /* SH4 + 2-hidden-layer tcnn head -> sigmoid rgb */
#define QF_HEAD_SG 2     /* BasicDecoder + spherical-Gaussian mixture -> sigmoid rgb */
#define QF_HEAD_SG_FEATURES 3 /* BasicDecoder output + density: [n, 3+7L+1] */

typedef struct qf_field_desc {
    qf_grid_desc grid;
    float aabb[6];        /* xmin,ymin,zmin,xmax,ymax,zmax (ngp.py:677-678) */
    int32_t head;         /* QF_HEAD_* */
    int32_t n_lobes;      /* SG heads: 1..QF_MAX_LOBES */
} qf_field_desc;

typedef struct qf_sg_head {
    const float *w1, *b1, *w2, *b2, *wout, *bout;
} qf_sg_head;

/* xyz [n,3] world positions, dirs [n,3] unit view directions (may be NULL for HEAD_NONE /
 * HEAD_SG_FEATURES).  n_device (or NULL; also on qf_field_forward_bf16, qf_deform_field_forward,
 * qf_texture_shade_points and, as total_device, qf_deform_resort_tiles): the point count lives in DEVICE memory
 * (int64; qf_tile_offsets' / qf_frame_offsets' total) and n is the capacity of the arrays -- min(*n_device, n) points
 * are processed.  A render-only frame needs no host wait between its tile pack and its field kernel this way, and is a
 * fixed sequence of launches into worst-case buffers (capturable as a HIP graph).
 * order: NULL, or a permutation of [0,n) giving the order in which points are PROCESSED
 * (16 consecutive slots share a wave pass; spatially coherent groups hit the caches better, see
 * qf_coherent_order); outputs are indexed by point, so results do not depend on it.
 * Outputs (any may be NULL when not produced by the head):
 *   rgb [n,3]; sigma [n] (density after exp(x-1)*selector); geo [n,15]; features [n,3+7L+1]. */
int qf_field_forward(const qf_field_desc *desc /* host */, const float *table,
                     const float *base_w, const float *head_ngp_w, const qf_sg_head *head_sg /* host */,
                     const float *xyz, const float *dirs, int64_t n, const int64_t *n_device, const int32_t *order,
                     float *rgb, float *sigma, float *geo, float *features,
                     float *enc_out /* [n,32] or NULL: the hash-grid encoding, for the training step's backward */,
                     void *stream);

/* bf16 variant (BASELINE config 3): hash tables as bf16x2 rows, MLP weights as bf16 (round-to-nearest-even copies of
 * the fp32 parameters, same layouts), activations rounded to bf16 between layers, fp32 accumulate on
 * v_mfma_f32_16x16x32_bf16.  tcnn itself runs these networks in fp16 (ngp.py:340-358), so this is the reduced
 * precision mode of the same calls.  Heads: QF_HEAD_NONE, QF_HEAD_NGP, QF_HEAD_SG.  SG biases b2 / bout stay fp32
 * (they initialise the accumulator); b1 rides in the first weight tile and is bf16.                              */
typedef struct qf_sg_head_bf16 {
    const uint16_t *w1, *b1, *w2;
    const float *b2;
    const uint16_t *wout;
    const float *bout;
} qf_sg_head_bf16;
int qf_field_forward_bf16(const qf_field_desc *desc /* host */, const uint16_t *table /* [rows,2] bf16 */,
                          const uint16_t *base_w, const uint16_t *head_ngp_w,
                          const qf_sg_head_bf16 *head_sg /* host */, const float *xyz, const float *dirs,
                          int64_t n, const int64_t *n_device, const int32_t *order, float *rgb, float *sigma, float *geo,
                          void *stream);

/* Backward of the two MLPs of NGPRadianceField (ngp.py:757-809), fused: recomputes the forward pass from the grid
 * encodings, back-propagates dL/drgb [n,3] and dL/ddensity [n] to dL/denc [n,32] (-> qf_grid_encode_backward) and
 * ACCUMULATES the weight gradients into grad_base_w [3072] / grad_head_w [7168] (zero them first; layouts as
 * base_w / head NGP above).  selector [n] uint8: the point is inside the aabb (density = exp(raw-1)*selector).
 * Replaces the backward of tcnn's FullyFusedMLP that torch autograd reaches in train_finetune.py:465-533.      */
int qf_ngp_mlp_backward(const float *enc /* [n,32] */, const float *dirs /* [n,3] */, const uint8_t *selector,
                        const float *d_rgb, const float *d_sigma, const float *base_w, const float *head_w,
                        int64_t n, float *d_enc /* [n,32] */, float *grad_base_w, float *grad_head_w,
                        void *stream);

/* The same for NGPRadianceFieldSGNew (ngp.py:404-470; the SG-fitting step train_fit_sg.py:439-461): base MLP +
 * BasicDecoder 15 -> 64 -> 64 -> (3+7L) with biases.  d_features [n, >= 3+7L] (row stride d_stride) comes from
 * qf_sg_features_to_rgb_backward.  grad_head: the six gradient arrays, same shapes as `head` (ACCUMULATED into, like
 * grad_base_w [3072]).                                                                           */
int qf_sg_mlp_backward(const float *enc /* [n,32] */, const uint8_t *selector, const float *d_features,
                       int64_t d_stride, const float *d_sigma, const float *base_w,
                       const qf_sg_head *head /* host */, int32_t n_lobes, int64_t n, float *d_enc,
                       float *grad_base_w, const qf_sg_head *grad_head /* host: device pointers, written */,
                       void *stream);

/* rgb = sigmoid(diffuse + sum_l c_l exp(|lambda_l| (a_l/|a_l| . d - 1))).
 * Replaces NGPRadianceFieldSGNew.features_to_rgb (ngp.py:456-461, discretize=False).
 * features [n, 3+7L] (row stride `feat_stride` floats), dirs [n,3] -> rgb [n,3].              */
int qf_sg_features_to_rgb(const float *features, int64_t feat_stride, const float *dirs,
                          int64_t n, int32_t n_lobes, float *rgb, void *stream);
/* Its backward (the SG-fitting step, train_fit_sg.py:439-461): d_rgb [n,3] -> d_features (row stride d_stride; the
 * first 3+7L columns of each row are written).                                                  */
int qf_sg_features_to_rgb_backward(const float *features, int64_t feat_stride, const float *dirs,
                                   const float *d_rgb, int64_t n, int32_t n_lobes, float *d_features,
                                   int64_t d_stride, void *stream);

/* Deformation field: examples/field.py Field.density (:186-203) as used at utils.py:555-566:
 * x01 = (x+scale)/(2 scale); cat[x01, grid(x01)] (35) -> hidden -> hidden -> 1, ReLU, biases.
 * w1 [hidden,35], b1, w2 [hidden,hidden], b2, wout [1,hidden], bout [1]; hidden must be 32.
 * order: NULL or a processing permutation as in qf_field_forward (results unchanged).          */
int qf_deform_field_forward(const qf_grid_desc *grid /* host */, const float *table, float scale,
                            int32_t hidden, const float *w1, const float *b1, const float *w2,
                            const float *b2, const float *wout, const float *bout,
                            const float *xyz, int64_t n, const int64_t *n_device /* or NULL, see qf_field_forward */,
                            const int32_t *order, float *out /* [n] */, float *enc_out /* [n,32] or NULL */,
                            void *stream);

/* Backward of the decoder of qf_deform_field_forward, fused (training: the deformation field of
 * train_finetune.py:387-399 is optimised together with the radiance field).  enc [n,32] = grid encoding of x01 [n,3],
 * d_out [n] = dL/dfield.  d_enc [n,32] -> qf_grid_encode_backward; d_x01 (optional) = the part of dL/dx01 that enters
 * through the first layer's three x01 columns.  The six gradient arrays (shapes of w1 [32,35], b1, w2 [32,32], b2,
 * wout [32], bout [1]) are ACCUMULATED into.                                                    */
int qf_deform_mlp_backward(const float *enc, const float *x01, const float *d_out, const float *w1,
                           const float *b1, const float *w2, const float *b2, const float *wout, int64_t n,
                           float *d_enc, float *d_x01, float *g_w1, float *g_b1, float *g_w2, float *g_b2,
                           float *g_wout, float *g_bout, void *stream);

/* One step of torch.optim.Adam (the reference's optimiser, train_finetune.py:402-417; amsgrad off) on one flat fp32
 * parameter tensor, in ONE launch: the update of torch's foreach implementation element for element --
 *   m += (1-b1)(g-m); v = v b2 + (1-b2) g g; p += -(lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * (weight_decay: g += wd p first; maximize: g = -g) -- where torch issues eleven launches and seven passes.  step = t,
 * counted from 1.  The hyper-parameters come as doubles: the derived scalars (1 - b1, 1 - b2, the step size, the bias
 * correction) are evaluated in double as torch's Python does and rounded to fp32 once.  All four arrays 16-byte aligned.  Values agree with torch to rounding (each operation individually
 * rounded here; torch's kernels leave contraction to the compiler).                                    */
int qf_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int32_t maximize, int64_t step, void *stream);

/* xyz_out = xyz + dh, ts_out = ts + dd with dd = tanh(f)*scaling*(1,1,1) . dir and dh = dd * dir (utils.py:566-571,
 * every operation rounded on its own as the reference's tensor ops are).  In place when xyz_out == xyz / ts_out == ts.
 * dh_out [n,3] or NULL: the reference's ``dh`` itself (what MeshFinetune.update_d accumulates, utils.py:570,598).  */
int qf_apply_deformation(const float *f /* [n] */, float scaling, const float *dirs, const float *xyz /* [n,3] */,
                         const float *ts /* [n] */, int64_t n, float *xyz_out, float *ts_out, float *dh_out,
                         void *stream);

/* ------------------------------------------------------------------------------------------
 * Packed compositing.
 * kaolin duck-types (kaolin.render.spc, called at utils.py:869-879, mesh_utils.py:407).      */
int qf_mark_pack_boundaries(const int64_t *ridx, int64_t n, uint8_t *boundary /* [n] bool */,
                            void *stream);
/* Segment heads -> out[segment, c] = sum_seg w*feats, weights[n] = exp(-excl cumsum tau)(1-exp(-tau)).
 * `seg_start` [n_seg] holds the index of the first sample of every segment (ascending),
 * exclusive != 0 as at utils.py:872.                                                          */
int qf_exponential_integration(const float *feats /* [n,c] */, int32_t c, const float *tau /* [n] */,
                               const int64_t *seg_start, int64_t n_seg, int64_t n, int32_t exclusive,
                               float *out /* [n_seg,c] */, float *weights /* [n] */, void *stream);
int qf_sum_reduce(const float *feats /* [n,c] */, int32_t c, const int64_t *seg_start,
                  int64_t n_seg, int64_t n, float *out /* [n_seg,c] */, void *stream);

/* Fused derive_properties (utils.py:863-898) for samples sorted by (ray, depth):
 * per ray tau = sigma*delta, w = exp(-excl cumsum tau)(1-exp(-tau)); writes the reference's
 * full-image buffers rgb [n_rays,3], alpha [n_rays,1], depth [n_rays,1] (rays without samples:
 * white unless bg_mode == black, alpha 0, depth 0) and weights [n].
 * bg_mode: 0 white, 1 black, 2 "random" (uses bkgd[3]); the double-alpha quirk is reproduced.
 * sample_index: NULL, or int32 [n]: sample i's colour and density are rgb_s[sample_index[i]] / sigma[sample_index[i]]
 * (they were produced in the field kernel's processing order, see qf_coherent_layout); depth, deltas, index_ray and
 * the weights output stay indexed by i.
 * A sample whose ray id lies outside [0, n_rays) contributes to no pixel (its weight is still written); n > 0
 * with n_rays == 0 is QF_ERR_INVALID_ARGUMENT (the reference raises IndexError there).               */
#define QF_BG_WHITE 0
#define QF_BG_BLACK 1
#define QF_BG_CUSTOM 2
#define QF_BG_NONE 3   /* no blend and no double alpha: rgb = sum(w c), rays without samples 0 -- the plain sums of
                        * nerfacc's accumulate_along_rays (field_rendering.py:100-156), used by the differentiable
                        * route of `rendering` */
int qf_derive_properties(const float *rgb_s /* [n,3] */, const float *sigma /* [n] */,
                         const float *depth /* [n] */, const float *deltas /* [n] or NULL */,
                         float delta_const, const int64_t *index_ray /* [n] */, int64_t n,
                         int64_t n_rays, int32_t bg_mode, const float *bkgd /* [3] or NULL */,
                         const int32_t *sample_index, float *out_rgb, float *out_alpha, float *out_depth,
                         float *weights, void *stream);

/* Backward of qf_derive_properties (training side: the loss of examples/train_finetune.py:489-533 back-propagates
 * through utils.py:139-186).  g_rgb [n_rays,3], g_alpha / g_depth [n_rays] or NULL -> grad_rgb_s [n,3],
 * grad_sigma [n], grad_depth [n] or NULL.  Same inputs as the forward; one thread per ray.  Samples whose ray id
 * is outside [0, n_rays) get zero gradients (the forward skips them too).                        */
int qf_derive_properties_backward(const float *rgb_s, const float *sigma, const float *depth,
                                  const float *deltas /* [n] or NULL */, float delta_const,
                                  const int64_t *index_ray, int64_t n, int64_t n_rays, int32_t bg_mode,
                                  const float *bkgd /* [3] or NULL */, const float *g_rgb,
                                  const float *g_alpha, const float *g_depth, float *grad_rgb_s,
                                  float *grad_sigma, float *grad_depth, void *stream);

/* nerfacc duck-types (nerfacc.pack / nerfacc.scan, imported at field_rendering.py:10-11).     */
int qf_pack_info(const int64_t *ray_indices /* sorted, [n] */, int64_t n, int64_t n_rays,
                 int64_t *packed_info /* [n_rays,2] = (start,count) */, void *stream);
/* mode 0: exclusive sum, 1: exclusive product, per chunk of packed_info.                      */
int qf_exclusive_scan(const float *x, const int64_t *packed_info, int64_t n_rays, int64_t n,
                      int32_t mode, float *out, void *stream);
/* out[ray, :] += w_i * values_i (values NULL: accumulate w).  field_rendering.py:483-573.
 * Deterministic: per-ray sequential over packed_info, no atomics.                            */
int qf_accumulate_along_rays(const float *weights, const float *values /* [n,c] or NULL */,
                             int32_t c, const int64_t *packed_info, int64_t n_rays, int64_t n,
                             float *out /* [n_rays,c], overwritten */, void *stream);
/* Fused render_weight_from_density + the three accumulate_along_rays of `rendering`
 * (field_rendering.py:100-156): trans/alphas/weights [n] and colors[n_rays,3],
 * opacities[n_rays,1], depths[n_rays,1] (already divided by clamp_min(opacity, eps) and with
 * bkgd*(1-opacity) added when bkgd != NULL).                                                  */
int qf_render_from_density(const float *t_starts, const float *t_ends, const float *sigmas,
                           const float *rgbs /* [n,3] */, const int64_t *packed_info,
                           int64_t n_rays, int64_t n, const float *bkgd /* [3] or NULL */,
                           float *weights, float *trans, float *alphas, float *colors,
                           float *opacities, float *depths, void *stream);

/* ------------------------------------------------------------------------------------------
 * Ray / mesh quadrature points.
 * Replaces trimesh RayMeshIntersector (mesh_utils.py:223,350-354), the OptiX module
 * build.lib.intersector (mesh_utils.py:77-96: Intersector(vertices, max_hits, device),
 * find_intersections, update_vertices) and the CPU sort of sampling_raytrace_numpy /
 * sampling_indexing (mesh_utils.py:359-381,394-403).                                          */
typedef struct qf_bvh qf_bvh; /* opaque; owns device memory */

/* tri_verts: HOST pointer, [n_tri][3][3] fp32 (the layout of mesh.vertices[mesh.faces]).
 * Builds a binned-SAH binary tree on the host, collapses it into the 8-wide tree the device traverses
 * (one 256-byte node = 8 child boxes, one leaf = up to 8 triangles) and uploads it.
 * n_tri < 2^28.                                                                                */
int qf_bvh_create(const float *tri_verts /* host */, int64_t n_tri, qf_bvh **out);
/* Same, with the level (1..32) below which the builder stops taking SAH splits and halves the index range
 * (qf_bvh_create uses 32).  Any value gives a valid tree and identical hits; small values trade traversal speed for
 * a shallower tree -- used by the tests to exercise the depth bound.                            */
int qf_bvh_create_ex(const float *tri_verts /* host */, int64_t n_tri, int32_t sah_depth, qf_bvh **out);
/* Same topology, new vertex positions (Intersector.update_vertices / train_finetune.py:716-718), HOST vertices.
 * Waits for the device (hipDeviceSynchronize) before the upload, so no traversal still reads the old tree.   */
int qf_bvh_refit(qf_bvh *bvh, const float *tri_verts /* host */, int64_t n_tri);
/* The same refit with DEVICE vertices, stream-ordered, no host round trip: d_tri_verts [n_tri][3][3] fp32 in the
 * ORIGINAL triangle order (vertices[faces] of the training loop, train_finetune.py:708-718).  One launch permutes
 * them into leaf order, then one launch per tree level recomputes the child boxes bottom-up.  The box inflation of
 * the build is reused (x1.25), valid while the mesh extent stays within 25 % of the build's.             */
int qf_bvh_refit_device(qf_bvh *bvh, const float *d_tri_verts /* device */, int64_t n_tri, void *stream);
void qf_bvh_destroy(qf_bvh *bvh);
int64_t qf_bvh_num_triangles(const qf_bvh *bvh);
int64_t qf_bvh_num_nodes(const qf_bvh *bvh);        /* binary build tree */
int64_t qf_bvh_num_wide_nodes(const qf_bvh *bvh);   /* the 8-wide tree that is traversed */
/* Depth of the deepest inner node of the binary tree (root = 1).  The builder guarantees <= 64 for any input: below
 * level 32 it halves the index range instead of taking the SAH split.                            */
int32_t qf_bvh_max_depth(const qf_bvh *bvh);
/* Exact bound of the wide traversal's per-ray stack for this tree (entries); the LDS stack is sized by it. */
int32_t qf_bvh_max_stack(const qf_bvh *bvh);
/* Host copies for inspection/tests: binary nodes as 16 floats each, wide nodes as 64 floats each (csrc/bvh.h),
 * leaf triangle ids.                                                                              */
int qf_bvh_copy_nodes(const qf_bvh *bvh, float *nodes_host, int64_t capacity_nodes);
int qf_bvh_copy_wide_nodes(const qf_bvh *bvh, float *nodes_host, int64_t capacity_nodes);
int qf_bvh_copy_tri_ids(const qf_bvh *bvh, int32_t *ids_host, int64_t capacity);

/* The reference's multi-hit rule (trimesh 3.23.5 ray_pyembree.RayMeshIntersector.intersects_id, called at
 * mesh_utils.py:350-354): after every hit the ray is re-originated `min_separation` past it and
 * the next closest-hit query starts there, so hits closer than that to the previously returned one -- and the second
 * copy of a duplicated face -- are never returned.  min_separation > 0 turns the rule on for every intersection
 * call on this handle (qf_bvh_intersect, qf_bvh_repair_overflow, qf_raster_intersect with sort_lists,
 * qf_filter_hits): with a ray's hits ascending in (t, tri), the first is kept and each later one iff
 * t > t_last_kept + min_separation (fp32), up to max_hits kept hits.  <= 0 (the default of a new handle): every hit
 * counts.  trimesh's value is clip(1e-4 * 100 / mesh.scale, 1e-8, inf) with mesh.scale = the bounding-box diagonal
 * (restated from memory: parity unpinned).                                                          */
int qf_bvh_set_min_separation(qf_bvh *bvh, float min_separation);
float qf_bvh_min_separation(const qf_bvh *bvh);

/* Up to max_hits nearest hits per ray (under the handle's min_separation rule), ascending by (t, triangle id).
 * rays_o, rays_d [n_rays,3]; hit_tri/hit_t [n_rays,max_hits] (unused slots -1 / +inf);
 * hit_count [n_rays].  image_width > 0: rays are a row-major image of that width and are
 * traversed in 8x4 pixel tiles (speed only; results identical).
 * Also the shape of Intersector.find_intersections (int[R*max_hits], -1 padded).             */
int qf_bvh_intersect(const qf_bvh *bvh, const float *rays_o, const float *rays_d, int64_t n_rays,
                     int32_t max_hits, int32_t image_width, int32_t *hit_tri, float *hit_t,
                     int32_t *hit_count, void *stream);
/* Applies the handle's min_separation rule to per-ray lists in ANY order that hold ALL hits of their ray
 * (hit_count <= max_hits; the lists of qf_raster_intersect after qf_bvh_repair_overflow): sorts each list ascending,
 * drops the hits the rule skips, pads, updates hit_count.  No launch when the rule is off.          */
int qf_filter_hits(const qf_bvh *bvh, int64_t n_rays, int32_t max_hits, int32_t *hit_tri, float *hit_t,
                   int32_t *hit_count, void *stream);

/* Camera-coherent variant for rays that are the row-major pixel grid of ONE pinhole camera in the reference's
 * convention (nerf_synthetic.py:341-358): camera_dir = ((x - cx + 0.5)/fx, -(y - cy + 0.5)/fy, -1),
 * ray = normalise(c2w[:3,:3] . camera_dir), origin = c2w[:3,3].  Every triangle is tested only against the pixels
 * of its projected screen box, with the same hit test on the same (rays_o, rays_d) values, so the result is
 * bit-identical to qf_bvh_intersect.  *overflow (device int32) is set to the number of hits that found their
 * ray's list full: when it is non-zero the lists are NOT guaranteed to hold the K nearest hits and the caller must
 * fall back to qf_bvh_intersect.  (hit_count and *overflow are zeroed by the call; with overflow == hit_count + n_rays
 * -- the counter stored right behind the counts -- that is one fill launch.)  The camera is only used to bound the
 * search, never for arithmetic (but see ray_flag below).
 * sort_lists = 1: lists come out ascending in (t, tri) and padded like qf_bvh_intersect; 0: left in arrival
 * order with raw counts, for qf_pack_samples (which sorts while it packs); 2: as 0, and the pass does not write
 * hit_tri at all (one scattered store less per hit) -- for a render-only frame whose tile pack (qf_pack_tiles with
 * tri_c = NULL) never reads the ids; hit_tri must still be valid memory: qf_bvh_repair_overflow writes the lists of
 * the rays it re-traverses.                                                                                     */
typedef struct qf_camera {
    float c2w[12];      /* row-major 3x4 camera-to-world (OpenGL axes: right, up, back | centre) */
    float fx, fy;       /* focal lengths in pixels */
    float cx, cy;       /* principal point (width/2, height/2 in the reference) */
    int32_t width, height;
} qf_camera;
/* cull_chunks != 0 (also on qf_raster_intersect_wide): for a camera that sees only PART of the scene -- the row bands
 * of a frame sharded over several GPUs -- the triangles (stored in BVH leaf order) are first culled in chunks of 64
 * against the camera's image with the triangle pass's own conservative screen-box test, and only the surviving
 * chunks are projected (two small extra launches; the per-frame set-up then shrinks with the band).  Same hits.
 * Culled calls on one handle must be issued on one stream (the handle owns the visible-chunk list).          */
/* ray_flag (or NULL; also on the two variants below) -- the route's precondition, VERIFIED: a device int32 the call
 * zeroes (in the same fill as the counts when it lies right behind the overflow counter: hit_count [n_rays] | overflow
 * | ray_flag) and then raises iff the rays are NOT this camera's pixel grid, i.e. iff for some ray i
 *   - the origin differs BITWISE from the camera centre c2w[:,3], or
 *   - the direction, projected with the pass's own projection, lies more than 0.02 px from the centre of pixel
 *     (i % width, i / width) or behind the camera (the guard band around a projected triangle is 0.25 px; a consistent
 *     fp32 ray reprojects to within 3e-3 px), or
 *   - | |d|^2 - 1 | > 1e-4 (qf_raster_intersect_slabs bins by distance, the re-origin rule compares t with a world
 *     distance).
 * That is what a consistent ray set is (nerf_synthetic.py:341-366); add_ray_direction_noise (:335-340), a stale or wrong
 * camera, another up_sample, another ray order all raise it.  While the flag is down the pass takes the origin from the
 * camera struct (the same value: one scattered 12-byte load less per candidate pixel, 17 % of the pass).  When it is
 * RAISED the pass returns without writing a hit (all counts stay 0) and qf_bvh_repair_overflow(traverse_all_flag =
 * ray_flag) traverses every ray through the BVH -- exact for any rays, no host round trip.  A caller that does not run
 * that repair must read the flag and call qf_bvh_intersect itself.  NULL: NO check -- the caller vouches for the rays,
 * and every ray's own origin is loaded.  (The check is one small launch before the pass, 24 B/ray streamed; with
 * cull_chunks it rides in the culling launch.)                                                                   */
int qf_raster_intersect(qf_bvh *bvh, const qf_camera *cam /* host */, const float *rays_o,
                        const float *rays_d, int64_t n_rays, int32_t max_hits, int32_t *hit_tri, float *hit_t,
                        int32_t *hit_count, int32_t *overflow, int32_t sort_lists, int32_t cull_chunks,
                        int32_t *ray_flag, void *stream);
/* The same pass for dense scenes, where most rays meet more than K = max_hits triangles (thin concentric shells):
 * up to wide_hits >= max_hits candidates per ray are collected in the scratch lists wide_tri / wide_t
 * ([wide_hits, n_rays], slot-major), then every ray's K nearest under (t, tri) -- the rule of qf_bvh_intersect -- go
 * to hit_tri / hit_t [n_rays, max_hits] in arrival order (as with sort_lists = 0) and hit_count is clamped to K.
 * Rays with more than wide_hits candidates keep their raw count (> K) for qf_bvh_repair_overflow; *overflow counts
 * the candidates beyond wide_hits.                                                                           */
int qf_raster_intersect_wide(qf_bvh *bvh, const qf_camera *cam /* host */, const float *rays_o,
                             const float *rays_d, int64_t n_rays, int32_t max_hits, int32_t wide_hits,
                             int32_t *wide_tri, float *wide_t, int32_t *hit_tri, float *hit_t, int32_t *hit_count,
                             int32_t *overflow, int32_t cull_chunks, int32_t *ray_flag, void *stream);
/* qf_raster_intersect_wide without collecting every crossing: the triangle chunks are binned by distance from the camera
 * into n_slabs (2..16) slabs of equal thickness and rasterised nearest slab first, one launch per slab; a hit is
 * accepted by the pass of the slab its t falls into, and a pixel that already holds (selection capacity + 1)
 * candidates when a pass starts is skipped -- its list is a complete depth prefix that contains its K nearest.  The
 * candidate lists are 8-byte keys (t bits << 32 | tri), wide_keys [wide_hits, n_rays] slot-major; wide_hits only has to
 * exceed the selection capacity by one slab's worth of crossings (not 4K).  Same result as qf_raster_intersect_wide:
 * hit_tri / hit_t [n_rays, max_hits] hold each ray's K nearest under (t, tri) in arrival order, rays that lost
 * candidates keep hit_count > max_hits for qf_bvh_repair_overflow.  Exact for unit-length rays from the camera centre
 * only (t = distance): pass ray_flag, which verifies exactly that, and run the repair with it.
 * One stream per handle, as with cull_chunks.                                                          */
int qf_raster_intersect_slabs(qf_bvh *bvh, const qf_camera *cam /* host */, const float *rays_o, const float *rays_d,
                              int64_t n_rays, int32_t max_hits, int32_t wide_hits, int32_t n_slabs, uint64_t *wide_keys,
                              int32_t *hit_tri, float *hit_t, int32_t *hit_count, int32_t *overflow, int32_t *ray_flag,
                              void *stream);
/* The fall-back, per ray: after qf_raster_intersect with sort_lists = 0 (raw counts), re-traverses exactly the rays
 * with hit_count > max_hits through the BVH (exact K nearest under the handle's min_separation rule; their lists and
 * counts are overwritten, in the layout of qf_bvh_intersect) and leaves every other ray's list alone.  No host round
 * trip; a frame with a handful of overflowing pixels costs a few microseconds instead of a whole-image traversal.
 * keep_mask [n_rays] uint64 / raw_count [n_rays] (both or neither, NULL to skip): when the handle's rule is on, the
 * same launch also decides it for every other ray WITHOUT rewriting its list (the lists of a frame are 128 MB; the
 * masks 8): bit i of keep_mask[r] = the i-th hit of ray r in (t, tri) order is kept, raw_count[r] = length of the
 * stored list, hit_count[r] = number kept.  qf_pack_samples takes the two arrays.  Ignored while the rule is off.
 * traverse_all_flag (or NULL): the ray_flag of the camera-coherent pass before it.  Raised (non-zero on the device when
 * this launch runs): EVERY ray is traversed -- the pass found that the rays are not its camera's pixel grid and wrote
 * nothing -- so the lists are those of qf_bvh_intersect whatever the rays were.                                  */
int qf_bvh_repair_overflow(const qf_bvh *bvh, const float *rays_o, const float *rays_d, int64_t n_rays,
                           int32_t max_hits, int32_t image_width, int32_t *hit_tri, float *hit_t,
                           int32_t *hit_count, uint64_t *keep_mask, int32_t *raw_count,
                           const int32_t *traverse_all_flag, void *stream);

/* Occupancy-grid ray marching: nerfacc 0.5.3 OccGridEstimator.sampling -> traverse_grids for one grid level and
 * cone_angle = 0 (examples/utils.py:137-147,266-285).  Samples are [t0 + k*step, t0 + (k+1)*step],
 * t0 = the ray's entry into `aabb` clipped to [near_plane (or t_min[r]), far_plane (or t_max[r])], kept iff the
 * sample's midpoint lies before the exit and inside an occupied cell of `binaries` [rx,ry,rz] (bool bytes, x-major).
 * Two passes around the caller's exclusive scan: _count fills count[n_rays]; _write fills t_starts / t_ends /
 * ray_indices at offsets[r].  aabb (6 floats) and resolution (3 ints) are HOST pointers; t_min / t_max may be NULL. */
int qf_grid_march_count(const float *aabb, const int32_t *resolution, const uint8_t *binaries,
                        const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                        int64_t n_rays, float near_plane, float far_plane, float step, int32_t *count, void *stream);
int qf_grid_march_write(const float *aabb, const int32_t *resolution, const uint8_t *binaries,
                        const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                        int64_t n_rays, float near_plane, float far_plane, float step, const int64_t *offsets,
                        float *t_starts, float *t_ends, int64_t *ray_indices, void *stream);

/* Full-image rays of a pinhole camera, row-major [H*W,3] origins and unit viewdirs, with the arithmetic of
 * SubjectLoader.fetch_data (datasets/nerf_synthetic.py:341-373).  opengl != 0: -y / -z camera axes (the
 * reference's NeRF-synthetic loader).                                                                            */
int qf_generate_rays(const qf_camera *cam /* host */, int32_t opengl, float *origins, float *viewdirs, void *stream);

/* out[index[i]] = max(out[index[i]], values[i]) for i < n; `out` [n_out] is read-modify-written (initialise it).
 * torch_scatter.scatter_max as used for triangle pruning (prune_mesh_after_finetuning.py:355-357).              */
int qf_scatter_max(const float *values, const int64_t *index, int64_t n, int64_t n_out, float *out, void *stream);

/* Offsets of the packed samples: ray_offset[r] = sum_{q<r} min(hit_count[q], max_hits) for r = 0..n_rays, i.e.
 * ray_offset[n_rays] is the total sample count (left in device memory, so the caller can start qf_pack_samples before
 * reading it back).  Replaces the index bookkeeping of mesh_utils.py:359-366 (np.argsort / boolean masks on the host).
 * temp: caller-provided device scratch of at least qf_sample_offsets_temp_bytes(n_rays) bytes.     */
int64_t qf_sample_offsets_temp_bytes(int64_t n_rays);
int qf_sample_offsets(const int32_t *hit_count, int64_t n_rays, int32_t max_hits,
                      int64_t *ray_offset /* [n_rays+1] */, void *temp, int64_t temp_bytes, void *stream);

/* One frame's offsets in three small launches: ray_offset [n_rays+1] exactly as qf_sample_offsets, and -- when
 * tile_base is not NULL (rays = a row-major width x height image) -- tile_base [ceil(w/8)*ceil(h/8)] = the exclusive
 * scan of the 8x8-tile sample totals that qf_coherent_layout takes (round 1: qf_tile_totals + a host-side cumsum).
 * temp: qf_frame_offsets_temp_bytes(n_rays) bytes of device scratch.
 * host_out (or NULL): device-accessible PINNED HOST memory, int64[4]; the scan writes [0] = total, [1] = *overflow_in,
 * [3] = *ray_flag_in there itself ([2] is the pack's), so the frame's readback needs no copy kernel -- the host waits
 * for an event recorded after this call.
 * overflow_in / ray_flag_in (or NULL -> 0): the device counter and the ray flag of qf_raster_intersect (policy only:
 * the repair launch has already acted on both).                                                    */
int64_t qf_frame_offsets_temp_bytes(int64_t n_rays);
int qf_frame_offsets(const int32_t *hit_count, int64_t n_rays, int32_t max_hits, int32_t width, int32_t height,
                     int64_t *ray_offset /* [n_rays+1] */, int64_t *tile_base /* or NULL */, void *temp,
                     int64_t temp_bytes, const int32_t *overflow_in, const int32_t *ray_flag_in, int64_t *host_out,
                     int32_t band_rows, void *stream);
/* band_rows (here and on qf_coherent_layout; 0 = off): the tile grid of the coherent order restarts every band_rows
 * rows, so that the samples of rows [b band_rows, (b+1) band_rows) are ONE contiguous run of the order -- positions
 * [ray_offset[b band_rows width], ray_offset[(b+1) band_rows width]), the same range they occupy ray-major.  With
 * band_rows = chunk / width the 160 000-ray windows of the reference's eval loop (generate_splits,
 * train_finetune.py:419-439) are then slices of ONE frame-wide layout instead of a layout search per window.
 * tile_base then has qf_banded_tile_count(width, height, band_rows) entries (a band's last tile row is partial when
 * band_rows is not a multiple of 8); qf_pack_tiles / qf_composite_tiles take unbanded frames only.                 */
int64_t qf_banded_tile_count(int32_t width, int32_t height, int32_t band_rows);

/* The same for a frame that is only rendered: tile_base (as above) and *total (device int64) alone -- what
 * qf_pack_tiles and qf_composite_tiles take -- in two launches, without the per-ray offsets.              */
int qf_tile_offsets(const int32_t *hit_count, int32_t max_hits, int32_t width, int32_t height, int64_t *tile_base,
                    int64_t *total, const int32_t *overflow_in, const int32_t *ray_flag_in, int64_t *host_out,
                    int32_t *zero_word /* or NULL: a device int32 the scan launch zeroes on the way -- the dropped-hit
                                          counter of the qf_pack_tiles that follows (dropped_is_zero) */, void *stream);

/* Packs the per-ray hit lists into the sample arrays sampling_raytrace_numpy returns
 * (mesh_utils.py:359-387), already sorted by (ray, depth): location = o + t d in float64,
 * dirs = d/(|d|+1e-7), depth = |location - o| (float64, rounded to fp32 at the end).
 * The per-ray lists may be in any order (they are sorted by (t, tri) first); counts above max_hits are clamped.
 * ray_offset [n_rays] = exclusive prefix sum of min(hit_count, max_hits).
 * inverse (NULL to skip; from qf_coherent_layout): sample -> position in the field kernel's processing order;
 * xyz_c / dirs_c [n,3] then receive a second copy of the positions / directions IN that order, so that
 * qf_field_forward can stream them (order = NULL) and write its outputs sequentially, and
 * qf_derive_properties picks colour and density back up through sample_index = inverse.  Measured: the indirection
 * through `order` costs the field kernel 10 % (two scattered sector reads and a scattered write per point).
 * depth_c (with inverse; NULL to skip): the depths in that order too -- with it qf_composite_tiles composites the
 * frame straight from the field kernel's outputs, and neither the inverse map nor index_ray is read again.
 * xyz / dirs / origins may then be NULL (all three) for a caller that only renders: the ray-major copies of the
 * positions are skipped; index_ray, depth and index_tri are always written.
 * keep_mask / raw_count (both or neither; from qf_bvh_repair_overflow): the stored list of ray r has raw_count[r]
 * entries of which the ones whose (t, tri)-sorted position has its bit set in keep_mask[r] are packed --
 * hit_count[r] (and ray_offset) already count only those.
 * close_flag (device int32, or NULL) with min_separation > 0 and no keep_mask: the OPTIMISTIC route of the re-origin
 * rule.  The lists are packed as if the rule dropped nothing and, once a list is sorted, that is verified exactly; if
 * some ray does have a hit within min_separation of its predecessor, *close_flag is set to 1 (the caller zeroes it)
 * and the caller must run qf_bvh_repair_overflow with keep_mask, the offsets and this call again.  A scene without
 * near-coincident faces never raises it and pays nothing for the rule.                              */
int qf_pack_samples(const float *rays_o, const float *rays_d, int64_t n_rays, int32_t max_hits,
                    const int32_t *hit_tri, const float *hit_t, const int32_t *hit_count,
                    const int64_t *ray_offset, float *xyz, float *dirs, int64_t *index_ray,
                    float *depth, int64_t *index_tri, float *origins, const int32_t *inverse,
                    float *xyz_c, float *dirs_c, float *depth_c /* [n] or NULL */, const uint64_t *keep_mask,
                    const int32_t *raw_count, float min_separation, int32_t *close_flag, void *stream);

/* qf_pack_samples for a frame that is only rendered (rays = a row-major width x height image): writes the positions,
 * unit directions and depths of the samples DIRECTLY in the coherent order below -- the order qf_field_forward streams
 * and qf_composite_tiles composites -- one wave per 8x8 tile, and nothing else: no ray-major arrays, no order, no
 * inverse map.  tile_base and total (= ray_offset + n_rays) from qf_frame_offsets on the same hit_count.
 * The re-origin rule: with keep_mask / raw_count (from qf_bvh_repair_overflow) as in qf_pack_samples; otherwise, with
 * min_separation > 0, it is applied HERE, on the sorted list (what qf_filter_hits does) -- no optimistic guess, no
 * second pass.  The hits it drops leave unused slots at the end of their tile (filled with a copy of a real sample so
 * that qf_field_forward can stream [0, *total) blindly): final_count [w*h] receives every pixel's kept count -- hand
 * THAT to qf_composite_tiles -- and *dropped (device int32) the frame's number of dropped hits (samples of the frame =
 * *total - *dropped).  Without host_out the call zeroes *dropped first and leaves the count in it.  host_out (pinned
 * host int64[3]): host_out[2] = the count after the call's kernels, for a caller that wants it without a copy; *dropped
 * must then be 0 on entry and is 0 again afterwards (a frame loop that keeps one counter pays no memset launch).
 * final_count may be given without the rule.
 * tri_c (or NULL): the samples' triangle ids in the same order, int32 (the baked-texture render looks its texels up
 * by triangle, utils.py:1055-1063 -- qf_texture_shade_points takes them as they are); without it hit_tri is not read at
 * all (may be NULL).
 * Values equal qf_pack_samples' xyz_c / dirs_c / depth_c bit for bit (position for position when nothing is dropped). */
int qf_pack_tiles(const float *rays_o, const float *rays_d, int32_t width, int32_t height, int32_t max_hits,
                  const int32_t *hit_tri, const float *hit_t, const int32_t *hit_count, const int64_t *tile_base,
                  const int64_t *total, float *xyz_c, float *dirs_c, float *depth_c, int32_t *tri_c /* or NULL */,
                  const uint64_t *keep_mask, const int32_t *raw_count, float min_separation, int32_t *final_count,
                  int32_t *dropped, int64_t *host_out,
                  int32_t dropped_is_zero /* non-zero, without host_out: *dropped is already zero (qf_tile_offsets
                                             zero_word) -- no memset launch; the count stays in *dropped */,
                  void *stream);

/* Spatially coherent PROCESSING order for qf_field_forward when the rays are a row-major width x height image:
 * (8x8 pixel tile, hit rank, pixel in tile).  Two steps around one exclusive scan the caller does:
 *   qf_tile_totals    : tile_total[tile] = sum of hit_count over the tile's pixels (tiles row-major, ceil(w/8) per row)
 *   qf_coherent_order : order[tile_base[tile] + slot] = ray_offset[ray] + k   (tile_base = exclusive scan of totals)
 * order is a permutation of [0, sum(hit_count)).  New on this platform (no reference counterpart): it only
 * changes which samples share a wave pass, never a result.                                                    */
int qf_tile_totals(const int32_t *hit_count, int32_t width, int32_t height, int64_t *tile_total, void *stream);
int qf_coherent_order(const int32_t *hit_count, const int64_t *ray_offset, const int64_t *tile_base,
                      int32_t width, int32_t height, int32_t *order, void *stream);
/* Same order together with its inverse: inverse[sample] = position (see qf_pack_samples).  order may be NULL
 * (a render-only frame streams the coherent copies and only needs the inverse).                  */
int qf_coherent_layout(const int32_t *hit_count, const int64_t *ray_offset, const int64_t *tile_base,
                       int32_t width, int32_t height, int32_t *order, int32_t *inverse,
                       int32_t band_rows /* as qf_frame_offsets' */, void *stream);

/* derive_properties (utils.py:863-898; the eval render of train_finetune.py:597-607) on a frame whose per-sample
 * colours, densities and depths are stored in the coherent order above (rgb_c / sigma_c = qf_field_forward's outputs
 * on xyz_c / dirs_c, depth_c from qf_pack_samples): one wave per 8x8 tile, every load a contiguous run, pixels
 * without samples get their background from the same launch.  hit_count [w*h] (pixel r has min(hit_count[r],
 * max_hits) samples) and tile_base as qf_frame_offsets / qf_coherent_layout took them; constant step delta_const
 * (find_deltas, mesh_utils.py:225-231).  Same values as qf_derive_properties on the ray-major arrays, bit for bit
 * (same per-ray operation order).  weights_c: the weights in the coherent order, or NULL.
 * out_packed (or NULL): [w*h, 5] = rgb | alpha | depth per pixel in ONE array instead of the three (which may then be
 * NULL) -- the 20 B/ray block a row band of a sharded frame sends to the other ranks.             */
int qf_composite_tiles(const float *rgb_c /* [n,3] */, const float *sigma_c /* [n] */, const float *depth_c /* [n] */,
                       float delta_const, const int32_t *hit_count /* [w*h] */, int32_t max_hits,
                       const int64_t *tile_base, int32_t width, int32_t height, int32_t bg_mode,
                       const float *bkgd /* [3] or NULL */, float *out_rgb, float *out_alpha, float *out_depth,
                       float *weights_c, float *out_packed, void *stream);

/* out_rows[y] = sum over the pixels of row y of min(hit_count, max_hits): quadrature points per pixel row of a frame
 * (band).  No reference counterpart: the cost profile the band-sharded renderer balances its cuts with.   */
int qf_row_sample_counts(const int32_t *hit_count, int32_t max_hits, int32_t width, int32_t height, float *out_rows,
                         void *stream);

/* One render-only camera frame as ONE host call: what FrameRenderer.render_async enqueues for a pinhole frame of the
 * reference's eval loop (the rays of nerf_synthetic.py:310-373 through render_image_finetune_with_occgrid,
 * examples/utils.py:510-620, scaling = 0), as a fixed sequence of this library's own launches:
 *   qf_raster_intersect (arrival order) -> qf_bvh_repair_overflow (no masks) -> qf_tile_offsets (zero_word = dropped)
 *   -> qf_pack_tiles (re-origin rule on the sorted lists; the dropped-hit count stays in *dropped)
 *   -> [field != NULL] qf_field_forward over min(*total, n_rays * max_hits) points -> qf_composite_tiles.
 * Nothing in between returns to the host: the sample count lives in total[0] (device), and its copy + the raster
 * overflow count travel to host_block[0..1] (pinned, device-writable; may be NULL) on their own.  Every pointer is
 * caller-owned device memory of the stated size; the results are those of the separate calls, bit for bit (the
 * function only composes them -- a row band of a frame sharded over 8 GPUs is ~0.3 ms of kernels, and a dozen
 * separately bound calls per band kept the host behind the GPU).  No reference counterpart.                       */
typedef struct qf_frame_job {
    const qf_camera *camera;            /* host; the rays are its pixel grid, row-major */
    const float *rays_o, *rays_d;       /* [n_rays,3] */
    int64_t n_rays;                     /* == camera->width * camera->height */
    int32_t max_hits;                   /* K */
    int32_t cull_chunks;                /* qf_raster_intersect's: the camera sees a part of the scene */
    float min_separation;               /* the tile pack's re-origin rule (0 = off) */
    int32_t bg_mode;                    /* QF_BG_* */
    float delta_const;                  /* render_step_size */
    int32_t reserved_;
    /* scratch */
    int32_t *hit_tri;                   /* [n_rays, K] */
    float *hit_t;                       /* [n_rays, K] */
    int32_t *hit_count;                 /* [n_rays + 2]: counts | raster overflow counter | ray flag */
    int32_t *final_count;               /* [n_rays]: counts after the rule (what the compositor walks) */
    int64_t *tile_base;                 /* [ceil(w/8) * ceil(h/8)] */
    int64_t *total;                     /* [3] device: slots | overflow | - */
    int64_t *host_block;                /* or NULL: pinned [4] */
    int32_t *dropped;                   /* [1]: hits the rule dropped (zeroed by the sequence itself) */
    /* the samples, in the coherent (tile, rank, pixel) order, at capacity n_rays * K */
    float *xyz_c, *dirs_c, *depth_c;
    int32_t *tri_c;                     /* or NULL */
    /* the field (NULL: stop after the samples) and its outputs at the same capacity */
    const qf_field_desc *field;         /* host */
    const float *table, *base_w, *head_ngp_w;
    const qf_sg_head *head_sg;          /* host, or NULL */
    float *rgb_c, *sigma_c;
    /* the image: three arrays or one packed [n_rays,5] (rgb | alpha | depth), as qf_composite_tiles */
    const float *bkgd;
    float *out_rgb, *out_alpha, *out_depth, *out_packed;
} qf_frame_job;
int qf_frame_render(qf_bvh *bvh, const qf_frame_job *job /* host */, void *stream);

/* The "before" evaluation of a frame in the coherent order (train_finetune.py:696; utils.py:555-572 + the re-sort of
 * mesh_utils.py:389-403): every sample is displaced along its ray by tanh(f) * scaling -- f_c [n] = the deformation
 * field's output at xyz_c (qf_deform_field_forward) -- and each ray's samples are put back in depth order (stable by
 * the new fp32 depth), all inside the tile layout: xyz_out / depth_out [n] take the place of xyz_c / depth_c for
 * qf_field_forward and qf_composite_tiles (directions do not change; out of place).  hit_count = the kept counts
 * (final_count of qf_pack_tiles), tile_base as there, total = the arrays' slot count (host value).  The displacement is qf_apply_deformation's, bit for bit. */
int qf_deform_resort_tiles(const float *f_c, float scaling, const float *xyz_c, const float *dirs_c,
                           const float *depth_c, const int32_t *hit_count, int32_t max_hits, const int64_t *tile_base,
                           int64_t total /* slots of the arrays */, const int64_t *total_device /* or NULL */,
                           int32_t width, int32_t height, float *xyz_out, float *depth_out, void *stream);

/* Stable per-ray re-sort by depth after deformation (sampling_indexing, mesh_utils.py:394-403):
 * perm[i] = source index of the sample that lands at i.  index_ray must be grouped by ray.   */
int qf_resort_by_depth(const int64_t *index_ray, const float *depth, int64_t n, int64_t *perm,
                       void *stream);

/* The whole of sampling_indexing (mesh_utils.py:389-412) in one launch: the re-sort above, the gathers of
 * points / depth / origins / vectors / index_tri through the permutation (index_ray is unchanged by a within-ray
 * sort) and kaolin's mark_pack_boundaries (:407).  perm and boundary may be NULL; so may out_origins / out_index_tri
 * (with their inputs): a caller that only renders reads neither (utils.py:574-577 discards both).
 * inverse (or NULL; from qf_split_layout / qf_coherent_layout on the same index_ray): the launch also writes the
 * re-sorted positions and directions a second time at out_points_c / out_vectors_c [inverse[i]] -- the copies
 * qf_field_forward streams (see qf_pack_samples).                                                 */
int qf_resort_samples(const int64_t *index_ray, const float *depth, int64_t n, const float *points,
                      const float *origins, const float *vectors, const int64_t *index_tri,
                      int64_t *perm, float *out_points, float *out_depth, float *out_origins,
                      float *out_vectors, int64_t *out_index_tri, uint8_t *boundary, const int32_t *inverse,
                      float *out_points_c, float *out_vectors_c, void *stream);

/* The coherent processing order of qf_coherent_layout for the samples of ONE SPLIT of a frame -- the windows of
 * generate_splits (train_finetune.py:419-439) that render_image_finetune_with_occgrid (utils.py:465-607) receives --
 * from nothing but the split's ray ids: index_ray [n] int64, ascending (grouped by ray, as the reference's loader
 * leaves them, mesh_utils.py:373-381), ids of a row-major width x height frame.  Four launches, no host round trip.
 * Scratch the caller provides: hit_count int32 [w*h], ray_offset int64 [w*h+1], tile_base int64
 * [ceil(w/8)*ceil(h/8)], invalid int32 [1].  Outputs: inverse int32 [n] (inverse[sample] = position), order int32 [n]
 * or NULL.  If the ids are NOT ascending or lie outside the frame, *invalid is set on the device and order / inverse
 * are the identity (the batch is then processed in its own order; results do not depend on the order).  */
int qf_split_layout(const int64_t *index_ray, int64_t n, int32_t width, int32_t height, int32_t *hit_count,
                    int64_t *ray_offset, int64_t *tile_base, int32_t *invalid, int32_t *order, int32_t *inverse,
                    void *stream);

/* MeshFinetune.update_d (mesh_utils.py:126-131; called per split by render_image_finetune_with_occgrid,
 * utils.py:596-600): cache_d[index_tri[i]] += d[i] * w[i], cache_w[index_tri[i]] += w[i] in one launch (fp32
 * atomics, as torch's index_add_).  cache [n_faces, 4]: cache_d in columns 0-2 and cache_w in column 3 of one
 * 16-byte row per triangle, so that a sample's four atomics are one memory request.  d == NULL: the displacement
 * is identically zero, only the weight column moves.
 * A triangle id outside [0, n_faces) is skipped (the reference's scatter_add raises on it); skipped (or NULL): a device
 * int32 that is INCREMENTED once per such sample -- the caller zeroes it and raises when it reads a non-zero count.
 * cache must be ordinary (coarse-grained) device memory: the kernel uses the hardware's fp32 add atomics
 * (unsafeAtomicAdd), which are not coherent on fine-grained / host-mapped allocations.                       */
int qf_mesh_update_d(const float *d /* [n,3] or NULL */, const float *w /* [n] */, const int64_t *index_tri, int64_t n,
                     int64_t n_faces, float *cache /* [n_faces,4] */, int32_t *skipped /* or NULL */, void *stream);

/* ------------------------------------------------------------------------------------------
 * Baked spherical-Gaussian textures.
 * Replaces trimesh.triangles.points_to_barycentric + the UV lookup (utils.py:1055-1063) and
 * FeatureCompression.get_features_from_texture_map (texture_utils.py:149-175) with the
 * dequantisers of ngp.py:245-281 and texture_utils.py:61-65.                                 */
typedef struct qf_texture_set {
    const uint8_t *alpha;                    /* [T,T] */
    const uint8_t *diffuse;                  /* [T,T,3] */
    const uint8_t *colors[QF_MAX_LOBES];     /* [T,T,3] each */
    const uint8_t *lambda_axis[QF_MAX_LOBES];/* [T,T,3] each: (lambda, azimuth, elevation) */
    int32_t texture_size;
    int32_t n_lobes;
    int32_t sigmoid_codec;                   /* 1 only when compression_type == "sigma" (B-7) */
    float lambda_thres;
} qf_texture_set;

/* vertices: float64 [V,3] (trimesh keeps float64); faces int64 [F,3]; uv fp32 [V,2] pre-scaled
 * by T as at test_baking_texture_images.py:325-328.  texel [n,2] int64 (row, col).           */
int qf_texel_indices(const double *vertices, const int64_t *faces, const float *uv,
                     const float *points, const int64_t *index_tri, int64_t n,
                     int32_t texture_size, int64_t *texel, void *stream);
/* The same lookup from a per-mesh table of 128-byte TRIANGLE RECORDS (corner, edges, their dot products and reciprocal
 * determinant in float64, the three corners' uv): qf_texel_records_pack builds records [n_faces * 128 bytes] once per
 * (mesh, uv); qf_texel_indices_packed then reads one line per sample instead of faces -> 3 vertices -> 3 uv.  Same
 * texels (the per-triangle values are the ones qf_texel_indices recomputes for every sample).             */
#define QF_TEXEL_TRIANGLE_RECORD_BYTES 128
int qf_texel_records_pack(const double *vertices, const int64_t *faces, const float *uv, int64_t n_faces,
                          void *records, void *stream);
int qf_texel_indices_packed(const void *records, const float *points, const int64_t *index_tri, int64_t n,
                            int32_t texture_size, int64_t *texel, void *stream);
/* texel [n,2] -> features [n, 3+7L+1] = [diffuse3 | (axis3, lambda, colour3)*L | sigma].      */
int qf_texture_fetch(const qf_texture_set *tex /* host */, const int64_t *texel, int64_t n,
                     float *features, void *stream);
/* Fused: texel fetch + dequantise + SG -> rgb [n,3], sigma [n] (no feature round trip).       */
int qf_texture_shade(const qf_texture_set *tex /* host */, const int64_t *texel,
                     const float *dirs, int64_t n, float *rgb, float *sigma, void *stream);

/* Device-resident form of the same texture set: one 64-byte record per texel,
 * [alpha | diffuse rgb | (lambda, azimuth, elevation, colour rgb) * n_lobes | zero pad], so a sample reads ONE
 * 64-byte sector instead of 2 + 2L scattered ones from the reference's planes (texture_utils.py:149-175 indexes
 * each plane separately).  qf_texture_pack builds records [T*T, QF_TEXEL_RECORD_BYTES] from the planes;
 * qf_texture_shade_packed == qf_texture_shade on them, bit for bit.                             */
#define QF_TEXEL_RECORD_BYTES 64
int qf_texture_pack(const qf_texture_set *tex /* host */, uint8_t *records, void *stream);
int qf_texture_shade_packed(const uint8_t *records, int32_t texture_size, int32_t n_lobes,
                            int32_t sigmoid_codec, float lambda_thres, const int64_t *texel,
                            const float *dirs, int64_t n, float *rgb, float *sigma, void *stream);
/* qf_texel_indices_packed + qf_texture_shade_packed in one launch: the texel of sample i is looked up from points[i] and
 * its triangle id through the triangle records inside the shading kernel -- no texel array is written and read back.
 * The baked-texture frame path's shading (utils.py:1055-1076).  Same values as the two calls.
 * The triangle ids come as EXACTLY ONE of index_tri (int64, the reference's dtype) or index_tri32 (int32, what
 * qf_pack_tiles writes: half the bytes per sample); the other is NULL.                                        */
int qf_texture_shade_points(const uint8_t *records, int32_t texture_size, int32_t n_lobes, int32_t sigmoid_codec,
                            float lambda_thres, const void *triangle_records, const float *points,
                            const int64_t *index_tri, const int32_t *index_tri32, const float *dirs, int64_t n,
                            const int64_t *n_device, float *rgb, float *sigma, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QF_HIP_H */
