// Dense Adam in ONE pass over the parameters (training side).
//
// The reference optimises the radiance field and the deformation field with torch.optim.Adam (train_finetune.py:402-417);
// torch's foreach implementation walks every tensor seven times (lerp, mul, addcmul, sqrt, div, add, addcdiv: 11 launches
// per step).  On the deformation field's T = 2^24 table (203 M parameters, 0.81 GB) that is 3.8 ms per step.  The update
// rule below is the same one, element for element and in the same order of operations; one launch reads p, g, m, v and
// writes p, m, v -- 5.7 GB, ~1.6 ms at the HBM rate, which is what "dense Adam" costs at that table size.  (A sparse /
// lazy variant would NOT be the reference's optimiser: with a zero gradient dense Adam still decays m and v and moves p
// along the remaining momentum, and after a few hundred steps nearly every row of the hashed levels has been touched.)
#include "qf_common.h"

namespace {

// Every scalar is evaluated on the host in DOUBLE, as torch's Python does, and rounded to fp32 once: 1 - beta2 computed in
// fp32 from the fp32 beta2 (0.999) is off by 1.3e-5 relative, which shows in exp_avg_sq after a few steps.
struct AdamArgs {
    float beta2, eps, weight_decay;
    float one_minus_beta1, one_minus_beta2;          // lerp weight, addcmul value
    float neg_step_size, bias_correction2_sqrt;      // -lr / (1 - beta1^t), sqrt(1 - beta2^t)
    int maximize;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a)
{
#pragma clang fp contract(off)
    if (a.maximize) g = -g;
    if (a.weight_decay != 0.0f) g = g + a.weight_decay * p;
    m = m + a.one_minus_beta1 * (g - m);                      // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (a.one_minus_beta2 * g) * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / a.bias_correction2_sqrt + a.eps;
    p = p + a.neg_step_size * (m / denom);                    // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, AdamArgs a)
{
    const int64_t n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride) {    // tail (n not a multiple of 4)
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, a);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

}  // namespace

extern "C" int qf_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                            double beta1, double beta2, double eps, double weight_decay, int32_t maximize, int64_t step,
                            void *stream)
{
    if (n < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return QF_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return QF_ERR_INVALID_ARGUMENT;                   // float4 accesses
    AdamArgs a;
    a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay; a.maximize = maximize ? 1 : 0;
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.neg_step_size = (float)(-(lr / (1.0 - pow(beta1, (double)step))));
    a.bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(qf_grid_1d((n + 3) / 4, 256, 16)), dim3(256), 0, qf_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}
