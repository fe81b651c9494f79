// Optimizer step of the adversarial update as three multi-tensor launches (include/a4nt_hip.h: a4nt_mt_*).
//
// Reference: train_adversarial_translator.py:306-312 (two torch.optim.RMSprop), :730 (clip_grad_norm at 10), :731 / :542
// (optimizer steps).  torch runs that as one launch per elementary operation and tensor-list chunk (mul, addcmul, sqrt,
// add, addcdiv; per-tensor norms, their norm, the scaling): 25 launches and 12 passes over the 99 MB of parameter state
// per training step.  Here: one pass that sums the squares of every gradient (per 16K-element chunk, fixed order), one
// workgroup that turns the partial sums into the total norm and the clipping factor (on the device: the host never
// reads it inside the step), one pass that scales the gradients in place when the factor is below one (what
// clip_grad_norm_ leaves in .grad), and one pass with the RMSprop update
//     sq = alpha * sq + (1 - alpha) * g * g;    p -= lr * g / (sqrt(sq) + eps)        (g += wd * p first when wd != 0)
// over all tensors of a parameter group.  HBM-bound: 4 bytes read per element for the norm, 12 read + 8 written for the
// update.
#include "common.h"

#define MT_CHUNK 16384            // elements per workgroup

struct MtTensor { float* p; float* g; float* sq; long long n; };      // one row of the caller's table (32 bytes)

// chunk c of the launch = elements [off, off + MT_CHUNK) of tensor ti: map[2c] = ti, map[2c+1] = off / MT_CHUNK
__device__ __forceinline__ bool mt_chunk(const MtTensor* tab, const int* map, MtTensor& t, long long& off, int& len) {
    const int ti = map[2 * blockIdx.x];
    off = (long long)map[2 * blockIdx.x + 1] * MT_CHUNK;
    t = tab[ti];
    const long long left = t.n - off;
    len = (int)(left < MT_CHUNK ? left : MT_CHUNK);
    return len > 0;
}

__global__ __launch_bounds__(256) void mt_sumsq_kernel(const MtTensor* tab, const int* map, float* partial) {
    __shared__ float red[4];
    MtTensor t; long long off; int len;
    float s = 0.f;
    if (mt_chunk(tab, map, t, off, len)) {
        const float* g = t.g + off;
        const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
        if (vec) {
            const int n4 = len >> 2;
            for (int i = threadIdx.x; i < n4; i += 256) {
                const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
                s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
            for (int i = 4 * n4 + threadIdx.x; i < len; i += 256) s += g[i] * g[i];
        } else {
            for (int i = threadIdx.x; i < len; i += 256) s += g[i] * g[i];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total 2-norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void mt_clip_coef_kernel(int n, const float* partial, float max_norm, float* out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = norm;
        out[1] = coef < 1.0f ? coef : 1.0f;
    }
}

__global__ __launch_bounds__(256) void mt_scale_kernel(const MtTensor* tab, const int* map, const float* coef_ptr) {
    const float coef = coef_ptr[1];
    if (!(coef < 1.0f)) return;                         // nothing to clip: no traffic at all
    MtTensor t; long long off; int len;
    if (!mt_chunk(tab, map, t, off, len)) return;
    float* g = t.g + off;
    for (int i = threadIdx.x; i < len; i += 256) g[i] *= coef;
}

__global__ __launch_bounds__(256) void mt_rmsprop_kernel(const MtTensor* tab, const int* map, float lr, float alpha, float eps,
                                                         float wd) {
    MtTensor t; long long off; int len;
    if (!mt_chunk(tab, map, t, off, len)) return;
    float* p = t.p + off; const float* g = t.g + off; float* sq = t.sq + off;
    const float oma = 1.0f - alpha;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(sq)) & 15) == 0;
    auto upd = [&](float& pv, float gv, float& sv) {
        if (wd != 0.f) gv = gv + wd * pv;
        sv = sv * alpha + oma * gv * gv;
        pv = pv - lr * (gv / (sqrtf(sv) + eps));
    };
    int done = 0;
    if (vec) {
        const int n4 = len >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
            const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
            f32x4 sv = reinterpret_cast<f32x4*>(sq)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], se = sv[e];
                upd(pe, gv[e], se);
                pv[e] = pe; sv[e] = se;
            }
            reinterpret_cast<f32x4*>(p)[i] = pv;
            reinterpret_cast<f32x4*>(sq)[i] = sv;
        }
        done = 4 * n4;
    }
    for (int i = done + threadIdx.x; i < len; i += 256) {
        float pv = p[i], sv = sq[i];
        upd(pv, g[i], sv);
        p[i] = pv; sq[i] = sv;
    }
}

// ---- C-ABI (include/a4nt_hip.h) ---------------------------------------------------------------------------------------
extern "C" int a4nt_mt_chunk_elems(void) { return MT_CHUNK; }

static bool mt_bad(int nchunks, const void* table, const int* map) { return nchunks < 0 || (nchunks > 0 && (!table || !map)); }

extern "C" int a4nt_mt_sumsq(int nchunks, const void* table, const int* map, float* partial, void* stream) {
    if (mt_bad(nchunks, table, map) || (nchunks > 0 && !partial)) { a4nt_set_error("a4nt_mt_sumsq: bad arguments"); return A4NT_ERR_ARG; }
    if (nchunks == 0) return A4NT_OK;
    hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const MtTensor*)table, map, partial);
    return a4nt_check_launch("mt_sumsq");
}
extern "C" int a4nt_mt_clip_coef(int nchunks, const float* partial, float max_norm, float* out2, void* stream) {
    if (nchunks < 0 || !out2 || (nchunks > 0 && !partial)) { a4nt_set_error("a4nt_mt_clip_coef: bad arguments"); return A4NT_ERR_ARG; }
    hipLaunchKernelGGL(mt_clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nchunks, partial, max_norm, out2);
    return a4nt_check_launch("mt_clip_coef");
}
extern "C" int a4nt_mt_scale(int nchunks, const void* table, const int* map, const float* coef2, void* stream) {
    if (mt_bad(nchunks, table, map) || !coef2) { a4nt_set_error("a4nt_mt_scale: bad arguments"); return A4NT_ERR_ARG; }
    if (nchunks == 0) return A4NT_OK;
    hipLaunchKernelGGL(mt_scale_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const MtTensor*)table, map, coef2);
    return a4nt_check_launch("mt_scale");
}
extern "C" int a4nt_mt_rmsprop(int nchunks, const void* table, const int* map, float lr, float alpha, float eps,
                               float weight_decay, void* stream) {
    if (mt_bad(nchunks, table, map)) { a4nt_set_error("a4nt_mt_rmsprop: bad arguments"); return A4NT_ERR_ARG; }
    if (nchunks == 0) return A4NT_OK;
    hipLaunchKernelGGL(mt_rmsprop_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const MtTensor*)table, map, lr,
                       alpha, eps, weight_decay);
    return a4nt_check_launch("mt_rmsprop");
}
