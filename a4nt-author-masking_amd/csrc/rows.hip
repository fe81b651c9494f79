// Row kernels over the vocabulary axis (HBM/L2-bound; one 256-thread workgroup per row):
//   * gumbel_fwd : scale -> log_softmax -> + Gumbel noise -> /tau -> softmax stats -> argmax,
//                  straight-through one-hot value, done / length bookkeeping of the free-running
//                  generator (reference models/char_translator.py:10-28, 375-392), greedy argmax
//   * gumbel_bwd : dlogits = (scale/tau) * p * (ds - <p, ds>)   (softmax + log_softmax backward;
//                  the log_softmax term vanishes because the softmax-backward rows sum to zero)
//   * nll_rows   : log-softmax + NLL of the teacher-forced / reconstruction pass, leaves
//                  (softmax - onehot) * gscale in place of the logits (fused CE backward)
#include "common.h"

#define ROWS_MAXW 16
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float m = red[0];
    for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
    return m;
}

// Philox4x32-10 (Salmon et al.), counter = (elem/4, row, step, 0), key = seed
__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                        unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float uniform_at(unsigned long long seed, unsigned step, unsigned row, unsigned v) {
    unsigned o[4];
    philox4(v >> 2, row, step, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
    return (float)(o[v & 3] >> 8) * (1.0f / 16777216.0f);      // [0,1) with 24 bits
}

struct GumbelFwdArgs {
    const float* logits; int ldl; int V;
    const float* U; int ldu;           // injected uniforms or null (then Philox(seed, step, row, v))
    unsigned long long seed; unsigned step;
    float scale, tau;
    int greedy;                        // 1: argmax of the raw logits, nothing else
    float* y; int ldy;                 // tempered noisy log-probs (saved for backward) or null
    float* stats;                      // [B][2] = (max y, sum exp(y - max))
    long long* tok;                    // [B] argmax
    float* val_in;                     // [B] value fed to the next decoder step (un-masked)
    float* val_out;                    // [B] value handed to consumers (0 once the row was done)
    int* gen_lens; unsigned char* done; long long end_c;
    float* dense; int ldd; int hard;   // optional dense sample row (generic path)
};

__global__ __launch_bounds__(1024) void gumbel_fwd_kernel(GumbelFwdArgs a) {
    __shared__ float red[ROWS_MAXW];
    __shared__ int redi[ROWS_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const float* x = a.logits + (long long)b * a.ldl;
    const int V = a.V;
    float best = -INFINITY; int besti = 0x7fffffff;

    if (a.greedy) {
        for (int v = tid; v < V; v += NT) { const float f = x[v]; if (f > best) { best = f; besti = v; } }
    } else {
        float m1 = -INFINITY;
        for (int v = tid; v < V; v += NT) m1 = fmaxf(m1, x[v] * a.scale);
        m1 = block_max(m1, red);
        float s1 = 0.f;
        for (int v = tid; v < V; v += NT) s1 += expf(x[v] * a.scale - m1);
        s1 = block_sum(s1, red);
        const float lse = logf(s1);
        float* y = a.y + (long long)b * a.ldy;
        if (a.U) {
            for (int v = tid; v < V; v += NT) {
                const float u = a.U[(long long)b * a.ldu + v];
                const float gn = -logf(-logf(u + 1e-20f) + 1e-20f);
                const float yy = (((x[v] * a.scale - m1) - lse) + gn) / a.tau;
                y[v] = yy;
                if (yy > best) { best = yy; besti = v; }
            }
        } else {
            // one Philox4x32-10 call feeds 4 consecutive vocabulary entries
            for (int v4 = tid * 4; v4 < V; v4 += NT * 4) {
                unsigned o[4];
                philox4((unsigned)(v4 >> 2), (unsigned)b, a.step, 0u, (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = v4 + i;
                    if (v < V) {
                        const float u = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
                        const float gn = -logf(-logf(u + 1e-20f) + 1e-20f);
                        const float yy = (((x[v] * a.scale - m1) - lse) + gn) / a.tau;
                        y[v] = yy;
                        if (yy > best) { best = yy; besti = v; }
                    }
                }
            }
        }
    }
    // block argmax, lowest index on ties
    {
        const float bm = block_max(best, red);
        int cand = (best == bm) ? besti : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
        __syncthreads();
        if ((tid & 63) == 0) redi[tid >> 6] = cand;
        __syncthreads();
        besti = redi[0];
        for (int i = 1; i < (NT >> 6); ++i) besti = min(besti, redi[i]);
        best = bm;
    }
    float hot = 1.0f, s2 = 1.0f;
    if (!a.greedy) {
        const float* y = a.y + (long long)b * a.ldy;
        float s = 0.f;
        // y was written by this workgroup (possibly other threads): the block_max barriers above order it
        for (int v = tid; v < V; v += NT) s += expf(y[v] - best);
        s2 = block_sum(s, red);
        const float p_hot = 1.0f / s2;                 // exp(0) / sum
        hot = a.hard ? ((1.0f - p_hot) + p_hot) : p_hot;   // (one_hot - ysft.data) + ysft at the hot index
    }
    const bool prev_done = a.done ? (a.done[b] != 0) : false;
    if (a.dense) {
        float* d = a.dense + (long long)b * a.ldd;
        if (a.hard) {
            for (int v = tid; v < V; v += NT) d[v] = (v == besti) ? hot : 0.f;
        } else {
            const float* y = a.y + (long long)b * a.ldy;
            for (int v = tid; v < V; v += NT) d[v] = expf(y[v] - best) / s2;
        }
    }
    if (tid == 0) {
        if (a.stats) { a.stats[2 * b] = best; a.stats[2 * b + 1] = s2; }
        a.tok[b] = besti;
        if (a.val_in) a.val_in[b] = hot;
        if (a.val_out) a.val_out[b] = prev_done ? 0.f : hot;
        if (a.gen_lens) a.gen_lens[b] += prev_done ? 0 : 1;
        if (a.done) a.done[b] = (prev_done || (long long)besti == a.end_c) ? 1 : 0;
    }
}

extern "C" int a4nt_gumbel_fwd(int B, int V, const float* logits, int ldl, const float* U, int ldu,
                               unsigned long long seed, unsigned step, float scale, float tau, int hard, int greedy,
                               float* y, int ldy, float* stats, long long* tok, float* val_in, float* val_out,
                               int* gen_lens, unsigned char* done, long long end_c, float* dense, int ldd,
                               void* stream) {
    if (B <= 0) return A4NT_OK;
    if (V <= 0 || !logits || !tok || (!greedy && (!y || tau <= 0.f))) {
        a4nt_set_error("a4nt_gumbel_fwd: bad arguments (V=%d tau=%g)", V, (double)tau);
        return A4NT_ERR_ARG;
    }
    GumbelFwdArgs a;
    a.logits = logits; a.ldl = ldl; a.V = V; a.U = U; a.ldu = ldu; a.seed = seed; a.step = step; a.scale = scale;
    a.tau = tau; a.greedy = greedy; a.y = y; a.ldy = ldy; a.stats = stats; a.tok = tok; a.val_in = val_in;
    a.val_out = val_out; a.gen_lens = gen_lens; a.done = done; a.end_c = end_c; a.dense = dense; a.ldd = ldd;
    a.hard = hard;
    hipLaunchKernelGGL(gumbel_fwd_kernel, dim3(B), dim3(V >= 4096 ? 1024 : 256), 0, (hipStream_t)stream, a);
    return a4nt_check_launch("gumbel_fwd");
}

// ds (in) -> dlogits (out; may be ds itself).  p = exp((y - stats[0]) * inv_tau) / stats[1]
// (inv_tau = 1 when y already holds the tempered log-probs; 1/tau when it holds z = scale*x + g).
// One row is only 40 KB, and 128 rows on 128 workgroups leave half the chip idle with every workgroup
// latency-bound; when out is a separate buffer, gridDim.y workgroups share a row: each computes the whole
// row's <p, ds> (the re-reads hit L2) and writes its own slice of the output.
__global__ __launch_bounds__(1024) void gumbel_bwd_kernel(int V, const float* ds, int ldd, const float* y, int ldy,
                                                         const float* stats, float coef, float inv_tau, float* out, int ldo) {
    __shared__ float red[ROWS_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const float* d = ds + (long long)b * ldd;
    const float* yy = y + (long long)b * ldy;
    float* o = out + (long long)b * ldo;
    const float m2 = stats[2 * b], inv = 1.0f / stats[2 * b + 1];
    float dot = 0.f;
    for (int v = tid; v < V; v += NT) dot += (expf((yy[v] - m2) * inv_tau) * inv) * d[v];
    dot = block_sum(dot, red);
    const int per = (V + gridDim.y - 1) / gridDim.y;
    const int v0 = blockIdx.y * per, v1 = min(V, v0 + per);
    for (int v = v0 + tid; v < v1; v += NT) o[v] = coef * (expf((yy[v] - m2) * inv_tau) * inv) * (d[v] - dot);
}

// The same for rows that fit the registers of one 1024-thread workgroup (V <= 1024 * NV, NV up to 32): one read of ds and y, every
// load of the row in flight at once, the probabilities kept for the output pass (no second read, no second exp).
template <int NV>
__global__ __launch_bounds__(1024) void gumbel_bwd_reg_kernel(int V, const float* ds, int ldd, const float* y, int ldy,
                                                             const float* stats, float coef, float inv_tau, float* out,
                                                             int ldo) {
    __shared__ float red[ROWS_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* d = ds + (long long)b * ldd;
    const float* yy = y + (long long)b * ldy;
    float* o = out + (long long)b * ldo;
    float dv[NV], yv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 1024 * i;
        dv[i] = v < V ? d[v] : 0.f;
        yv[i] = v < V ? yy[v] : -INFINITY;
    }
    const float m2 = stats[2 * b], inv = 1.0f / stats[2 * b + 1];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        yv[i] = expf((yv[i] - m2) * inv_tau) * inv;          // exp(-inf) = 0 past the end of the row
        dot += yv[i] * dv[i];
    }
    dot = block_sum(dot, red);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 1024 * i;
        if (v < V) o[v] = coef * yv[i] * (dv[i] - dot);
    }
}

extern "C" int a4nt_gumbel_bwd(int B, int V, const float* ds, int ldd, const float* y, int ldy, const float* stats,
                               float coef, float inv_tau, float* out, int ldo, void* stream) {
    if (B <= 0) return A4NT_OK;
    if (V <= 0 || !ds || !y || !stats || !out) { a4nt_set_error("a4nt_gumbel_bwd: bad arguments"); return A4NT_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (V > 2048 && V <= 1024 * 32) {
        if (V <= 1024 * 10)
            hipLaunchKernelGGL((gumbel_bwd_reg_kernel<10>), dim3(B), dim3(1024), 0, s, V, ds, ldd, y, ldy, stats, coef, inv_tau, out, ldo);
        else if (V <= 1024 * 16)
            hipLaunchKernelGGL((gumbel_bwd_reg_kernel<16>), dim3(B), dim3(1024), 0, s, V, ds, ldd, y, ldy, stats, coef, inv_tau, out, ldo);
        else
            hipLaunchKernelGGL((gumbel_bwd_reg_kernel<32>), dim3(B), dim3(1024), 0, s, V, ds, ldd, y, ldy, stats, coef, inv_tau, out, ldo);
        return a4nt_check_launch("gumbel_bwd");
    }
    // in place: one workgroup per row (another workgroup of the row would still be reading what this one overwrites)
    const int q = (out != ds && V >= 4096 && B * 4 <= 2048) ? 4 : 1;
    hipLaunchKernelGGL(gumbel_bwd_kernel, dim3(B, q), dim3(V >= 4096 ? 1024 : 256), 0, s, V, ds, ldd, y, ldy,
                       stats, coef, inv_tau, out, ldo);
    return a4nt_check_launch("gumbel_bwd");
}

// Fused log-softmax + NLL + backward seed.  target < 0: row ignored (loss 0, gradient 0).
__global__ __launch_bounds__(256) void nll_rows_kernel(int V, float* logits, int ld, const long long* target,
                                                       float gscale, float* loss_rows, int write_grad) {
    __shared__ float red[ROWS_MAXW];
    const int r = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    float* x = logits + (long long)r * ld;
    const long long tg = target[r];
    if (tg < 0) {
        if (write_grad) for (int v = tid; v < V; v += NT) x[v] = 0.f;
        if (tid == 0) loss_rows[r] = 0.f;
        return;
    }
    // one pass for max and sum: per-thread running (max, sum of exp relative to it), rescaled when the max moves
    float ml = -INFINITY, sl = 0.f;
    for (int v = tid; v < V; v += NT) {
        const float xv = x[v];
        const float mn = fmaxf(ml, xv);
        sl = sl * expf(ml - mn) + expf(xv - mn);         // expf(-inf) = 0 on the first element
        ml = mn;
    }
    const float m = block_max(ml, red);
    float s = (ml == -INFINITY) ? 0.f : sl * expf(ml - m);
    s = block_sum(s, red);
    const float lse = m + logf(s);
    if (tid == 0) loss_rows[r] = lse - x[tg];
    if (write_grad) {
        __syncthreads();
        const float inv = 1.0f / s;
        for (int v = tid; v < V; v += NT) {
            const float p = expf(x[v] - m) * inv;
            x[v] = (p - ((long long)v == tg ? 1.f : 0.f)) * gscale;
        }
    }
}

extern "C" int a4nt_nll_rows(int N, int V, float* logits, int ld, const long long* target, float gscale,
                             float* loss_rows, int write_grad, void* stream) {
    if (N <= 0) return A4NT_OK;
    if (V <= 0 || !logits || !target || !loss_rows) { a4nt_set_error("a4nt_nll_rows: bad arguments"); return A4NT_ERR_ARG; }
    hipLaunchKernelGGL(nll_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, V, logits, ld, target, gscale,
                       loss_rows, write_grad);
    return a4nt_check_launch("nll_rows");
}


// Row kernel behind a4nt_sgemm_nll: merges the per-wave (max, sum exp) partials the projection's epilogue left for
// the row, gives the row's loss and turns the logits into (softmax - onehot) * gscale in place: ONE read and one
// write of the row; the loss itself costs no pass over the logits.  target < 0: row ignored (loss 0, gradient 0).
__global__ __launch_bounds__(256) void nll_grad_rows_kernel(int V, float* logits, int ld, const long long* target,
                                                            const float* part, int nparts, float gscale,
                                                            float* loss_rows, int write_grad) {
    __shared__ float red[ROWS_MAXW];
    const int r = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    float* x = logits + (long long)r * ld;
    const long long tg = target[r];
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    const int V4 = vec ? (V & ~3) : 0;
    if (tg < 0) {
        if (write_grad) {
            for (int v = 4 * tid; v < V4; v += 4 * NT) *reinterpret_cast<f32x4*>(x + v) = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int v = V4 + tid; v < V; v += NT) x[v] = 0.f;
        }
        if (tid == 0) loss_rows[r] = 0.f;
        return;
    }
    const float* pr = part + (long long)r * nparts * 2;
    float ml = -INFINITY;
    for (int p = tid; p < nparts; p += NT) ml = fmaxf(ml, pr[2 * p]);
    const float m = block_max(ml, red);
    float sl = 0.f;
    for (int p = tid; p < nparts; p += NT) {
        const float mp = pr[2 * p];
        if (mp != -INFINITY) sl += pr[2 * p + 1] * expf(mp - m);
    }
    const float s = block_sum(sl, red);
    const float lse = m + logf(s);
    if (tid == 0) loss_rows[r] = lse - x[tg];
    if (write_grad) {
        __syncthreads();                       // x[tg] is read above, overwritten below
        const float inv = 1.0f / s;
        for (int v = 4 * tid; v < V4; v += 4 * NT) {
            f32x4 q = *reinterpret_cast<const f32x4*>(x + v);
            q.x = (expf(q.x - m) * inv - ((long long)v == tg ? 1.f : 0.f)) * gscale;
            q.y = (expf(q.y - m) * inv - ((long long)(v + 1) == tg ? 1.f : 0.f)) * gscale;
            q.z = (expf(q.z - m) * inv - ((long long)(v + 2) == tg ? 1.f : 0.f)) * gscale;
            q.w = (expf(q.w - m) * inv - ((long long)(v + 3) == tg ? 1.f : 0.f)) * gscale;
            *reinterpret_cast<f32x4*>(x + v) = q;
        }
        for (int v = V4 + tid; v < V; v += NT) x[v] = (expf(x[v] - m) * inv - ((long long)v == tg ? 1.f : 0.f)) * gscale;
    }
}

extern "C" int a4nt_nll_grad_rows(int N, int V, float* logits, int ld, const long long* target, const float* part,
                                  int nparts, float gscale, float* loss_rows, int write_grad, void* stream) {
    if (N <= 0) return A4NT_OK;
    if (V <= 0 || !logits || !target || !part || !loss_rows || nparts != 2 * ((V + 127) / 128)) {
        a4nt_set_error("a4nt_nll_grad_rows: bad arguments (V=%d, nparts=%d)", V, nparts);
        return A4NT_ERR_ARG;
    }
    hipLaunchKernelGGL(nll_grad_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, V, logits, ld, target, part, nparts,
                       gscale, loss_rows, write_grad);
    return a4nt_check_launch("nll_grad_rows");
}
