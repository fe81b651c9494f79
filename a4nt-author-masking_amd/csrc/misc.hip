// Small HBM-bound helpers of the hot path: embedding gather / scatter-add (token ids and
// straight-through sample values), column sums for bias gradients, split-K slab reduction, the
// element-wise LSTM cell backward used by the free-running decoder's reverse loop, and the
// leaky-relu + max-over-time epilogue of the CharCNN classifier.
#include "common.h"

// out[n][:] = (scale ? scale[n] : 1) * W[ids[n]][:]
__global__ void embed_fwd_kernel(long long N, int E, const long long* ids, const float* scale, const float* W,
                                 float* out, int ldo) {
    const long long total = N * E;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / E; const int e = (int)(i - n * E);
        const float s = scale ? scale[n] : 1.f;
        out[n * ldo + e] = s * W[ids[n] * (long long)E + e];
    }
}
extern "C" int a4nt_embed_fwd(long long N, int E, int V, const long long* ids, const float* scale, const float* W,
                              float* out, int ldo, void* stream) {
    if (N <= 0) return A4NT_OK;
    if (E <= 0 || V <= 0 || !ids || !W || !out) { a4nt_set_error("a4nt_embed_fwd: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = N * E;
    const int grid = (int)min((long long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, E, ids, scale, W, out, ldo);
    return a4nt_check_launch("embed_fwd");
}

// dW[ids[n]][:] += (scale ? scale[n] : 1) * dout[n][:]   (rows with ids[n] == padding_idx skipped)
__global__ void embed_bwd_kernel(long long N, int E, const long long* ids, const float* scale, const float* dout,
                                 int ldd, float* dW, long long padding_idx) {
    const long long total = N * E;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / E; const int e = (int)(i - n * E);
        const long long id = ids[n];
        if (id == padding_idx) continue;
        const float s = scale ? scale[n] : 1.f;
        if (s == 0.f) continue;
        atomicAdd(&dW[id * (long long)E + e], s * dout[n * ldd + e]);
    }
}
extern "C" int a4nt_embed_bwd(long long N, int E, int V, const long long* ids, const float* scale, const float* dout,
                              int ldd, float* dW, long long padding_idx, void* stream) {
    if (N <= 0) return A4NT_OK;
    if (E <= 0 || V <= 0 || !ids || !dout || !dW) { a4nt_set_error("a4nt_embed_bwd: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = N * E;
    const int grid = (int)min((long long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, E, ids, scale, dout, ldd, dW,
                       padding_idx);
    return a4nt_check_launch("embed_bwd");
}

// out[n] (+)= sum_m A[m][n].  Deterministic: the row chunks of a column block write partial sums into a workspace
// ([chunks][N]) and a second launch adds them in chunk order (no atomics: the same bits on every run).
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* A, int lda, float* part, int rows_per,
                                                     int direct, int accumulate) {
    // one workgroup = 256 columns x rows_per rows: lane l of every wave owns columns 4l..4l+3 of the block (one 16-byte
    // load per row when the row is 16-byte aligned, 1 KB per wave instruction), the 4 waves take every 4th row
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + 4 * lane;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const bool vec = (c + 3 < N) && ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    if (vec) {
        for (int m = m0 + w; m < m1; m += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(A + (long long)m * lda + c);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    } else if (c < N) {
        for (int m = m0 + w; m < m1; m += 4) {
            const float* p = A + (long long)m * lda + c;
            s0 += p[0];
            if (c + 1 < N) s1 += p[1];
            if (c + 2 < N) s2 += p[2];
            if (c + 3 < N) s3 += p[3];
        }
    }
    red[w][4 * lane] = s0; red[w][4 * lane + 1] = s1; red[w][4 * lane + 2] = s2; red[w][4 * lane + 3] = s3;
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc >= N) return;
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (direct) part[cc] = (accumulate ? part[cc] : 0.f) + v;          // a single chunk: `part` is the output itself
    else part[(long long)blockIdx.y * N + cc] = v;
}
// 64 columns x 4 chunk groups per workgroup: group q adds chunks q, q+4, ... (independent loads in flight), the four
// group sums are added in a fixed order
__global__ __launch_bounds__(256) void colsum_finish_kernel(int N, int chunks, const float* part, float* out, int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        int k = q;
        for (; k + 4 < chunks; k += 8) {
            s0 += part[(long long)k * N + c];
            s1 += part[(long long)(k + 4) * N + c];
        }
        if (k < chunks) s0 += part[(long long)k * N + c];
    }
    red[q][cl] = s0 + s1;
    __syncthreads();
    if (q == 0 && c < N) out[c] = (accumulate ? out[c] : 0.f) + ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}
static int colsum_chunks(int M, int N) {
    const int cb = (N + 255) / 256;
    return max(1, min((M + 31) / 32, 2048 / cb));
}
extern "C" long long a4nt_colsum_ws_floats(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int chunks = colsum_chunks(M, N);
    return chunks > 1 ? (long long)chunks * N : 0;
}
extern "C" int a4nt_colsum(int M, int N, const float* A, int lda, float* out, int accumulate, float* ws,
                           long long ws_floats, void* stream) {
    if (N <= 0) return A4NT_OK;
    if (!A || !out || M < 0) { a4nt_set_error("a4nt_colsum: bad arguments"); return A4NT_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        if (!accumulate && hipMemsetAsync(out, 0, (size_t)N * 4, s) != hipSuccess) return a4nt_check_launch("colsum memset");
        return A4NT_OK;
    }
    const int cb = (N + 255) / 256;
    int chunks = colsum_chunks(M, N);
    if (!ws) chunks = 1;                                   // no workspace: one workgroup per column block (slow, still exact)
    else if (ws_floats < (long long)chunks * N) chunks = (int)max(1ll, ws_floats / N);
    const int rows_per = (M + chunks - 1) / chunks;
    chunks = (M + rows_per - 1) / rows_per;
    if (chunks == 1) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1), dim3(256), 0, s, M, N, A, lda, out, rows_per, 1, accumulate ? 1 : 0);
        return a4nt_check_launch("colsum");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, chunks), dim3(256), 0, s, M, N, A, lda, ws, rows_per, 0, 0);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(256), 0, s, N, chunks, ws, out, accumulate ? 1 : 0);
    return a4nt_check_launch("colsum");
}

// out[r][c] = sum_s slabs[s*stride + r*lds + c0 + c] (+ add[r][c]),  r < R, c < C
// optional second output (a4nt_slab_reduce2): out2[r][c - c2] = out[r][c] + add2[r][c - c2] for the columns c >= c2
struct Slab2 { const float* add2; int lda2; float* out2; int ldo2; int c2; };
__global__ void slab_reduce_kernel(int S, long long stride, int R, int C, const float* slabs, int lds, int c0,
                                   const float* add, int lda, float* out, int ldo, Slab2 x) {
    const long long total = (long long)R * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        float v = add ? add[(long long)r * lda + c] : 0.f;
        for (int s = 0; s < S; ++s) v += slabs[s * stride + (long long)r * lds + c0 + c];
        out[(long long)r * ldo + c] = v;
        if (x.out2 && c >= x.c2) x.out2[(long long)r * x.ldo2 + c - x.c2] = v + x.add2[(long long)r * x.lda2 + c - x.c2];
    }
}
// the same with 16-byte accesses (every row start and leading dimension 16-byte aligned, C a multiple of 4): the
// per-step reductions of the decoder's reverse loop are latency bound -- S independent 16-byte loads per thread
// in flight instead of S dependent-looking 4-byte ones
template <int SMAX>
__global__ __launch_bounds__(256) void slab_reduce4_kernel(int S, long long stride, int R, int C4, const float* slabs,
                                                           int lds, int c0, const float* add, int lda, float* out, int ldo,
                                                           Slab2 x) {
    const int total = R * C4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / C4, c = 4 * (i - r * C4);
        const float* p = slabs + (long long)r * lds + c0 + c;
        f32x4 v[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < S) v[s] = *reinterpret_cast<const f32x4*>(p + s * stride);
        f32x4 acc = add ? *reinterpret_cast<const f32x4*>(add + (long long)r * lda + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < S) { acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w; }
        *reinterpret_cast<f32x4*>(out + (long long)r * ldo + c) = acc;
        if (x.out2 && c >= x.c2) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(x.add2 + (long long)r * x.lda2 + c - x.c2);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            *reinterpret_cast<f32x4*>(x.out2 + (long long)r * x.ldo2 + c - x.c2) = acc;
        }
    }
}
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static int slab_reduce_launch(int S, long long stride, int R, int C, const float* slabs, int lds, int c0,
                              const float* add, int lda, float* out, int ldo, Slab2 x, void* stream) {
    if (R <= 0 || C <= 0) return A4NT_OK;
    if (S < 1 || !slabs || !out) { a4nt_set_error("a4nt_slab_reduce: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = (long long)R * C;
    const bool x_ok = !x.out2 || ((x.lda2 & 3) == 0 && (x.ldo2 & 3) == 0 && (x.c2 & 3) == 0 && al16(x.add2) && al16(x.out2));
    if (S <= 16 && (C & 3) == 0 && (lds & 3) == 0 && (c0 & 3) == 0 && (ldo & 3) == 0 && (stride & 3) == 0 && al16(slabs) &&
        al16(out) && (!add || ((lda & 3) == 0 && al16(add))) && x_ok && total / 4 < 0x7fffffffll) {
        const int n4 = (int)(total / 4);
        const int bs = n4 <= 65536 ? 64 : 256;          // small (per-step) reductions: one wave per workgroup, on every CU
        const int grid = min(2048, (n4 + bs - 1) / bs);
        if (S <= 8) hipLaunchKernelGGL((slab_reduce4_kernel<8>), dim3(grid), dim3(bs), 0, (hipStream_t)stream, S, stride, R, C / 4, slabs, lds, c0, add, lda, out, ldo, x);
        else hipLaunchKernelGGL((slab_reduce4_kernel<16>), dim3(grid), dim3(bs), 0, (hipStream_t)stream, S, stride, R, C / 4, slabs, lds, c0, add, lda, out, ldo, x);
        return a4nt_check_launch("slab_reduce");
    }
    const int grid = (int)min((long long)1024, (total + 255) / 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, S, stride, R, C, slabs, lds, c0,
                       add, lda, out, ldo, x);
    return a4nt_check_launch("slab_reduce");
}
extern "C" int a4nt_slab_reduce(int S, long long stride, int R, int C, const float* slabs, int lds, int c0,
                                const float* add, int lda, float* out, int ldo, void* stream) {
    Slab2 x = {nullptr, 0, nullptr, 0, 0};
    return slab_reduce_launch(S, stride, R, C, slabs, lds, c0, add, lda, out, ldo, x, stream);
}
extern "C" int a4nt_slab_reduce2(int S, long long stride, int R, int C, const float* slabs, int lds, float* out, int ldo,
                                 int c2, const float* add2, int lda2, float* out2, int ldo2, void* stream) {
    if (!add2 || !out2 || c2 < 0 || c2 > C) { a4nt_set_error("a4nt_slab_reduce2: bad arguments"); return A4NT_ERR_ARG; }
    Slab2 x = {add2, lda2, out2, ldo2, c2};
    return slab_reduce_launch(S, stride, R, C, slabs, lds, 0, nullptr, 0, out, ldo, x, stream);
}

// Element-wise LSTM cell backward of one decoder step:
//   dh = sum_s dh_slabs[s] (split-K partials of dlogits*W^T) + dh_b ; dc = dc_io + dh*o*(1-tanh(c)^2) ;
//   dgates (pre-activation) ; dc_io = dc*f
__global__ void lstm_cell_bwd_kernel(int B, int H, const float* gates, const float* c_t, const float* c_prev,
                                     const float* dh_slabs, int n_slabs, long long slab_stride, const float* dh_b,
                                     int ldb, float* dc_io, float* dgates) {
    const long long total = (long long)B * H;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), u = (int)(i - (long long)b * H);
        const float* pg = gates + (long long)b * 4 * H + u;
        const float ig = pg[0], fg = pg[H], gg = pg[2 * H], og = pg[3 * H];
        float dh = dh_b ? dh_b[(long long)b * ldb + u] : 0.f;
        for (int s = 0; s < n_slabs; ++s) dh += dh_slabs[s * slab_stride + i];
        const float tc = tanhf_(c_t[i]);
        const float cp = c_prev ? c_prev[i] : 0.f;
        const float dc = dc_io[i] + dh * og * (1.f - tc * tc);
        float* pd = dgates + (long long)b * 4 * H + u;
        pd[0] = dc * gg * ig * (1.f - ig);
        pd[H] = dc * cp * fg * (1.f - fg);
        pd[2 * H] = dc * ig * (1.f - gg * gg);
        pd[3 * H] = dh * tc * og * (1.f - og);
        dc_io[i] = dc * fg;
    }
}
// four units per thread through 16-byte accesses, all loads of a thread issued before the first use (same
// arithmetic per element, same order of the slab sum)
template <int SMAX>
__global__ __launch_bounds__(256) void lstm_cell_bwd4_kernel(int B, int H, const float* gates, const float* c_t,
                                                             const float* c_prev, const float* dh_slabs, int n_slabs,
                                                             long long slab_stride, const float* dh_b, int ldb,
                                                             float* dc_io, float* dgates) {
    const int H4q = H >> 2, total = B * H4q;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H4q, u = 4 * (i - b * H4q);
        const long long e = (long long)b * H + u;
        const float* pg = gates + (long long)b * 4 * H + u;
        const f32x4 ig = *reinterpret_cast<const f32x4*>(pg), fg = *reinterpret_cast<const f32x4*>(pg + H);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(pg + 2 * H), og = *reinterpret_cast<const f32x4*>(pg + 3 * H);
        const f32x4 ct = *reinterpret_cast<const f32x4*>(c_t + e);
        const f32x4 cp = c_prev ? *reinterpret_cast<const f32x4*>(c_prev + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 dci = *reinterpret_cast<const f32x4*>(dc_io + e);
        f32x4 sl[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < n_slabs) sl[s] = *reinterpret_cast<const f32x4*>(dh_slabs + s * slab_stride + e);
        f32x4 dh = dh_b ? *reinterpret_cast<const f32x4*>(dh_b + (long long)b * ldb + u) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < n_slabs) { dh.x += sl[s].x; dh.y += sl[s].y; dh.z += sl[s].z; dh.w += sl[s].w; }
        f32x4 di, df, dg, dO, dcn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tc = tanhf_(ct[k]);
            const float dc = dci[k] + dh[k] * og[k] * (1.f - tc * tc);
            di[k] = dc * gg[k] * ig[k] * (1.f - ig[k]);
            df[k] = dc * cp[k] * fg[k] * (1.f - fg[k]);
            dg[k] = dc * ig[k] * (1.f - gg[k] * gg[k]);
            dO[k] = dh[k] * tc * og[k] * (1.f - og[k]);
            dcn[k] = dc * fg[k];
        }
        float* pd = dgates + (long long)b * 4 * H + u;
        *reinterpret_cast<f32x4*>(pd) = di; *reinterpret_cast<f32x4*>(pd + H) = df;
        *reinterpret_cast<f32x4*>(pd + 2 * H) = dg; *reinterpret_cast<f32x4*>(pd + 3 * H) = dO;
        *reinterpret_cast<f32x4*>(dc_io + e) = dcn;
    }
}
extern "C" int a4nt_lstm_cell_bwd(int B, int H, const float* gates, const float* c_t, const float* c_prev,
                                  const float* dh_slabs, int n_slabs, long long slab_stride, const float* dh_b, int ldb,
                                  float* dc_io, float* dgates, void* stream) {
    if (B <= 0 || H <= 0) return A4NT_OK;
    if (!gates || !c_t || !dc_io || !dgates || (n_slabs > 0 && !dh_slabs)) { a4nt_set_error("a4nt_lstm_cell_bwd: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = (long long)B * H;
    if ((H & 3) == 0 && n_slabs <= 16 && (slab_stride & 3) == 0 && al16(gates) && al16(c_t) && (!c_prev || al16(c_prev)) &&
        (n_slabs == 0 || al16(dh_slabs)) && (!dh_b || ((ldb & 3) == 0 && al16(dh_b))) && al16(dc_io) && al16(dgates) &&
        total / 4 < 0x7fffffffll) {
        const int n4 = (int)(total / 4);
        const int bs = n4 <= 65536 ? 64 : 256;
        const int grid = min(2048, (n4 + bs - 1) / bs);
        if (n_slabs <= 4) hipLaunchKernelGGL((lstm_cell_bwd4_kernel<4>), dim3(grid), dim3(bs), 0, (hipStream_t)stream, B, H, gates, c_t, c_prev, dh_slabs, n_slabs, slab_stride, dh_b, ldb, dc_io, dgates);
        else hipLaunchKernelGGL((lstm_cell_bwd4_kernel<16>), dim3(grid), dim3(bs), 0, (hipStream_t)stream, B, H, gates, c_t, c_prev, dh_slabs, n_slabs, slab_stride, dh_b, ldb, dc_io, dgates);
        return a4nt_check_launch("lstm_cell_bwd");
    }
    const int grid = (int)min((long long)1024, (total + 255) / 256);
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, H, gates, c_t, c_prev,
                       dh_slabs, n_slabs, slab_stride, dh_b, ldb, dc_io, dgates);
    return a4nt_check_launch("lstm_cell_bwd");
}

// CharCNN epilogue.  pre: [Tout][B][F] conv pre-activations (time major, bias included).
// feat[b][f] = max over the padded time axis of leaky_relu(pre), where `pad` zero positions take
// part in the max iff pad > 0 (reference char_cnn.py:84-92 pads branches to a common length with
// zeros before the global max-pool).  argmax[b][f] = winning time step, or -1 when a zero pad won.
__global__ void lrelu_maxtime_kernel(int Tout, int B, int F, const float* pre, int ldp, float slope, int pad,
                                     float* feat, int ldf, int* argmax) {
    const long long total = (long long)B * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i - (long long)b * F);
        float best = pad > 0 ? 0.f : -INFINITY; int bi = -1;
        for (int t = 0; t < Tout; ++t) {
            float v = pre[((long long)t * B + b) * ldp + f];
            v = v > 0.f ? v : v * slope;
            if (v > best) { best = v; bi = t; }
        }
        feat[(long long)b * ldf + f] = best;
        argmax[i] = bi;
    }
}
extern "C" int a4nt_lrelu_maxtime(int Tout, int B, int F, const float* pre, int ldp, float slope, int pad, float* feat,
                                  int ldf, int* argmax, void* stream) {
    if (B <= 0 || F <= 0) return A4NT_OK;
    if (Tout <= 0 || !pre || !feat || !argmax) { a4nt_set_error("a4nt_lrelu_maxtime: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = (long long)B * F;
    const int grid = (int)min((long long)1024, (total + 255) / 256);
    hipLaunchKernelGGL(lrelu_maxtime_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Tout, B, F, pre, ldp, slope,
                       pad, feat, ldf, argmax);
    return a4nt_check_launch("lrelu_maxtime");
}
// dpre[t][b][f] = (t == argmax[b][f]) ? dfeat[b][f] * lrelu'(pre) : 0     (in place over pre)
__global__ void lrelu_maxtime_bwd_kernel(int Tout, int B, int F, float* pre, int ldp, float slope, const float* dfeat,
                                         int ldf, const int* argmax) {
    const long long total = (long long)Tout * B * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F); const long long tb = i / F; const int b = (int)(tb % B); const int t = (int)(tb / B);
        float* p = pre + tb * ldp + f;
        float g = 0.f;
        if (argmax[(long long)b * F + f] == t) g = dfeat[(long long)b * ldf + f] * (*p > 0.f ? 1.f : slope);
        *p = g;
    }
}
extern "C" int a4nt_lrelu_maxtime_bwd(int Tout, int B, int F, float* pre, int ldp, float slope, const float* dfeat,
                                      int ldf, const int* argmax, void* stream) {
    if (B <= 0 || F <= 0 || Tout <= 0) return A4NT_OK;
    if (!pre || !dfeat || !argmax) { a4nt_set_error("a4nt_lrelu_maxtime_bwd: bad arguments"); return A4NT_ERR_ARG; }
    const long long total = (long long)Tout * B * F;
    const int grid = (int)min((long long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(lrelu_maxtime_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Tout, B, F, pre, ldp, slope,
                       dfeat, ldf, argmax);
    return a4nt_check_launch("lrelu_maxtime_bwd");
}

// ---------------------------------------------------------------------------------------------
// binary cross-entropy with logits on one column of a [N][C] logit matrix against a constant target
// (train_adversarial_translator.py:487-489, :690: BCEWithLogitsLoss of the discriminator's score for one author
// against ones / zeros, optionally weighted per sample), mean over the N samples.  torch runs it as a dozen
// element-wise launches on 128 numbers; here: one launch each way.
//   loss = mean_n w_n * (max(x_n, 0) - x_n * y + log1p(exp(-|x_n|)))
//   dx_n = g * w_n * (sigmoid(x_n) - y) / N        (written into column `col` of dlogits, zeros elsewhere)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_logits_fwd_kernel(int N, int C, int col, const float* x, float y, const float* w,
                                                             float* loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float v = x[(long long)n * C + col];
        const float l = fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
        s += w ? w[n] * l : l;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)N;
}
__global__ __launch_bounds__(256) void bce_logits_bwd_kernel(int N, int C, int col, const float* x, float y, const float* w,
                                                             const float* gout, float* dx) {
    const float g = gout[0] / (float)N;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * C; i += gridDim.x * 256) {
        const int n = i / C, c = i - n * C;
        float d = 0.f;
        if (c == col) {
            const float v = x[i];
            d = g * (1.0f / (1.0f + expf(-v)) - y);
            if (w) d *= w[n];
        }
        dx[i] = d;
    }
}
extern "C" int a4nt_bce_logits_fwd(int N, int C, int col, const float* logits, float target, const float* weight, float* loss,
                                   void* stream) {
    if (N <= 0 || C <= 0 || col < 0 || col >= C || !logits || !loss) { a4nt_set_error("a4nt_bce_logits_fwd: bad arguments"); return A4NT_ERR_ARG; }
    hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, N, C, col, logits, target, weight, loss);
    return a4nt_check_launch("bce_logits_fwd");
}
extern "C" int a4nt_bce_logits_bwd(int N, int C, int col, const float* logits, float target, const float* weight,
                                   const float* grad_loss, float* dlogits, void* stream) {
    if (N <= 0 || C <= 0 || col < 0 || col >= C || !logits || !grad_loss || !dlogits) { a4nt_set_error("a4nt_bce_logits_bwd: bad arguments"); return A4NT_ERR_ARG; }
    const int grid = (N * C + 255) / 256 < 64 ? (N * C + 255) / 256 : 64;
    hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, C, col, logits, target, weight,
                       grad_loss, dlogits);
    return a4nt_check_launch("bce_logits_bwd");
}
