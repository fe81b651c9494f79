// Fused output projection + Gumbel-softmax sampling for the free-running decoder
// (reference models/char_translator.py:377-392: dec_out = p_rnn.mm(W)+b; gumbel_softmax_sample;
//  argmax; done / gen_lens bookkeeping).
//
// One decoder step multiplies only B <= 128 rows by the [H,V] projection, so the generic 64x64 /
// 128x128 tilings leave the chip badly balanced (314 or 79 workgroups on 256 CUs).  Here every
// workgroup takes ALL rows and a 16*NT-column strip of the vocabulary (209 strips at V = 10001),
// runs the contraction on v_mfma_f32_16x16x4_f32 and finishes the sampling in the epilogue:
//     z = scale*(h W + b) + gumbel(u)                      (greedy: z = h W + b)
// softmax((log_softmax(x)+g)/tau) == softmax(z/tau) (shift invariance), so the row statistics the
// sampler needs are max z, arg max z and sum exp((z - max)/tau).  Each strip emits those three per
// row; a tiny finalize kernel merges the strips (online-softmax merge), derives the straight-through
// value and updates done / gen_lens.  z is stored once (saved for the backward pass); the logits
// never make a round trip through HBM.
#include "gemm_stage.h"

#define PG_BM 128
#define PG_BK 32
#define PG_BC 64          // staged B columns (NT <= 4)

struct ProjGumbelArgs {
    const float* h; int ldh;
    const float* W; int ldw;
    const float* bias;
    int M, N, K;
    float scale, inv_tau;
    int greedy;
    const float* U; int ldu;
    unsigned long long seed; unsigned step;
    float* z; int ldz;
    float* part;            // [M][nstrips][4] = (max z, sum exp((z-max)/tau), arg max as int bits, unused)
    int nstrips;
};

__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
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

template <int NT>
__device__ __forceinline__ void pg_mma(const float* as, const float* bs, int w, int fr, int fq, f32x4 (&acc)[2][NT]) {
#pragma unroll
    for (int kk = 0; kk < PG_BK; kk += 4) {
        float a[2], b[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = as[(kk + fq) * (PG_BM + LPAD) + (2 * w + i) * 16 + fr];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = bs[(kk + fq) * (PG_BC + LPAD) + nt * 16 + fr];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[nt], acc[i][nt], 0, 0, 0);
    }
}

template <int NT, bool EB>
__device__ __forceinline__ void pg_kloop(const ProjGumbelArgs& a, int m0, int c0, float (*As)[PG_BK * (PG_BM + LPAD)],
                                         float (*Bs)[PG_BK * (PG_BC + LPAD)], int w, int fr, int fq, f32x4 (&acc)[2][NT]) {
    constexpr int PF = 2;
    StageKC<PG_BM, PG_BK> sa[PF];
    StageMC<PG_BC, PG_BK> sb[PF];
    const int nk = a.K / PG_BK;
    const int kfull = nk * PG_BK;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        sa[u].template load<false>(a.h, a.ldh, m0, a.M, u * PG_BK, kfull);
        sb[u].template load<EB>(a.W, a.ldw, c0, a.N, u * PG_BK, kfull);
    }
    if (nk > 0) { sa[0].store(As[0]); sb[0].store(Bs[0]); }
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            const int cur = kt & 1;
            sa[u].template load<false>(a.h, a.ldh, m0, a.M, (kt + PF) * PG_BK, kfull);
            sb[u].template load<EB>(a.W, a.ldw, c0, a.N, (kt + PF) * PG_BK, kfull);
            pg_mma<NT>(As[cur], Bs[cur], w, fr, fq, acc);
            sa[(u + 1) % PF].store(As[cur ^ 1]);
            sb[(u + 1) % PF].store(Bs[cur ^ 1]);
            __syncthreads();
        }
    }
    if (kfull < a.K) {                       // partial last K tile
        sa[0].template load<true>(a.h, a.ldh, m0, a.M, kfull, a.K);
        sb[0].template load<true>(a.W, a.ldw, c0, a.N, kfull, a.K);
        sa[0].store(As[0]); sb[0].store(Bs[0]);
        __syncthreads();
        pg_mma<NT>(As[0], Bs[0], w, fr, fq, acc);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void proj_gumbel_kernel(ProjGumbelArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][PG_BK * (PG_BM + LPAD)];
    __shared__ __attribute__((aligned(16))) float Bs[2][PG_BK * (PG_BC + LPAD)];
    constexpr int SW = 16 * NT;
    const int strip = blockIdx.x;
    const int m0 = blockIdx.y * PG_BM, c0 = strip * SW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (a.N - c0 >= PG_BC) pg_kloop<NT, false>(a, m0, c0, As, Bs, w, fr, fq, acc);
    else pg_kloop<NT, true>(a, m0, c0, As, Bs, w, fr, fq, acc);

    // ---- epilogue: C layout of a 16x16 tile: col = lane&15, row = 4*(lane>>4) + reg -----------------
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = c0 + nt * 16 + fr;
        bv[nt] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rbase = m0 + 32 * w + 16 * i + 4 * fq;          // rows rbase .. rbase+3 (multiple of 4)
        float zv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c0 + nt * 16 + fr;
            float g4[4] = {0.f, 0.f, 0.f, 0.f};
            if (!a.greedy) {
                if (a.U) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = rbase + e;
                        const float u = (row < a.M && col < a.N) ? a.U[(long long)row * a.ldu + col] : 0.5f;
                        g4[e] = -logf(-logf(u + 1e-20f) + 1e-20f);
                    }
                } else {
                    // one Philox4x32-10 call = the 4 rows (rbase..rbase+3) of this column
                    unsigned o[4];
                    philox4x32((unsigned)col, (unsigned)(rbase >> 2), a.step, 0u, (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = (float)(o[e] >> 8) * (1.0f / 16777216.0f);
                        g4[e] = -__logf(-__logf(u + 1e-20f) + 1e-20f);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = rbase + e;
                const float x = acc[i][nt][e] + bv[nt];
                float zz = a.greedy ? x : (x * a.scale + g4[e]);
                const bool valid = (row < a.M) && (col < a.N);
                if (valid && a.z) a.z[(long long)row * a.ldz + col] = zz;
                zv[nt][e] = valid ? zz : -INFINITY;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float v = zv[nt][e];
                if (v > m) { m = v; mi = c0 + nt * 16 + fr; }
            }
            // reduce over the 16 lanes that share this row (lanes differ in fr only)
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float om = __shfl_xor(m, o);
                const int oi = __shfl_xor(mi, o);
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
            }
            float s = 0.f;
            if (!a.greedy) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) s += (zv[nt][e] == -INFINITY) ? 0.f : expf((zv[nt][e] - m) * a.inv_tau);
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
            }
            const int row = rbase + e;
            if (fr == 0 && row < a.M) {
                float* p = a.part + ((long long)row * a.nstrips + strip) * 4;
                p[0] = m; p[1] = s; p[2] = __int_as_float(mi); p[3] = 0.f;
            }
        }
    }
}

// Merge the strips of one row (online-softmax merge), then the sampler's bookkeeping.
struct PgFinalArgs {
    const float* part; int nstrips; float inv_tau; int greedy; int hard;
    float* stats; long long* tok; float* val_in; float* val_out; int* gen_lens; unsigned char* done; long long end_c;
};
__global__ __launch_bounds__(64) void proj_gumbel_final_kernel(PgFinalArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* p = a.part + (long long)b * a.nstrips * 4;
    float m = -INFINITY; int mi = 0x7fffffff;
    for (int s = lane; s < a.nstrips; s += 64) {
        const float om = p[4 * s]; const int oi = __float_as_int(p[4 * s + 2]);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o); const int oi = __shfl_xor(mi, o);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float sum = 0.f;
    if (!a.greedy) {
        for (int s = lane; s < a.nstrips; s += 64) {
            const float sm = p[4 * s];
            if (sm != -INFINITY) sum += p[4 * s + 1] * expf((sm - m) * a.inv_tau);
        }
        sum = wave_sum(sum);
    }
    if (lane == 0) {
        float hot = 1.0f, s2 = 1.0f;
        if (!a.greedy) {
            s2 = sum;
            const float p_hot = 1.0f / s2;
            hot = a.hard ? ((1.0f - p_hot) + p_hot) : p_hot;
        }
        const bool prev_done = a.done ? (a.done[b] != 0) : false;
        if (a.stats) { a.stats[2 * b] = m; a.stats[2 * b + 1] = s2; }
        a.tok[b] = mi;
        if (a.val_in) a.val_in[b] = hot;
        if (a.val_out) a.val_out[b] = prev_done ? 0.f : hot;
        if (a.gen_lens) a.gen_lens[b] += prev_done ? 0 : 1;
        if (a.done) a.done[b] = (prev_done || (long long)mi == a.end_c) ? 1 : 0;
    }
}

static int pg_nt(int N) {
    for (int nt = 1; nt <= 4; ++nt)
        if ((long long)16 * nt * 256 >= N) return nt;
    return 4;
}

extern "C" long long a4nt_proj_gumbel_part_floats(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int sw = 16 * pg_nt(N);
    return (long long)M * ((N + sw - 1) / sw) * 4;
}

extern "C" int a4nt_proj_gumbel_fwd(int M, int N, int K, const float* h, int ldh, const float* W, int ldw,
                                    const float* bias, float scale, float tau, int hard, int greedy, const float* U,
                                    int ldu, unsigned long long seed, unsigned step, float* z, int ldz, float* part,
                                    float* stats, long long* tok, float* val_in, float* val_out, int* gen_lens,
                                    unsigned char* done, long long end_c, void* stream) {
    if (M <= 0) return A4NT_OK;
    if (N <= 0 || K <= 0 || !h || !W || !part || !tok || (!greedy && tau <= 0.f)) {
        a4nt_set_error("a4nt_proj_gumbel_fwd: bad arguments (N=%d K=%d tau=%g)", N, K, (double)tau);
        return A4NT_ERR_ARG;
    }
    const int nt = pg_nt(N), sw = 16 * nt;
    ProjGumbelArgs a;
    a.h = h; a.ldh = ldh; a.W = W; a.ldw = ldw; a.bias = bias; a.M = M; a.N = N; a.K = K;
    a.scale = scale; a.inv_tau = greedy ? 1.f : 1.f / tau; a.greedy = greedy; a.U = U; a.ldu = ldu; a.seed = seed;
    a.step = step; a.z = z; a.ldz = ldz; a.part = part; a.nstrips = (N + sw - 1) / sw;
    dim3 grid(a.nstrips, (M + PG_BM - 1) / PG_BM), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (nt == 1) hipLaunchKernelGGL(proj_gumbel_kernel<1>, grid, block, 0, s, a);
    else if (nt == 2) hipLaunchKernelGGL(proj_gumbel_kernel<2>, grid, block, 0, s, a);
    else if (nt == 3) hipLaunchKernelGGL(proj_gumbel_kernel<3>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(proj_gumbel_kernel<4>, grid, block, 0, s, a);
    int rc = a4nt_check_launch("proj_gumbel");
    if (rc != A4NT_OK) return rc;
    PgFinalArgs f;
    f.part = part; f.nstrips = a.nstrips; f.inv_tau = a.inv_tau; f.greedy = greedy; f.hard = hard; f.stats = stats;
    f.tok = tok; f.val_in = val_in; f.val_out = val_out; f.gen_lens = gen_lens; f.done = done; f.end_c = end_c;
    hipLaunchKernelGGL(proj_gumbel_final_kernel, dim3(M), dim3(64), 0, s, f);
    return a4nt_check_launch("proj_gumbel_final");
}
