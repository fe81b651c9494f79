// Skinny GEMM for the free-running decoder's per-step products (M = batch <= a few hundred rows):
//   C[M,N] (+)= A[M,K] * B   (transB = 0: B is [K][N];  transB = 1: B is stored [N][K])   (+ bias[N])
//
// With so few rows the LDS tilings of sgemm.hip are bound by their own load -> LDS -> barrier ->
// ds_read chain, not by the matrix pipe (see projgumbel.hip, which shares this scheme).  Here there is no
// LDS and no barrier: every WAVE owns a 16-row x (16*CT)-column output tile for one K slab and feeds
// v_mfma_f32_16x16x4_f32 from registers, with loads running PF 16-k blocks ahead in a register ring.
//   A:            lane (row m, quad q) loads A[m][kb+4q .. +3] (one dwordx4); register i is the operand of
//                 k-step i, whose k set is {kb+4q+i}: a permutation of the block's 16 k's;
//   B, transB=0:  for k = kb+4q+i lane (n, q) loads B[k][c0+CT*n .. +CT-1] (dwordx4 [+ dword]); register c feeds
//                 column tile c = the strided column set {c0 + CT*n + c};
//   B, transB=1:  lane (n, q) loads B[c0+16c+n][kb+4q .. +3] for each column tile c (contiguous tiles).
// CT in {4, 5} and the K split are chosen so that the number of waves is just under a multiple of the
// chip's 1024 SIMDs.  Split-K writes slabs (C + z*slab_stride), reduced by the consumer.
#include "gemm_stage.h"

#define SK_PF 8

struct SkinnyArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    int M, N, K;
    int beta;
    int rt, ct, ks, kchunk;
    long long slab_stride;
};

template <int CT>
struct SkBlock {
    f32x4 a;
    f32x4 b[4];     // transB=0: k-step i -> column tiles 0..3;  transB=1: column tile c (0..3) -> k-steps 0..3
    f32x4 bx;       // fifth column tile (CT = 5): transB=0: k-steps 0..3 of tile 4;  transB=1: same
};

template <bool TB, int CT, bool TAIL>
__device__ __forceinline__ void sk_load(SkBlock<CT>& blk, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned aoff,
                                        const unsigned (&boff)[5], unsigned ldb_bytes, int kb, int kend, int fq, bool on) {
    // kb: first k of the block.  TAIL: the block crosses kend, every element is range-checked on k.
    if (!TAIL) {
        blk.a = bl4(ra, on ? aoff + (unsigned)kb * 4u : OOB);
        if (!TB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned bo = on ? boff[0] + (unsigned)(kb + i) * ldb_bytes : OOB;
                blk.b[i] = bl4(rb, bo);
                if (CT == 5) blk.bx[i] = bl1(rb, on ? bo + 16u : OOB);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) blk.b[c] = bl4(rb, on ? boff[c] + (unsigned)kb * 4u : OOB);
            if (CT == 5) blk.bx = bl4(rb, on ? boff[4] + (unsigned)kb * 4u : OOB);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool kv = on && (kb + 4 * fq + i < kend);
            blk.a[i] = bl1(ra, kv ? aoff + (unsigned)(kb + i) * 4u : OOB);
            if (!TB) {
                const unsigned bo = kv ? boff[0] + (unsigned)(kb + i) * ldb_bytes : OOB;
                blk.b[i] = bl4(rb, bo);
                if (CT == 5) blk.bx[i] = bl1(rb, kv ? bo + 16u : OOB);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) blk.b[c][i] = bl1(rb, kv ? boff[c] + (unsigned)(kb + i) * 4u : OOB);
                if (CT == 5) blk.bx[i] = bl1(rb, kv ? boff[4] + (unsigned)(kb + i) * 4u : OOB);
            }
        }
    }
}

template <bool TB, int CT>
__device__ __forceinline__ void sk_mma(const SkBlock<CT>& blk, f32x4 (&acc)[CT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(blk.a[i], TB ? blk.b[c][i] : blk.b[i][c], acc[c], 0, 0, 0);
        if (CT == 5) acc[CT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(blk.a[i], blk.bx[i], acc[CT - 1], 0, 0, 0);
    }
}

template <bool TB, int CT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // wave id -> (row tile fastest, then strip, then K slab): the waves of a workgroup share their B strip
    const long long gw = (long long)blockIdx.x * 4 + w;
    const int r = (int)(gw % a.rt);
    const int sidx = (int)((gw / a.rt) % a.ct);
    const int z = (int)(gw / ((long long)a.rt * a.ct));
    if (z >= a.ks) return;                                   // no barriers in this kernel
    const int m0 = 16 * r, c0 = sidx * 16 * CT;
    const int k0 = z * a.kchunk, kend = min(a.K, k0 + a.kchunk);

    __amdgpu_buffer_rsrc_t ra = make_rsrc(a.A, clamp_bytes(((long long)(a.M - 1) * a.lda + a.K) * 4));
    const long long bbytes = TB ? ((long long)(a.N - 1) * a.ldb + a.K) * 4 : ((long long)(a.K - 1) * a.ldb + a.N) * 4;
    __amdgpu_buffer_rsrc_t rb = make_rsrc(a.B, clamp_bytes(bbytes));
    const unsigned aoff = (unsigned)(min(m0 + fr, a.M - 1) * a.lda + 4 * fq) * 4u;
    const unsigned ldb_bytes = (unsigned)a.ldb * 4u;
    unsigned boff[5];
    if (!TB) {
        boff[0] = (unsigned)(4 * fq) * ldb_bytes + (unsigned)(c0 + CT * fr) * 4u;
        boff[1] = boff[2] = boff[3] = boff[4] = 0u;
    } else {
#pragma unroll
        for (int c = 0; c < 5; ++c) boff[c] = (unsigned)(min(c0 + 16 * c + fr, a.N - 1) * a.ldb + 4 * fq) * 4u;
    }

    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nfull = (kend - k0) >> 4;                      // whole 16-k blocks of this slab
    SkBlock<CT> ring[SK_PF];
#pragma unroll
    for (int u = 0; u < SK_PF; ++u) sk_load<TB, CT, false>(ring[u], ra, rb, aoff, boff, ldb_bytes, k0 + 16 * u, kend, fq, u < nfull);
    for (int j0 = 0; j0 < nfull; j0 += SK_PF) {
#pragma unroll
        for (int u = 0; u < SK_PF; ++u) {
            if (j0 + u < nfull) {                            // wave-uniform
                sk_mma<TB, CT>(ring[u], acc);
                sk_load<TB, CT, false>(ring[u], ra, rb, aoff, boff, ldb_bytes, k0 + 16 * (j0 + u + SK_PF), kend, fq,
                                       j0 + u + SK_PF < nfull);
            }
        }
    }
    if (k0 + 16 * nfull < kend) {                            // partial last block of the slab
        SkBlock<CT> t;
        sk_load<TB, CT, true>(t, ra, rb, aoff, boff, ldb_bytes, k0 + 16 * nfull, kend, fq, true);
        sk_mma<TB, CT>(t, acc);
    }

    // ---- epilogue: C layout of a 16x16 tile: col = lane&15, row = 4*(lane>>4) + reg -------------------
    float* Cz = a.C + (long long)z * a.slab_stride;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = TB ? (c0 + 16 * c + fr) : (c0 + CT * fr + c);
        if (col >= a.N) continue;
        const float bv = (a.bias && z == 0) ? a.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + 4 * fq + e;
            if (row < a.M) {
                float* p = Cz + (long long)row * a.ldc + col;
                float v = acc[c][e] + bv;
                if (a.beta) v += *p;
                *p = v;
            }
        }
    }
}

// waves needed / column-tile count / K split for a shape: minimise rounds(waves / 1024 SIMDs) * work per wave
static void sk_plan(int M, int N, int K, int ks_in, int* ct_out, int* ks_out) {
    const int rt = (M + 15) / 16;
    double best = 1e30; int bct = 4, bks = 1;
    for (int CT = 4; CT <= 5; ++CT) {
        const int ct = (N + 16 * CT - 1) / (16 * CT);
        for (int ks = (ks_in > 0 ? ks_in : 1); ks <= (ks_in > 0 ? ks_in : 64); ++ks) {
            const int kchunk = (((K + ks - 1) / ks) + 15) / 16 * 16;
            if (ks_in <= 0 && ks > 1) {
                if (kchunk < 64) break;
                if ((long long)(ks - 1) * kchunk >= K) continue;        // would leave an empty slab
            }
            const long long waves = (long long)rt * ct * ks;
            const double rounds = (double)((waves + 1023) / 1024);
            // per wave: kchunk/4 k-steps of CT MFMAs (32 cycles each) + a fixed prologue/epilogue; slabs cost a pass
            double t = rounds * (CT * (kchunk / 4) * 32.0 + 6000.0) + (ks > 1 ? 300.0 * ks : 0.0);
            if (t < best) { best = t; bct = CT; bks = ks; }
        }
    }
    *ct_out = bct; *ks_out = bks;
}

// Suggested split-K factor for a4nt_sgemm_skinny (the caller provides ksplit slabs of C when it is > 1).
extern "C" int a4nt_sgemm_skinny_ksplit(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    int ct, ks;
    sk_plan(M, N, K, 0, &ct, &ks);
    return ks;
}

extern "C" int a4nt_sgemm_skinny(int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                                 int ldc, const float* bias, int beta, int ksplit, long long slab_stride, void* stream) {
    if (M <= 0 || N <= 0) return A4NT_OK;
    if (K <= 0 || !A || !B || !C || ksplit < 1 || (ksplit > 1 && (beta || slab_stride < (long long)(M - 1) * ldc + N))) {
        a4nt_set_error("a4nt_sgemm_skinny: bad arguments (K=%d ksplit=%d beta=%d)", K, ksplit, beta);
        return A4NT_ERR_ARG;
    }
    int CT, ks;
    sk_plan(M, N, K, ksplit, &CT, &ks);
    SkinnyArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.bias = bias; a.M = M; a.N = N; a.K = K; a.beta = beta;
    a.rt = (M + 15) / 16; a.ct = (N + 16 * CT - 1) / (16 * CT); a.ks = ksplit;
    a.kchunk = (((K + ksplit - 1) / ksplit) + 15) / 16 * 16;
    a.slab_stride = slab_stride;
    if (ksplit > 1 && (long long)(ksplit - 1) * a.kchunk >= K) {
        a4nt_set_error("a4nt_sgemm_skinny: ksplit=%d leaves empty slabs at K=%d", ksplit, K);
        return A4NT_ERR_ARG;
    }
    const long long waves = (long long)a.rt * a.ct * a.ks;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (!transB) {
        if (CT == 4) hipLaunchKernelGGL((skinny_kernel<false, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((skinny_kernel<false, 5>), grid, block, 0, s, a);
    } else {
        if (CT == 4) hipLaunchKernelGGL((skinny_kernel<true, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((skinny_kernel<true, 5>), grid, block, 0, s, a);
    }
    return a4nt_check_launch("sgemm_skinny");
}
