// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
//   C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
//
// Used for every time-batched contraction of the hot path: LSTM input gates x*W_ih^T, vocab
// projection h*W, soft-embedding samp*E, and all the weight / data gradients of those.
// 256 threads = 4 waves as 2x2; each wave owns WM x WN tiles of 32x32; LDS tiles are stored
// k-major ([BK][BM+4]) so that an MFMA operand fetch is one conflict-free ds_read_b32 per lane.
// Global->LDS staging goes through registers and is double buffered (one barrier per K tile).
// Arbitrary M, N, K and leading dimensions: rows that are not 16-byte aligned (V = 10001) fall back
// to dword loads, everything is bounds-checked.  Split-K writes per-slice slabs (deterministic;
// the consumer sums them).
#include "common.h"

#define BK 16
#define LPAD 4

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int beta;            // 0: overwrite, 1: accumulate into C
    int kchunk;          // K range per blockIdx.z (multiple of BK)
    long long slab;      // C offset (floats) per blockIdx.z
    int tiles_n;
};

// Stage one [ROWS x BK] operand tile, global element (r, k) at base[r*ld + k]  ("k contiguous").
template <int ROWS>
struct StageKC {
    static constexpr int NV = ROWS / 64;
    f32x4 v[NV];
    __device__ __forceinline__ void load(const float* base, int ld, int r0, int rmax, int k0, int kmax, bool aligned) {
        const int tid = threadIdx.x;
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int r = r0 + (tid >> 2) + 64 * p;
            const int k = k0 + kq;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (r < rmax) {
                const float* src = base + (long long)r * ld + k;
                if (aligned && k + 3 < kmax) {
                    t = *reinterpret_cast<const f32x4*>(src);
                } else {
                    if (k + 0 < kmax) t.x = src[0];
                    if (k + 1 < kmax) t.y = src[1];
                    if (k + 2 < kmax) t.z = src[2];
                    if (k + 3 < kmax) t.w = src[3];
                }
            }
            v[p] = t;
        }
    }
    __device__ __forceinline__ void store(float* S /* [BK][ROWS+LPAD] */) const {
        const int tid = threadIdx.x;
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int r = (tid >> 2) + 64 * p;
            S[(kq + 0) * (ROWS + LPAD) + r] = v[p].x;
            S[(kq + 1) * (ROWS + LPAD) + r] = v[p].y;
            S[(kq + 2) * (ROWS + LPAD) + r] = v[p].z;
            S[(kq + 3) * (ROWS + LPAD) + r] = v[p].w;
        }
    }
};

// Stage one [BK x COLS] operand tile, global element (k, c) at base[k*ld + c]  ("mn contiguous").
template <int COLS>
struct StageMC {
    static constexpr int PER_ROW = COLS / 4;
    static constexpr int NV = (BK * PER_ROW) / 256;
    f32x4 v[NV];
    __device__ __forceinline__ void load(const float* base, int ld, int c0, int cmax, int k0, int kmax, bool aligned) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 256 * p;
            const int k = k0 + idx / PER_ROW;
            const int c = c0 + (idx % PER_ROW) * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k < kmax) {
                const float* src = base + (long long)k * ld + c;
                if (aligned && c + 3 < cmax) {
                    t = *reinterpret_cast<const f32x4*>(src);
                } else {
                    if (c + 0 < cmax) t.x = src[0];
                    if (c + 1 < cmax) t.y = src[1];
                    if (c + 2 < cmax) t.z = src[2];
                    if (c + 3 < cmax) t.w = src[3];
                }
            }
            v[p] = t;
        }
    }
    __device__ __forceinline__ void store(float* S /* [BK][COLS+LPAD] */) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 256 * p;
            const int k = idx / PER_ROW;
            const int c = (idx % PER_ROW) * 4;
            *reinterpret_cast<f32x4*>(&S[k * (COLS + LPAD) + c]) = v[p];
        }
    }
};

template <bool TA, bool TB, int WM, int WN>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) float As[2][BK * (BM + LPAD)];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * (BN + LPAD)];

    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    float* C = g.C + (long long)blockIdx.z * g.slab;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (32 * WM), wn = (wave & 1) * (32 * WN);

    const bool a_al = ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((g.lda & 3) == 0);
    const bool b_al = ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) && ((g.ldb & 3) == 0);

    typename std::conditional<TA, StageMC<BM>, StageKC<BM>>::type sa;
    typename std::conditional<TB, StageKC<BN>, StageMC<BN>>::type sb;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        sa.load(g.A, g.lda, m0, g.M, kbeg, kend, a_al);
        sb.load(g.B, g.ldb, n0, g.N, kbeg, kend, b_al);
        sa.store(As[0]);
        sb.store(Bs[0]);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * BK;
            sa.load(g.A, g.lda, m0, g.M, k0, kend, a_al);
            sb.load(g.B, g.ldb, n0, g.N, k0, kend, b_al);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = as[kr * (BM + LPAD) + wm + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = bs[kr * (BN + LPAD) + wn + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(As[cur ^ 1]);
            sb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool add_bias = (g.bias != nullptr) && (blockIdx.z == 0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float* dst = C + (long long)row * g.ldc + col;
                    float v = acc[i][j][r] + bv;
                    if (g.beta) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <bool TA, bool TB, int WM, int WN>
static int launch_sgemm(GemmArgs g, int ksplit, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(tiles_m * g.tiles_n, 1, ksplit);
    hipLaunchKernelGGL((sgemm_kernel<TA, TB, WM, WN>), grid, dim3(256), 0, s, g);
    return a4nt_check_launch("sgemm");
}

// C-ABI entry: see include/a4nt_hip.h
extern "C" int a4nt_sgemm(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                          int ldb, float* C, int ldc, const float* bias, int beta, int ksplit, long long slab_stride,
                          void* stream) {
    if (M <= 0 || N <= 0) return A4NT_OK;
    if (K < 0 || !A || !B || !C || ksplit < 1 || (transA && transB)) {
        a4nt_set_error("a4nt_sgemm: bad arguments (M=%d N=%d K=%d ksplit=%d tA=%d tB=%d)", M, N, K, ksplit, transA, transB);
        return A4NT_ERR_ARG;
    }
    if (ksplit > 1 && (beta || slab_stride < (long long)M * ldc)) {
        a4nt_set_error("a4nt_sgemm: split-K needs beta=0 and slab_stride >= M*ldc");
        return A4NT_ERR_ARG;
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.beta = beta;
    int kchunk = (K + ksplit - 1) / ksplit;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    if (kchunk == 0) kchunk = BK;
    g.kchunk = kchunk;
    g.slab = slab_stride;
    g.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    // tile choice: big 128x128 tiles when they still fill the chip, 64x64 otherwise
    const long long big_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * ksplit;
    const bool big = big_tiles >= 200;
    if (!transA && !transB) return big ? launch_sgemm<false, false, 2, 2>(g, ksplit, s) : launch_sgemm<false, false, 1, 1>(g, ksplit, s);
    if (!transA && transB) return big ? launch_sgemm<false, true, 2, 2>(g, ksplit, s) : launch_sgemm<false, true, 1, 1>(g, ksplit, s);
    return big ? launch_sgemm<true, false, 2, 2>(g, ksplit, s) : launch_sgemm<true, false, 1, 1>(g, ksplit, s);
}
