// C[M,N] (+)= A[M,K] * B[N,K]^T for few rows and K <= 512: the per-step "d(sample) += d(emb) * E^T" product of the
// generator's backward pass (M = the batch of one decoder step, N = the vocabulary, K = the embedding size).
//
// The 64x64-tile GEMM is latency bound on this shape (one tile per CU, 32 dependent K tiles).  Here the
// small operand is stationary instead: a workgroup of 8 waves owns 128 rows, wave w keeps its 16 rows of A
// over the WHOLE K in registers (K/16 dwordx4 per lane = 128 VGPRs at K = 512) and the workgroup streams
// 16-column blocks of B (16 rows of [N][K], 2 KB contiguous each) through a double-buffered LDS tile, so
// every B element is fetched once per 128 rows and each block costs one barrier for K/4
// v_mfma_f32_16x16x4_f32 per wave.  k is permuted inside 16-element groups (the same permutation on both
// operands): lane (r, q) holds elements 16j+4q+i, i = 0..3, of its row as one dwordx4.
//
// Measured (M=128, N=10001, K=512, accumulate): 26 us against 30 us for the 64x64-tile kernel.  About 8 us of
// that is fixed (launch, the A load, the first block); each further block costs 5 us against 3.5 us of
// MFMA issue time, and 626 blocks over 256 workgroups leave a partial third round.  Rotating the k order per
// workgroup (so that the workgroups do not read the same lines of A at once) changed nothing.
#include "common.h"
#include "gemm_stage.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct NtRowsArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc, beta;
    int nblk;                // 16-column blocks
};

template <int KJ>            // K <= 16*KJ, KJ % 4 == 0
__global__ __launch_bounds__(512) void nt_rows_kernel(NtRowsArgs g) {
    constexpr int KP = 16 * KJ, ROW = KP + 4;        // +4: lanes r, r+8 of a ds_read_b128 fall in different bank groups
    constexpr int NV = (KJ + 7) / 8;                 // float4 per thread per block (16 rows x KP/4 float4 over 512 threads)
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: keeps the buffer descriptors in SGPRs
    const int r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.y * 128 + 16 * w;

    // B staging: float4 index idx = tid + 512*p -> row idx / (KP/4), float4 column idx % (KP/4)
    f32x4 st[NV];
    auto load_block = [&](int blk) {
        const int n0 = blk * 16;
        const int nrows = blk < g.nblk ? min(16, g.N - n0) : 0;
        const float* origin = g.B + (long long)(nrows > 0 ? n0 : 0) * g.ldb;
        const long long bytes = nrows > 0 ? ((long long)(nrows - 1) * g.ldb + g.K) * 4 : 0;
        __amdgpu_buffer_rsrc_t rs = make_rsrc(origin, clamp_bytes(bytes));
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 512 * p;
            const int n = idx / (KP / 4), k = (idx % (KP / 4)) * 4;
            st[p] = bl4(rs, (n < nrows && k < g.K && idx < 16 * (KP / 4)) ? (unsigned)(n * g.ldb + k) * 4u : OOB);
        }
    };
    auto store_block = [&](float* S) {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 512 * p;
            const int n = idx / (KP / 4), k = (idx % (KP / 4)) * 4;
            if (idx < 16 * (KP / 4)) *reinterpret_cast<f32x4*>(&S[n * ROW + k]) = st[p];
        }
    };

    int blk = blockIdx.x;
    load_block(blk);                                           // first B block ahead of A: it reaches LDS while A streams in
    __builtin_amdgcn_sched_barrier(0);
    // stationary A fragments (the MFMAs of the first block wait for them one k-group at a time)
    f32x4 a[KJ];
    {
        const int nrows = min(16, g.M - row0);
        const float* origin = g.A + (long long)max(0, min(row0, g.M - 1)) * g.lda;
        const long long bytes = nrows > 0 ? ((long long)(nrows - 1) * g.lda + g.K) * 4 : 0;
        __amdgpu_buffer_rsrc_t rs = make_rsrc(origin, clamp_bytes(bytes));
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int k = 16 * j + 4 * q;
            a[j] = bl4(rs, (r < nrows && k < g.K) ? (unsigned)(r * g.lda + k) * 4u : OOB);
        }
    }

    __builtin_amdgcn_sched_barrier(0);

    store_block(Bs[0]);
    __syncthreads();
    for (int it = 0; blk < g.nblk; blk += gridDim.x, ++it) {
        const float* S = Bs[it & 1];
        load_block(blk + gridDim.x);                           // next block: in flight under the MFMAs (zeros past the end)
        const int n0 = blk * 16;
        // previous C values (beta): C/D layout of the 16x16 tile: col = lane&15, row = 4*(lane>>4) + i
        const int col = n0 + r;
        float cold[4];
        {                                                       // branch-free: nothing waits on these before the MFMAs
            const int nrows = min(16, g.M - row0);
            const float* origin = g.C + (long long)max(0, min(row0, g.M - 1)) * g.ldc;
            const long long bytes = (nrows > 0 && g.beta) ? ((long long)(nrows - 1) * g.ldc + g.N) * 4 : 0;
            __amdgpu_buffer_rsrc_t rc = make_rsrc(origin, clamp_bytes(bytes));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cold[i] = bl1(rc, (4 * q + i < nrows && col < g.N) ? (unsigned)((4 * q + i) * g.ldc + col) * 4u : OOB);
        }
        // the loads above stay above the MFMA block (hipcc otherwise sinks them next to their use, after it)
        __builtin_amdgcn_sched_barrier(0);
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* Sl = &S[r * ROW + 4 * q];
        auto ldb = [&](int j) { return *reinterpret_cast<const f32x4*>(Sl + 16 * j); };
        // B fragments two k-groups ahead of the MFMAs that use them
        f32x4 b0 = ldb(0), b1 = ldb(1), b2 = ldb(2), b3 = ldb(3);
#pragma unroll
        for (int j = 0; j < KJ; j += 4) {
            f32x4 n0 = b0, n1 = b1, n2 = b2, n3 = b3;
            if (j + 4 < KJ) {
                n0 = ldb(j + 4);
                n1 = ldb(j + 5);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1].x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1].y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1].z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1].w, b1.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 4 < KJ) {
                n2 = ldb(j + 6);
                n3 = ldb(j + 7);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 2].x, b2.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 3].x, b3.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 2].y, b2.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 3].y, b3.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 2].z, b2.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 3].z, b3.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 2].w, b2.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 3].w, b3.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = n0; b1 = n1; b2 = n2; b3 = n3;
        }
        if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * q + i;
                if (row < g.M) g.C[(long long)row * g.ldc + col] = acc0[i] + acc1[i] + cold[i];
            }
        }
        store_block(Bs[(it & 1) ^ 1]);
        __syncthreads();
    }
}

// C-ABI entry: see include/a4nt_hip.h
extern "C" int a4nt_sgemm_nt_rows_ok(int M, int N, int K, int lda, int ldb) {
    return M > 0 && N > 0 && K >= 16 && K <= 512 && (K % 16) == 0 && (lda % 4) == 0 && (ldb % 4) == 0;
}

extern "C" int a4nt_sgemm_nt_rows(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                  int beta, void* stream) {
    if (M <= 0 || N <= 0) return A4NT_OK;
    if (!A || !B || !C || !a4nt_sgemm_nt_rows_ok(M, N, K, lda, ldb) || (((uintptr_t)A | (uintptr_t)B) & 15)) {
        a4nt_set_error("a4nt_sgemm_nt_rows: needs 16 <= K <= 512, K %% 16 == 0, 16-byte aligned A and B rows "
                       "(M=%d N=%d K=%d lda=%d ldb=%d)", M, N, K, lda, ldb);
        return A4NT_ERR_ARG;
    }
    NtRowsArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.beta = beta ? 1 : 0;
    g.nblk = (N + 15) / 16;
    const int rb = (M + 127) / 128;
    int gx = 256 / rb;                                   // one workgroup per CU: A is loaded once per workgroup
    if (gx < 1) gx = 1;
    if (gx > g.nblk) gx = g.nblk;
    dim3 grid(gx, rb);
    hipStream_t s = (hipStream_t)stream;
    if (K <= 128) hipLaunchKernelGGL((nt_rows_kernel<8>), grid, dim3(512), 0, s, g);
    else if (K <= 256) hipLaunchKernelGGL((nt_rows_kernel<16>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((nt_rows_kernel<32>), grid, dim3(512), 0, s, g);
    return a4nt_check_launch("sgemm_nt_rows");
}
