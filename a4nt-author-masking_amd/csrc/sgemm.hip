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
// Few-tile products (the per-step decode GEMMs: 128 rows) run 64x64 tiles with KG "k-groups" of 4 waves
// per workgroup: every group walks its own part of the workgroup's K range through its own LDS buffers
// and the groups' accumulators are summed through LDS at the end.  That puts 2-4 waves on each SIMD of
// a CU that holds one tile (the single-group kernel was latency bound there) without more C slabs.
#include "common.h"

#ifndef BK
#define BK 16  // 32 was measured: two workgroups per CU instead of three, 113 against 124 TFLOP/s on the K = 512 shapes
#endif
#ifndef PFB
#define PFB 2   // K tiles in flight, 128x128 tiles
#endif
#ifndef PFS
#define PFS 4   // K tiles in flight, 64x64 tiles (small-M decode GEMMs are latency bound)
#endif

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int beta;            // 0: overwrite, 1: accumulate into C
    int kchunk;          // K range per blockIdx.z (multiple of BK)
    long long slab;      // C offset (floats) per blockIdx.z
    int tiles_n;
    float* part;         // NLL epilogue: [M][nparts][2] per-row (max, sum exp) over each wave's 64 columns, or null
    int nparts;
    float* colpart;      // CS instantiation: [gridDim.z][N] column sums of B over this K slab (bias gradient), or null
};

#include "gemm_stage.h"

template <bool T, int R> struct StageSel { typedef StageKC<R, BK> type; };
template <int R> struct StageSel<true, R> { typedef StageMC<R, BK> type; };

template <int WM, int WN>
__device__ __forceinline__ void mma_tile(const float* as, const float* bs, int lane, int wm, int wn, f32x16 (&acc)[WM][WN]) {
    mma_tile_k<WM, WN, BK>(as, bs, lane, wm, wn, acc);
}

// K loop.  EA / EB: the "mn contiguous" operand of this workgroup is an edge tile (dword loads).
// Full K tiles run through a PF-deep register ring with branch-free loads (tiles past the end load
// zeros); a partial last K tile is staged synchronously afterwards.
// [kbeg, kend) is this k-group's range (possibly empty); nk and partial are workgroup-uniform: the loop
// trip count (the largest group's full tiles) and whether some group has a partial last tile -- every
// group runs the same barriers, a group with less work multiplies zeros.
// CS (B stored [K][N] only): every staged B tile is also added, per thread, into colacc -- the thread's four columns
// (tid % 32) * 4 .. + 3 of the tile, over the k rows it stages -- for the column sums of B (sgemm_kernel's CS epilogue).
template <bool TA, bool TB, int WM, int WN, int PF, bool EA, bool EB, bool CS = false>
__device__ __forceinline__ void k_loop(const GemmArgs& g, int m0, int n0, int kbeg, int kend, int nk, bool partial,
                                       float (*As)[BK * (64 * WM + LPAD)],
                                       float (*Bs)[BK * (64 * WN + LPAD)], int lane, int wm, int wn,
                                       f32x16 (&acc)[WM][WN], f32x4& colacc) {
    auto cs_add = [&](const typename StageSel<!TB, 64 * WN>::type& st) {
        if (CS) {
#pragma unroll
            for (int p = 0; p < StageSel<!TB, 64 * WN>::type::NV; ++p) {
                colacc.x += st.v[p].x; colacc.y += st.v[p].y; colacc.z += st.v[p].z; colacc.w += st.v[p].w;
            }
        }
    };
    constexpr int BM = 64 * WM, BN = 64 * WN;
    typename StageSel<TA, BM>::type sa[PF];      // TA: A stored [K][M] -> mn contiguous
    typename StageSel<!TB, BN>::type sb[PF];     // !TB: B stored [K][N] -> mn contiguous
    const int kfull = kbeg + max(0, (kend - kbeg) / BK) * BK;   // ring loads see [kbeg, kfull): anything later reads as zeros
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        sa[u].template load<TA && EA>(g.A, g.lda, m0, g.M, kbeg + u * BK, kfull);
        sb[u].template load<!TB && EB>(g.B, g.ldb, n0, g.N, kbeg + u * BK, kfull);
    }
    if (nk > 0) {
        sa[0].store(As[0]);
        sb[0].store(Bs[0]);
        cs_add(sb[0]);
    }
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            const int cur = kt & 1;
            // slot u is free (tile kt already sits in LDS): refill it with tile kt+PF
            sa[u].template load<TA && EA>(g.A, g.lda, m0, g.M, kbeg + (kt + PF) * BK, kfull);
            sb[u].template load<!TB && EB>(g.B, g.ldb, n0, g.N, kbeg + (kt + PF) * BK, kfull);
            mma_tile<WM, WN>(As[cur], Bs[cur], lane, wm, wn, acc);
            sa[(u + 1) % PF].store(As[cur ^ 1]);       // tile kt+1 (zeros when past the end)
            sb[(u + 1) % PF].store(Bs[cur ^ 1]);
            cs_add(sb[(u + 1) % PF]);
            __syncthreads();
        }
    }
    if (partial) {                                      // partial last K tile (zeros for the groups without one)
        sa[0].template load<true>(g.A, g.lda, m0, g.M, kfull, kend);
        sb[0].template load<true>(g.B, g.ldb, n0, g.N, kfull, kend);
        sa[0].store(As[0]);
        sb[0].store(Bs[0]);
        cs_add(sb[0]);
        __syncthreads();
        mma_tile<WM, WN>(As[0], Bs[0], lane, wm, wn, acc);
    }
}

// Epilogue of one workgroup tile at (m0, n0): C (+)= acc (+ bias), bounds-checked; NLL: per row the (max, sum of exp)
// of this wave's 32*WN columns into g.part (see a4nt_sgemm_nll).
// C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <int WM, int WN, bool NLL>
__device__ __forceinline__ void store_tile(const GemmArgs& g, float* C, f32x16 (&acc)[WM][WN], int m0, int n0, int tn,
                                           int lane, int wave, int wm, int wn, bool add_bias) {
    float bcol[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn + 32 * j + (lane & 31);
        bcol[j] = (add_bias && col < g.N) ? g.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            if (col >= g.N) continue;
            const float bv = bcol[j];
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
    if (NLL) {
        // log-softmax statistics of the logits just produced, while they are still in the accumulators: per row the
        // (max, sum of exp relative to it) over this wave's 32*WN columns.  A row of the 32x32 tile lives in one
        // register index across the 32 lanes of a half wave: butterfly over those lanes.  The row kernel
        // (a4nt_nll_grad_rows) merges the partials of a row, so the loss needs no pass of its own over the logits.
        const int p = tn * 2 + (wave & 1);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[WN];
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + wn + 32 * j + (lane & 31);
                    v[j] = col < g.N ? acc[i][j][r] + bcol[j] : -INFINITY;
                    m = fmaxf(m, v[j]);
                }
                // over the 32 lanes of the half wave: the xor-16 partner through the crossbar, then DPP rotations inside
                // the 16-lane rows (lane 0 of a row: the same operands in the same order as the xor butterfly)
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, pg_dpp_f<PG_ROR(8)>(m));
                m = fmaxf(m, pg_dpp_f<PG_ROR(4)>(m));
                m = fmaxf(m, pg_dpp_f<PG_ROR(2)>(m));
                m = fmaxf(m, pg_dpp_f<PG_ROR(1)>(m));
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < WN; ++j) e += (v[j] == -INFINITY) ? 0.f : expf(v[j] - m);
                e += __shfl_xor(e, 16);
                e += pg_dpp_f<PG_ROR(8)>(e);
                e += pg_dpp_f<PG_ROR(4)>(e);
                e += pg_dpp_f<PG_ROR(2)>(e);
                e += pg_dpp_f<PG_ROR(1)>(e);
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if ((lane & 31) == 0 && row < g.M) {
                    float* q = g.part + ((long long)row * g.nparts + p) * 2;
                    q[0] = m; q[1] = e;
                }
            }
    }
}

template <bool TA, bool TB, int WM, int WN, int PF, int KG, bool NLL = false, bool CS = false>
__global__ __launch_bounds__(256 * KG) void sgemm_kernel(GemmArgs g) {
    static_assert(!CS || (!TB && KG == 1 && WN == 2), "column sums: B stored [K][N], 128-wide tiles");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int ASZ = BK * (BM + LPAD), BSZ = BK * (BN + LPAD);
    static_assert(KG == 1 || (WM == 1 && WN == 1), "k-groups: 64x64 tiles only");
    __shared__ __attribute__((aligned(16))) float smem[KG * 2 * (ASZ + BSZ)];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;   // wave-uniform: descriptors stay in SGPRs
    float (*As)[ASZ] = reinterpret_cast<float (*)[ASZ]>(smem + grp * 2 * (ASZ + BSZ));
    float (*Bs)[BSZ] = reinterpret_cast<float (*)[BSZ]>(smem + grp * 2 * (ASZ + BSZ) + 2 * ASZ);

    // Tile order.  Workgroups are dealt to the 8 XCDs round robin (block b -> XCD b % 8; speed only, nothing depends
    // on it), and each XCD has its own 4 MB L2: XCD x takes a contiguous eighth of a banded tile order (bands of 8
    // tile columns, row-major inside a band), so the ~96 tiles an XCD runs at a time share about 12 A row panels and
    // 8 B column panels (5 MB at K = 512) instead of 48 + 2 spread over the whole matrix.
    int tile = blockIdx.x, zz = blockIdx.z;
    if (gridDim.z > 1 && (gridDim.z & 7) == 0 && gridDim.x <= 32) {
        // few tiles, K split 8- or 16-fold (the per-step decode products: 128 rows against a [512 x 10001] weight):
        // every tile of a K slab on ONE XCD, so that the slab's part of the weight (1/8 or 1/16 of it) is fetched
        // into that XCD's L2 once -- by tiles, each XCD would pull its column panels over the whole K, 2.4x the bytes
        const int L = blockIdx.z * gridDim.x + blockIdx.x, x = L & 7, q = L >> 3;
        zz = x + 8 * (q / (int)gridDim.x);
        tile = q % (int)gridDim.x;
    } else {
        const int nt = gridDim.x, x = tile & 7, i = tile >> 3;
        tile = x * (nt >> 3) + min(x, nt & 7) + i;
    }
    int tm, tn;
    {
        const int tiles_m = gridDim.x / g.tiles_n;
        const int band_tiles = tiles_m * 8, band = tile / band_tiles;
        const int wb = min(8, g.tiles_n - 8 * band);            // width of this band
        const int idx = tile - band * band_tiles;
        tm = idx / wb;
        tn = 8 * band + idx % wb;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wg_kbeg = zz * g.kchunk;
    const int wg_kend = min(g.K, wg_kbeg + g.kchunk);
    // k-groups: ceil(tiles/KG) K tiles each
    const int wg_len = max(0, wg_kend - wg_kbeg);
    const int tg = ((wg_len + BK - 1) / BK + KG - 1) / KG;
    const int kbeg = wg_kbeg + grp * tg * BK;
    const int kend = min(wg_kend, kbeg + tg * BK);
    const int nk = min(tg, wg_len / BK);               // group 0 has the most full tiles
    const bool partial = (wg_len % BK) != 0;
    float* C = g.C + (long long)zz * g.slab;

    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = (wave >> 1) * (32 * WM), wn = (wave & 1) * (32 * WN);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // workgroup-uniform edge flags of the "mn contiguous" operands (K-contiguous ones need none)
    const bool ea = TA && (g.M - m0 < BM);
    const bool eb = !TB && (g.N - n0 < BN);
    f32x4 colacc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!ea && !eb) k_loop<TA, TB, WM, WN, PF, false, false, CS>(g, m0, n0, kbeg, kend, nk, partial, As, Bs, lane, wm, wn, acc, colacc);
    else if (ea && !eb) k_loop<TA, TB, WM, WN, PF, true, false, CS>(g, m0, n0, kbeg, kend, nk, partial, As, Bs, lane, wm, wn, acc, colacc);
    else if (!ea && eb) k_loop<TA, TB, WM, WN, PF, false, true, CS>(g, m0, n0, kbeg, kend, nk, partial, As, Bs, lane, wm, wn, acc, colacc);
    else k_loop<TA, TB, WM, WN, PF, true, true, CS>(g, m0, n0, kbeg, kend, nk, partial, As, Bs, lane, wm, wn, acc, colacc);

    if (CS) {
        // column sums of B over this workgroup's K slab: the 8 threads that stage the same four columns (tid % 32) hold
        // the k rows tid / 32, tid / 32 + 8, ...; summed through LDS in a fixed order; written by the first tile row only
        __syncthreads();                               // the staging buffers are free
        float* red = smem;                             // [8][128]
        *reinterpret_cast<f32x4*>(&red[(threadIdx.x >> 5) * 128 + (threadIdx.x & 31) * 4]) = colacc;
        __syncthreads();
        if (tm == 0 && threadIdx.x < 128 && n0 + (int)threadIdx.x < g.N) {
            const float* r = red + threadIdx.x;
            g.colpart[(long long)zz * g.N + n0 + threadIdx.x] =
                ((r[0] + r[128]) + (r[256] + r[384])) + ((r[512] + r[640]) + (r[768] + r[896]));
        }
        __syncthreads();
    }

    if (KG > 1) {                                      // sum the k-groups' accumulators into group 0
        __syncthreads();                               // every group is done with its staging buffers
        float* red = smem;                             // [KG-1][4 waves][16][64]
        if (grp > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((grp - 1) * 4 + wave) * 16 + r) * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int q = 0; q < KG - 1; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += red[((q * 4 + wave) * 16 + r) * 64 + lane];
    }

    store_tile<WM, WN, NLL>(g, C, acc, m0, n0, tn, lane, wave, wm, wn, (g.bias != nullptr) && (zz == 0));
}

template <bool TA, bool TB, int WM, int WN, int PF, int KG = 1>
static int launch_sgemm(GemmArgs g, int ksplit, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(tiles_m * g.tiles_n, 1, ksplit);
    hipLaunchKernelGGL((sgemm_kernel<TA, TB, WM, WN, PF, KG>), grid, dim3(256 * KG), 0, s, g);
    return a4nt_check_launch("sgemm");
}

// C-ABI entry: see include/a4nt_hip.h
// Workgroup tile edge a4nt_sgemm picks: 128x128 tiles when they still fill the chip, 64x64 otherwise.
extern "C" int a4nt_sgemm_tile(int M, int N, int ksplit) {
    const long long big_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * (ksplit < 1 ? 1 : ksplit);
    return big_tiles >= 200 ? 128 : 64;
}

// k-groups of a 64x64-tile launch: two once the launch is small enough to leave most CUs with one tile
// (measured: 28 -> 23 us for the [128 x 10001] x [10001 x 512] split-K product of a decoder step; four
// groups were never better), while each group keeps at least 4 K tiles.  a4nt_sgemm_set_kgroups(n > 0) pins it (benchmarks), 0 = this rule.
static int g_kgroups = 0;
extern "C" int a4nt_sgemm_set_kgroups(int n) {
    if (n != 0 && n != 1 && n != 2 && n != 4) {
        a4nt_set_error("a4nt_sgemm_set_kgroups: %d (0 = automatic, 1, 2 or 4)", n);
        return A4NT_ERR_ARG;
    }
    g_kgroups = n;
    return A4NT_OK;
}
static int small_kgroups(int M, int N, int kchunk, int ksplit) {
    if (g_kgroups) return g_kgroups;
    const long long wgs = (long long)((M + 63) / 64) * ((N + 63) / 64) * ksplit;
    const int ktiles = (kchunk + BK - 1) / BK;
    int kg = wgs <= 768 ? 2 : 1;
    while (kg > 1 && ktiles / kg < 4) kg >>= 1;
    return kg;
}

template <bool TA, bool TB>
static int launch_small(GemmArgs g, int ksplit, hipStream_t s) {
    switch (small_kgroups(g.M, g.N, g.kchunk, ksplit)) {
        case 4: return launch_sgemm<TA, TB, 1, 1, PFS, 4>(g, ksplit, s);
        case 2: return launch_sgemm<TA, TB, 1, 1, PFS, 2>(g, ksplit, s);
        default: return launch_sgemm<TA, TB, 1, 1, PFS, 1>(g, ksplit, s);
    }
}

static thread_local bool g_tail_split = true;

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
    // Tail split: a launch of t 128x128 tiles takes ceil(t/256) rounds on the 256 CUs; when the last round would be
    // mostly empty (1600 tiles = 6.25 rounds for [12800 x 2048], 800 = 3.125 for [25600 x 512]) the rows of that
    // partial round go to a second launch, which then runs 64x64 tiles on every CU.
    if (ksplit == 1 && g_tail_split) {
        const long long tn = (N + 127) / 128, tm = (M + 127) / 128, t = tm * tn;
        const long long rem = t % 256;
        if (t > 256 && rem > 0 && rem <= 80) {
            const long long rows_tail = (rem + tn - 1) / tn;                 // tile rows of the partial round
            const long long m1 = (tm - rows_tail) * 128;
            if (m1 > 0 && (tm - rows_tail) * tn >= 256 && rows_tail * tn <= 96) {
                const float* A2 = transA ? A + m1 : A + m1 * (long long)lda;
                g_tail_split = false;                                        // (the two halves are not split again)
                int rc = a4nt_sgemm(transA, transB, (int)m1, N, K, A, lda, B, ldb, C, ldc, bias, beta, 1, 0, stream);
                if (rc == A4NT_OK)
                    rc = a4nt_sgemm(transA, transB, M - (int)m1, N, K, A2, lda, B, ldb, C + m1 * (long long)ldc, ldc, bias, beta, 1, 0, stream);
                g_tail_split = true;
                return rc;
            }
        }
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
    g.part = nullptr; g.nparts = 0; g.colpart = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool big = a4nt_sgemm_tile(M, N, ksplit) == 128;
    if (!transA && !transB) return big ? launch_sgemm<false, false, 2, 2, PFB>(g, ksplit, s) : launch_small<false, false>(g, ksplit, s);
    if (!transA && transB) return big ? launch_sgemm<false, true, 2, 2, PFB>(g, ksplit, s) : launch_small<false, true>(g, ksplit, s);
    return big ? launch_sgemm<true, false, 2, 2, PFB>(g, ksplit, s) : launch_small<true, false>(g, ksplit, s);
}

// Vocabulary projection with the log-softmax statistics in its epilogue (include/a4nt_hip.h).
extern "C" int a4nt_sgemm_nll_parts(int N) { return N > 0 ? 2 * ((N + 127) / 128) : -1; }
extern "C" int a4nt_sgemm_nll(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              const float* bias, float* part, void* stream) {
    if (M <= 0 || N <= 0) return A4NT_OK;
    if (K <= 0 || !A || !B || !C || !part || lda < K || ldb < N || ldc < N) {
        a4nt_set_error("a4nt_sgemm_nll: bad arguments (M=%d N=%d K=%d)", M, N, K);
        return A4NT_ERR_ARG;
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.beta = 0;
    g.kchunk = ((K + BK - 1) / BK) * BK;
    g.slab = 0;
    g.part = part; g.nparts = a4nt_sgemm_nll_parts(N); g.colpart = nullptr;
    constexpr int BM = 128, BN = 128;
    const int tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    hipLaunchKernelGGL((sgemm_kernel<false, false, 2, 2, PFB, 1, true>), dim3(tiles_m * g.tiles_n, 1, 1), dim3(256), 0,
                       (hipStream_t)stream, g);
    return a4nt_check_launch("sgemm_nll");
}

// Weight gradient and bias gradient of a linear layer in one pass (include/a4nt_hip.h): C = A^T B and, from the B tiles
// as they are staged, the column sums of B.
extern "C" int a4nt_sgemm_tn_colsum(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                    int ksplit, long long slab_stride, float* colpart, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !colpart || ksplit < 1 || lda < M || ldb < N ||
        (ksplit > 1 && slab_stride < (long long)M * ldc)) {
        a4nt_set_error("a4nt_sgemm_tn_colsum: bad arguments (M=%d N=%d K=%d ksplit=%d)", M, N, K, ksplit);
        return A4NT_ERR_ARG;
    }
    if (a4nt_sgemm_tile(M, N, ksplit) != 128) {
        a4nt_set_error("a4nt_sgemm_tn_colsum: %d x %d output in %d slabs is a few-tile product (a4nt_sgemm + a4nt_colsum)", M, N, ksplit);
        return A4NT_ERR_ARG;
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.beta = 0;
    int kchunk = (K + ksplit - 1) / ksplit;
    g.kchunk = ((kchunk + BK - 1) / BK) * BK;
    g.slab = slab_stride;
    g.part = nullptr; g.nparts = 0; g.colpart = colpart;
    const int tiles_m = (M + 127) / 128;
    g.tiles_n = (N + 127) / 128;
    hipLaunchKernelGGL((sgemm_kernel<true, false, 2, 2, PFB, 1, false, true>), dim3(tiles_m * g.tiles_n, 1, ksplit), dim3(256), 0,
                       (hipStream_t)stream, g);
    return a4nt_check_launch("sgemm_tn_colsum");
}
