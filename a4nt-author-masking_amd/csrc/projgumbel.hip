// Fused output projection + Gumbel-softmax sampling for the free-running decoder
// (reference models/char_translator.py:377-392: dec_out = p_rnn.mm(W)+b; gumbel_softmax_sample;
//  argmax; done / gen_lens bookkeeping).
//
// One decoder step multiplies only B <= 128 rows by the [H,V] projection: 8 x 626 MFMA tiles of 16x16
// for 1024 SIMDs.  LDS-staged tilings (the generic 64x64 / 128x128 kernels, and three LDS versions of
// this one: 128x48 and 32x64 tiles with 32-deep k-tiles, 64x80 with 128-deep chunks) all took 27-33 us
// for the 1.3 GFLOP whatever the tile shape or occupancy: with so few rows every wave's critical path
// is load -> LDS -> barrier -> ds_read -> MFMA -> epilogue, and nothing overlaps it.
// This version has no LDS and no barrier.  Every WAVE owns a 16-row x 80-column output tile
// (8 x 126 = 1008 waves, one per SIMD: 98 % of the 1024 SIMDs carry exactly five MFMA tiles) and feeds
// v_mfma_f32_16x16x4_f32 straight from registers:
//   A (h rows):   lane (row m, quad q) loads h[m][16j+4q .. +3] with one dwordx4; register i is the A operand
//                 of k-step i, whose k set is therefore {16j+4q+i}: a permutation of the block's 16 k's;
//   B (W [K][N]): for that k (= 16j+4q+i) lane (n, q) loads W[k][c0+5n .. 5n+4] (dwordx4 + dword; the 16 lanes
//                 of a quad cover 320 contiguous bytes); register c is the B operand of column tile c, i.e.
//                 column tile c holds the columns {c0 + 5n + c}: a strided column set, which the epilogue
//                 simply accounts for.
// Loads run eight 16-k blocks ahead in a register ring (the W stream comes from HBM / Infinity Cache).
// The sampling is finished in the epilogue:
//     z = scale*(h W + b) + gumbel(u)                      (greedy: z = h W + b)
// softmax((log_softmax(x)+g)/tau) == softmax(z/tau) (shift invariance), so the row statistics the
// sampler needs are max z, arg max z and sum exp((z - max)/tau).  Each wave emits those three per row
// for its 80-column strip; a tiny finalize kernel merges the strips (online-softmax merge), derives the
// straight-through value and updates done / gen_lens.  z is stored once (saved for the backward pass);
// the logits never make a round trip through HBM.
#include "gemm_stage.h"
#include "pg_epilogue.h"

#define PG_DEPTH 8        // 16-k blocks in flight per wave

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
    int zrow0, zrows;       // z holds rows [zrow0, zrow0+zrows) only (the rows whose backward will run)
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

// PACKED: W comes as the image a4nt_proj_gumbel_pack writes once per generation: for every (strip, 16-k block) the
// 20 B operands of each lane (4 k-steps x 5 column tiles) are contiguous, so a block is 5 aligned dwordx4 per lane
// and 5 KB contiguous per wave, instead of 4 x (unaligned dwordx4 + dword) gathered at a 20-byte lane stride from
// rows whose stride (V = 10001 floats) is not a multiple of 16 bytes -- the address pipe, not the matrix pipe, was
// pacing the kernel.
template <int RT, bool PACKED>
struct PgBlock {            // operands of one 16-k block
    f32x4 a[RT];            // A: k-steps 0..3 of each of this wave's RT row tiles
    f32x4 b[5];             // B: element 5*i + c = k-step i, column tile c (unpacked: b[i] = tiles 0..3, b[4][i] = tile 4)
};

template <int RT, bool PACKED>
__device__ __forceinline__ void pg_load(PgBlock<RT, PACKED>& blk, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                        const unsigned (&aoff)[RT], unsigned boff, unsigned ldw_bytes, int kb, bool on) {
    // kb: first k of the block; `on` false (past the end of K) loads nothing that is used
#pragma unroll
    for (int r = 0; r < RT; ++r) blk.a[r] = bl4(ra, on ? aoff[r] + (unsigned)kb * 4u : OOB);
    if (PACKED) {
        // boff = byte offset of this lane inside block 0 of its strip; blocks are 64 lanes x 80 bytes apart
        const unsigned bo = on ? boff + (unsigned)(kb >> 4) * (64u * 80u) : OOB;
#pragma unroll
        for (int v = 0; v < 5; ++v) blk.b[v] = bl4(rb, on ? bo + 16u * v : OOB);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned bo = on ? boff + (unsigned)(kb + i) * ldw_bytes : OOB;
            blk.b[i] = bl4(rb, bo);
            blk.b[4][i] = bl1(rb, on ? bo + 16u : OOB);
        }
    }
}

template <int RT, bool PACKED>
__device__ __forceinline__ void pg_mma(const PgBlock<RT, PACKED>& blk, f32x4 (&acc)[RT][5]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float bv = PACKED ? blk.b[(5 * i + c) >> 2][(5 * i + c) & 3] : (c < 4 ? blk.b[i][c] : blk.b[4][i]);
                acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(blk.a[r][i], bv, acc[r][c], 0, 0, 0);
            }
}

// RT = row tiles per wave: 1 up to 128 rows (one wave per SIMD); 2 above (256 rows: again 1008 waves, and
// a B fragment now feeds two MFMAs, which keeps the CU's address pipe -- 4 waves x 9 loads per 16-k block --
// below the matrix pipe's time; two waves per SIMD with RT = 1 measured 47 us against 2 x 25 us).
// Also measured: two k-groups per workgroup (8 waves, each set of 4 on half of K, summed through LDS; same loads
// and MFMAs per CU, two waves per SIMD): 47.2 against 47.9 us at 256 rows, 32.0 against 29.9 at 128 -- latency is
// not what bounds this kernel, the per-lane unaligned B loads are.
// NW = 1: a second set of 4 waves ("noise waves") draws the Philox Gumbel noise of the four MFMA waves' tiles while
// those multiply, and hands it over through LDS: with one wave per SIMD the noise (10 Philox calls and 80
// logarithms per lane at RT = 2) ran after the k loop and cost 6.8 us of a 43 us launch; placed inside the k loop
// of the same wave it cost the same (hipcc keeps it a block of its own between the MFMA blocks).  On a second wave
// of the SIMD the hardware overlaps it with the MFMAs by itself.
template <int RT, bool PACKED, int NW>
__global__ __launch_bounds__(256 * (1 + NW)) void proj_gumbel_kernel(ProjGumbelArgs a) {
    constexpr int BM = 64 * RT;                              // rows per workgroup: wave w owns rows [16*RT*w, +16*RT)
    // 1-D grid; the row blocks of one strip sit 8 apart in block id (id % 8 = strip % 8), so under round-robin
    // placement they land on the same XCD and share the strip of W in that XCD's L2 (speed only)
    const int nrb = (a.M + BM - 1) / BM;
    const int grp = blockIdx.x / (8 * nrb), rem = blockIdx.x % (8 * nrb);
    const int strip = grp * 8 + (rem & 7), rblk = rem >> 3;
    const int lane = threadIdx.x & 63;
    const int w = NW ? (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3) : (int)(threadIdx.x >> 6);
    const bool noise_wave = NW && __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = rblk * BM + 16 * RT * w, c0 = strip * PG_BC;
    if (strip >= a.nstrips) return;                          // workgroup-uniform
    const bool active = m0 < a.M;                            // wave-uniform
    if (!NW && !active) return;                              // (with noise waves every wave reaches the barrier)
    __shared__ __attribute__((aligned(16))) float sn[NW ? 4 * RT * 5 * 256 : 4];
    if (noise_wave) {
        if (active) {
#pragma unroll
            for (int t = 0; t < RT * 5; ++t) {
                const int r = t / 5, c = t % 5;
                unsigned o[4];        // one Philox4x32-10 call = the 4 rows (rbase..rbase+3) of this column
                philox4x32((unsigned)(c0 + 5 * fr + c), (unsigned)((m0 + 16 * r + 4 * fq) >> 2), a.step, 0u,
                           (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
                f32x4 gv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (float)(o[e] >> 8) * (1.0f / 16777216.0f);
                    gv[e] = -__logf(-__logf(u + 1e-20f) + 1e-20f);
                }
                *reinterpret_cast<f32x4*>(&sn[((w * RT * 5) + t) * 256 + 4 * lane]) = gv;
            }
        }
        __syncthreads();
        return;
    }

    // A: rows m0+16r+fr (clamped), k offset 4*fq inside a block.  B: columns c0 + 5*fr .. +4, k row 4*fq + i.
    __amdgpu_buffer_rsrc_t ra = make_rsrc(a.h, clamp_bytes(((long long)(a.M - 1) * a.ldh + a.K) * 4));
    const int nkb = a.K >> 4;
    __amdgpu_buffer_rsrc_t rb = PACKED ? make_rsrc(a.W, clamp_bytes((long long)a.nstrips * nkb * 64 * 80))
                                       : make_rsrc(a.W, clamp_bytes(((long long)(a.K - 1) * a.ldw + a.N) * 4));
    unsigned aoff[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) aoff[r] = (unsigned)(min(m0 + 16 * r + fr, a.M - 1) * a.ldh + 4 * fq) * 4u;
    const unsigned ldw_bytes = (unsigned)a.ldw * 4u;
    const unsigned boff = PACKED ? ((unsigned)strip * nkb * 64u + lane) * 80u
                                 : (unsigned)(4 * fq) * ldw_bytes + (unsigned)(c0 + 5 * fr) * 4u;

    f32x4 acc[RT][5];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float bv[5];                                             // bias of this lane's 5 columns: in flight under the k loop
#pragma unroll
    for (int c = 0; c < 5; ++c) bv[c] = (a.bias && c0 + 5 * fr + c < a.N) ? a.bias[c0 + 5 * fr + c] : 0.f;
    const bool philox = !a.greedy && !a.U;
    constexpr int DEPTH = RT > 1 ? PG_DEPTH / 2 : PG_DEPTH;      // RT = 2: twice the MFMAs per block, half the ring
    PgBlock<RT, PACKED> ring[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) pg_load(ring[u], ra, rb, aoff, boff, ldw_bytes, 16 * u, u < nkb);
    for (int j0 = 0; j0 < nkb && active; j0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (j0 + u < nkb) {                              // wave-uniform
                pg_mma(ring[u], acc);
                pg_load(ring[u], ra, rb, aoff, boff, ldw_bytes, 16 * (j0 + u + DEPTH), j0 + u + DEPTH < nkb);
            }
        }
    }
    // Gumbel noise: from the noise waves (LDS), or drawn here
    float gn[RT * 5][4];
    if (NW) {
        __syncthreads();
        if (!active) return;
#pragma unroll
        for (int t = 0; t < RT * 5; ++t) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(&sn[((w * RT * 5) + t) * 256 + 4 * lane]);
#pragma unroll
            for (int e = 0; e < 4; ++e) gn[t][e] = gv[e];
        }
    } else {
#pragma unroll
        for (int t = 0; t < RT * 5; ++t) {
            const int r = t / 5, c = t % 5;
#pragma unroll
            for (int e = 0; e < 4; ++e) gn[t][e] = 0.f;
            if (philox) {
                unsigned o[4];
                philox4x32((unsigned)(c0 + 5 * fr + c), (unsigned)((m0 + 16 * r + 4 * fq) >> 2), a.step, 0u,
                           (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (float)(o[e] >> 8) * (1.0f / 16777216.0f);
                    gn[t][e] = -__logf(-__logf(u + 1e-20f) + 1e-20f);
                }
            }
        }
    }

    // injected uniforms (parity tests): the same transform, from memory
    if (!a.greedy && a.U) {
#pragma unroll
        for (int t = 0; t < RT * 5; ++t) {
            const int r = t / 5, c = t % 5;
            const int col = c0 + 5 * fr + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + 16 * r + 4 * fq + e;
                const float u = (row < a.M && col < a.N) ? a.U[(long long)row * a.ldu + col] : 0.5f;
                gn[t][e] = -logf(-logf(u + 1e-20f) + 1e-20f);
            }
        }
    }
    PgEpi ep;
    ep.M = a.M; ep.N = a.N; ep.scale = a.scale; ep.inv_tau = a.inv_tau; ep.greedy = a.greedy;
    ep.m0 = m0; ep.c0 = c0; ep.fr = fr; ep.fq = fq;
    ep.z_on = a.z != nullptr; ep.zrow0 = a.zrow0; ep.zrows = a.zrows; ep.ldz = a.ldz;
    ep.rz = make_rsrc(a.z, a.z ? clamp_bytes(((long long)(a.zrows - 1) * a.ldz + a.N) * 4) : 0u);
    ep.rpart = make_rsrc(a.part, clamp_bytes((long long)a.M * a.nstrips * 16));
    ep.nstrips = a.nstrips; ep.strip = strip;
    pg_epilogue<RT, false>(ep, acc, bv, gn);
}

// Merge the strips of one row (online-softmax merge), then the sampler's bookkeeping (PgFinalArgs: common.h; the
// decoder step kernel of lstm.hip does the same merge in its prologue when the finalisation is deferred to it).
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

// Packed image of W [K][N] for proj_gumbel_kernel<.., true>: float ((strip*nkb + kb)*64 + lane)*20 + 5*i + c
// = W[16*kb + 4*(lane>>4) + i][80*strip + 5*(lane&15) + c] (0 past the last column).
__global__ __launch_bounds__(64) void proj_gumbel_pack_kernel(int K, int N, const float* W, int ldw, float* Wp, int nkb) {
    const int strip = blockIdx.x / nkb, kb = blockIdx.x % nkb, lane = threadIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    float* dst = Wp + ((long long)blockIdx.x * 64 + lane) * 20;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int k = 16 * kb + 4 * fq + i, col = PG_BC * strip + 5 * fr + c;
            dst[5 * i + c] = (k < K && col < N) ? W[(long long)k * ldw + col] : 0.f;
        }
}
extern "C" long long a4nt_proj_gumbel_packed_floats(int K, int N) {
    if (K <= 0 || N <= 0 || (K & 15)) return 0;
    return (long long)((N + PG_BC - 1) / PG_BC) * (K >> 4) * 64 * 20;
}
extern "C" int a4nt_proj_gumbel_pack(int K, int N, const float* W, int ldw, float* Wp, void* stream) {
    if (K <= 0 || N <= 0 || (K & 15) || !W || !Wp || ldw < N || (reinterpret_cast<uintptr_t>(Wp) & 15)) {
        a4nt_set_error("a4nt_proj_gumbel_pack: bad arguments (K=%d must be a multiple of 16, N=%d, ldw=%d)", K, N, ldw);
        return A4NT_ERR_ARG;
    }
    const int nkb = K >> 4, nstrips = (N + PG_BC - 1) / PG_BC;
    hipLaunchKernelGGL(proj_gumbel_pack_kernel, dim3(nstrips * nkb), dim3(64), 0, (hipStream_t)stream, K, N, W, ldw, Wp, nkb);
    return a4nt_check_launch("proj_gumbel_pack");
}

extern "C" long long a4nt_proj_gumbel_part_floats(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (long long)M * ((N + PG_BC - 1) / PG_BC) * 4;
}

extern "C" int a4nt_proj_gumbel_fwd(int M, int N, int K, const float* h, int ldh, const float* W, int ldw,
                                    const float* bias, float scale, float tau, int hard, int greedy, const float* U,
                                    int ldu, unsigned long long seed, unsigned step, float* z, int ldz, int zrow0,
                                    int zrows, float* part,
                                    float* stats, long long* tok, float* val_in, float* val_out, int* gen_lens,
                                    unsigned char* done, long long end_c, void* stream) {
    if (M <= 0) return A4NT_OK;
    if (N <= 0 || K <= 0 || (K & 15) || !h || !W || !part || (!greedy && tau <= 0.f) ||
        (reinterpret_cast<uintptr_t>(h) & 15) || (ldh & 3)) {
        a4nt_set_error("a4nt_proj_gumbel_fwd: bad arguments (N=%d, K=%d must be a multiple of 16, tau=%g, h 16-byte "
                       "aligned with ldh a multiple of 4)", N, K, (double)tau);
        return A4NT_ERR_ARG;
    }
    ProjGumbelArgs a;
    a.h = h; a.ldh = ldh; a.W = W; a.ldw = ldw; a.bias = bias; a.M = M; a.N = N; a.K = K;
    a.scale = scale; a.inv_tau = greedy ? 1.f : 1.f / tau; a.greedy = greedy; a.U = U; a.ldu = ldu; a.seed = seed;
    a.step = step; a.z = z; a.ldz = ldz; a.zrow0 = zrow0; a.zrows = zrows; a.part = part; a.nstrips = (N + PG_BC - 1) / PG_BC;
    const int rt = M > 128 ? 2 : 1;
    const int nrb = (M + 64 * rt - 1) / (64 * rt);
    dim3 grid(((a.nstrips + 7) / 8) * 8 * nrb), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool packed = ldw == 0;                   // W is the image written by a4nt_proj_gumbel_pack
    if (packed && (reinterpret_cast<uintptr_t>(W) & 15)) { a4nt_set_error("a4nt_proj_gumbel_fwd: packed W must be 16-byte aligned"); return A4NT_ERR_ARG; }
    const bool nw = !greedy && !U;                  // Philox noise: drawn by a second set of waves
    if (nw) block = dim3(512);
#define PG_LAUNCH(RT_, PK_, NW_) hipLaunchKernelGGL((proj_gumbel_kernel<RT_, PK_, NW_>), grid, block, 0, s, a)
    if (rt == 1) { if (packed) { if (nw) PG_LAUNCH(1, true, 1); else PG_LAUNCH(1, true, 0); } else { if (nw) PG_LAUNCH(1, false, 1); else PG_LAUNCH(1, false, 0); } }
    else         { if (packed) { if (nw) PG_LAUNCH(2, true, 1); else PG_LAUNCH(2, true, 0); } else { if (nw) PG_LAUNCH(2, false, 1); else PG_LAUNCH(2, false, 0); } }
#undef PG_LAUNCH
    int rc = a4nt_check_launch("proj_gumbel");
    if (rc != A4NT_OK || !tok) return rc;            // tok == null: the strips stay in `part` for a later finalisation
    return a4nt_proj_gumbel_finalize(M, N, part, tau, hard, greedy, stats, tok, val_in, val_out, gen_lens, done, end_c, stream);
}

extern "C" int a4nt_proj_gumbel_finalize(int M, int N, const float* part, float tau, int hard, int greedy, float* stats,
                                         long long* tok, float* val_in, float* val_out, int* gen_lens, unsigned char* done,
                                         long long end_c, void* stream) {
    if (M <= 0) return A4NT_OK;
    if (N <= 0 || !part || !tok || (!greedy && tau <= 0.f)) {
        a4nt_set_error("a4nt_proj_gumbel_finalize: bad arguments (N=%d, tau=%g)", N, (double)tau);
        return A4NT_ERR_ARG;
    }
    PgFinalArgs f;
    f.part = part; f.nstrips = (N + PG_BC - 1) / PG_BC; f.inv_tau = greedy ? 1.f : 1.f / tau; f.greedy = greedy; f.hard = hard;
    f.stats = stats; f.tok = tok; f.val_in = val_in; f.val_out = val_out; f.gen_lens = gen_lens; f.done = done; f.end_c = end_c;
    hipLaunchKernelGGL(proj_gumbel_final_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, f);
    return a4nt_check_launch("proj_gumbel_final");
}
