// The free-running decoder's forward loop as ONE persistent launch (include/a4nt_hip.h: a4nt_gen_persist_fwd).
//
// Reference: the generation loop of CharTranslator.forward_advers_gen (models/char_translator.py:375-399): per step a
// single-step LSTM, the output projection, the Gumbel-softmax sample and the done / length bookkeeping.  The per-step
// path (a4nt_gen_loop_fwd) runs that as two launches per step -- lstm_dec_kernel and proj_gumbel_kernel -- and every
// decoder-step launch re-reads its slice of W_hh into registers (100 times per generation; 16 us per launch for
// 1.7 us of matrix work).  Here the same arithmetic (same operand order per element: the results are bit-identical to
// the per-step kernels') runs inside one launch of at most 256 workgroups of 8 waves, one per CU:
//   * waves 4-7 ("L waves") are an LSTM workgroup of the decoder-step kernel at 32 rows per group: unit block j of row
//     pair gp, W_hh slice resident in VGPRs for the WHOLE generation, cell state in registers;
//   * waves 0-3 ("P waves") are the fused projection kernel at two row tiles per wave: wave w owns the 32 rows of row
//     pair 4*rblk + w and the 80 columns of one vocabulary strip, W streamed from the packed image through a register
//     ring whose first blocks are already in flight when h_t arrives; they draw the Philox Gumbel noise of their own
//     tile into an LDS slot of their own while the L waves run their phase (the matrix pipe is idle then).
// Steps are chained by data flow, not by kernel boundaries: per row pair an arrival counter hc (the 32-row h_t is
// published: NW arrivals per step) and pc (the strips of a step's projection are in `part`: nstrips arrivals per
// step).  The recurrent product of step t+1 (which needs only h_t) runs on the L waves UNDER the projection of step
// t; the critical path between two projections is poll -> merge of the strips -> table gather -> cell update ->
// publish.  Hand-off protocol: common.h (sc1 payload stores, every storing wave drains, workgroup barrier, ONE lane
// adds to the counter; consumers poll with sc1 loads and read the payload with sc1 loads only).  Every wave of a
// workgroup executes the same four barriers per step whatever its role.  All waits are bounded (timeout word).
// Measured on MI355X (tools/gen_fwd_time.py): 54.5 us per step against 50.6 for the step launches -- what residency saves
// (the 5 us weight fetch, the launch) the two in-launch hand-offs and the register pressure of two roles in one kernel
// give back -- so this path is opt-in (gen_decode.PERSIST), kept as the measured alternative.
#include "common.h"
#include "gemm_stage.h"
#include "pg_epilogue.h"

// Diagnostic build only (-DGP_STAMP, tools/gen_fwd_time.py): per-phase s_memtime sums of workgroup 0 (P wave 0 and
// L wave 0) in spare words of the sync block that no kernel code reads.  Never defined in the product.
#ifdef GP_STAMP
#define GST_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = 0;
#define GST_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GST(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - st_prev; st_prev = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define GST_FLUSH(off) do { if (bid == 0 && lane == 0 && w == 0) { unsigned long long* d_ = (unsigned long long*)(a.sync + 32 * 16 + (off)); for (int k_ = 0; k_ < 8; ++k_) d_[k_] = st_acc[k_]; } } while (0)
#define GST_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define GST_DECL
#define GST_BEGIN()
#define GST(k)
#define GST_FLUSH(off)
#define GST_WAITV()
#endif

#define GP_PSTR 68
#define GP_DEPTH 4        // 16-k blocks in flight per P wave (two row tiles per block)

struct GenPersistArgs {
    int B, H, V, t0, t1;
    int G, NW, nstrips, nrb;        // row pairs, unit blocks (H/16), vocabulary strips, 128-row blocks
    const float* cg; const float* ptab; const float* whh;
    const long long* tok_in; const float* val_in0;     // token / value that feed step t0 (final in memory)
    const float* c_prev;                               // cell state before step t0 ([B][H]) or null = zeros
    float* hs_all; float* gates; float* cs;            // [n_max+1][B][H], [n_max][B][4H], [n_max][B][H]
    float* hx;                                         // exchange [2][G][NW][2][256]
    const float* Wp; const float* bias;                // packed projection image, bias [V] or null
    float scale, inv_tau; int hard, greedy, philox;
    unsigned long long seed; unsigned step0;
    float* z; int ldz, zrow0, zrows;
    float* part;                                       // [B][nstrips][4]
    float* stats; long long* tok; float* val_in; float* val_out; int* gen_lens; unsigned char* done; long long end_c;
    unsigned* sync;                                    // pc[gp] at 32*gp, hc[gp] at 32*(8+gp), timeout word at 32*16
};

__device__ __forceinline__ void gp_philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
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

// One lane polls, the wave learns the answer (the wave that polled loads only after its poll has matched).
__device__ __forceinline__ bool wave_wait_ge(const unsigned* ctr, unsigned target, unsigned* tmo, int lane) {
    int ok = 1;
    if (lane == 0) ok = wait_ge(ctr, target, tmo) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

struct GpBlock { f32x4 a[2]; f32x4 b[5]; };

template <int NJ>       // H = 64 * NJ
__global__ __launch_bounds__(512) void gen_persist_kernel(GenPersistArgs a) {
    constexpr int MT = 2, RT = 2;
    __shared__ __attribute__((aligned(16))) float sn[4 * RT * 5 * 256];          // Gumbel noise of the four P waves' tiles
    __shared__ __attribute__((aligned(16))) float lpart[4][MT][16 * GP_PSTR];    // K-quarter partial gate tiles
    __shared__ __attribute__((aligned(16))) float hstage[MT][256];
    __shared__ int s_ok;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_l = wv >= 4;
    const int w = wv & 3;
    const int bid = blockIdx.x;
    const int H = a.H, H4 = 4 * a.H;
    const int fr = lane & 15, fq = lane >> 4;
    const int nsteps = a.t1 - a.t0;

    unsigned* tmo = a.sync + 32 * 16;
    const unsigned blk_bytes = 256 * 4;
    const long long hx_par = (long long)a.G * a.NW * MT * 256;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(a.hx, (unsigned)(2 * hx_par * 4));
    __amdgpu_buffer_rsrc_t rpart = make_rsrc(a.part, clamp_bytes((long long)a.B * a.nstrips * 16));

    // ---- projection role of this workgroup (P waves compute it, L waves draw its noise) --------------------------
    // both 128-row blocks of a strip sit 8 apart in block id (one XCD under round-robin placement: speed only)
    const int px = bid & 7, pq = bid >> 3;
    const int strip = (a.nrb == 2 ? (pq >> 1) : pq) * 8 + px;
    const int rblk = a.nrb == 2 ? (pq & 1) : 0;
    const bool p_valid = strip < a.nstrips && (a.nrb == 2 || pq * 8 + px < a.nstrips);
    const int gpw = 4 * rblk + w;                       // row pair of P wave w (and of L wave w's noise tile)
    const int m0 = 32 * gpw, c0 = strip * PG_BC;
    const bool p_active = p_valid && m0 < a.B;

    if (is_l) {
        // ================================ L waves: LSTM unit block j of row pair gpl ================================
        const int ltid = tid & 255;
        const int gpl = bid % a.G, j = bid / a.G;
        const bool l_valid = j < a.NW;
        const int u0 = 16 * j, r0 = 32 * gpl;
        const int crow = ltid >> 4, cunit = ltid & 15;
        unsigned* pc = a.sync + 32 * gpl;
        unsigned* hc = a.sync + 32 * (8 + gpl);

        f32x4 wreg[NJ][4];
        if (l_valid) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
                    wreg[jj][gt] = *reinterpret_cast<const f32x4*>(a.whh + (long long)(gt * H + u0 + fr) * H + w * (H >> 2) + 16 * jj + 4 * fq);
        } else {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) wreg[jj][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float c_reg[MT], cgv[MT][4], valr[MT];
        long long tokr[MT];
        int glen_r[MT]; bool done_r[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
            const bool rv = l_valid && rb < a.B;
            c_reg[mt] = (rv && a.c_prev) ? a.c_prev[(long long)rb * H + u0 + cunit] : 0.f;
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) cgv[mt][gt] = rv ? a.cg[(long long)rb * H4 + gt * H + u0 + cunit] : 0.f;
            tokr[mt] = rv ? a.tok_in[rb] : 0;
            valr[mt] = rv ? a.val_in0[rb] : 0.f;
            glen_r[mt] = (rv && a.gen_lens) ? a.gen_lens[rb] : 0;
            done_r[mt] = (rv && a.done) ? (a.done[rb] != 0) : false;
        }

        // recurrent product of one step: h_{t-1} blocks of the row pair (parity par) x the resident W_hh slice
        auto recurrent = [&](const int par) {
            f32x4 acc[4][MT];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[gt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            constexpr int CHF = NJ < 4 ? NJ : 4;
#pragma unroll
            for (int jc = 0; jc < NJ / CHF; ++jc) {
                f32x4 af[CHF][MT];
#pragma unroll
                for (int j8 = 0; j8 < CHF; ++j8)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const unsigned blk = (unsigned)(((par * a.G + gpl) * a.NW + (w * NJ + jc * CHF + j8)) * MT + mt);
                        af[j8][mt] = ld16_sc1_s(rs, (unsigned)((fr * 16 + fq * 4) * 4), blk * blk_bytes);
                    }
#pragma unroll
                for (int j8 = 0; j8 < CHF; ++j8)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[gt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j8][mt][i], wreg[jc * CHF + j8][gt][i], acc[gt][mt], 0, 0, 0);
            }
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) lpart[w][mt][(4 * fq + i) * GP_PSTR + gt * 16 + fr] = acc[gt][mt][i];
        };

        if (l_valid) recurrent(a.t0 & 1);                 // h_{t0-1}: published by an earlier launch, or zeros

        GST_DECL
        for (int s = 0; s < nsteps; ++s) {
            const int t = a.t0 + s;
            const int par_w = (t & 1) ^ 1;
            GST_BEGIN();
            // the strips of step t-1's projection, all of them, for this row pair
            if (ltid == 0) {
                int ok = 1;
                if (s > 0 && l_valid) ok = wait_ge(pc, (unsigned)(a.nstrips * s), tmo) ? 1 : 0;
                if (ld_flag(tmo) != 0u) ok = 0;
                s_ok = ok;
            }
            __syncthreads();                                                             // barrier 1
            if (!s_ok) break;
            GST(0);
            if (l_valid) {
                if (s > 0) {
                    // merge the strips of the previous step (the arithmetic of proj_gumbel_final_kernel; every workgroup
                    // of the row pair repeats it, the first one writes the step's results)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int rb = r0 + 16 * mt + crow;
                        const unsigned rowoff = (unsigned)(min(rb, a.B - 1) * a.nstrips) * 16u;
                        const int npass = (a.nstrips + 127) >> 7;
                        f32x4 v[8];
                        auto load_pass = [&](int ps) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int sx = 128 * ps + 16 * u + cunit;
                                v[u] = ld16_sc1(rpart, rowoff + (unsigned)min(sx, a.nstrips - 1) * 16u);
                                if (sx >= a.nstrips) { v[u].x = -INFINITY; v[u].y = 0.f; v[u].z = __int_as_float(0x7fffffff); }
                            }
                        };
                        float m = -INFINITY; int mi = 0x7fffffff;
                        for (int ps = npass - 1; ps >= 0; --ps) {
                            load_pass(ps);
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const float om = v[u].x; const int oi = __float_as_int(v[u].z);
                                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
                            }
                        }
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) {
                            const float om = __shfl_xor(m, o); const int oi = __shfl_xor(mi, o);
                            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
                        }
                        float hot = 1.0f, s2 = 1.0f;
                        if (!a.greedy) {
                            float sum = 0.f;
                            for (int ps = 0; ps < npass; ++ps) {
                                if (ps > 0) load_pass(ps);
#pragma unroll
                                for (int u = 0; u < 8; ++u)
                                    if (v[u].x != -INFINITY) sum += v[u].y * expf((v[u].x - m) * a.inv_tau);
                            }
#pragma unroll
                            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                            s2 = sum;
                            const float p_hot = 1.0f / s2;
                            hot = a.hard ? ((1.0f - p_hot) + p_hot) : p_hot;
                        }
                        tokr[mt] = mi; valr[mt] = hot;
                        if (j == 0 && cunit == 0 && rb < a.B) {
                            const long long o = (long long)(t - 1) * a.B + rb;
                            const bool prev_done = done_r[mt];
                            if (a.stats) { a.stats[2 * o] = m; a.stats[2 * o + 1] = s2; }
                            a.tok[o] = mi;
                            if (a.val_in) a.val_in[o] = hot;
                            if (a.val_out) a.val_out[o] = prev_done ? 0.f : hot;
                            glen_r[mt] += prev_done ? 0 : 1;
                            done_r[mt] = prev_done || (long long)mi == a.end_c;
                            if (a.gen_lens) a.gen_lens[rb] = glen_r[mt];
                            if (a.done) a.done[rb] = done_r[mt] ? 1 : 0;
                        }
                    }
                }
                GST(1);
                // input gates: loop-invariant part + scaled table row of the token (clamped for the address only: a
                // token read after a timed-out wait may be anything)
                float o_i[MT], o_f[MT], o_g[MT], o_o[MT], o_c[MT], o_h[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int rb = r0 + 16 * mt + crow;
                    float xg[4] = {0.f, 0.f, 0.f, 0.f};
                    if (rb < a.B) {
                        const long long tk = max(0ll, min(tokr[mt], (long long)a.V - 1));
                        const float* pt = a.ptab + tk * (long long)H4 + u0 + cunit;
                        const float vv = valr[mt];
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt) xg[gt] = cgv[mt][gt] + vv * pt[gt * H];
                    }
                    float pre[4];
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        const int o = crow * GP_PSTR + gt * 16 + cunit;
                        pre[gt] = xg[gt] + ((lpart[0][mt][o] + lpart[1][mt][o]) + (lpart[2][mt][o] + lpart[3][mt][o]));
                    }
                    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
                    const float cn = fg * c_reg[mt] + ig * gg;
                    const float hn = og * tanhf_(cn);
                    c_reg[mt] = cn;
                    o_i[mt] = ig; o_f[mt] = fg; o_g[mt] = gg; o_o[mt] = og; o_c[mt] = cn; o_h[mt] = hn;
                    hstage[mt][crow * 16 + cunit] = hn;
                }
                GST(2);
                __syncthreads();                                                         // barrier 2
                if (w < MT) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&hstage[w][4 * lane]);
                    const unsigned blk = (unsigned)(((par_w * a.G + gpl) * a.NW + j) * MT + w);
                    st16_sc1_s(rs, (unsigned)(lane * 16), blk * blk_bytes, v);
                    drain_vmem();
                }
                __syncthreads();                                                         // barrier 3
                if (ltid == 0) add_flag(hc, 1u);
                GST(3);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int rb = r0 + 16 * mt + crow;
                    if (rb < a.B) {
                        const long long o = (long long)t * a.B + rb;
                        a.hs_all[(o + a.B) * H + u0 + cunit] = o_h[mt];
                        float* pg = a.gates + o * H4 + u0 + cunit;
                        pg[0] = o_i[mt]; pg[H] = o_f[mt]; pg[2 * H] = o_g[mt]; pg[3 * H] = o_o[mt];
                        a.cs[o * H + u0 + cunit] = o_c[mt];
                    }
                }
            } else {
                __syncthreads();                                                         // barrier 2
                __syncthreads();                                                         // barrier 3
            }
            GST(4);
            // the recurrent product of the NEXT step needs only h_t: under the projection of this step
            if (l_valid && s + 1 < nsteps) {
                wave_wait_ge(hc, (unsigned)(a.NW * (s + 1)), tmo, lane);
                GST(5);
                recurrent(par_w);
            }
            GST(6);
            __syncthreads();                                                             // barrier 4: the P waves' strips are drained
            GST(7);
        }
        GST_FLUSH(32);
        return;
    }

    // ==================================== P waves: projection tile (strip, row pair gpw) ==============================
    const int nkb = a.H >> 4;
    __amdgpu_buffer_rsrc_t rb_ = make_rsrc(a.Wp, clamp_bytes((long long)a.nstrips * nkb * 64 * 80));
    const unsigned boff = ((unsigned)strip * nkb * 64u + lane) * 80u;
    unsigned* hcw = a.sync + 32 * (8 + gpw);
    GpBlock ring[GP_DEPTH];
    float bv[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) bv[c] = (p_active && a.bias && c0 + 5 * fr + c < a.V) ? a.bias[c0 + 5 * fr + c] : 0.f;
    PgEpi ep;
    ep.M = a.B; ep.N = a.V; ep.scale = a.scale; ep.inv_tau = a.inv_tau; ep.greedy = a.greedy;
    ep.m0 = m0; ep.c0 = c0; ep.fr = fr; ep.fq = fq;
    ep.z_on = a.z != nullptr; ep.zrow0 = a.zrow0; ep.zrows = a.zrows; ep.ldz = a.ldz;
    ep.rpart = rpart; ep.nstrips = a.nstrips; ep.strip = strip;
    auto load_b = [&](GpBlock& blk, const int kb16, const bool on) {
        const unsigned bo = on ? boff + (unsigned)kb16 * (64u * 80u) : OOB;
#pragma unroll
        for (int v = 0; v < 5; ++v) blk.b[v] = bl4(rb_, on ? bo + 16u * v : OOB);
    };
    auto load_a = [&](GpBlock& blk, const int kb16, const int par, const bool on) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const unsigned hb = (unsigned)(((par * a.G + gpw) * a.NW + kb16) * MT + r);
            blk.a[r] = ld16_sc1_s(rs, on ? (unsigned)((fr * 16 + fq * 4) * 4) : OOB, on ? hb * blk_bytes : 0u);
        }
    };
    if (p_active) {
#pragma unroll
        for (int u = 0; u < GP_DEPTH; ++u) load_b(ring[u], u, u < nkb);
    }
    GST_DECL
    for (int s = 0; s < nsteps; ++s) {
        const int t = a.t0 + s;
        const int par_w = (t & 1) ^ 1;
        GST_BEGIN();
        __syncthreads();                                                                 // barrier 1
        if (!s_ok) break;
        GST(0);
        // Gumbel noise of this step's tile, drawn while the L waves merge / update (the matrix pipe is idle then) into
        // an LDS slot only this wave reads back (no hand-over, no registers held across the k loop)
        if (p_active && a.philox) {
#pragma unroll
            for (int tt = 0; tt < RT * 5; ++tt) {
                const int r = tt / 5, c = tt % 5;
                unsigned o[4];
                gp_philox4x32((unsigned)(c0 + 5 * fr + c), (unsigned)((m0 + 16 * r + 4 * fq) >> 2), a.step0 + (unsigned)t, 0u,
                              (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
                f32x4 gv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (float)(o[e] >> 8) * (1.0f / 16777216.0f);
                    gv[e] = -__logf(-__logf(u + 1e-20f) + 1e-20f);
                }
                *reinterpret_cast<f32x4*>(&sn[((w * RT * 5) + tt) * 256 + 4 * lane]) = gv;
            }
        }
        GST(7);
        __syncthreads();                                                                 // barrier 2
        __syncthreads();                                                                 // barrier 3
        GST(3);
        f32x4 acc[RT][5];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p_active) {
            wave_wait_ge(hcw, (unsigned)(a.NW * (s + 1)), tmo, lane);                    // h_t of this wave's row pair
            GST(1);
#pragma unroll
            for (int u = 0; u < GP_DEPTH; ++u) load_a(ring[u], u, par_w, u < nkb);
            for (int j0 = 0; j0 < nkb; j0 += GP_DEPTH) {
#pragma unroll
                for (int u = 0; u < GP_DEPTH; ++u) {
                    if (j0 + u < nkb) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < RT; ++r)
#pragma unroll
                                for (int c = 0; c < 5; ++c)
                                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u].a[r][i], ring[u].b[(5 * i + c) >> 2][(5 * i + c) & 3], acc[r][c], 0, 0, 0);
                        const bool on = j0 + u + GP_DEPTH < nkb;
                        load_a(ring[u], j0 + u + GP_DEPTH, par_w, on);
                        load_b(ring[u], j0 + u + GP_DEPTH, on);
                    }
                }
            }
        }
        GST(2);
        if (p_active) {
            float gn[RT * 5][4];
#pragma unroll
            for (int tt = 0; tt < RT * 5; ++tt) {
                f32x4 gv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.philox) gv = *reinterpret_cast<const f32x4*>(&sn[((w * RT * 5) + tt) * 256 + 4 * lane]);
#pragma unroll
                for (int e = 0; e < 4; ++e) gn[tt][e] = gv[e];
            }
            ep.rz = make_rsrc(a.z ? a.z + (long long)t * a.zrows * a.ldz : nullptr,
                              a.z ? clamp_bytes(((long long)(a.zrows - 1) * a.ldz + a.V) * 4) : 0u);
            pg_epilogue<RT, true>(ep, acc, bv, gn);
            GST(4);
            drain_vmem();
            GST(5);
        }
        __syncthreads();                                                                 // barrier 4
        GST(6);
        if (tid == 0 && p_valid) {
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4)
                if (32 * (4 * rblk + w4) < a.B) add_flag(a.sync + 32 * (4 * rblk + w4), 1u);
        }
        if (p_active && s + 1 < nsteps) {
#pragma unroll
            for (int u = 0; u < GP_DEPTH; ++u) load_b(ring[u], u, u < nkb);
        }
    }
    GST_FLUSH(8);
}

// ---- host side -------------------------------------------------------------------------------------------------------
static int gp_geometry(int B, int H, int V, int* G, int* NW, int* nstrips, int* nrb, int* grid) {
    if (B <= 0 || B > 256 || V <= 0) return 0;
    if (H != 64 && H != 128 && H != 256 && H != 512) return 0;
    *G = (B + 31) / 32; *NW = H / 16;
    *nstrips = (V + PG_BC - 1) / PG_BC;
    *nrb = B > 128 ? 2 : 1;
    const int gl = *G * *NW, gp = 8 * *nrb * ((*nstrips + 7) / 8);
    *grid = gl > gp ? gl : gp;
    return *grid <= 256;
}

extern "C" int a4nt_gen_persist_ok(int B, int H, int V) {
    int G, NW, ns, nrb, grid;
    return gp_geometry(B, H, V, &G, &NW, &ns, &nrb, &grid);
}
extern "C" long long a4nt_gen_persist_xch_floats(int B, int H) {
    if (B <= 0 || B > 256 || H <= 0 || (H & 15)) return -1;
    return 2ll * ((B + 31) / 32) * (H / 16) * 2 * 256;
}
extern "C" int a4nt_gen_persist_sync_words(void) { return 32 * 16 + 64; }

extern "C" int a4nt_gen_persist_fwd(const a4nt_gen_fwd_args* a, int t0, int t1, float* xch, unsigned* sync, void* stream) {
    if (!a || t0 < 0 || t1 > a->n_max || t0 >= t1 || !a->cg || !a->ptab || !a->tok0 || !a->ones || !a->whh || !a->hs_all ||
        !a->gates || !a->cs || !a->W || !a->part || !a->stats || !a->tok || !a->val_in || !a->val_out || !a->gen_lens ||
        !a->done || !xch || !sync) {
        a4nt_set_error("a4nt_gen_persist_fwd: bad arguments (steps [%d,%d))", t0, t1);
        return A4NT_ERR_ARG;
    }
    GenPersistArgs g;
    int grid;
    if (!gp_geometry(a->B, a->H, a->V, &g.G, &g.NW, &g.nstrips, &g.nrb, &grid)) {
        a4nt_set_error("a4nt_gen_persist_fwd: B=%d H=%d V=%d does not fit one launch (B <= 256, H in 64/128/256/512, "
                       "at most 256 workgroups): use a4nt_gen_loop_fwd", a->B, a->H, a->V);
        return A4NT_ERR_ARG;
    }
    if (a->ldw != 0 || a->U || (reinterpret_cast<uintptr_t>(a->W) & 15) || (reinterpret_cast<uintptr_t>(xch) & 15) ||
        (reinterpret_cast<uintptr_t>(a->part) & 15) || (reinterpret_cast<uintptr_t>(a->whh) & 15) || (!a->greedy && a->tau <= 0.f)) {
        a4nt_set_error("a4nt_gen_persist_fwd: needs the packed projection image (ldw = 0), Philox or greedy sampling (no "
                       "injected uniforms), 16-byte aligned W / xch / part / whh, tau > 0");
        return A4NT_ERR_ARG;
    }
    const long long B = a->B, H = a->H;
    g.B = a->B; g.H = a->H; g.V = a->V; g.t0 = t0; g.t1 = t1;
    g.cg = a->cg; g.ptab = a->ptab; g.whh = a->whh;
    g.tok_in = t0 == 0 ? a->tok0 : a->tok + (t0 - 1) * B;
    g.val_in0 = t0 == 0 ? a->ones : a->val_in + (t0 - 1) * B;
    g.c_prev = t0 == 0 ? nullptr : a->cs + (t0 - 1) * B * H;
    g.hs_all = a->hs_all; g.gates = a->gates; g.cs = a->cs; g.hx = xch;
    g.Wp = a->W; g.bias = a->bias; g.scale = a->scale; g.inv_tau = a->greedy ? 1.f : 1.f / a->tau; g.hard = a->hard;
    g.greedy = a->greedy; g.philox = a->greedy ? 0 : 1; g.seed = a->seed; g.step0 = a->step0;
    g.z = a->z; g.ldz = a->ldz; g.zrow0 = a->zrow0; g.zrows = a->z ? a->zrows : 0;
    g.part = a->part; g.stats = a->stats; g.tok = a->tok; g.val_in = a->val_in; g.val_out = a->val_out;
    g.gen_lens = a->gen_lens; g.done = a->done; g.end_c = a->end_c; g.sync = sync;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, (size_t)(32 * 16) * 4, s) != hipSuccess) return a4nt_check_launch("gen_persist memset");
    switch (a->H) {
        case 64: hipLaunchKernelGGL((gen_persist_kernel<1>), dim3(grid), dim3(512), 0, s, g); break;
        case 128: hipLaunchKernelGGL((gen_persist_kernel<2>), dim3(grid), dim3(512), 0, s, g); break;
        case 256: hipLaunchKernelGGL((gen_persist_kernel<4>), dim3(grid), dim3(512), 0, s, g); break;
        default: hipLaunchKernelGGL((gen_persist_kernel<8>), dim3(grid), dim3(512), 0, s, g); break;
    }
    int rc = a4nt_check_launch("gen_persist");
    if (rc != A4NT_OK) return rc;
    // the last step of the range is finalised by a launch of its own (as in a4nt_gen_loop_fwd)
    const int t = t1 - 1;
    return a4nt_proj_gumbel_finalize(a->B, a->V, a->part, a->tau, a->hard, a->greedy, a->stats + t * B * 2, a->tok + t * B,
                                     a->val_in + t * B, a->val_out + t * B, a->gen_lens, a->done, a->end_c, stream);
}
