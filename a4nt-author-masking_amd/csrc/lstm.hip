// Persistent LSTM sweeps for gfx950: recurrent weights resident in VGPRs across all timesteps,
// recurrent gate contraction on v_mfma_f32_16x16x4_f32 (exact f32), gate non-linearity / cell
// update fused behind it, hidden state exchanged between workgroups through L2 with write-through
// stores and an agent-scope arrival counter (common.h).
//
// Decomposition (forward).  The batch is cut into G independent row groups of R = 16*MT rows;
// group g is served by NW = H/16 workgroups, workgroup j owning hidden units [16j, 16j+16), i.e.
// the 64 gate columns {gate*H + 16j + n}.  Inside a workgroup wave w owns the K quarter
// [w*H/4, (w+1)*H/4) of the contraction and keeps W_hh[those 64 rows][that quarter] as MFMA
// B operands in registers (H/4 VGPRs: 128 at H = 512).  Per step each wave pulls its quarter of
// h_{t-1} for the group's rows straight into MFMA A fragments (1 KiB coalesced sc1 loads of the
// 16x16 blocks the owning workgroups published), multiplies, and the four partial 16x64 gate
// tiles are summed through LDS by the cell-update threads (one (row, unit) pair each), which add
// the precomputed input gates, apply sigmoid/tanh, update c (kept in a register for the whole
// sweep), emit h_t / the saved activations, and publish the workgroup's 16x16 h block for the
// next step.  Only the NW workgroups of one group synchronise with each other; groups never do.
//
// Backward: workgroup j turns the dgates of its own 16 units into a partial dh_{t-1} over all H
// columns with its 64 rows of W_hh, and the partials are reduce-scattered between the workgroups
// (see lstm_bwd_kernel).
//
// Sequences of different lengths: a row simply freezes (state kept, output zero, gradient zero)
// once t >= len[row]; no sorting or packing is needed by the kernels.
#include "common.h"
#include <atomic>

// Diagnostic build only (-DA4NT_STAMP, tools/lstm_stamp.py): per-phase s_memtime sums of workgroup 0,
// written to spare words of the sync block that no kernel code reads.  Never defined in the product.
#ifdef A4NT_STAMP
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = 0;
#define STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - st_prev; st_prev = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH(sync, G) do { if (bid == 0 && threadIdx.x == 0) { unsigned long long* d_ = (unsigned long long*)((sync) + 32 * (G) + 8); for (int k_ = 0; k_ < 8; ++k_) d_[k_] = st_acc[k_]; } } while (0)
#define STAMP_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define STAMP_DECL
#define STAMP_BEGIN()
#define STAMP(k)
#define STAMP_FLUSH(sync, G)
#define STAMP_WAITV()
#endif

#define PSTR 68   // LDS row stride (floats) of the partial gate tiles: conflict-free MFMA-layout stores

// Block index -> (row group g, workgroup j of the group): g = bid % G, so the workgroups of a group share
// blockIdx % 8 (one XCD under round-robin placement: speed only).  Measured (tools/micro/census.hip): with two
// workgroups per CU, blocks b and b + 256 share a CU, i.e. two workgroups of the SAME group at G = 16; giving the
// second one another group instead (and a half-step head start) was tried and is slower (bwd 10.1 -> 11.7 us/step).
__device__ __forceinline__ void decode_bid(const int bid, const int G, int& g, int& j) {
    g = bid % G;
    j = bid / G;
}

// Block -> (problem, block inside the problem) of a two-sweep launch.  The row groups of the two problems are
// interleaved as ONE list of Ga + Gb groups (block b: combined group b % (Ga + Gb), unit block b / (Ga + Gb)), exactly
// the map of a single sweep over all those groups: with two workgroups per CU the blocks b and b + 256 that share a CU
// then belong to the SAME row group of the same problem whenever Ga + Gb divides 256, as in a single 512-workgroup
// sweep.  (Problem a on the first nA blocks and problem b on the rest put one workgroup of each problem on every CU:
// measured slower for the backward sweep, 8.1 against 7.2 us per step at 2 x 128 rows, like every other mix of row
// groups on a CU.)
__device__ __forceinline__ bool pair_block(const int bid, const int Ga, const int Gb, int& local) {
    const int gc = bid % (Ga + Gb), jc = bid / (Ga + Gb);
    const bool first = gc < Ga;
    local = first ? jc * Ga + gc : jc * Gb + (gc - Ga);
    return first;
}

struct LstmFwdArgs {
    int T, B, H, G, NW;
    const float* gx;        // [T][B][4H] input-gate pre-activations (biases included)
    const float* whh;       // [4H][H]
    const float* h0;        // [B][H] or null
    const float* c0;        // [B][H] or null
    const int* lens;        // [B] or null
    const float* cg;        // decode: [B][4H] loop-invariant part of the input gates
    const float* ptab;      // decode: [V][4H] per-token input-gate table (E * W_ie^T)
    const long long* tok;   // decode: [B]
    const float* val;       // decode: [B] straight-through sample value (scales the table row)
    float* hs;              // [T][B][H]
    float* gates;           // [T][B][4H] post-activation gates (may alias gx) or null
    float* cs;              // [T][B][H] or null
    float* hN; float* cN;   // [B][H] or null
    float* hx;              // exchange [2][G][NW][MT][256]
    unsigned* sync;         // [G] counters, 32 words apart
    unsigned* tmo;          // sticky timeout word (sync[32*G] for caller-provided sync blocks)
    int mode;               // 0: persistent sequence sweep, 1: single decode step
    int step0;              // decode: absolute step index (selects the exchange parity)
    // decode, optional: the previous step's sampler is finished here instead of by proj_gumbel_final_kernel -- tok / val
    // are then outputs (of step step0 - 1), written by the first workgroup of every row group
    PgFinalArgs fin;        // fin.part == null: tok / val above are inputs
};

template <int MT, int NJ, bool DEC = false>
__device__ __forceinline__ void lstm_fwd_body(const LstmFwdArgs& a, const int bid) {
    __shared__ __attribute__((aligned(16))) float part[4][MT][16 * PSTR];
    __shared__ __attribute__((aligned(16))) float hstage[MT][256];
    __shared__ int s_ok;

    int g, j;
    if (DEC && (a.NW & 7) == 0) {
        // A single decoder step has no hand-off inside the launch (h_{t-1} was published by the previous launch), so
        // its placement is free: XCD x (= bid % 8) takes the unit blocks j in [x*NW/8, (x+1)*NW/8) of EVERY row group.
        // Each XCD then reads one eighth of W_hh per step, and the same eighth at every step, instead of all of it
        // for its own row group (8 x 4 MB per launch at H = 512: most of the launch was that fetch).
        const int per = a.NW >> 3, x = bid & 7, q = bid >> 3;
        j = x * per + q % per;
        g = q / per;
    } else {
        decode_bid(bid, a.G, g, j);
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H = a.H, H4 = 4 * a.H;
    constexpr int nj = NJ;                 // 16-wide k blocks in this wave's K quarter (H = 64*NJ)
    const int u0 = 16 * j;
    const int r0 = g * 16 * MT;
    const int fr = lane & 15, fq = lane >> 4;      // MFMA fragment row/col and k-quad
    const int crow = tid >> 4, cunit = tid & 15;   // cell-update (row, unit) of this thread

    // decode with a deferred sampler: merge the vocabulary strips of the previous step's projection for this
    // workgroup's rows (16 lanes per row, 8 strips per lane and pass, all loads of a pass in flight together; the
    // arithmetic of proj_gumbel_final_kernel, which every workgroup of the row group repeats instead of waiting for
    // one more launch).  Placed ahead of the weight loads so that both latencies overlap.
    long long tokr[MT];
    float valr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { tokr[mt] = 0; valr[mt] = 0.f; }
    if (DEC) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
            if (a.fin.part) {
                const PgFinalArgs& f = a.fin;
                const f32x4* p = reinterpret_cast<const f32x4*>(f.part) + (long long)min(rb, a.B - 1) * f.nstrips;
                const int npass = (f.nstrips + 127) >> 7;
                f32x4 v[8];
                auto load_pass = [&](int ps) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int s = 128 * ps + 16 * u + cunit;
                        v[u] = p[min(s, f.nstrips - 1)];
                        if (s >= f.nstrips) { v[u].x = -INFINITY; v[u].y = 0.f; v[u].z = __int_as_float(0x7fffffff); }
                    }
                };
                float m = -INFINITY; int mi = 0x7fffffff;
                for (int ps = npass - 1; ps >= 0; --ps) {           // pass 0 last: its strips stay in registers
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
                if (!f.greedy) {
                    float sum = 0.f;
                    for (int ps = 0; ps < npass; ++ps) {
                        if (ps > 0) load_pass(ps);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (v[u].x != -INFINITY) sum += v[u].y * expf((v[u].x - m) * f.inv_tau);
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    s2 = sum;
                    const float p_hot = 1.0f / s2;
                    hot = f.hard ? ((1.0f - p_hot) + p_hot) : p_hot;
                }
                tokr[mt] = mi; valr[mt] = hot;
                if (j == 0 && cunit == 0 && rb < a.B) {
                    const bool prev_done = f.done ? (f.done[rb] != 0) : false;
                    if (f.stats) { f.stats[2 * rb] = m; f.stats[2 * rb + 1] = s2; }
                    f.tok[rb] = mi;
                    if (f.val_in) f.val_in[rb] = hot;
                    if (f.val_out) f.val_out[rb] = prev_done ? 0.f : hot;
                    if (f.gen_lens) f.gen_lens[rb] += prev_done ? 0 : 1;
                    if (f.done) f.done[rb] = (prev_done || (long long)mi == f.end_c) ? 1 : 0;
                }
            } else if (rb < a.B) {
                tokr[mt] = a.tok[rb]; valr[mt] = a.val[rb];
            }
        }
    }

    // ---- recurrent weights -> registers --------------------------------------------------
    f32x4 wreg[NJ][4];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
            wreg[jj][gt] = *reinterpret_cast<const f32x4*>(a.whh + (long long)(gt * H + u0 + fr) * H + w * (H >> 2) + 16 * jj + 4 * fq);

    unsigned* ctr = a.sync + 32 * g;
    unsigned* tmo = a.tmo;
    const unsigned blk_bytes = 256 * 4;
    const long long hx_par = (long long)a.G * a.NW * MT * 256;   // floats per parity buffer
    __amdgpu_buffer_rsrc_t rs = make_rsrc(a.hx, (unsigned)(2 * hx_par * 4));

    // ---- per-thread recurrent state --------------------------------------------------------
    float c_reg[MT], h_reg[MT];
    int len_r[MT];
    int Tg = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
        const bool rv = rb < a.B;
        c_reg[mt] = (rv && a.c0) ? a.c0[(long long)rb * H + u0 + cunit] : 0.f;
        h_reg[mt] = (rv && a.h0) ? a.h0[(long long)rb * H + u0 + cunit] : 0.f;
        len_r[mt] = rv ? (a.lens ? min(a.lens[rb], a.T) : a.T) : 0;
    }
    if (a.lens) {
        for (int r = 0; r < 16 * MT; ++r) {
            const int rb = r0 + r;
            if (rb < a.B) Tg = max(Tg, min(a.lens[rb], a.T));
        }
    } else {
        Tg = a.T;
    }

    // optional loop-invariant input-gate term (sequence mode): gates_x[t][b] = gx[t][b] + cg[b]
    float cgv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
            cgv[mt][gt] = (!DEC && a.cg && rb < a.B) ? a.cg[(long long)rb * H4 + gt * H + u0 + cunit] : 0.f;
    }

    int par_r = 0;
    if (!DEC) {
        // publish h0 as "step -1" into parity 0
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hstage[mt][crow * 16 + cunit] = h_reg[mt];
        __syncthreads();
        if (w < MT) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&hstage[w][4 * lane]);
            const long long blk = ((long long)(0 * a.G + g) * a.NW + j) * MT + w;
            st16_sc1(rs, (unsigned)(blk * blk_bytes + lane * 16), v);
            drain_vmem();
        }
        __syncthreads();
        if (tid == 0) add_flag(ctr, 1u);
    } else {
        par_r = a.step0 & 1;
    }

    STAMP_DECL
    for (int t = 0; t < Tg; ++t) {
        const int par_w = par_r ^ 1;
        STAMP_BEGIN();
        // (a) input gates of this step for the cell-update threads (independent of h: issue early)
        float xg[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) xg[mt][gt] = 0.f;
            if (rb < a.B) {
                if (!DEC) {
                    const float* p = a.gx + ((long long)t * a.B + rb) * H4 + u0 + cunit;
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) xg[mt][gt] = p[gt * H] + cgv[mt][gt];
                } else {
                    const float* pc = a.cg + (long long)rb * H4 + u0 + cunit;
                    const float* pt = a.ptab + tokr[mt] * (long long)H4 + u0 + cunit;
                    const float vv = valr[mt];
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) xg[mt][gt] = pc[gt * H] + vv * pt[gt * H];
                }
            }
        }
        // (b) wait until every workgroup of the group has published h_{t-1}
        if (!DEC) {
            if (tid == 0) s_ok = wait_ge(ctr, (unsigned)(a.NW * (t + 1)), tmo) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;
        }
        STAMP(0);
        // (c)+(d) A fragments straight from the exchange buffer, recurrent MFMA
        f32x4 acc[4][MT];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[gt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // A blocks are pulled CHF at a time so that H = 1024 (16 blocks per wave, 256 weight registers) still fits
        constexpr int CHF = (NJ * MT > 16) ? (16 / MT) : NJ;
#pragma unroll
        for (int jc = 0; jc < NJ / CHF; ++jc) {
            f32x4 af[CHF][MT];
#pragma unroll
            for (int j8 = 0; j8 < CHF; ++j8)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const long long blk = ((long long)(par_r * a.G + g) * a.NW + (w * nj + jc * CHF + j8)) * MT + mt;
                    af[j8][mt] = ld16_sc1(rs, (unsigned)(blk * blk_bytes + (fr * 16 + fq * 4) * 4));
                }
            if (jc == NJ / CHF - 1) { STAMP_WAITV(); STAMP(1); }
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
        // (e) partial gate tiles -> LDS   (C layout: col = lane&15, row = 4*(lane>>4) + reg)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) part[w][mt][(4 * fq + i) * PSTR + gt * 16 + fr] = acc[gt][mt][i];
        STAMP(2);
        __syncthreads();
        STAMP(3);
        // (f) cell update, one (row, unit) per thread and row tile
        float o_i[MT], o_f[MT], o_g[MT], o_o[MT], o_c[MT], o_h[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float pre[4];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                const int o = crow * PSTR + gt * 16 + cunit;
                pre[gt] = xg[mt][gt] + ((part[0][mt][o] + part[1][mt][o]) + (part[2][mt][o] + part[3][mt][o]));
            }
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
            const float cn = fg * c_reg[mt] + ig * gg;
            const float hn = og * tanhf_(cn);
            const bool active = t < len_r[mt];
            if (active) { c_reg[mt] = cn; h_reg[mt] = hn; }
            o_i[mt] = ig; o_f[mt] = fg; o_g[mt] = gg; o_o[mt] = og; o_c[mt] = cn; o_h[mt] = active ? hn : 0.f;
            hstage[mt][crow * 16 + cunit] = h_reg[mt];
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
        // (g) publish this workgroup's h block(s) for step t+1 FIRST (critical path of every peer) ...
        if (w < MT) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&hstage[w][4 * lane]);
            const long long blk = ((long long)(par_w * a.G + g) * a.NW + j) * MT + w;
            st16_sc1(rs, (unsigned)(blk * blk_bytes + lane * 16), v);
            drain_vmem();
        }
        STAMP(6);
        __syncthreads();
        if (!DEC && tid == 0) add_flag(ctr, 1u);
        STAMP(7);
        // ... then the outputs nobody waits for inside this launch (h_t, saved activations)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
            if (rb < a.B) {
                const long long o = ((long long)t * a.B + rb);
                a.hs[o * H + u0 + cunit] = o_h[mt];
                if (a.gates) {
                    float* pg = a.gates + o * H4 + u0 + cunit;
                    pg[0] = o_i[mt]; pg[H] = o_f[mt]; pg[2 * H] = o_g[mt]; pg[3 * H] = o_o[mt];
                }
                if (a.cs) a.cs[o * H + u0 + cunit] = o_c[mt];
            }
        }
        par_r = par_w;
    }
    STAMP_FLUSH(a.sync, a.G);

    // final states and zero tail of the outputs for groups that finished early
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
        if (rb < a.B) {
            if (a.hN) a.hN[(long long)rb * H + u0 + cunit] = h_reg[mt];
            if (a.cN) a.cN[(long long)rb * H + u0 + cunit] = c_reg[mt];
            for (int t = Tg; t < a.T; ++t) a.hs[((long long)t * a.B + rb) * H + u0 + cunit] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward sweep
// ---------------------------------------------------------------------------------------------
template <int MT, int NJ>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(LstmFwdArgs a) { lstm_fwd_body<MT, NJ>(a, blockIdx.x); }
// the single free-running decoder step (mode = 1) under a name of its own, so that profiles tell the 100 short steps
// of a generation from the full sweeps
template <int MT, int NJ>
__global__ __launch_bounds__(256) void lstm_dec_kernel(LstmFwdArgs a) { lstm_fwd_body<MT, NJ, true>(a, blockIdx.x); }

// Two independent sweeps in one launch (a4nt_lstm_seq_fwd_pair): the row groups of the two problems share the grid
// (pair_block), each problem with its own exchange buffer and counters.  Neither sweep fills the chip alone (128 rows
// = 256 of 512 resident workgroups), and each step of a sweep is mostly hand-off latency: side by side,
// one sweep's MFMA phase runs under the other's waits.
template <int MT, int NJ>
__global__ __launch_bounds__(256) void lstm_fwd2_kernel(LstmFwdArgs a, LstmFwdArgs b, int nA) {
    int local;
    const bool first = pair_block((int)blockIdx.x, a.G, b.G, local);
    const LstmFwdArgs s = first ? a : b;
    lstm_fwd_body<MT, NJ>(s, local);
}

struct LstmBwdArgs {
    int T, B, H, G, NW;
    const float* whh;       // [4H][H]
    const float* gates;     // [T][B][4H] saved post-activation gates
    const float* cs;        // [T][B][H]
    const float* c0;        // [B][H] or null
    const int* lens;        // [B] or null
    const float* dhs;       // [T][B][H] or null
    const float* dhN; const float* dcN;   // [B][H] or null
    float* dgates;          // [T][B][4H] (may alias gates)
    float* dh0; float* dc0; // [B][H] or null
    float* gxch;            // exchange [2][G][NW consumers][NW producers][MT][256]
    unsigned* sync;
    unsigned* tmo;
    int need_dh0;           // run the recurrent product of the last step too
    float* dbias;           // [G][4H] per-row-group column sums of dgates (the bias gradient's partials) or null; 2-D kernels only
};

template <int MT, int NJ>
__device__ __forceinline__ void lstm_bwd_body(const LstmBwdArgs& a, const int bid) {
    // Workgroup j of a row group owns hidden units [16j, 16j+16): it computes dgates_t for those units
    // (64 of the 4H gate rows) and multiplies them, straight from LDS, with ITS 64 rows of W_hh, which
    // gives a partial dh_{t-1} for ALL H columns; the partials are exchanged as 16x16 blocks and every
    // workgroup sums the NW blocks of its own 16 columns (a reduce-scatter per step).  The payload per
    // step is H floats per row in each direction, as in the forward sweep (an all-gather of dgates would
    // move 4H).  Wave w owns output columns [w*H/4, (w+1)*H/4).
    __shared__ __attribute__((aligned(16))) float sA[MT][16 * 68];      // dgates tile [row][gate*16 + unit]
    __shared__ __attribute__((aligned(16))) float red[4][MT][16 * 20];  // per-wave partial sums [col][row]
    __shared__ int s_ok;

    int g, j;
    decode_bid(bid, a.G, g, j);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H = a.H, H4 = 4 * a.H;
    const int u0 = 16 * j;
    const int r0 = g * 16 * MT;
    const int fr = lane & 15, fq = lane >> 4;
    const int crow = tid >> 4, cunit = tid & 15;

    // B operands: wreg[ct][gate][i] (lane n = fr, quad fq) = W_hh[gate*H + u0 + 4*fq + i][w*H/4 + 16*ct + fr]
    // (the k order inside a gate's 16 units is permuted the same way on the A side)
    float wreg[NJ][4][4];
#pragma unroll
    for (int ct = 0; ct < NJ; ++ct)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wreg[ct][gt][i] = a.whh[(long long)(gt * H + u0 + 4 * fq + i) * H + w * (H >> 2) + 16 * ct + fr];

    unsigned* ctr = a.sync + 32 * g;
    unsigned* tmo = a.tmo;
    const unsigned blk_bytes = 256 * 4;
    const long long x_par = (long long)a.G * a.NW * a.NW * MT * 256;    // floats per parity
    __amdgpu_buffer_rsrc_t rs = make_rsrc(a.gxch, (unsigned)(2 * x_par * 4));

    float dh_rec[MT], dc_next[MT];
    int len_r[MT];
    int Tg = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
        dh_rec[mt] = 0.f; dc_next[mt] = 0.f;
        len_r[mt] = (rb < a.B) ? (a.lens ? min(a.lens[rb], a.T) : a.T) : 0;
    }
    if (a.lens) {
        for (int r = 0; r < 16 * MT; ++r) {
            const int rb = r0 + r;
            if (rb < a.B) Tg = max(Tg, min(a.lens[rb], a.T));
        }
    } else {
        Tg = a.T;
    }

    // rows that never ran: their dgates are zero for t in [Tg, T)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
        if (rb < a.B)
            for (int t = Tg; t < a.T; ++t) {
                float* pd = a.dgates + ((long long)t * a.B + rb) * H4 + u0 + cunit;
                pd[0] = 0.f; pd[H] = 0.f; pd[2 * H] = 0.f; pd[3 * H] = 0.f;
            }
    }

    // Saved activations, cell states and upstream dh of a step do not depend on the recurrence: they are loaded one
    // step ahead (issued before the wait for the peers' partials), with clamped, always-valid addresses so that the
    // loads are unconditional; validity is applied to the values.
    struct StepIn { float ig, fg, gg, og, cp, dhs; };
    auto load_step = [&](int t, StepIn (&p)[MT]) {
        const int tc = max(t, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = min(r0 + 16 * mt + crow, a.B - 1);
            const long long o = ((long long)tc * a.B + rb);
            const float* pg = a.gates + o * H4 + u0 + cunit;
            p[mt].ig = pg[0]; p[mt].fg = pg[H]; p[mt].gg = pg[2 * H]; p[mt].og = pg[3 * H];
            if (tc > 0) p[mt].cp = a.cs[(o - a.B) * H + u0 + cunit];
            else p[mt].cp = a.c0 ? a.c0[(long long)rb * H + u0 + cunit] : 0.f;
            p[mt].dhs = a.dhs ? a.dhs[o * H + u0 + cunit] : 0.f;
        }
    };
    StepIn cur[MT];
    float ct_cur[MT];           // c_t of the current step = c_prev of the step processed before it
    load_step(Tg - 1, cur);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = min(r0 + 16 * mt + crow, a.B - 1);
        ct_cur[mt] = Tg > 0 ? a.cs[((long long)(Tg - 1) * a.B + rb) * H + u0 + cunit] : 0.f;
    }

    int s = 0;   // published steps so far
    STAMP_DECL
    for (int t = Tg - 1; t >= 0; --t, ++s) {
        const int par = s & 1;
        STAMP_BEGIN();
        float dpo[MT][4];
        // (a) cell backward for this thread's (row, unit)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
            float dp[4] = {0.f, 0.f, 0.f, 0.f};
            if (rb < a.B && t < len_r[mt]) {
                const float ig = cur[mt].ig, fg = cur[mt].fg, gg = cur[mt].gg, og = cur[mt].og;
                const float ct = ct_cur[mt], cp = cur[mt].cp;
                float dh = dh_rec[mt] + cur[mt].dhs;
                float dc = dc_next[mt];
                if (t == len_r[mt] - 1) {
                    if (a.dhN) dh += a.dhN[(long long)rb * H + u0 + cunit];
                    if (a.dcN) dc += a.dcN[(long long)rb * H + u0 + cunit];
                }
                const float tc = tanhf_(ct);
                dc += dh * og * (1.f - tc * tc);
                dp[0] = dc * gg * ig * (1.f - ig);
                dp[1] = dc * cp * fg * (1.f - fg);
                dp[2] = dc * ig * (1.f - gg * gg);
                dp[3] = dh * tc * og * (1.f - og);
                dc_next[mt] = dc * fg;
            }
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) { sA[mt][crow * 68 + gt * 16 + cunit] = dp[gt]; dpo[mt][gt] = dp[gt]; }
        }
        if (t == 0 && !a.need_dh0) {
            // last step: nothing to exchange, just emit dgates_0
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int rb = r0 + 16 * mt + crow;
                if (rb < a.B) {
                    float* pd = a.dgates + ((long long)rb) * H4 + u0 + cunit;
                    pd[0] = dpo[mt][0]; pd[H] = dpo[mt][1]; pd[2 * H] = dpo[mt][2]; pd[3 * H] = dpo[mt][3];
                }
            }
            break;
        }
        __syncthreads();
        STAMP(0);
        // (b) partial dh_{t-1}[:, this wave's columns] = dgates_t[:, own 64] * W_hh[own 64 rows, those columns]
        // column tiles go CHB at a time (bounds the accumulator registers at H = 1024 / MT = 4); the A
        // fragments are re-read from LDS per gate and chunk
        constexpr int CHB = (NJ * MT > 16) ? ((NJ == 16 ? 8 : 16) / MT) : NJ;
#pragma unroll
        for (int cc = 0; cc < NJ / CHB; ++cc) {
            f32x4 acc[CHB][MT];
#pragma unroll
            for (int c8 = 0; c8 < CHB; ++c8)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[c8][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                f32x4 af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(&sA[mt][fr * 68 + gt * 16 + 4 * fq]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c8 = 0; c8 < CHB; ++c8)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[c8][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][i], wreg[cc * CHB + c8][gt][i], acc[c8][mt], 0, 0, 0);
            }
            // publish: block (consumer c, producer j, mt), layout [col][row] (C fragment order: one 1 KiB store per wave)
#pragma unroll
            for (int c8 = 0; c8 < CHB; ++c8)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int c = w * NJ + cc * CHB + c8;
                    const long long blk = ((((long long)par * a.G + g) * a.NW + c) * a.NW + j) * MT + mt;
                    st16_sc1(rs, (unsigned)(blk * blk_bytes + (fr * 16 + 4 * fq) * 4), acc[c8][mt]);
                }
        }
        STAMP(1);
        drain_vmem();
        __syncthreads();
        if (tid == 0) add_flag(ctr, 1u);
        STAMP(2);
        // dgates_t for the time-batched weight-gradient GEMMs: nobody waits for it inside this launch
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rb = r0 + 16 * mt + crow;
            if (rb < a.B) {
                float* pd = a.dgates + ((long long)t * a.B + rb) * H4 + u0 + cunit;
                pd[0] = dpo[mt][0]; pd[H] = dpo[mt][1]; pd[2 * H] = dpo[mt][2]; pd[3 * H] = dpo[mt][3];
            }
        }
        StepIn nxt[MT];
        load_step(t - 1, nxt);                   // in flight under the wait below
        STAMP(3);
        if (tid == 0) s_ok = wait_ge(ctr, (unsigned)(a.NW * (s + 1)), tmo) ? 1 : 0;
        __syncthreads();
        if (!s_ok) break;
        STAMP(4);
        // (c) sum the NW partial blocks of this workgroup's 16 columns: wave w takes producers [w*NJ, (w+1)*NJ)
        f32x4 ps[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ps[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < NJ / CHB; ++cc) {
            f32x4 ld[CHB][MT];
#pragma unroll
            for (int p8 = 0; p8 < CHB; ++p8)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const long long blk = ((((long long)par * a.G + g) * a.NW + j) * a.NW + (w * NJ + cc * CHB + p8)) * MT + mt;
                    ld[p8][mt] = ld16_sc1(rs, (unsigned)(blk * blk_bytes + lane * 16));
                }
#pragma unroll
            for (int p8 = 0; p8 < CHB; ++p8)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ps[mt].x += ld[p8][mt].x; ps[mt].y += ld[p8][mt].y; ps[mt].z += ld[p8][mt].z; ps[mt].w += ld[p8][mt].w;
                }
        }
        STAMP_WAITV();
        STAMP(5);
        // lane l holds column l>>2, rows 4*(l&3)..+3
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(&red[w][mt][(lane >> 2) * 20 + 4 * (lane & 3)]) = ps[mt];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int o = cunit * 20 + crow;
            dh_rec[mt] = (red[0][mt][o] + red[1][mt][o]) + (red[2][mt][o] + red[3][mt][o]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { ct_cur[mt] = cur[mt].cp; cur[mt] = nxt[mt]; }
        STAMP(6);
        // the next iteration's barrier after (a) orders these LDS reads before red[] / sA[] are rewritten
    }

    STAMP_FLUSH(a.sync, a.G);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rb = r0 + 16 * mt + crow;
        if (rb < a.B) {
            if (a.dh0) a.dh0[(long long)rb * H + u0 + cunit] = dh_rec[mt];
            if (a.dc0) a.dc0[(long long)rb * H + u0 + cunit] = dc_next[mt];
        }
    }
}

template <int MT, int NJ>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(LstmBwdArgs a) { lstm_bwd_body<MT, NJ>(a, blockIdx.x); }

template <int MT, int NJ>
__global__ __launch_bounds__(256) void lstm_bwd2_kernel(LstmBwdArgs a, LstmBwdArgs b, int nA) {
    int local;
    const bool first = pair_block((int)blockIdx.x, a.G, b.G, local);
    const LstmBwdArgs s = first ? a : b;
    lstm_bwd_body<MT, NJ>(s, local);
}

// ---------------------------------------------------------------------------------------------
// backward sweep, 2-D decomposition with 16*MT rows per group and LDS-staged step inputs
// ---------------------------------------------------------------------------------------------
// The NW = H/16 workgroups of a row group form a P x 4 grid: workgroup (i, q) holds the W_hh block
// [4 gates x units 64i..64i+63][columns q*H/4 .. (q+1)*H/4) (H/4 weight registers per lane).  Per step and row tile it
//   * sums, for ITS 64 units, the P partial dh blocks the workgroups (., q*) of that column range published
//     (wave w: unit tile 4i+w, P x MT contiguous 1 KiB blocks; the sum lands in the lanes that own those
//     (row, unit) pairs -- no LDS round trip),
//   * runs the cell backward for 16 rows x 64 units (4 pairs per thread; the 4 workgroups of a unit block do this
//     redundantly: element-wise work on data that is already in LDS), dgates -> LDS [16][256],
//   * multiplies that tile with its weight block: a partial dh over H/4 columns,
//   * publishes it: 16 x H/4 floats per step instead of the 16 x H of the 1-D scheme (lstm_bwd_body, still used for
//     H <= 128), where every workgroup publishes a partial over ALL columns -- at H = 512 a quarter of the
//     write-through bytes, 2 instead of 8 stores per wave to drain before the signal.
// Same hand-off protocol as lstm_bwd_body; one arrival counter per group.  Further:
//   * MT row tiles of 16 rows share the workgroup's weight registers (32 or 64 rows per group: one workgroup per CU
//     at 256 rows and H = 512; the only geometry that fits 256 rows at H = 1024).  A tile is processed as
//     cell backward -> dgates tile in LDS -> MFMA with the weight block -> publish, one tile after the other through
//     ONE LDS tile, so the accumulators are NT fragments whatever MT (the 1-D kernel's 64-row instantiation spilled).
//   * The saved activations, cell states and upstream dh of a step never pass through registers: every wave
//     fetches, for its own 16 units x 16 rows and all six arrays, the NEXT step's values straight into LDS
//     (global_load_lds_dwordx4: 6 instructions per wave and tile) right after the cell backward of the current step
//     freed that slot, so they travel under the MFMA chain.  No prefetch registers, no compiler-placed wait on them
//     at the loop's back edge (the register version waited vmcnt(0) there), and the loads are whole 64-byte pieces
//     instead of dwords.  The slots are private to a wave (it stages what its own lanes consume), so no barrier is
//     needed between the fetch and the use.
// LDS-DMA writes lane-linear: lane p's 16 bytes land at chunk p of the slot.  Chunk p holds (row, unit quad) =
// (8*(p>>5) + 4*((p>>2)&1) + ((p>>3)&3), p&3): with that order the cell backward's reads (lane = unit, rows
// 4*(lane&3)+j) spread over all 32 banks (row-major chunks would put the four row groups on the same 16 banks).
template <int NT, int MT>
__device__ __forceinline__ void lstm_bwdx_body(const LstmBwdArgs& a, const int bid) {
    constexpr int SA = 260;                                            // LDS row stride of the dgates tile
    __shared__ __attribute__((aligned(16))) float sA[16 * SA];         // [row][gate*64 + unit], one row tile
    __shared__ __attribute__((aligned(16))) float sIn[MT][4][6][256];  // [tile][wave][i,f,g,o,c_prev,dhs][chunk*4 + e]
    // H = 1024 with 64 rows per group: 256 weight registers leave no room for all of the per-row recurrent state, so
    // c_t and d(c) of the thread's 16 (row, unit) pairs live in LDS words of its own ([.][tid]: conflict-free)
    constexpr bool ST_LDS = (NT * MT >= 16);
    __shared__ float sSt[ST_LDS ? 2 * MT * 4 : 1][256];
    __shared__ int s_len[16 * MT];                                     // row lengths of the group (0 past the batch)
    __shared__ int s_ok;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    int g, jw;
    decode_bid(bid, a.G, g, jw);
    const int i_blk = jw >> 2, q = jw & 3;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: SGPR)
    const int H = a.H, H4 = 4 * a.H, P = a.H >> 6, NTILES = a.H >> 4;
    const int u0 = 64 * i_blk;
    const int r0 = g * 16 * MT;
    const int fr = lane & 15, fq = lane >> 4;
    const int ul = 16 * w + (lane >> 2);          // this thread's unit inside the block ...
    const int rj0 = 4 * (lane & 3);               // ... and its 4 rows rj0..rj0+3 of every row tile
    const int col0 = q * (H >> 2) + w * 16 * NT;  // first output column of this wave

    float wreg[NT][16][4];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int kb = 0; kb < 16; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                wreg[c][kb][e] = a.whh[(long long)((kb >> 2) * H + u0 + 16 * (kb & 3) + 4 * fq + e) * H + col0 + 16 * c + fr];

    unsigned* ctr = a.sync + 32 * g;
    unsigned* tmo = a.tmo;
    const unsigned blk_bytes = 256 * 4;
    const long long x_par = (long long)a.G * NTILES * P * MT * 256;    // floats per parity
    __amdgpu_buffer_rsrc_t rs = make_rsrc(a.gxch, (unsigned)(2 * x_par * 4));

    constexpr int NST = ST_LDS ? 1 : MT;
    float dh_rec[MT][4], dc_next[NST][4], ct_cur[NST][4];
    int Tg = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dh_rec[m][j] = 0.f;
            if (ST_LDS) sSt[(MT + m) * 4 + j][tid] = 0.f; else dc_next[m][j] = 0.f;
        }
    if (tid < 16 * MT) {
        const int rb = r0 + tid;
        s_len[tid] = (rb < a.B) ? (a.lens ? min(a.lens[rb], a.T) : a.T) : 0;
    }
    __syncthreads();
    for (int r = 0; r < 16 * MT; ++r) Tg = max(Tg, s_len[r]);
    // dgates of the steps no row of this group ran (each of the four workgroups of a unit block: one gate)
    for (int t = Tg; t < a.T; ++t)
        for (int idx = tid; idx < 16 * MT * 64; idx += 256) {
            const int row = idx >> 6, cc = idx & 63;
            if (r0 + row < a.B) a.dgates[((long long)t * a.B + r0 + row) * H4 + q * H + u0 + cc] = 0.f;
        }

    // ---- step inputs: LDS-DMA ---------------------------------------------------------------------------------
    // source of this lane's chunk: row prow of the tile, units u0 + 16w + 4*(lane&3) .. +3 (rows past the batch are
    // clamped to the last row: fetched, never used)
    const int prow = 8 * (lane >> 5) + 4 * ((lane >> 2) & 1) + ((lane >> 3) & 3);
    const int pun = u0 + 16 * w + 4 * (lane & 3);
    int src_row[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) src_row[m] = min(r0 + 16 * m + prow, a.B - 1);
    // where the cell backward finds (row rj0 + j, unit ul) in a slot: chunk 8*(j + 4*(row>>3)) + 4*((row>>2)&1) + quad
    const int rd0 = 4 * (8 * (4 * ((lane & 3) >> 1)) + 4 * (lane & 1) + (lane >> 4)) + ((lane >> 2) & 3);   // + 32*j
    if (!a.dhs) {                            // no upstream dh: that array's slots stay zero for the whole sweep
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(&sIn[m][w][5][4 * lane]) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto fetch = [&](const int t, const int m) {
        const long long o = (long long)t * a.B + src_row[m];
        const float* pg = a.gates + o * H4 + pun;
        __builtin_amdgcn_global_load_lds((gptr_t)(pg), (lptr_t)&sIn[m][w][0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pg + H), (lptr_t)&sIn[m][w][1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pg + 2 * H), (lptr_t)&sIn[m][w][2][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pg + 3 * H), (lptr_t)&sIn[m][w][3][0], 16, 0, 0);
        if (t > 0) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a.cs + (o - a.B) * H + pun), (lptr_t)&sIn[m][w][4][0], 16, 0, 0);
        } else if (a.c0) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a.c0 + (long long)src_row[m] * H + pun), (lptr_t)&sIn[m][w][4][0], 16, 0, 0);
        } else {
            *reinterpret_cast<f32x4*>(&sIn[m][w][4][4 * lane]) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (a.dhs) __builtin_amdgcn_global_load_lds((gptr_t)(a.dhs + o * H + pun), (lptr_t)&sIn[m][w][5][0], 16, 0, 0);
    };
    if (Tg > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            fetch(Tg - 1, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = min(r0 + 16 * m + rj0 + j, a.B - 1);
                const float c_last = a.cs[((long long)(Tg - 1) * a.B + rb) * H + u0 + ul];
                if (ST_LDS) sSt[m * 4 + j][tid] = c_last; else ct_cur[m][j] = c_last;
            }
        }
    }

    // (rows past the batch or past their length hold zeros in the tile: the running column sum takes every row)
    float bsum = 0.f;
    auto store_dgates = [&](const int t, const int m) {
        for (int idx = tid; idx < 16 * 64; idx += 256) {
            const int row = idx >> 6, cc = idx & 63;
            const float v = sA[row * SA + q * 64 + cc];
            bsum += v;
            if (r0 + 16 * m + row < a.B)
                a.dgates[((long long)t * a.B + r0 + 16 * m + row) * H4 + q * H + u0 + cc] = v;
        }
    };

    int s = 0;
    for (int t = Tg - 1; t >= 0; --t, ++s) {
        const int par = s & 1;
        const bool last = (t == 0 && !a.need_dh0);     // nothing to exchange after the last step's cell backward
        // this step's inputs were fetched a step ago (or by the prologue); everything this wave has in flight is older
        // than its last wait except in the first iteration
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m > 0) __syncthreads();               // the previous tile's MFMA and dgates reads of sA are done
            // (a) cell backward of this thread's 4 (row, unit) pairs of tile m -> dgates tile in LDS
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = r0 + 16 * m + rj0 + j;
                const float* in = &sIn[m][w][0][rd0 + 32 * j];
                const float ig = in[0], fg = in[256], gg = in[512], og = in[768], cp = in[1024], dhs = in[1280];
                float dp[4] = {0.f, 0.f, 0.f, 0.f};
                const int len_rj = s_len[16 * m + rj0 + j];          // 0 for rows past the batch
                if (t < len_rj) {
                    const float ct = ST_LDS ? sSt[m * 4 + j][tid] : ct_cur[ST_LDS ? 0 : m][j];
                    float dh = dh_rec[m][j] + dhs;
                    float dc = ST_LDS ? sSt[(MT + m) * 4 + j][tid] : dc_next[ST_LDS ? 0 : m][j];
                    if (t == len_rj - 1) {
                        if (a.dhN) dh += a.dhN[(long long)rb * H + u0 + ul];
                        if (a.dcN) dc += a.dcN[(long long)rb * H + u0 + ul];
                    }
                    const float tc = tanhf_(ct);
                    dc += dh * og * (1.f - tc * tc);
                    dp[0] = dc * gg * ig * (1.f - ig);
                    dp[1] = dc * cp * fg * (1.f - fg);
                    dp[2] = dc * ig * (1.f - gg * gg);
                    dp[3] = dh * tc * og * (1.f - og);
                    if (ST_LDS) sSt[(MT + m) * 4 + j][tid] = dc * fg; else dc_next[ST_LDS ? 0 : m][j] = dc * fg;
                }
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) sA[(rj0 + j) * SA + gt * 64 + ul] = dp[gt];
                if (ST_LDS) sSt[m * 4 + j][tid] = cp; else ct_cur[ST_LDS ? 0 : m][j] = cp;
            }
            __syncthreads();
            // the slot of tile m is free (its reads are behind this wave): fetch the next step's values into it
            if (t > 0) fetch(t - 1, m);
            if (!last) {
                // (b) partial dh_{t-1}[tile m][:, this wave's 16*NT columns] = dgates tile [16 x 256] * weight block
                f32x4 acc[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(&sA[fr * SA + (kb >> 2) * 64 + 16 * (kb & 3) + 4 * fq]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < NT; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wreg[c][kb][e], acc[c], 0, 0, 0);
                }
                // publish: block (column tile, producer unit block, row tile), [col][row] fragment order
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const int ctg = (col0 >> 4) + c;
                    const unsigned blk = (unsigned)((((par * a.G + g) * NTILES + ctg) * P + i_blk) * MT + m);   // wave-uniform
                    st16_sc1_s(rs, (unsigned)((fr * 16 + 4 * fq) * 4), blk * blk_bytes, acc[c]);
                }
            }
            // dgates_t of tile m for the time-batched weight-gradient GEMMs (the four workgroups that share the unit
            // block write one gate each).  Nobody waits for it inside this launch: the last tile's goes out behind the
            // signal, so that the drain in front of the signal does not wait for these stores
            if (m + 1 < MT || last) store_dgates(t, m);
        }
        if (last) break;
        drain_vmem();
        __syncthreads();
        if (tid == 0) add_flag(ctr, 1u);
        store_dgates(t, MT - 1);
        if (tid == 0) s_ok = wait_ge(ctr, (unsigned)(a.NW * (s + 1)), tmo) ? 1 : 0;
        __syncthreads();
        if (!s_ok) break;
        // (c) dh_{t-1} of this wave's unit tile, per row tile: sum of the P producers' blocks, in producer order; the
        // loads of the next row tile are issued before the sums of this one
        {
            constexpr int PP = 4 * NT;               // = P
            constexpr int DB = (NT < 4) ? 2 : 1;           // H = 1024: one set of 16 blocks (64 registers) in flight
            f32x4 v[DB][PP];
            auto ldp = [&](const int m, f32x4 (&d)[PP]) {
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const unsigned blk = (unsigned)((((par * a.G + g) * NTILES + (u0 >> 4) + w) * P + p) * MT + m);   // wave-uniform
                    d[p] = ld16_sc1_s(rs, (unsigned)(lane * 16), blk * blk_bytes);
                }
            };
            ldp(0, v[0]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (DB == 2 && m + 1 < MT) ldp(m + 1, v[(m + 1) & (DB - 1)]);
                f32x4 ps = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < PP; ++p) { ps.x += v[m & (DB - 1)][p].x; ps.y += v[m & (DB - 1)][p].y; ps.z += v[m & (DB - 1)][p].z; ps.w += v[m & (DB - 1)][p].w; }
                if (DB == 1 && m + 1 < MT) ldp(m + 1, v[0]);
                dh_rec[m][0] = ps.x; dh_rec[m][1] = ps.y; dh_rec[m][2] = ps.z; dh_rec[m][3] = ps.w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the wave (early exits included)

    if (a.dbias) {
        // bias gradient of this row group: the four waves hold the sums of rows w, w+4, .. of the same 64 columns
        __syncthreads();
        sA[tid] = bsum;
        __syncthreads();
        if (tid < 64)
            a.dbias[(long long)g * H4 + q * H + u0 + tid] = (sA[tid] + sA[64 + tid]) + (sA[128 + tid] + sA[192 + tid]);
    }
    if (q == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = r0 + 16 * m + rj0 + j;
                if (rb < a.B) {
                    if (a.dh0) a.dh0[(long long)rb * H + u0 + ul] = dh_rec[m][j];
                    if (a.dc0) a.dc0[(long long)rb * H + u0 + ul] = ST_LDS ? sSt[(MT + m) * 4 + j][tid] : dc_next[ST_LDS ? 0 : m][j];
                }
            }
    }
}

// (second launch bound = waves per SIMD the register allocation must leave room for: two row groups per CU for
// 16-row groups up to H = 512; 32- and 64-row groups and H = 1024 take one workgroup per CU)
template <int NT, int MT>
__global__ __launch_bounds__(256, (NT <= 2 && MT == 1) ? 2 : 1) void lstm_bwdx_kernel(LstmBwdArgs a) { lstm_bwdx_body<NT, MT>(a, blockIdx.x); }

template <int NT, int MT>
__global__ __launch_bounds__(256, (NT <= 2 && MT == 1) ? 2 : 1) void lstm_bwdx2_kernel(LstmBwdArgs a, LstmBwdArgs b, int nA) {
    int local;
    const bool first = pair_block((int)blockIdx.x, a.G, b.G, local);
    const LstmBwdArgs s = first ? a : b;
    lstm_bwdx_body<NT, MT>(s, local);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Workgroup budget of one sweep.  256 = one workgroup per CU.  The MT = 1 kernels up to H = 512 fit two
// per CU (<= 224 VGPR+AGPR, 18 KB LDS), and with 512 co-resident workgroups two independent row groups
// share every CU, so one group's hand-off latency hides behind the other's MFMA phase (B = 256,
// H = 512: 6.7 us/step against 8.8 us/step for 256 workgroups of MT = 2).  The H = 1024 kernels (NJ = 16)
// hold 256 weight registers per lane: one workgroup per CU, whatever MT.  Every workgroup of a sweep
// must be resident at once (they wait for each other), so the static caps below are cut to what
// hipOccupancyMaxActiveBlocksPerMultiprocessor reports for the instantiation on the initialised device
// (a4nt_lstm_probe_occupancy, called by a4nt_init): a geometry that cannot be co-resident is an error at
// geometry time, not a spin that runs into A4NT_SPIN_LIMIT.
// a4nt_lstm_set_max_workgroups(256) restores one workgroup per CU for every instantiation (data
// parallelism: leaves register space for RCCL's kernels).
#define A4NT_MAX_DEV 16
static int g_max_wgs[A4NT_MAX_DEV + 1];       // per device (last slot: no current device); 0 = the default, 512
// [device][mt index 0..2][nj index 0..4]: co-resident workgroups (CUs x occupancy, min over the forward and the
// backward kernel); 0 = not probed (no device initialised: the static caps alone apply).  [..][1] = pair kernels.
static int g_resident[A4NT_MAX_DEV][2][3][5];
static int ilog2(int v) { int r = 0; while ((1 << (r + 1)) <= v) ++r; return r; }
extern "C" int a4nt_lstm_set_max_workgroups(int n) {
    if (n != 256 && n != 512) { a4nt_set_error("a4nt_lstm_set_max_workgroups: 256 or 512"); return A4NT_ERR_ARG; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= A4NT_MAX_DEV) { (void)hipGetLastError(); dev = A4NT_MAX_DEV; }
    g_max_wgs[dev] = n;                   // a setting of the calling thread's current device
    return A4NT_OK;
}
static int budget(int mt, int nj, int pair) {
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < A4NT_MAX_DEV;
    if (!have_dev) { (void)hipGetLastError(); dev = A4NT_MAX_DEV; }      // no device (CPU-only host): static caps
    const int max_wgs = g_max_wgs[dev] ? g_max_wgs[dev] : 512;
    int cap = (mt == 1 && nj < 16) ? max_wgs : 256;
    if (have_dev) {
        const int res = g_resident[dev][pair][ilog2(mt)][ilog2(nj)];
        if (res > 0 && res < cap) cap = res;
    }
    return cap;
}
static int pick_mt(int B, int NW, int* G) {
    for (int mt = 1; mt <= 4; mt *= 2) {
        const int g = (B + 16 * mt - 1) / (16 * mt);
        if ((long long)g * NW <= budget(mt, NW / 4, 0)) { *G = g; return mt; }
    }
    return 0;
}

#define A4NT_DISPATCH_LSTM(KERNEL, mt, nj, grid, block, s, a)                                                   \
    do {                                                                                                         \
        if (mt == 1 && nj == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, a);                         \
        else if (mt == 1 && nj == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, a);                    \
        else if (mt == 1 && nj == 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, s, a);                    \
        else if (mt == 1 && nj == 8) hipLaunchKernelGGL((KERNEL<1, 8>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 8) hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, s, a);                    \
        else if (mt == 4 && nj == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, s, a);                    \
        else if (mt == 4 && nj == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, s, a);                    \
        else if (mt == 4 && nj == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, s, a);                    \
        else if (mt == 4 && nj == 8) hipLaunchKernelGGL((KERNEL<4, 8>), grid, block, 0, s, a);                    \
        else if (mt == 1 && nj == 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, block, 0, s, a);                  \
        else if (mt == 2 && nj == 16) hipLaunchKernelGGL((KERNEL<2, 16>), grid, block, 0, s, a);                  \
        else hipLaunchKernelGGL((KERNEL<4, 16>), grid, block, 0, s, a);                                           \
    } while (0)

// (the 1-D backward kernel serves H = 64 / 128 only)
#define A4NT_DISPATCH_LSTM_SMALL(KERNEL, mt, nj, grid, block, s, a)                                             \
    do {                                                                                                         \
        if (mt == 1 && nj == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, a);                         \
        else if (mt == 1) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, a);                               \
        else if (mt == 2 && nj == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, a);                    \
        else if (mt == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, a);                               \
        else if (nj == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, s, a);                               \
        else hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, s, a);                                            \
    } while (0)

// (every instantiation up to H = 512)
#define A4NT_DISPATCH_LSTM_TO8(KERNEL, mt, nj, grid, block, s, a)                                               \
    do {                                                                                                         \
        if (mt == 1 && nj == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, a);                         \
        else if (mt == 1 && nj == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, a);                    \
        else if (mt == 1 && nj == 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, s, a);                    \
        else if (mt == 1) hipLaunchKernelGGL((KERNEL<1, 8>), grid, block, 0, s, a);                               \
        else if (mt == 2 && nj == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, a);                    \
        else if (mt == 2 && nj == 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, s, a);                    \
        else if (mt == 2) hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, s, a);                               \
        else if (nj == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, s, a);                               \
        else if (nj == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, s, a);                               \
        else if (nj == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, s, a);                               \
        else hipLaunchKernelGGL((KERNEL<4, 8>), grid, block, 0, s, a);                                            \
    } while (0)

static bool bad_h(int H) { return !(H == 64 || H == 128 || H == 256 || H == 512 || H == 1024); }

// Co-residency probe (called by a4nt_init with `device` current): what the occupancy calculator says about every
// instantiation, stored per device for budget().  Fails when some instantiation cannot hold one workgroup per CU.
template <typename KF, typename KB> static int occ_pair(KF kf, KB kb) {
    int nf = 0, nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, kf, 256, 0) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kb, 256, 0) != hipSuccess) return -1;
    return nf < nb ? nf : nb;
}
#define A4NT_PROBE(mt, nj)                                                                                        \
    do {                                                                                                          \
        const int o = occ_pair(lstm_fwd_kernel<mt, nj>, lstm_bwd_kernel<mt, nj>);                                 \
        if (o < 1) { a4nt_set_error("a4nt_init: LSTM sweep <MT=%d,NJ=%d> cannot be resident on device %d (occupancy %d)", mt, nj, device, o); return A4NT_ERR_ARG; } \
        g_resident[device][0][ilog2(mt)][ilog2(nj)] = o * num_cus;                                                \
    } while (0)
#define A4NT_PROBE2(mt, nj)                                                                                       \
    do {                                                                                                          \
        const int o = occ_pair(lstm_fwd2_kernel<mt, nj>, lstm_bwd2_kernel<mt, nj>);                               \
        if (o < 1) { a4nt_set_error("a4nt_init: paired LSTM sweep <MT=%d,NJ=%d> cannot be resident on device %d (occupancy %d)", mt, nj, device, o); return A4NT_ERR_ARG; } \
        g_resident[device][1][ilog2(mt)][ilog2(nj)] = o * num_cus;                                                \
    } while (0)
static int scratch_lookup(int device);
#define A4NT_COMMA ,
#define A4NT_PROBE_K(mt, nj, pair, KF, KB)                                                                        \
    do {                                                                                                          \
        const int o = occ_pair(KF, KB);                                                                           \
        if (o < 1) { a4nt_set_error("a4nt_init: LSTM sweep <MT=%d,NJ=%d> cannot be resident on device %d (occupancy %d)", mt, nj, device, o); return A4NT_ERR_ARG; } \
        g_resident[device][pair][ilog2(mt)][ilog2(nj)] = o * num_cus;                                             \
    } while (0)
int a4nt_lstm_probe_occupancy(int device, int num_cus) {
    if (device < 0 || device >= A4NT_MAX_DEV) return A4NT_OK;        // beyond the table: static caps
    const int rcs = scratch_lookup(device);                          // a4nt_lstm_seq_fwd_noalloc's per-device buffer
    if (rcs != A4NT_OK) return rcs;
    A4NT_PROBE(1, 1); A4NT_PROBE(1, 2);
    A4NT_PROBE(2, 1); A4NT_PROBE(2, 2);
    A4NT_PROBE(4, 1); A4NT_PROBE(4, 2);
    // H >= 256: the backward sweep is the 2-D kernel
#define A4NT_PROBE_X(mt, nj) A4NT_PROBE_K(mt, nj, 0, lstm_fwd_kernel<mt A4NT_COMMA nj>, lstm_bwdx_kernel<nj / 4 A4NT_COMMA mt>)
    A4NT_PROBE_X(1, 4); A4NT_PROBE_X(1, 8); A4NT_PROBE_X(1, 16);
    A4NT_PROBE_X(2, 4); A4NT_PROBE_X(2, 8); A4NT_PROBE_X(2, 16);
    A4NT_PROBE_X(4, 4); A4NT_PROBE_X(4, 8); A4NT_PROBE_X(4, 16);
    A4NT_PROBE2(1, 1); A4NT_PROBE2(1, 2);
    A4NT_PROBE2(2, 1); A4NT_PROBE2(2, 2);
#define A4NT_PROBE_X2(mt, nj) A4NT_PROBE_K(mt, nj, 1, lstm_fwd2_kernel<mt A4NT_COMMA nj>, lstm_bwdx2_kernel<nj / 4 A4NT_COMMA mt>)
    A4NT_PROBE_X2(1, 4); A4NT_PROBE_X2(1, 8);
    A4NT_PROBE_X2(2, 4); A4NT_PROBE_X2(2, 8);
    return A4NT_OK;
}
// co-resident workgroups the probe found for the single-sweep instantiation (0 = not probed)
extern "C" int a4nt_lstm_resident_workgroups(int device, int mt, int H) {
    if (device < 0 || device >= A4NT_MAX_DEV || (mt != 1 && mt != 2 && mt != 4) || bad_h(H)) return -1;
    return g_resident[device][0][ilog2(mt)][ilog2(H / 64)];
}

extern "C" int a4nt_lstm_geometry(int B, int H, int* mt_out, int* groups_out) {
    if (B <= 0 || bad_h(H)) { a4nt_set_error("a4nt_lstm: H must be 64, 128, 256, 512 or 1024 (the kernels are instantiated for these), got H=%d B=%d", H, B); return A4NT_ERR_ARG; }
    int G = 0;
    const int mt = pick_mt(B, H / 16, &G);
    if (!mt) { a4nt_set_error("a4nt_lstm: batch %d too large for one sweep at H=%d (%d co-resident workgroups of 64 rows x 16 units at most)", B, H, budget(4, H / 64, 0)); return A4NT_ERR_ARG; }
    *mt_out = mt; *groups_out = G;
    return A4NT_OK;
}

// floats needed for the exchange buffer and unsigned words for the sync block
extern "C" long long a4nt_lstm_xch_floats(int B, int H, int backward) {
    int mt, G;
    if (a4nt_lstm_geometry(B, H, &mt, &G) != A4NT_OK) return -1;
    // rows padded to the largest group size (64): covers the sweep's geometry under any workgroup budget and the
    // decoder step's own choice of rows per workgroup (dec_geometry)
    const long long per = (long long)((B + 63) / 64) * 64 * H;
    return 2 * per * (backward ? (H / 16) : 1);
}
extern "C" int a4nt_lstm_sync_words(int B, int H) {
    int mt, G;
    if (a4nt_lstm_geometry(B, H, &mt, &G) != A4NT_OK) return -1;
    return 32 * G + 32;
}

static bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
// the 2-D backward sweep fetches its step inputs with 16-byte LDS-DMA loads
static bool bwdx_misaligned(const LstmBwdArgs& a) {
    return misaligned(a.gates) || misaligned(a.cs) || (a.c0 && misaligned(a.c0)) || (a.dhs && misaligned(a.dhs));
}
#define A4NT_DISPATCH_BWDX(KERNEL, mt, nt, grid, block, s, a)                                                    \
    do {                                                                                                         \
        if (nt == 1 && mt == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, a);                         \
        else if (nt == 1 && mt == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, a);                    \
        else if (nt == 1) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, s, a);                               \
        else if (nt == 2 && mt == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, a);                    \
        else if (nt == 2 && mt == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, a);                    \
        else if (nt == 2) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, s, a);                               \
        else if (mt == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, s, a);                               \
        else if (mt == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, s, a);                               \
        else hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, s, a);                                            \
    } while (0)

static int launch_seq_fwd(const char* what, int T, int B, int H, const float* gx, const float* gc, const float* whh,
                          const float* h0, const float* c0, const int* lens, float* hs, float* gates, float* cs, float* hN,
                          float* cN, float* xch, unsigned* sync, unsigned* tmo, int mt, int G, hipStream_t s) {
    if (T <= 0 || !gx || !whh || !hs || !xch || !sync || misaligned(whh) || misaligned(xch)) {
        a4nt_set_error("%s: bad arguments (T=%d, null or misaligned pointer)", what, T);
        return A4NT_ERR_ARG;
    }
    // the counters start at zero; the timeout word is sticky (cleared by whoever reports it)
    if (hipMemsetAsync(sync, 0, (size_t)(32 * G) * 4, s) != hipSuccess) return a4nt_check_launch(what);
    LstmFwdArgs a;
    a.T = T; a.B = B; a.H = H; a.G = G; a.NW = H / 16;
    a.gx = gx; a.whh = whh; a.h0 = h0; a.c0 = c0; a.lens = lens;
    a.cg = gc; a.ptab = nullptr; a.tok = nullptr; a.val = nullptr;
    a.hs = hs; a.gates = gates; a.cs = cs; a.hN = hN; a.cN = cN; a.hx = xch; a.sync = sync; a.tmo = tmo; a.mode = 0; a.step0 = 0;
    a.fin = PgFinalArgs();
    dim3 grid(G * a.NW), block(256);
    A4NT_DISPATCH_LSTM(lstm_fwd_kernel, mt, H / 64, grid, block, s, a);
    return a4nt_check_launch(what);
}

extern "C" int a4nt_lstm_seq_fwd(int T, int B, int H, const float* gx, const float* gc, const float* whh, const float* h0,
                                 const float* c0, const int* lens, float* hs, float* gates, float* cs, float* hN,
                                 float* cN, float* xch, unsigned* sync, void* stream) {
    int mt, G;
    int rc = a4nt_lstm_geometry(B, H, &mt, &G);
    if (rc != A4NT_OK) return rc;
    return launch_seq_fwd("a4nt_lstm_seq_fwd", T, B, H, gx, gc, whh, h0, c0, lens, hs, gates, cs, hN, cN, xch, sync,
                          sync ? sync + 32 * G : nullptr, mt, G, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// allocation-free forward sweep (include/a4nt_hip.h: a4nt_lstm_seq_fwd_noalloc)
// ---------------------------------------------------------------------------------------------
// The hand-off buffer and the arrival counters of the sweep live in a file-scope device array instead of a
// caller-provided workspace.  Any single-sweep geometry needs at most
//   2 parities x (rows padded to 16*mt) x H floats <= 2 x (256 workgroups x 64 rows x 16 units) x 4 B = 2 MiB
// of hand-off blocks (the 512-workgroup mode has mt = 1: half of that) and 32 x groups counter words with at most
// 128 groups (H = 64, mt = 1), plus one timeout word.  Every GPU the library runs on holds its own copy of the
// array until the process ends.
#define A4NT_SCRATCH_XCH_FLOATS (2 * 256 * 64 * 16)
#define A4NT_SCRATCH_SYNC_WORDS (32 * 128 + 32)
#define A4NT_SCRATCH_BYTES (A4NT_SCRATCH_XCH_FLOATS * 4 + A4NT_SCRATCH_SYNC_WORDS * 4)
static __device__ __align__(256) unsigned char scratch[A4NT_SCRATCH_BYTES];
// device address of `scratch` per device (the name itself is a host-side stand-in, never a kernel argument)
// (atomic: two host threads may look the same device up at once; both get the same address)
static std::atomic<void*> g_scratch_ptr[A4NT_MAX_DEV];

// `device` must be current.  Called by a4nt_init; the entry point repeats it for a device that was never initialised.
static int scratch_lookup(int device) {
    if (device < 0 || device >= A4NT_MAX_DEV) { a4nt_set_error("a4nt: device index %d beyond the %d this library tracks", device, A4NT_MAX_DEV); return A4NT_ERR_ARG; }
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return A4NT_OK;
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        a4nt_set_error("a4nt: hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        (void)hipGetLastError();
        return A4NT_ERR_LAUNCH;
    }
    g_scratch_ptr[device].store(ptr, std::memory_order_release);
    return A4NT_OK;
}

extern "C" int a4nt_lstm_seq_fwd_noalloc(int T, int B, int H, const float* gx, const float* gc, const float* whh,
                                         const float* h0, const float* c0, const int* lens, float* hs, float* gates,
                                         float* cs, float* hN, float* cN, unsigned* timeout_word, void* stream) {
    int mt, G;
    int rc = a4nt_lstm_geometry(B, H, &mt, &G);
    if (rc != A4NT_OK) return rc;
    const long long need = 2LL * G * (H / 16) * mt * 256;
    if (need > A4NT_SCRATCH_XCH_FLOATS || 32 * G + 32 > A4NT_SCRATCH_SYNC_WORDS) {
        a4nt_set_error("a4nt_lstm_seq_fwd_noalloc: B=%d, H=%d needs %lld hand-off floats and %d groups; the static scratch holds %d floats and %d groups",
                       B, H, need, G, A4NT_SCRATCH_XCH_FLOATS, A4NT_SCRATCH_SYNC_WORDS / 32 - 1);
        return A4NT_ERR_ARG;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return a4nt_check_launch("a4nt_lstm_seq_fwd_noalloc: hipGetDevice");
    rc = scratch_lookup(dev);
    if (rc != A4NT_OK) return rc;
    unsigned char* base = (unsigned char*)g_scratch_ptr[dev].load(std::memory_order_acquire);      // the looked-up device address, never `scratch`
    float* xch = (float*)base;
    unsigned* sync = (unsigned*)(base + (size_t)A4NT_SCRATCH_XCH_FLOATS * 4);
    unsigned* tmo = timeout_word ? timeout_word : sync + (A4NT_SCRATCH_SYNC_WORDS - 32);
    hipStream_t s = (hipStream_t)stream;
    // the array keeps the previous call's contents: counters are cleared by launch_seq_fwd, the hand-off blocks are
    // all written before they are read (parity 0 by the h0 publication, parity 1 by step 0); the internal timeout
    // word (no caller word) must start clear, or a stale 1 would end this sweep early
    if (!timeout_word && hipMemsetAsync(tmo, 0, 4, s) != hipSuccess) return a4nt_check_launch("a4nt_lstm_seq_fwd_noalloc memset");
    return launch_seq_fwd("a4nt_lstm_seq_fwd_noalloc", T, B, H, gx, gc, whh, h0, c0, lens, hs, gates, cs, hN, cN, xch, sync,
                          tmo, mt, G, s);
}

// (A decoder step takes the sweep's geometry.  Larger row tiles -- fewer row groups, so fewer re-reads of W_hh per
// launch -- were measured and are slower: 256 rows as 256 workgroups of 32 rows 55.3 us per decoding step against 53.8
// as 512 workgroups of 16 rows; the step is bound by the latency of a workgroup's own weight fetch, merge and MFMA
// chain, not by the L2 traffic of the fetches.)
static int launch_dec_step(const char* what, int step, int B, int H, const float* cg, const float* ptab, const long long* tok,
                           const float* val, const float* whh, const float* c_prev, float* h_out, float* gates_out,
                           float* c_out, float* xch, const PgFinalArgs& fin, void* stream) {
    int mt, G;
    int rc = a4nt_lstm_geometry(B, H, &mt, &G);
    if (rc != A4NT_OK) return rc;
    // Nothing is handed over inside a decoder step's launch, so no residency rule binds it: at H = 1024 it runs 32
    // rows per workgroup at most (the 64-row instantiation would carry the merge prologue on top of 256 weight
    // registers and 64 accumulator registers, and spill).  A function of (B, H) alone: every step of a generation
    // lays the hand-off buffer out the same way.
    if (H == 1024 && mt == 4) { mt = 2; G = (B + 31) / 32; }
    if (!cg || !ptab || !whh || !h_out || !xch || misaligned(whh) || misaligned(xch)) {
        a4nt_set_error("%s: null or misaligned pointer", what);
        return A4NT_ERR_ARG;
    }
    LstmFwdArgs a;
    a.T = 1; a.B = B; a.H = H; a.G = G; a.NW = H / 16;
    a.gx = nullptr; a.whh = whh; a.h0 = nullptr; a.c0 = c_prev; a.lens = nullptr;
    a.cg = cg; a.ptab = ptab; a.tok = tok; a.val = val;
    a.hs = h_out; a.gates = gates_out; a.cs = c_out; a.hN = nullptr; a.cN = nullptr; a.hx = xch; a.sync = nullptr; a.tmo = nullptr;
    a.mode = 1; a.step0 = step;
    a.fin = fin;
    dim3 grid(G * a.NW), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (mt == 2 && H == 1024) hipLaunchKernelGGL((lstm_dec_kernel<2, 16>), grid, block, 0, s, a);
    else if (mt == 1 && H == 1024) hipLaunchKernelGGL((lstm_dec_kernel<1, 16>), grid, block, 0, s, a);
    else A4NT_DISPATCH_LSTM_TO8(lstm_dec_kernel, mt, H / 64, grid, block, s, a);
    return a4nt_check_launch(what);
}

extern "C" int a4nt_lstm_dec_step(int step, int B, int H, const float* cg, const float* ptab, const long long* tok,
                                  const float* val, const float* whh, const float* c_prev, float* h_out,
                                  float* gates_out, float* c_out, float* xch, void* stream) {
    if (!tok || !val) { a4nt_set_error("a4nt_lstm_dec_step: null tok / val"); return A4NT_ERR_ARG; }
    return launch_dec_step("a4nt_lstm_dec_step", step, B, H, cg, ptab, tok, val, whh, c_prev, h_out, gates_out, c_out, xch,
                           PgFinalArgs(), stream);
}

extern "C" int a4nt_lstm_dec_step_merge(int step, int B, int H, const float* cg, const float* ptab, const float* whh,
                                        const float* c_prev, float* h_out, float* gates_out, float* c_out, float* xch,
                                        int V, const float* part, float tau, int hard, int greedy, float* stats,
                                        long long* tok, float* val_in, float* val_out, int* gen_lens, unsigned char* done,
                                        long long end_c, void* stream) {
    if (V <= 0 || !part || !tok || (!greedy && tau <= 0.f) || (reinterpret_cast<uintptr_t>(part) & 15)) {
        a4nt_set_error("a4nt_lstm_dec_step_merge: bad sampler arguments (V=%d, tau=%g, part 16-byte aligned)", V, (double)tau);
        return A4NT_ERR_ARG;
    }
    PgFinalArgs f;
    f.part = part; f.nstrips = (V + PG_BC - 1) / PG_BC; f.inv_tau = greedy ? 1.f : 1.f / tau; f.greedy = greedy; f.hard = hard;
    f.stats = stats; f.tok = tok; f.val_in = val_in; f.val_out = val_out; f.gen_lens = gen_lens; f.done = done; f.end_c = end_c;
    return launch_dec_step("a4nt_lstm_dec_step_merge", step, B, H, cg, ptab, nullptr, nullptr, whh, c_prev, h_out, gates_out, c_out,
                           xch, f, stream);
}

static int seq_bwd(int T, int B, int H, const float* whh, const float* gates, const float* cs,
                   const float* c0, const int* lens, const float* dhs, const float* dhN,
                   const float* dcN, float* dgates, float* dh0, float* dc0, float* dbias, float* xch, unsigned* sync,
                   void* stream);
extern "C" int a4nt_lstm_seq_bwd(int T, int B, int H, const float* whh, const float* gates, const float* cs,
                                 const float* c0, const int* lens, const float* dhs, const float* dhN,
                                 const float* dcN, float* dgates, float* dh0, float* dc0, float* xch, unsigned* sync,
                                 void* stream) {
    return seq_bwd(T, B, H, whh, gates, cs, c0, lens, dhs, dhN, dcN, dgates, dh0, dc0, nullptr, xch, sync, stream);
}
extern "C" int a4nt_lstm_bwd_bias_rows(int B, int H) {
    int mt, G;
    if (H < 256 || a4nt_lstm_geometry(B, H, &mt, &G) != A4NT_OK) return 0;
    return G;
}
extern "C" int a4nt_lstm_seq_bwd_bias(int T, int B, int H, const float* whh, const float* gates, const float* cs,
                                      const float* c0, const int* lens, const float* dhs, const float* dhN,
                                      const float* dcN, float* dgates, float* dh0, float* dc0, float* dbias_part,
                                      float* xch, unsigned* sync, void* stream) {
    if (dbias_part && H < 256) { a4nt_set_error("a4nt_lstm_seq_bwd_bias: bias partials need H >= 256 (a4nt_lstm_bwd_bias_rows)"); return A4NT_ERR_ARG; }
    return seq_bwd(T, B, H, whh, gates, cs, c0, lens, dhs, dhN, dcN, dgates, dh0, dc0, dbias_part, xch, sync, stream);
}
static int seq_bwd(int T, int B, int H, const float* whh, const float* gates, const float* cs,
                   const float* c0, const int* lens, const float* dhs, const float* dhN,
                   const float* dcN, float* dgates, float* dh0, float* dc0, float* dbias, float* xch, unsigned* sync,
                   void* stream) {
    int mt, G;
    int rc = a4nt_lstm_geometry(B, H, &mt, &G);
    if (rc != A4NT_OK) return rc;
    if (T <= 0 || !whh || !gates || !cs || !dgates || !xch || !sync || misaligned(xch)) {
        a4nt_set_error("a4nt_lstm_seq_bwd: bad arguments (T=%d, null or misaligned pointer)", T);
        return A4NT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, (size_t)(32 * G) * 4, s) != hipSuccess) return a4nt_check_launch("lstm_seq_bwd memset");
    LstmBwdArgs a;
    a.T = T; a.B = B; a.H = H; a.G = G; a.NW = H / 16;
    a.whh = whh; a.gates = gates; a.cs = cs; a.c0 = c0; a.lens = lens; a.dhs = dhs; a.dhN = dhN; a.dcN = dcN;
    a.dgates = dgates; a.dh0 = dh0; a.dc0 = dc0; a.gxch = xch; a.sync = sync; a.tmo = sync + 32 * G;
    a.need_dh0 = (dh0 != nullptr) ? 1 : 0;
    a.dbias = dbias;
    dim3 grid(G * a.NW), block(256);
    if (H >= 256) {
        if (bwdx_misaligned(a)) { a4nt_set_error("a4nt_lstm_seq_bwd: gates / cs / c0 / dhs must be 16-byte aligned at H >= 256"); return A4NT_ERR_ARG; }
        A4NT_DISPATCH_BWDX(lstm_bwdx_kernel, mt, H / 256, grid, block, s, a);
    } else {
        A4NT_DISPATCH_LSTM_SMALL(lstm_bwd_kernel, mt, H / 64, grid, block, s, a);
    }
    return a4nt_check_launch("lstm_seq_bwd");
}

// ---------------------------------------------------------------------------------------------
// two sweeps in one launch
// ---------------------------------------------------------------------------------------------
// (a4nt_lstm_fwd_problem / a4nt_lstm_bwd_problem: include/a4nt_hip.h)

#define A4NT_DISPATCH_LSTM2(KERNEL, mt, nj, grid, block, s, a, b, nA)                                             \
    do {                                                                                                         \
        if (mt == 1 && nj == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, a, b, nA);                  \
        else if (mt == 1 && nj == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, a, b, nA);             \
        else if (mt == 1 && nj == 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, s, a, b, nA);             \
        else if (mt == 1 && nj == 8) hipLaunchKernelGGL((KERNEL<1, 8>), grid, block, 0, s, a, b, nA);             \
        else if (mt == 2 && nj == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, a, b, nA);             \
        else if (mt == 2 && nj == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, a, b, nA);             \
        else if (mt == 2 && nj == 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, s, a, b, nA);             \
        else hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, s, a, b, nA);                                     \
    } while (0)

// Geometry of a pair: both sweeps use the same rows-per-workgroup (16*mt), and together they must fit the
// workgroup budget.  Returns A4NT_ERR_ARG (no error text) when the pair does not fit: run them one by one.
extern "C" int a4nt_lstm_pair_geometry(int B0, int B1, int H, int* mt_out, int* g0_out, int* g1_out) {
    if (B0 <= 0 || B1 <= 0 || bad_h(H) || H > 512) return A4NT_ERR_ARG;
    const int NW = H / 16;
    for (int mt = 1; mt <= 2; mt *= 2) {
        const int g0 = (B0 + 16 * mt - 1) / (16 * mt), g1 = (B1 + 16 * mt - 1) / (16 * mt);
        if ((long long)(g0 + g1) * NW <= budget(mt, NW / 4, 1)) { *mt_out = mt; *g0_out = g0; *g1_out = g1; return A4NT_OK; }
    }
    return A4NT_ERR_ARG;
}
// exchange floats / sync words of one sweep of a pair (mt from a4nt_lstm_pair_geometry)
extern "C" long long a4nt_lstm_pair_xch_floats(int B, int H, int mt, int backward) {
    if (B <= 0 || bad_h(H) || (mt != 1 && mt != 2)) return -1;
    const long long G = (B + 16 * mt - 1) / (16 * mt);
    return 2 * G * (H / 16) * mt * 256 * (backward ? (H / 16) : 1);
}
extern "C" int a4nt_lstm_pair_sync_words(int B, int mt) {
    if (B <= 0 || (mt != 1 && mt != 2)) return -1;
    return 32 * ((B + 16 * mt - 1) / (16 * mt)) + 32;
}

extern "C" int a4nt_lstm_seq_fwd_pair(int H, const a4nt_lstm_fwd_problem* p0, const a4nt_lstm_fwd_problem* p1, void* stream) {
    if (!p0 || !p1) { a4nt_set_error("a4nt_lstm_seq_fwd_pair: null problem"); return A4NT_ERR_ARG; }
    int mt, G[2];
    if (a4nt_lstm_pair_geometry(p0->B, p1->B, H, &mt, &G[0], &G[1]) != A4NT_OK) {
        a4nt_set_error("a4nt_lstm_seq_fwd_pair: %d + %d rows at H=%d do not fit one launch", p0->B, p1->B, H);
        return A4NT_ERR_ARG;
    }
    const a4nt_lstm_fwd_problem* p[2] = {p0, p1};
    LstmFwdArgs a[2];
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < 2; ++i) {
        if (p[i]->T <= 0 || !p[i]->gx || !p[i]->whh || !p[i]->hs || !p[i]->xch || !p[i]->sync || misaligned(p[i]->whh) ||
            misaligned(p[i]->xch) || (i == 1 && (p0->xch == p1->xch || p0->sync == p1->sync))) {
            a4nt_set_error("a4nt_lstm_seq_fwd_pair: bad arguments in problem %d (T=%d, null, misaligned or shared scratch)", i, p[i]->T);
            return A4NT_ERR_ARG;
        }
        if (hipMemsetAsync(p[i]->sync, 0, (size_t)(32 * G[i]) * 4, s) != hipSuccess) return a4nt_check_launch("lstm_seq_fwd_pair memset");
        LstmFwdArgs& q = a[i];
        q.T = p[i]->T; q.B = p[i]->B; q.H = H; q.G = G[i]; q.NW = H / 16;
        q.gx = p[i]->gx; q.whh = p[i]->whh; q.h0 = p[i]->h0; q.c0 = p[i]->c0; q.lens = p[i]->lens;
        q.cg = p[i]->gc; q.ptab = nullptr; q.tok = nullptr; q.val = nullptr;
        q.hs = p[i]->hs; q.gates = p[i]->gates; q.cs = p[i]->cs; q.hN = p[i]->hN; q.cN = p[i]->cN; q.hx = p[i]->xch;
        q.sync = p[i]->sync; q.tmo = p[i]->sync + 32 * G[i]; q.mode = 0; q.step0 = 0;
        q.fin = PgFinalArgs();
    }
    const int nA = G[0] * (H / 16);
    dim3 grid((G[0] + G[1]) * (H / 16)), block(256);
    A4NT_DISPATCH_LSTM2(lstm_fwd2_kernel, mt, H / 64, grid, block, s, a[0], a[1], nA);
    return a4nt_check_launch("lstm_seq_fwd_pair");
}

extern "C" int a4nt_lstm_seq_bwd_pair(int H, const a4nt_lstm_bwd_problem* p0, const a4nt_lstm_bwd_problem* p1, void* stream) {
    if (!p0 || !p1) { a4nt_set_error("a4nt_lstm_seq_bwd_pair: null problem"); return A4NT_ERR_ARG; }
    int mt, G[2];
    if (a4nt_lstm_pair_geometry(p0->B, p1->B, H, &mt, &G[0], &G[1]) != A4NT_OK) {
        a4nt_set_error("a4nt_lstm_seq_bwd_pair: %d + %d rows at H=%d do not fit one launch", p0->B, p1->B, H);
        return A4NT_ERR_ARG;
    }
    const a4nt_lstm_bwd_problem* p[2] = {p0, p1};
    LstmBwdArgs a[2];
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < 2; ++i) {
        if (p[i]->T <= 0 || !p[i]->whh || !p[i]->gates || !p[i]->cs || !p[i]->dgates || !p[i]->xch || !p[i]->sync ||
            misaligned(p[i]->xch) || (i == 1 && (p0->xch == p1->xch || p0->sync == p1->sync))) {
            a4nt_set_error("a4nt_lstm_seq_bwd_pair: bad arguments in problem %d (T=%d, null, misaligned or shared scratch)", i, p[i]->T);
            return A4NT_ERR_ARG;
        }
        if (hipMemsetAsync(p[i]->sync, 0, (size_t)(32 * G[i]) * 4, s) != hipSuccess) return a4nt_check_launch("lstm_seq_bwd_pair memset");
        LstmBwdArgs& q = a[i];
        q.T = p[i]->T; q.B = p[i]->B; q.H = H; q.G = G[i]; q.NW = H / 16;
        q.whh = p[i]->whh; q.gates = p[i]->gates; q.cs = p[i]->cs; q.c0 = p[i]->c0; q.lens = p[i]->lens;
        q.dhs = p[i]->dhs; q.dhN = p[i]->dhN; q.dcN = p[i]->dcN;
        q.dgates = p[i]->dgates; q.dh0 = p[i]->dh0; q.dc0 = p[i]->dc0; q.gxch = p[i]->xch; q.sync = p[i]->sync; q.tmo = p[i]->sync + 32 * G[i];
        q.need_dh0 = (p[i]->dh0 != nullptr) ? 1 : 0;
        q.dbias = H >= 256 ? p[i]->dbias : nullptr;
    }
    const int nA = G[0] * (H / 16);
    dim3 grid((G[0] + G[1]) * (H / 16)), block(256);
    if (H >= 256) {
        if (bwdx_misaligned(a[0]) || bwdx_misaligned(a[1])) { a4nt_set_error("a4nt_lstm_seq_bwd_pair: gates / cs / c0 / dhs must be 16-byte aligned at H >= 256"); return A4NT_ERR_ARG; }
        if (mt == 1 && H == 256) hipLaunchKernelGGL((lstm_bwdx2_kernel<1, 1>), grid, block, 0, s, a[0], a[1], nA);
        else if (mt == 2 && H == 256) hipLaunchKernelGGL((lstm_bwdx2_kernel<1, 2>), grid, block, 0, s, a[0], a[1], nA);
        else if (mt == 1) hipLaunchKernelGGL((lstm_bwdx2_kernel<2, 1>), grid, block, 0, s, a[0], a[1], nA);
        else hipLaunchKernelGGL((lstm_bwdx2_kernel<2, 2>), grid, block, 0, s, a[0], a[1], nA);
    } else {
        if (mt == 1 && H == 64) hipLaunchKernelGGL((lstm_bwd2_kernel<1, 1>), grid, block, 0, s, a[0], a[1], nA);
        else if (mt == 1) hipLaunchKernelGGL((lstm_bwd2_kernel<1, 2>), grid, block, 0, s, a[0], a[1], nA);
        else if (H == 64) hipLaunchKernelGGL((lstm_bwd2_kernel<2, 1>), grid, block, 0, s, a[0], a[1], nA);
        else hipLaunchKernelGGL((lstm_bwd2_kernel<2, 2>), grid, block, 0, s, a[0], a[1], nA);
    }
    return a4nt_check_launch("lstm_seq_bwd_pair");
}
