// Host loops of the free-running decoder in native code (include/a4nt_hip.h: a4nt_gen_loop_fwd / a4nt_gen_loop_bwd).
//
// The generation loop of CharTranslator.forward_advers_gen (reference models/char_translator.py:375-399) and the
// reverse loop of its backward pass are 100 sequential steps of 3 resp. 6 short kernels (10-40 us each).  Driven
// from Python that is 900 launches per training step at 10-20 us of interpreter time each: more than the kernels
// take, so the GPU idles between them on any but the fastest host.  These two entry points run the same per-step
// sequence of the same kernels (the C-ABI entry points of this library, in the same order, on the caller's stream)
// from a C++ loop: ~3 us of host time per launch, the loop stays ahead of the GPU.  No allocation, no
// synchronisation: every buffer is the caller's, the early-exit test stays with the caller (who asks for the
// steps in chunks when an end token is in play).
#include "common.h"

// out (+)= A[M,K] * B[N,K]^T, routed like ops.gemm: the A-stationary kernel for few rows and K <= 512, else a4nt_sgemm
static int gemm_nt_auto(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int beta,
                        void* stream) {
    if (M >= 96 && M <= 256 && N >= 4096 && a4nt_sgemm_nt_rows_ok(M, N, K, lda, ldb) &&
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0)
        return a4nt_sgemm_nt_rows(M, N, K, A, lda, B, ldb, C, ldc, beta, stream);
    return a4nt_sgemm(0, 1, M, N, K, A, lda, B, ldb, C, ldc, nullptr, beta, 1, 0, stream);
}

extern "C" int a4nt_gen_loop_fwd(const a4nt_gen_fwd_args* a, int t0, int t1, void* stream) {
    if (!a || t0 < 0 || t1 > a->n_max || t0 > t1 || a->B <= 0 || a->H <= 0 || a->V <= 0 || !a->cg || !a->ptab || !a->tok0 ||
        !a->ones || !a->whh || !a->hs_all || !a->gates || !a->cs || !a->xch || !a->W || !a->part || !a->stats || !a->tok ||
        !a->val_in || !a->val_out || !a->gen_lens || !a->done) {
        a4nt_set_error("a4nt_gen_loop_fwd: bad arguments (steps [%d,%d))", t0, t1);
        return A4NT_ERR_ARG;
    }
    const long long B = a->B, H = a->H;
    for (int t = t0; t < t1; ++t) {
        const float* cp = t == 0 ? nullptr : a->cs + (t - 1) * B * H;
        float* h_t = a->hs_all + (long long)(t + 1) * B * H;
        int rc;
        if (t == t0) {
            // the step before this chunk (or the start token) is final in memory
            const long long* tk = t == 0 ? a->tok0 : a->tok + (t - 1) * B;
            const float* vl = t == 0 ? a->ones : a->val_in + (t - 1) * B;
            rc = a4nt_lstm_dec_step(t, a->B, a->H, a->cg, a->ptab, tk, vl, a->whh, cp, h_t, a->gates + t * B * 4 * H,
                                    a->cs + t * B * H, a->xch, stream);
        } else {
            // the sampler of step t-1 is finished in this step's prologue (no launch of its own)
            rc = a4nt_lstm_dec_step_merge(t, a->B, a->H, a->cg, a->ptab, a->whh, cp, h_t, a->gates + t * B * 4 * H,
                                          a->cs + t * B * H, a->xch, a->V, a->part, a->tau, a->hard, a->greedy,
                                          a->stats + (t - 1) * B * 2, a->tok + (t - 1) * B, a->val_in + (t - 1) * B,
                                          a->val_out + (t - 1) * B, a->gen_lens, a->done, a->end_c, stream);
        }
        if (rc != A4NT_OK) return rc;
        const float* U = (a->U && t < a->n_noise) ? a->U[t] : nullptr;
        float* z = a->z ? a->z + (long long)t * a->zrows * a->ldz : nullptr;
        const bool last = t == t1 - 1;                 // the chunk's last step is finalised by a launch of its own
        rc = a4nt_proj_gumbel_fwd(a->B, a->V, a->H, h_t, a->H, a->W, a->ldw, a->bias, a->scale, a->tau, a->hard, a->greedy, U,
                                  U ? a->ldu : 0, a->seed, a->step0 + (unsigned)t, z, z ? a->ldz : 0, a->zrow0,
                                  z ? a->zrows : 0, a->part, last ? a->stats + t * B * 2 : nullptr, last ? a->tok + t * B : nullptr,
                                  last ? a->val_in + t * B : nullptr, last ? a->val_out + t * B : nullptr,
                                  last ? a->gen_lens : nullptr, last ? a->done : nullptr, a->end_c, stream);
        if (rc != A4NT_OK) return rc;
    }
    return A4NT_OK;
}

extern "C" int a4nt_gen_loop_bwd(const a4nt_gen_bwd_args* a, void* stream) {
    if (!a || a->n <= 0 || a->B <= 0 || a->H <= 0 || a->E <= 0 || a->V <= 0 || !a->S || !a->D || !a->y || !a->stats ||
        !a->emb_W || !a->W_out || !a->Wcat || !a->gates || !a->cs || !a->dc || !a->dG || !a->rec || !a->dh_slabs ||
        a->ks_v < 1 || a->ks_r < 1 || (a->ks_r > 1 && !a->rec_slabs) || (a->fold && !a->abuf)) {
        a4nt_set_error("a4nt_gen_loop_bwd: bad arguments");
        return A4NT_ERR_ARG;
    }
    const long long B = a->B, H = a->H, E = a->E, HE = H + E;
    const int n = a->n;
    for (int t = n - 1; t >= 0; --t) {
        float* S_t = a->S + (long long)t * B * a->lds;
        float* D_t = a->D + (long long)t * B * a->ldd;
        const float* rec_next = t < n - 1 ? a->rec + (long long)(t + 1) * B * HE : nullptr;
        int rc;
        // S_t += (d(emb_{t+1}) + fold_t) * E^T: the sum is in abuf (written with rec_{t+1}); the last step has fold alone
        const float* A_t = nullptr; int lda_t = 0;
        if (a->fold) { A_t = rec_next ? a->abuf : a->fold + (long long)t * B * E; lda_t = (int)E; }
        else if (rec_next) { A_t = rec_next + H; lda_t = (int)HE; }
        if (A_t) {
            rc = gemm_nt_auto(a->B, a->V, a->E, A_t, lda_t, a->emb_W, a->E, S_t, a->lds, 1, stream);
            if (rc != A4NT_OK) return rc;
        }
        rc = a4nt_gumbel_bwd(a->B, a->V, S_t, a->lds, a->y + (long long)t * B * a->ldy, a->ldy, a->stats + t * B * 2, a->coef,
                             a->inv_tau, D_t, a->ldd, stream);
        if (rc != A4NT_OK) return rc;
        // dh_t (split-K partials) = dlogits_t * W_out^T
        rc = a4nt_sgemm(0, 1, a->B, a->H, a->V, D_t, a->ldd, a->W_out, a->ldw, a->dh_slabs, a->H, nullptr, 0, a->ks_v,
                        a->ks_v > 1 ? B * H : 0, stream);
        if (rc != A4NT_OK) return rc;
        rc = a4nt_lstm_cell_bwd(a->B, a->H, a->gates + t * B * 4 * H, a->cs + t * B * H, t > 0 ? a->cs + (t - 1) * B * H : nullptr,
                                a->dh_slabs, a->ks_v, B * H, rec_next, (int)HE, a->dc, a->dG + t * B * 4 * H, stream);
        if (rc != A4NT_OK) return rc;
        // rec_t = dgates_t * [W_hh | W_ie]
        float* rec_t = a->rec + (long long)t * B * HE;
        if (a->ks_r > 1) {
            rc = a4nt_sgemm(0, 0, a->B, (int)HE, 4 * a->H, a->dG + t * B * 4 * H, 4 * a->H, a->Wcat, (int)HE, a->rec_slabs, (int)HE,
                            nullptr, 0, a->ks_r, B * HE, stream);
            if (rc != A4NT_OK) return rc;
            if (a->fold && t > 0)
                rc = a4nt_slab_reduce2(a->ks_r, B * HE, a->B, (int)HE, a->rec_slabs, (int)HE, rec_t, (int)HE, (int)H,
                                       a->fold + (long long)(t - 1) * B * E, (int)E, a->abuf, (int)E, stream);
            else
                rc = a4nt_slab_reduce(a->ks_r, B * HE, a->B, (int)HE, a->rec_slabs, (int)HE, 0, nullptr, 0, rec_t, (int)HE, stream);
        } else if (a->B <= 256 && HE >= 2048) {      // few rows x wide weight: the LDS-free kernel (as ops.gemm routes it)
            rc = a4nt_sgemm_skinny(0, a->B, (int)HE, 4 * a->H, a->dG + t * B * 4 * H, 4 * a->H, a->Wcat, (int)HE, rec_t, (int)HE,
                                   nullptr, 0, 1, 0, stream);
        } else {
            rc = a4nt_sgemm(0, 0, a->B, (int)HE, 4 * a->H, a->dG + t * B * 4 * H, 4 * a->H, a->Wcat, (int)HE, rec_t, (int)HE, nullptr,
                            0, 1, 0, stream);
        }
        if (rc != A4NT_OK) return rc;
        if (a->fold && t > 0 && a->ks_r <= 1) {      // rec_t was written by the product itself: the sum takes a launch of its own
            rc = a4nt_slab_reduce2(1, 0, a->B, (int)HE, rec_t, (int)HE, rec_t, (int)HE, (int)H,
                                   a->fold + (long long)(t - 1) * B * E, (int)E, a->abuf, (int)E, stream);
            if (rc != A4NT_OK) return rc;
        }
    }
    return A4NT_OK;
}
