/*
 * a4nt_hip.h -- C ABI of liba4nt_hip.so: the hand-written gfx950 (MI355X) kernels behind the
 * A4NT adversarial-training hot path.
 *
 * The reference (rakshithShetty/A4NT-author-masking) has no FFI layer: its hot path is PyTorch
 * ops called from four Python model files.  Each entry point below therefore cites the reference
 * call site(s) whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory the caller owns (the Python host allocates it
 *     with torch and passes tensor.data_ptr()); the library never allocates or frees;
 *   - matrices are row-major fp32 with an explicit leading dimension (in elements);
 *   - `stream` is a hipStream_t; kernels are enqueued on it and the call returns immediately;
 *   - return value: 0 = ok, 1 = bad argument, 2 = HIP launch error, 3 = persistent-kernel
 *     hand-off timeout; a4nt_last_error() returns the message of the calling thread's last error;
 *   - token ids are int64 (torch.LongTensor), lengths int32.
 */
#ifndef A4NT_HIP_H
#define A4NT_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* a4nt_last_error(void);
int a4nt_version(void);
/* Verifies `device` is gfx950 with 256 CUs and asks the occupancy calculator
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor) how many workgroups of every persistent LSTM sweep
 * instantiation can be resident at once; a4nt_lstm_geometry never hands out more than that, so a sweep
 * whose workgroups wait for each other is refused at geometry time instead of spinning into its timeout.
 * Replaces nothing in the reference (it hard-codes .cuda(), char_lstm.py:98). */
int a4nt_init(int device, int* num_cus);

/* C[M,N] (+)= op(A) * op(B) (+ bias[N]) on v_mfma_f32_32x32x2_f32.
 *   transA=0: A is [M][K];  transA=1: A is stored [K][M].   transB=0: B is [K][N]; transB=1: [N][K].
 *   beta in {0,1}.  ksplit>1 writes ksplit partial slabs at C + z*slab_stride (beta must be 0).
 * Replaces: x.mm(W) of the vocab projection (char_translator.py:279,377), the soft-embedding
 * contractions (char_translator.py:214,299,329,381; char_lstm.py:289; char_cnn.py:73), the
 * x*W_ih^T half of nn.LSTM (char_translator.py:182-197; char_lstm.py:148-163), nn.Conv1d as
 * shifted GEMMs' weight gradients (char_cnn.py:84), the classifier heads (char_lstm.py:329; char_cnn.py:96) and
 * every autograd backward of those. */
int a4nt_sgemm(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
               float* C, int ldc, const float* bias, int beta, int ksplit, long long slab_stride, void* stream);
/* Workgroup tile edge (128 or 64) a4nt_sgemm uses for this shape; lets the caller label a launch for
 * profiling.  Nothing in the reference corresponds (torch picks its BLAS kernel internally). */
int a4nt_sgemm_tile(int M, int N, int ksplit);
/* k-groups (sets of 4 waves, each walking its own share of the workgroup's K range) of the 64x64-tile
 * launches: 0 = chosen per launch (default), 1 / 2 / 4 = pinned, for benchmarks.  Results differ only
 * in the fp32 summation order over K.  Nothing in the reference corresponds. */
int a4nt_sgemm_set_kgroups(int n);

/* C[M,N] (+)= A[M,K] * B[N,K]^T for few rows and 16 <= K <= 512 (K % 16 == 0, A and B rows 16-byte aligned):
 * A stays in registers (16 rows x K per wave), 16-column blocks of B stream through LDS once per 128 rows.
 * Replaces: the d(sample) += d(emb) * E^T step of autograd's backward of samp.mm(embedding) inside the
 * sampling loop (char_translator.py:381-385), one decoder step at a time.
 * a4nt_sgemm_nt_rows_ok tells whether a shape qualifies (otherwise use a4nt_sgemm). */
int a4nt_sgemm_nt_rows_ok(int M, int N, int K, int lda, int ldb);
int a4nt_sgemm_nt_rows(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       int beta, void* stream);

/* C[M,N] (+)= A[M,K] * op(B) (+ bias) for few rows (M = the batch of ONE decoder step): no LDS, one wave
 * per 16-row x 64/80-column tile and K slab, v_mfma_f32_16x16x4_f32 fed from a register ring.
 * transB=0: B is [K][N]; transB=1: B is stored [N][K].  ksplit>1 writes slabs at C + z*slab_stride
 * (beta must be 0); a4nt_sgemm_skinny_ksplit suggests the split that fills the chip's 1024 SIMDs.
 * Replaces, per generated token: emb = samp.mm(char_emb.weight) backward (char_translator.py:381),
 * dec_out = p_rnn.mm(W) backward (:377) and the decoder nn.LSTM step's backward (:398). */
int a4nt_sgemm_skinny_ksplit(int M, int N, int K);
int a4nt_sgemm_skinny(int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                      int ldc, const float* bias, int beta, int ksplit, long long slab_stride, void* stream);

/* Geometry of the persistent LSTM sweeps for a batch of B rows and hidden size H
 * (H in {64, 128, 256, 512, 1024}): rows per group = 16*mt, groups*(H/16) workgroups <= the budget of the
 * instantiation: 512 (two workgroups per CU) for mt = 1 and H <= 512, 256 otherwise, cut to what a4nt_init
 * found co-resident on the current device. */
int a4nt_lstm_geometry(int B, int H, int* mt_out, int* groups_out);
/* 512 (default: two workgroups per CU where the instantiation allows it; needs the GPU to itself) or
 * 256 (one per CU for every instantiation: leaves room for other streams' kernels, e.g. RCCL).  A setting of
 * the calling thread's current device, read at launch time by every sweep entry point: change it from one host
 * thread per device at a time (two threads launching sweeps on one device while one of them switches the budget
 * see each other's value). */
int a4nt_lstm_set_max_workgroups(int n);
/* Co-resident workgroups a4nt_init found for the <mt, H> single-sweep kernels on `device` (0: not probed). */
int a4nt_lstm_resident_workgroups(int device, int mt, int H);
long long a4nt_lstm_xch_floats(int B, int H, int backward);   /* exchange-buffer size (floats): rows padded to 64, so
                                                                 * one buffer serves every rows-per-group the budget may pick */
int a4nt_lstm_sync_words(int B, int H);                       /* sync-block size (uint32 words) */

/* Persistent forward sweep of ONE LSTM layer over T steps (W_hh resident in VGPRs, recurrent
 * contraction on v_mfma_f32_16x16x4_f32, fused gate non-linearity / cell update).
 *   gx    [T][B][4H]  x_t*W_ih^T + b_ih + b_hh (gate order i,f,g,o), from a4nt_sgemm
 *   gc    [B][4H] or NULL: loop-invariant term added to every step's gates (the decoder's
 *                     ctxt*W_ic^T, char_translator.py:258, hoisted out of the time loop)
 *   whh   [4H][H]     16-byte aligned
 *   h0,c0 [B][H] or NULL (zeros);  lens [B] or NULL (every row runs T steps; rows need no order)
 *   hs    [T][B][H]   outputs, zero where t >= lens[b] (pad_packed_sequence semantics)
 *   gates [T][B][4H]  post-activation gates saved for BPTT (may alias gx) or NULL
 *   cs    [T][B][H]   cell states saved for BPTT or NULL
 *   hN,cN [B][H]      state at each row's own last step or NULL
 *   xch, sync         scratch sized by a4nt_lstm_xch_floats(B,H,0) / a4nt_lstm_sync_words
 * Replaces: nn.LSTM over a PackedSequence, one layer (char_translator.py:175-198 as used at
 * :221,:266,:305,:335; char_lstm.py:142-164 as used at :296). */
int a4nt_lstm_seq_fwd(int T, int B, int H, const float* gx, const float* gc, const float* whh, const float* h0, const float* c0,
                      const int* lens, float* hs, float* gates, float* cs, float* hN, float* cN, float* xch,
                      unsigned* sync, void* stream);

/* Allocation-free variant of a4nt_lstm_seq_fwd.  No counterpart in the reference and none among the workspace
 * arguments above: it is for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same arguments and results (bit for bit) as a4nt_lstm_seq_fwd
 * without `xch` / `sync`: the hand-off buffer and the arrival counters live in a static device array inside the
 * library (2 MiB + 16.5 KiB per GPU, held until the process ends), which covers every geometry
 * a4nt_lstm_geometry accepts (rows padded to the group size x H <= 262144, at most 128 row groups); anything
 * larger is refused with A4NT_ERR_ARG.  The call allocates and frees nothing.
 *   timeout_word  device word the sweep sets to 1 when a bounded hand-off wait expired (sticky; the caller
 *                 zeroes and reads it), or NULL: an internal word is used and nothing is reported.
 * Every call on a GPU shares that GPU's array: calls on one stream are ordered and safe (each call clears the
 * counters it needs); calls that overlap on one GPU (two streams, two threads) race on it. */
int a4nt_lstm_seq_fwd_noalloc(int T, int B, int H, const float* gx, const float* gc, const float* whh, const float* h0,
                              const float* c0, const int* lens, float* hs, float* gates, float* cs, float* hN, float* cN,
                              unsigned* timeout_word, void* stream);

/* Persistent backward sweep (BPTT) of one layer.  dgates [T][B][4H] receives the gradient w.r.t.
 * the pre-activation gates (zero where t >= lens[b]; may alias gates); weight / input gradients
 * are time-batched a4nt_sgemm calls over it.  dhs / dhN / dcN may be NULL.  dh0 NULL skips the
 * last recurrent product.  xch sized by a4nt_lstm_xch_floats(B,H,1).
 * At H >= 256 gates, cs, c0 and dhs must be 16-byte aligned (the sweep fetches its step inputs with 16-byte
 * LDS-DMA loads; A4NT_ERR_ARG otherwise).  All arrays are addressed with 64-bit arithmetic: T*B*4H*4 bytes may
 * exceed 4 GiB.
 * Replaces: autograd backward of the nn.LSTM calls cited above. */
int a4nt_lstm_seq_bwd(int T, int B, int H, const float* whh, const float* gates, const float* cs, const float* c0,
                      const int* lens, const float* dhs, const float* dhN, const float* dcN, float* dgates,
                      float* dh0, float* dc0, float* xch, unsigned* sync, void* stream);
/* The same sweep that also leaves the bias gradient's partial sums: dbias_part [a4nt_lstm_bwd_bias_rows(B, H)][4H] =
 * per row group, the column sums of dgates over all steps and the group's rows (the caller adds the few rows: that IS
 * d(b_ih) = d(b_hh)), accumulated from the dgates tiles while they are in LDS -- no pass over the [T*B, 4H] matrix.
 * H >= 256 (the 2-D sweep kernels); a4nt_lstm_bwd_bias_rows returns 0 where the partials are not available.
 * Replaces: the bias part of nn.LSTM's BPTT (models/char_translator.py:182-197, models/char_lstm.py:148-163). */
int a4nt_lstm_bwd_bias_rows(int B, int H);
int a4nt_lstm_seq_bwd_bias(int T, int B, int H, const float* whh, const float* gates, const float* cs,
                           const float* c0, const int* lens, const float* dhs, const float* dhN, const float* dcN,
                           float* dgates, float* dh0, float* dc0, float* dbias_part, float* xch, unsigned* sync,
                           void* stream);

/* Two independent sweeps in ONE launch: workgroups of both problems are co-resident, so one sweep's MFMA
 * phase runs under the other's hand-off waits (two 128-row sweeps at H = 512 each fill half of the 512
 * resident workgroups).  Fields as the arguments of a4nt_lstm_seq_fwd / a4nt_lstm_seq_bwd; every problem has
 * its own xch / sync scratch, sized by a4nt_lstm_pair_xch_floats / a4nt_lstm_pair_sync_words with the mt
 * a4nt_lstm_pair_geometry returns (it fails, without an error text, when the pair does not fit one launch or
 * H > 512: run the sweeps one by one then).  Results are those of the single-sweep entry points, bit for bit.
 * Nothing in the reference corresponds to the pairing itself (its nn.LSTM calls run one after another):
 * the pair is the discriminator's layer over the generated batch next to the reconstruction encoder /
 * decoder layer over the same batch in the generator update (adv_step.g_update). */
typedef struct {
    int T, B;
    const float* gx; const float* gc; const float* whh; const float* h0; const float* c0; const int* lens;
    float* hs; float* gates; float* cs; float* hN; float* cN; float* xch; unsigned* sync;
} a4nt_lstm_fwd_problem;
typedef struct {
    int T, B;
    const float* whh; const float* gates; const float* cs; const float* c0; const int* lens;
    const float* dhs; const float* dhN; const float* dcN;
    float* dgates; float* dh0; float* dc0; float* xch; unsigned* sync;
    float* dbias;       /* [groups of this problem][4H] bias-gradient partials (H >= 256) or NULL: a4nt_lstm_seq_bwd_bias */
} a4nt_lstm_bwd_problem;
int a4nt_lstm_pair_geometry(int B0, int B1, int H, int* mt_out, int* g0_out, int* g1_out);
long long a4nt_lstm_pair_xch_floats(int B, int H, int mt, int backward);
int a4nt_lstm_pair_sync_words(int B, int mt);
int a4nt_lstm_seq_fwd_pair(int H, const a4nt_lstm_fwd_problem* p0, const a4nt_lstm_fwd_problem* p1, void* stream);
int a4nt_lstm_seq_bwd_pair(int H, const a4nt_lstm_bwd_problem* p0, const a4nt_lstm_bwd_problem* p1, void* stream);
/* Synchronises `stream` and reports A4NT_ERR_TIMEOUT if a sweep raised its timeout word (the word after the
 * `groups` counters of the sync block).  The word is sticky: launches clear the counters only, so a timeout
 * of any earlier sweep on this sync block is still visible; the caller clears it after reporting. */
int a4nt_lstm_check_timeout(const unsigned* sync, int groups, void* stream);

/* One step of the free-running decoder (first layer): gates = cg[b] + val[b]*ptab[tok[b]] +
 * h_{t-1}*W_hh^T, fused cell update.  h_{t-1} is read from / h_t published to xch (fragment
 * layout, parity = step&1; zero xch before step 0).  c_prev NULL = zeros.
 * Replaces: the single-step nn.LSTM call inside the generation loop (char_translator.py:360,398)
 * with the loop-invariant ctxt part hoisted (cg) and emb*W_ie^T tabulated per token (ptab). */
int a4nt_lstm_dec_step(int step, int B, int H, const float* cg, const float* ptab, const long long* tok,
                       const float* val, const float* whh, const float* c_prev, float* h_out, float* gates_out,
                       float* c_out, float* xch, void* stream);
/* The same step with the PREVIOUS step's sampler finished in its prologue: a4nt_proj_gumbel_fwd(tok = NULL) left
 * the per-strip partials of that step in `part`; every workgroup merges the strips of its own 16 rows (the
 * arithmetic of a4nt_proj_gumbel_finalize) to get the token and the straight-through value it needs, and the first
 * workgroup of each row group writes the step's results (stats, tok, val_in, val_out, gen_lens, done: the
 * arguments of a4nt_proj_gumbel_finalize, all for step - 1).  Saves one launch per decoding step.
 * Replaces: the same lines as a4nt_lstm_dec_step plus the bookkeeping of char_translator.py:384-392. */
int a4nt_lstm_dec_step_merge(int step, int B, int H, const float* cg, const float* ptab, const float* whh,
                             const float* c_prev, float* h_out, float* gates_out, float* c_out, float* xch, int V,
                             const float* part, float tau, int hard, int greedy, float* stats, long long* tok,
                             float* val_in, float* val_out, int* gen_lens, unsigned char* done, long long end_c,
                             void* stream);

/* The free-running decoder's host loops in native code: steps [t0, t1) of the generation loop
 * (a4nt_lstm_dec_step + a4nt_proj_gumbel_fwd per step) and the whole reverse loop of its backward pass (per step,
 * last to first: S_t += d(emb_{t+1}) E^T; a4nt_gumbel_bwd; dlogits_t W_out^T in ks_v split-K slabs;
 * a4nt_lstm_cell_bwd; dgates_t [W_hh | W_ie] in ks_r slabs + a4nt_slab_reduce).  Exactly the kernel sequence the
 * per-step entry points give, issued from a C++ loop (about 3 us of host time per launch instead of the 10-20 us a
 * Python wrapper costs: 900 launches per training step), on `stream`; nothing is allocated or synchronised.
 * All arrays are the caller's, time major and contiguous unless a leading dimension is given:
 *   forward : hs_all [n_max+1][B][H] (row 0 = zeros), gates [n_max][B][4H], cs [n_max][B][H], stats [n_max][B][2],
 *             tok / val_in / val_out [n_max][B], z [n_max][zrows][ldz] or NULL, U = host array of n_noise device
 *             pointers ([B][ldu] uniforms per step) or NULL; the other fields as in a4nt_lstm_dec_step /
 *             a4nt_proj_gumbel_fwd
 *   backward: S, D [n][B][lds / ldd] (d(sample) in, dlogits out), y [n][B][ldy], rec [n][B][H+E] (out),
 *             dG [n][B][4H] (out), dc [B][H] (zeros in), dh_slabs [ks_v][B][H], rec_slabs [ks_r][B][H+E] scratch
 * Replaces: the Python-level loop of models/char_translator.py:375-399 and autograd's walk back through it. */
typedef struct {
    int B, H, V, n_max;
    const float* cg; const float* ptab; const long long* tok0; const float* ones; const float* whh;
    float* hs_all; float* gates; float* cs; float* xch;
    const float* W; int ldw; const float* bias; float scale, tau; int hard, greedy;
    const float* const* U; int n_noise, ldu;
    unsigned long long seed; unsigned step0;
    float* z; int ldz, zrow0, zrows;
    float* part; float* stats; long long* tok; float* val_in; float* val_out; int* gen_lens; unsigned char* done;
    long long end_c;
} a4nt_gen_fwd_args;
typedef struct {
    int n, B, H, E, V;
    float* S; int lds; float* D; int ldd; const float* y; int ldy; const float* stats; float coef, inv_tau;
    const float* emb_W; const float* W_out; int ldw; const float* Wcat;
    const float* gates; const float* cs; float* dc; float* dG; float* rec;
    float* dh_slabs; int ks_v; float* rec_slabs; int ks_r;
    /* consumers of the samples that embedded them with emb_W itself (the reconstruction encoder): their d(embedding)
     * [n][B][E], summed, joins d(emb input of step t+1) BEFORE the contraction with emb_W^T instead of being contracted
     * on its own, time-batched (S_t += (rec_{t+1}[:, H:] + fold_t) emb_W^T); NULL = none.  abuf: [B][E] scratch. */
    const float* fold; float* abuf;
} a4nt_gen_bwd_args;
int a4nt_gen_loop_fwd(const a4nt_gen_fwd_args* a, int t0, int t1, void* stream);
int a4nt_gen_loop_bwd(const a4nt_gen_bwd_args* a, void* stream);

/* Steps [t0, t1) of the generation loop as ONE persistent launch (csrc/genpersist.hip): the decoder's W_hh slices stay
 * in registers for the whole range instead of being re-read by every decoder-step launch, the projection's first
 * weight blocks are in flight when h_t arrives, and the recurrent product of step t+1 runs under the projection of
 * step t; steps are chained by per-row-pair arrival counters (bounded waits; the sticky timeout word is
 * sync[32*16], as for the sweeps).  Same arithmetic per element as a4nt_gen_loop_fwd: bit-identical results.
 * Limits (a4nt_gen_persist_ok): B <= 256, H in {64, 128, 256, 512}, at most 256 workgroups (V <= 10240 above 128
 * rows), a->W = the packed image of a4nt_proj_gumbel_pack (a->ldw = 0), Philox or greedy sampling (a->U = NULL).
 * xch: a4nt_gen_persist_xch_floats(B, H) floats, zeroed before step 0 (the hand-off layout differs from
 * a4nt_lstm_dec_step's: do not mix the two paths inside one generation); sync: a4nt_gen_persist_sync_words() words,
 * zeroed once by the caller (launches clear the counters).  One launch at a time per (xch, sync), every workgroup
 * must be resident: nothing else may occupy the device's CUs meanwhile.
 * Replaces: the same lines as a4nt_gen_loop_fwd (models/char_translator.py:375-399). */
int a4nt_gen_persist_ok(int B, int H, int V);
long long a4nt_gen_persist_xch_floats(int B, int H);
int a4nt_gen_persist_sync_words(void);
int a4nt_gen_persist_fwd(const a4nt_gen_fwd_args* a, int t0, int t1, float* xch, unsigned* sync, void* stream);

/* Binary cross-entropy with logits of column `col` of logits [N][C] against the constant `target` (0 or 1), optional
 * per-sample weights [N], mean over N -> loss[0]; backward: dlogits [N][C] = grad_loss[0] * d(loss)/d(logits) (zeros
 * outside the column).  One launch each way.
 * Replaces: nn.BCEWithLogitsLoss on eval_out[:, targ_aid] (train_adversarial_translator.py:487-489, :690). */
int a4nt_bce_logits_fwd(int N, int C, int col, const float* logits, float target, const float* weight, float* loss,
                        void* stream);
int a4nt_bce_logits_bwd(int N, int C, int col, const float* logits, float target, const float* weight,
                        const float* grad_loss, float* dlogits, void* stream);

/* Optimizer step of an update as multi-tensor launches (csrc/optim.hip).  `table`: device array of
 * {float* p, float* g, float* sq, long long n} rows, one per tensor (32 bytes each); `map`: device int pairs
 * (tensor row, chunk index inside it), one per workgroup, chunks of a4nt_mt_chunk_elems() elements.
 *   a4nt_mt_sumsq     partial[c] = sum of g^2 over chunk c (fixed order)
 *   a4nt_mt_clip_coef out2[0] = sqrt(sum of partial) = the global 2-norm, out2[1] = min(1, max_norm / (norm + 1e-6))
 *   a4nt_mt_scale     g *= coef2[1] where it is below 1 (what clip_grad_norm_ leaves in .grad); no traffic otherwise
 *   a4nt_mt_rmsprop   (g += wd*p;) sq = alpha*sq + (1-alpha)*g*g; p -= lr * g / (sqrt(sq) + eps)
 * Nothing is read back by the host.  Replaces: torch.nn.utils.clip_grad_norm and the two torch.optim.RMSprop steps of
 * train_adversarial_translator.py:306-312, :542, :730-731 (one launch per elementary operation there). */
int a4nt_mt_chunk_elems(void);
int a4nt_mt_sumsq(int nchunks, const void* table, const int* map, float* partial, void* stream);
int a4nt_mt_clip_coef(int nchunks, const float* partial, float max_norm, float* out2, void* stream);
int a4nt_mt_scale(int nchunks, const void* table, const int* map, const float* coef2, void* stream);
int a4nt_mt_rmsprop(int nchunks, const void* table, const int* map, float lr, float alpha, float eps, float weight_decay,
                    void* stream);

/* Element-wise cell backward of one decoder step (reverse loop of the generator):
 * dh = sum of n_slabs split-K partials [B][H] (slab_stride floats apart) + dh_b [B][ldb] (may be NULL). */
int a4nt_lstm_cell_bwd(int B, int H, const float* gates, const float* c_t, const float* c_prev, const float* dh_slabs,
                       int n_slabs, long long slab_stride, const float* dh_b, int ldb, float* dc_io, float* dgates,
                       void* stream);

/* Gumbel-softmax sampling row kernel: scale, log_softmax, Gumbel noise (injected uniforms U or
 * Philox(seed, step, row, v)), /tau, softmax statistics, argmax, straight-through one-hot value,
 * gen_lens / done bookkeeping; greedy=1: plain argmax of the logits.
 * Replaces: sample_gumbel + gumbel_softmax_sample (char_translator.py:10-28) and the loop body
 * char_translator.py:379-392. */
int a4nt_gumbel_fwd(int B, int V, const float* logits, int ldl, const float* U, int ldu, unsigned long long seed,
                    unsigned step, float scale, float tau, int hard, int greedy, float* y, int ldy, float* stats,
                    long long* tok, float* val_in, float* val_out, int* gen_lens, unsigned char* done,
                    long long end_c, float* dense, int ldd, void* stream);
/* out = dlogits = coef * p * (ds - <p,ds>), p = softmax(y) from the saved y / stats.  out may be ds (in
 * place: one workgroup per row); a separate out lets four workgroups share a row.
 * Replaces: autograd backward of gumbel_softmax_sample (char_translator.py:17-28). */
int a4nt_gumbel_bwd(int B, int V, const float* ds, int ldd, const float* y, int ldy, const float* stats, float coef,
                    float inv_tau, float* out, int ldo, void* stream);

/* Fused output projection + Gumbel-softmax sampling of one free-running decoder step:
 * z = scale*(h W + b) + gumbel(u) (greedy: z = h W + b), per-row arg max / softmax(z/tau) statistics,
 * straight-through value, done / gen_lens bookkeeping.  One wave per 16-row x 80-column output tile
 * (v_mfma_f32_16x16x4_f32 fed from registers, no LDS), strips merged by a finalize kernel.  K must be a
 * multiple of 16, h 16-byte aligned with ldh a multiple of 4.  z [zrows][ldz] holds rows zrow0.. of the
 * batch (the rows whose backward pass will run; zrow0 = 0, zrows = M for all) and is saved for
 * a4nt_gumbel_bwd(inv_tau = 1/tau) (may be NULL); part is scratch of a4nt_proj_gumbel_part_floats().
 * Replaces: dec_out = p_rnn.mm(W)+b; gumbel_softmax_sample(dec_out*scale, tau); argmax; done/len
 * updates (char_translator.py:377-392) -- softmax((log_softmax(x)+g)/tau) == softmax((x+g)/tau). */
/* W may be given as the packed image a4nt_proj_gumbel_pack writes (pass ldw = 0): per (80-column strip, 16-k block)
 * the B operands of every lane are contiguous, so the kernel reads 5 aligned 16-byte words per lane and block
 * instead of gathering 20-byte pieces from rows whose stride is not a multiple of 16 bytes.  Pack once per
 * generation (the weights do not change inside it); a4nt_proj_gumbel_packed_floats gives the image size. */
long long a4nt_proj_gumbel_packed_floats(int K, int N);
int a4nt_proj_gumbel_pack(int K, int N, const float* W, int ldw, float* Wp, void* stream);
long long a4nt_proj_gumbel_part_floats(int M, int N);
int a4nt_proj_gumbel_fwd(int M, int N, int K, const float* h, int ldh, const float* W, int ldw, const float* bias,
                         float scale, float tau, int hard, int greedy, const float* U, int ldu,
                         unsigned long long seed, unsigned step, float* z, int ldz, int zrow0, int zrows, float* part,
                         float* stats,
                         long long* tok, float* val_in, float* val_out, int* gen_lens, unsigned char* done,
                         long long end_c, void* stream);
/* The finalisation alone (merge of the strips in `part` + the bookkeeping above).  a4nt_proj_gumbel_fwd runs it
 * unless tok is NULL; then the caller finishes the step later, with this call or inside the next decoder step
 * (a4nt_lstm_dec_step_merge).  N = the projection's column count (it fixes the number of strips). */
int a4nt_proj_gumbel_finalize(int M, int N, const float* part, float tau, int hard, int greedy, float* stats,
                              long long* tok, float* val_in, float* val_out, int* gen_lens, unsigned char* done,
                              long long end_c, void* stream);

/* Log-softmax + NLL over N rows of logits that some other producer wrote (no statistics available): leaves
 * (softmax - onehot)*gscale in place of the logits when write_grad.  target < 0 ignores the row.
 * Replaces: nn.CrossEntropyLoss over packed scores (train_translator.py:112-113;
 * train_adversarial_translator.py:661-664) and its backward. */
int a4nt_nll_rows(int N, int V, float* logits, int ld, const long long* target, float gscale, float* loss_rows,
                  int write_grad, void* stream);

/* Fused output projection + log-softmax / NLL.  a4nt_sgemm_nll: C[M,N] = A[M,K] * B[K,N] + bias on 128x128 MFMA
 * tiles whose epilogue, besides writing the logits once, reduces per row the (max, sum exp) of each wave's 64
 * columns while they are still in the accumulators: part [M][a4nt_sgemm_nll_parts(N)][2].  a4nt_nll_grad_rows then
 * merges a row's partials into its log-sum-exp, writes loss_rows[n] = lse - logit[target[n]] and, with write_grad,
 * turns the row into (softmax - onehot) * gscale in place: one read and one write of the [M,N] logits in total, no
 * pass of its own for the loss.  target < 0 ignores the row (loss 0, gradient 0).
 * Replaces: dec_out.mm(decoder_W) + b -> nn.CrossEntropyLoss over the packed scores and its backward seed
 * (char_translator.py:279-284; train_translator.py:112-113; train_adversarial_translator.py:661-664). */
int a4nt_sgemm_nll_parts(int N);
int a4nt_sgemm_nll(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, float* part, void* stream);
int a4nt_nll_grad_rows(int N, int V, float* logits, int ld, const long long* target, const float* part, int nparts,
                       float gscale, float* loss_rows, int write_grad, void* stream);

/* Embedding gather out[n] = scale[n]*W[ids[n]] and its scatter-add backward (padding_idx row skipped).
 * Replaces: nn.Embedding lookups (char_translator.py:212,250,327,349,389; char_lstm.py:280;
 * char_cnn.py:71) and, with scale = straight-through sample value, samp.mm(char_emb.weight) for
 * one-hot samples (char_translator.py:381; char_lstm.py:289). */
int a4nt_embed_fwd(long long N, int E, int V, const long long* ids, const float* scale, const float* W, float* out,
                   int ldo, void* stream);
int a4nt_embed_bwd(long long N, int E, int V, const long long* ids, const float* scale, const float* dout, int ldd,
                   float* dW, long long padding_idx, void* stream);

/* out[n] (+)= sum_m A[m][n]  (bias gradients: the .sum(0) autograd takes for an nn.Linear / nn.LSTM bias,
 * char_translator.py:279, char_lstm.py:148).  Deterministic: row chunks write partial sums into `ws`
 * (a4nt_colsum_ws_floats(M, N) floats of device memory, contents undefined before and after) and a second
 * launch adds them in chunk order.  ws may be null or smaller: fewer chunks, slower, the same kind of result. */
long long a4nt_colsum_ws_floats(int M, int N);
int a4nt_colsum(int M, int N, const float* A, int lda, float* out, int accumulate, float* ws, long long ws_floats,
                void* stream);
/* Weight and bias gradient of a linear layer in one pass: C[M,N] = A^T * B with A stored [K][M] and B stored [K][N]
 * (the transA=1, transB=0 case of a4nt_sgemm on 128x128 tiles, same ksplit / slab_stride meaning), and
 * colpart[z][n] = sum over K slab z of B[k][n], taken from the B tiles as they are staged for the product, so the
 * [K,N] matrix (the d(logits) of the output layer: K = tokens, N = vocabulary) is not read a second time for
 * a4nt_colsum.  colpart holds ksplit * N floats; the caller adds the ksplit rows (a4nt_colsum over [ksplit][N]).
 * Shapes whose output is a few tiles only (a4nt_sgemm_tile(M, N, ksplit) != 128) are refused: A4NT_ERR_ARG.
 * Replaces: autograd's backward of dec_out.mm(decoder_W) + decoder_b for W and b together (char_translator.py:279,377). */
int a4nt_sgemm_tn_colsum(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         int ksplit, long long slab_stride, float* colpart, void* stream);
/* out[r][c] = sum_s slabs[s*stride + r*lds + c0 + c] (+ add[r][c])  (split-K reduction). */
int a4nt_slab_reduce(int S, long long stride, int R, int C, const float* slabs, int lds, int c0, const float* add,
                     int lda, float* out, int ldo, void* stream);
/* The same reduction with a second output: out2[r][c - c2] = out[r][c] + add2[r][c - c2] for the columns c >= c2 (the
 * reverse loop of the generator: rec_t and, in the same launch, the A operand of the next d(sample) product). */
int a4nt_slab_reduce2(int S, long long stride, int R, int C, const float* slabs, int lds, float* out, int ldo, int c2,
                      const float* add2, int lda2, float* out2, int ldo2, void* stream);

/* Conv1d over time-major activations, all K taps accumulated in the MFMA accumulators of one launch, with bias,
 * leaky_relu, placement into a (zero-padded / cropped, concatenated) output and the max over time in the epilogue.
 *   x [T*B][ldx] (row t*B + b, C channels)      w [K][C][F] taps (nn.Conv1d weight [F][C][K] re-laid)
 *   y[t][b][f] = bias[f] + sum_k sum_c x[t + k - pad][b][c] * w[k][c][f],  t in [0, Tout = T + 2*pad - K + 1)
 *   (pad < 0 crops: rows outside [0, T) simply read as zero)      act = 1: y = leaky_relu(y, slope)
 *   out (may be NULL): steps [t_lo, t_hi) are written at row ((t - t_lo + t_off)*B + b), columns [0, F), leading
 *       dimension ldo (point it at a column slice of the concatenated layer output); beta = 1 accumulates
 *   maxkey (may be NULL): [B][F] 64-bit keys (a4nt_maxkey_init before, a4nt_maxkey_decode after): running max over
 *       steps [t_lo, t_hi) of y as (ordered value bits, step) -- order independent, hence deterministic.
 * With the taps flipped and transposed (w'[k][f][c] = W[f][c][K-1-k], pad' = K-1-pad, x = d(pre)) the same call
 * computes the data gradient.  Replaces: nn.Conv1d + FN.leaky_relu + FN.pad + torch.cat + FN.max_pool1d
 * (char_cnn.py:83-92) and the conv-over-hidden-states head (char_lstm.py:75,324), and their autograd backward. */
int a4nt_conv1d_tm(int T, int B, int C, int F, int K, int pad, const float* x, int ldx, const float* w,
                   const float* bias, int act, float slope, float* out, int ldo, int t_lo, int t_hi, int t_off,
                   int beta, unsigned long long* maxkey, void* stream);
/* keys = "nothing yet" (with_pad = 0) or "a zero-pad position" (value 0, wins ties; with_pad = 1). */
int a4nt_maxkey_init(int B, int F, unsigned long long* keys, int with_pad, void* stream);
/* feat[b][f] = winning value, argmax[b][f] = winning step or -1 when a zero-pad position won. */
int a4nt_maxkey_decode(int B, int F, const unsigned long long* keys, float* feat, int ldf, int* argmax, void* stream);
/* dpre [Tout][B][ldp] = 0 except dpre[argmax[b][f]][b][f] = dfeat[b][f] * (feat[b][f] > 0 ? 1 : slope):
 * backward of leaky_relu + max over time from the decoded keys (leaky_relu keeps the sign).
 * Replaces: autograd backward of FN.max_pool1d(FN.leaky_relu(.)) (char_cnn.py:84-92). */
int a4nt_maxtime_scatter_bwd(int Tout, int B, int F, const float* dfeat, int ldd, const float* feat, int ldf,
                             const int* argmax, float slope, float* dpre, int ldp, void* stream);

/* CharCNN epilogue: leaky_relu + (zero-padded) max over time of [Tout][B][F] conv pre-activations,
 * and its backward (in place over pre).  Replaces: FN.leaky_relu + FN.pad + FN.max_pool1d
 * (char_cnn.py:84-92). */
int a4nt_lrelu_maxtime(int Tout, int B, int F, const float* pre, int ldp, float slope, int pad, float* feat, int ldf,
                       int* argmax, void* stream);
int a4nt_lrelu_maxtime_bwd(int Tout, int B, int F, float* pre, int ldp, float slope, const float* dfeat, int ldf,
                           const int* argmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif
