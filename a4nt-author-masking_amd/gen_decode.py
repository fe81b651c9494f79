"""Free-running generator loop of CharTranslator.forward_advers_gen as ONE fused autograd operator,
and the compact representation of its straight-through samples.

Reference: models/char_translator.py:347-401 (loop), :10-28 (Gumbel-softmax).

What is fused / restructured for MI355X (numerically equivalent within fp32 rounding):
  * the decoder input of every step is [ctxt, emb]; ctxt is constant, so ctxt*W_ic^T (+ both biases)
    is computed once (cg) and emb*W_ie^T is tabulated per token once per call (ptab = E*W_ie^T), so
    a step is: gather + recurrent MFMA + fused cell update (a4nt_lstm_dec_step);
  * with gumbel_hard the sample is one_hot*val (val = (1-p)+p at the arg-max), so samples travel
    as (token, value) pairs (SoftSamples) and `samp.mm(embedding)` in any consumer is a scaled gather;
  * the backward never materialises d(sample): consumers register their d(embedding) [T,B,E] and
    embedding matrix with the operator, whose reverse loop contracts them over the vocabulary
    inside the per-step Gumbel-softmax backward (a4nt_sgemm + a4nt_gumbel_bwd);
  * no host sync per step: `done` / `gen_lens` live on the device; the host looks at them every
    CHECK_EVERY steps (reading the answer one period later, without draining the GPU) and once at the end (the
    reference syncs every step, :393).
"""
import torch
from torch.autograd import Function

from . import ops
from .ops import gemm, gemm_auto, slab_reduce, colsum, embed_fwd, embed_bwd

CHECK_EVERY = 16
PACK_W = True             # re-lay the output projection out for the fused kernel once per generation
# kernel labels of ops.TIMER that live inside the two loops: when bench.py times one of them the loops are driven
# from Python (per-launch events); otherwise from the library's C++ loops (a4nt_gen_loop_fwd / _bwd): same kernels,
# same order, a fraction of the host time per launch
_LOOP_LABELS = ('proj_gumbel', 'sgemm_nt_64', 'sgemm_nn_64', 'sgemm_nt_rows', 'skinny_nn', 'skinny_nt')


PY_LOOPS = False          # benchmarks: drive the loops from Python even when nothing is being timed
FOLD_SAME_EMBEDDING = True   # backward: see _GenDecode.backward (False: every consumer contracted time-batched, as in r02)
# The forward loop as one persistent launch (a4nt_gen_persist_fwd, csrc/genpersist.hip).  False (default): the per-step
# launches -- measured on the box (tools/gen_fwd_time.py, 256 rows, H = 512, V = 10001): 54.5 us per step for the
# persistent launch against 50.6 for the two launches per step; DESIGN.md section 7a has the phase stamps.  True:
# whenever the shape fits; 'auto': above 128 rows.  Same results bit for bit either way (tests/test_hip_persist.py).
PERSIST = False
_persist_ws = {}


def _persist_workspace(dev, B, H):
    """Hand-off buffer and counters of the persistent generation kernel (one per device and shape)."""
    from . import _lib
    key = (dev.index, B, H)
    if key not in _persist_ws:
        lib = _lib.load()
        xch = torch.zeros(int(lib.a4nt_gen_persist_xch_floats(B, H)), device=dev)
        sync = torch.zeros(int(lib.a4nt_gen_persist_sync_words()), device=dev, dtype=torch.int32)
        _persist_ws[key] = (xch, sync)
        ops._sync_blocks.append((sync, 16))        # its timeout word is reported by ops.check_timeouts like a sweep's
    return _persist_ws[key]


def _native_loops():
    return not PY_LOOPS and (ops.TIMER is None or ops.TIMER['name'] not in _LOOP_LABELS)


class _GenHandle(object):
    """Side channel between the generation operator and the consumers of its samples."""

    def __init__(self):
        self.contribs = []        # (demb [n,B,E], W [V,E]) registered by consumers' backward
        self.dense_grads = []     # dense d(sample) [n,B,V] (API path that materialised the samples)
        self.grad_filter = False  # GradFilter of the reference driver (train_adversarial_translator.py:25-38)


class SoftSamples(object):
    """Straight-through one-hot samples as (tok [T,B] int64, val [T,B] fp32).

    Row t, column b stands for the dense row  val[t,b] * one_hot(tok[t,b])  of the reference's
    [T,B,V] sample tensor (val is 0 for rows generated after the sequence finished,
    char_translator.py:384).  `origin` = (handle, t_offset, perm): row t >= t_offset, column b came
    from generator step t - t_offset, generator batch column perm[b].
    """

    def __init__(self, tok, val, vocab, origin=None, const_rows=0, vocab_map=None):
        self.tok, self.val, self.vocab, self.origin = tok, val, vocab, origin
        self.const_rows = const_rows      # leading rows that are constant one-hot rows (prepend_token)
        # [V, V'] matrix applied to every row before a consumer sees it (samples.mm(vocab_map), the vocabulary
        # map of the language-model loss): consumers embed with vocab_map @ W, tokens stay in this vocabulary
        self.vocab_map = vocab_map

    @property
    def shape(self):
        return (self.tok.shape[0], self.tok.shape[1], self.vocab)

    def size(self, i):
        return self.shape[i]

    def detach(self):
        return SoftSamples(self.tok, self.val.detach(), self.vocab, None)

    def index_select_batch(self, idx):
        """samples.index_select(1, idx) of the reference (train_adversarial_translator.py:143)."""
        origin = None
        if self.origin is not None:
            h, off, perm = self.origin
            origin = (h, off, idx if perm is None else perm.index_select(0, idx))
        return SoftSamples(self.tok.index_select(1, idx), self.val.index_select(1, idx), self.vocab, origin,
                           self.const_rows)

    def prepend_token(self, token_id):
        """torch.cat([one_hot(start), samples]) of the reference (train_adversarial_translator.py:158)."""
        B = self.tok.shape[1]
        t0 = torch.full((1, B), int(token_id), dtype=torch.int64, device=self.tok.device)
        v0 = torch.ones((1, B), dtype=torch.float32, device=self.val.device)
        origin = None
        if self.origin is not None:
            h, off, perm = self.origin
            origin = (h, off + 1, perm)
        return SoftSamples(torch.cat([t0, self.tok]), torch.cat([v0, self.val]), self.vocab, origin,
                           self.const_rows + 1)

    def set_grad_filter(self):
        """The reference's GradFilter on this sample tensor: its backward keeps, per batch column, only the
        time step whose gradient has the largest absolute sum over the vocabulary."""
        if self.origin is not None:
            self.origin[0].grad_filter = True
        return self

    def rows(self, r0, r1, n=None, differentiable=True):
        """Batch columns [r0, r1) and the first n steps.  With differentiable=True the result keeps this
        object's origin, whose frame must be exactly those columns (forward_advers_gen(grad_rows=(r0, r1)))."""
        n = self.tok.shape[0] if n is None else n
        val = self.val[:n, r0:r1] if differentiable else self.val[:n, r0:r1].detach()
        return SoftSamples(self.tok[:n, r0:r1].contiguous(), val.contiguous(), self.vocab,
                           self.origin if differentiable else None)

    def to_dense(self):
        return _DenseFromCompact.apply(self.val, self.tok, self.vocab, self.origin)


def _register(origin, demb):
    """demb: [T',B',E] gradient w.r.t. val*W[tok] in the (shifted, permuted) frame of a SoftSamples."""
    handle, off, perm, W = origin
    g = demb[off:]
    if perm is not None:
        full = torch.empty_like(g)
        full.index_copy_(1, perm, g)
        g = full
    handle.contribs.append((g.contiguous(), W))


class _SampleEmbed(Function):
    """emb[t,b,:] = val[t,b] * W[tok[t,b],:]  ==  samples.view(T*B,V).mm(W) for one-hot samples
    (char_translator.py:214,329,381; char_lstm.py:289; char_cnn.py:73)."""

    @staticmethod
    def forward(ctx, val, tok, W, origin):
        ctx.save_for_backward(val, tok, W)
        ctx.origin = origin
        return embed_fwd(tok, W, val.contiguous())

    @staticmethod
    def backward(ctx, demb):
        val, tok, W = ctx.saved_tensors
        demb = demb.contiguous()
        dW = None
        if ctx.needs_input_grad[2]:
            dW = torch.zeros_like(W)
            embed_bwd(tok, demb, dW, val.contiguous(), -1)     # a dense mm has no padding_idx
        if ctx.origin is not None and ctx.needs_input_grad[0]:
            _register(ctx.origin + (W,), demb)
        dval = torch.zeros_like(val) if ctx.needs_input_grad[0] else None
        return dval, None, dW, None


def sample_embed(samples, W):
    if samples.vocab_map is not None:
        W = ops.linear_kn(samples.vocab_map, W)
    return _SampleEmbed.apply(samples.val, samples.tok, W, samples.origin)


def mapped(samples, vocab_map):
    """samples.mm(vocab_map) for every row, kept compact."""
    assert samples.vocab_map is None
    return SoftSamples(samples.tok, samples.val, vocab_map.shape[1], samples.origin, samples.const_rows, vocab_map)


class _DenseFromCompact(Function):
    """Materialise the reference's dense [T,B,V] sample tensor (API compatibility path)."""

    @staticmethod
    def forward(ctx, val, tok, vocab, origin):
        T, B = tok.shape
        dense = torch.zeros((T, B, vocab), device=val.device, dtype=torch.float32)
        dense.view(T * B, vocab).scatter_(1, tok.reshape(-1, 1), val.reshape(-1, 1))
        ctx.origin = origin
        return dense

    @staticmethod
    def backward(ctx, dd):
        if ctx.origin is not None:
            handle, off, perm = ctx.origin
            g = dd[off:]
            if perm is not None:
                full = torch.empty_like(g)
                full.index_copy_(1, perm, g)
                g = full
            handle.dense_grads.append(g.contiguous())
        return torch.zeros(dd.shape[:2], device=dd.device), None, None, None


class _GenDecode(Function):
    @staticmethod
    def forward(ctx, ctxt, tok0, emb_W, w_ih, w_hh, b_ih, b_hh, W_out, b_out, cfg, holder):
        from . import _lib
        lib = _lib.load()
        dev = ctxt.device
        B, Hc = ctxt.shape
        H = w_hh.shape[1]
        V, E = emb_W.shape
        n_max = int(cfg['n_max'])
        greedy = bool(cfg['greedy'])
        need_grad = bool(cfg['need_grad']) and not greedy
        # rows [r0, r1) are the ones whose backward pass will run (default: all); the others are generated
        # in the same launches but nothing is kept for them (adv_step.fused_update: the discriminator
        # update's no-grad batch and the generator update's batch share one decoding loop)
        r0, r1 = cfg.get('grad_rows') or (0, B)
        noise = cfg.get('noise')          # list of [B,V] uniforms or None
        w_ic = w_ih[:, :Hc]
        w_ie = w_ih[:, Hc:].contiguous()
        w_hh_c = w_hh.contiguous()
        cg = gemm(ctxt.contiguous(), w_ic, transB=True, bias=(b_ih + b_hh))            # [B,4H]
        ptab = gemm(emb_W, w_ie, transB=True)                                          # [V,4H]
        xch, _, _ = ops._lstm_ws(dev, B, H, False)
        xch.zero_()
        hs_all = torch.zeros((n_max + 1, B, H), device=dev)
        gates = torch.empty((n_max, B, 4 * H), device=dev)
        cs = torch.empty((n_max, B, H), device=dev)
        # z = scale*logits + gumbel, saved per step for the backward pass only
        ybuf = torch.empty((n_max, r1 - r0, V), device=dev) if need_grad else None
        stats = torch.empty((n_max, B, 2), device=dev)
        tok = torch.zeros((n_max, B), dtype=torch.int64, device=dev)
        val_in = torch.ones((n_max, B), device=dev)
        val_out = torch.zeros((n_max, B), device=dev)
        gen_lens = torch.zeros(B, dtype=torch.int32, device=dev)
        done = torch.zeros(B, dtype=torch.uint8, device=dev)
        part = torch.empty(int(lib.a4nt_proj_gumbel_part_floats(B, V)), device=dev)
        # the projection matrix re-laid out for the fused kernel's loads, once for all n_max steps
        W_packed = ops.proj_gumbel_pack(W_out) if (PACK_W and n_max >= 4 and H % 16 == 0) else None
        ones = torch.ones(B, device=dev)
        tok0 = tok0.contiguous()
        s = ops._s()
        ran = 0
        pending = None
        seed, step0 = int(cfg.get('seed', 0)), int(cfg.get('step0', 0))
        U_list = None
        if noise is not None and not greedy:
            U_list = [u.contiguous() for u in noise[:n_max]]     # injected uniforms (parity tests); Philox beyond the list
            for U in U_list:
                assert tuple(U.shape) == (B, V)
        native = _native_loops()
        if native:
            import ctypes
            fa = _lib.GenFwdArgs()
            fa.B, fa.H, fa.V, fa.n_max = B, H, V, n_max
            fa.cg, fa.ptab, fa.tok0, fa.ones, fa.whh = ops._p(cg), ops._p(ptab), ops._p(tok0), ops._p(ones), ops._p(w_hh_c)
            fa.hs_all, fa.gates, fa.cs, fa.xch = ops._p(hs_all), ops._p(gates), ops._p(cs), ops._p(xch)
            Wk = W_out if W_packed is None else W_packed
            if W_packed is None:
                Wk = ops._rowmajor(W_out)
            fa.W, fa.ldw, fa.bias = ops._p(Wk), (ops._ld(Wk) if W_packed is None else 0), ops._p(b_out)
            fa.scale, fa.tau, fa.hard, fa.greedy = float(cfg['scale']), float(cfg['tau']), int(cfg['hard']), int(greedy)
            if U_list:
                arr = (ctypes.c_void_p * len(U_list))(*[u.data_ptr() for u in U_list])
                fa.U, fa.n_noise, fa.ldu = arr, len(U_list), V
            else:
                fa.U, fa.n_noise, fa.ldu = None, 0, 0
            fa.seed, fa.step0 = seed, step0
            fa.z, fa.ldz, fa.zrow0, fa.zrows = (ops._p(ybuf), V, r0, r1 - r0) if need_grad else (None, 0, r0, 0)
            fa.part, fa.stats, fa.tok, fa.val_in, fa.val_out = ops._p(part), ops._p(stats), ops._p(tok), ops._p(val_in), ops._p(val_out)
            fa.gen_lens, fa.done, fa.end_c = ops._p(gen_lens), ops._p(done), int(cfg['end_c'])
        persist = None
        if native and PERSIST and W_packed is not None and not U_list and (PERSIST != 'auto' or B > 128) \
                and lib.a4nt_gen_persist_ok(B, H, V):
            persist = _persist_workspace(dev, B, H)
            persist[0].zero_()
        chunk = n_max if cfg['end_c'] < 0 else CHECK_EVERY
        for c0 in range(0, n_max, chunk):
            c1 = min(n_max, c0 + chunk)
            if persist is not None:
                _lib.check(lib.a4nt_gen_persist_fwd(ctypes.byref(fa), c0, c1, ops._p(persist[0]), ops._p(persist[1]), s),
                           'a4nt_gen_persist_fwd')
            elif native:
                _lib.check(lib.a4nt_gen_loop_fwd(ctypes.byref(fa), c0, c1, s), 'a4nt_gen_loop_fwd')
            else:
                for t in range(c0, c1):
                    tk = tok0 if t == 0 else tok[t - 1]
                    vl = ones if t == 0 else val_in[t - 1]
                    cp = None if t == 0 else cs[t - 1]
                    _lib.check(lib.a4nt_lstm_dec_step(t, B, H, ops._p(cg), ops._p(ptab), ops._p(tk), ops._p(vl),
                                                      ops._p(w_hh_c), ops._p(cp), ops._p(hs_all[t + 1]), ops._p(gates[t]),
                                                      ops._p(cs[t]), ops._p(xch), s), 'a4nt_lstm_dec_step')
                    U = U_list[t] if (U_list is not None and t < len(U_list)) else None
                    ops.proj_gumbel(hs_all[t + 1], W_out, b_out, U, seed, step0 + t, cfg['scale'], cfg['tau'], cfg['hard'],
                                    greedy, ybuf[t] if need_grad else None, part, stats[t], tok[t], val_in[t], val_out[t],
                                    gen_lens, done, cfg['end_c'], zrow0=r0, packed=W_packed)
            ran = c1
            if cfg['end_c'] >= 0 and ran < n_max:
                # "all sentences ended?" is read one period late: the answer asked for CHECK_EVERY steps ago has
                # long arrived, so the host never waits for the GPU to drain (steps run past the end are trimmed
                # below, as the steps between two checks already were)
                if pending is not None:
                    pending[0].synchronize()
                    if bool(pending[1][0]):
                        break
                flag = torch.empty(1, dtype=torch.bool, pin_memory=True)
                flag.copy_(done.all().view(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending = (ev, flag)
        # the one read-back of a generation: every later consumer of the lengths (the length sort, the row split of
        # fused_update) takes this host copy (ops.host_lens) instead of draining the GPU again.  Without an end token
        # (end_c < 0, the reference's default) no row can finish early: every length is n_max, known without asking
        # the GPU, and the host keeps running ahead of it through the whole training step.
        if cfg['end_c'] < 0:
            gl_host = torch.full((B,), ran, dtype=torch.int32)
        else:
            gl_host = gen_lens.cpu()
        holder['lens_host'] = gl_host
        n_out = int(gl_host.max())                  # = steps the reference loop would have run
        n_out = max(1, min(n_out, ran))
        if noise is not None and not greedy:
            # the reference stops drawing noise when its loop breaks: consume exactly n_out draws
            assert n_out <= len(noise), 'injected noise list shorter than the generated length'
            del noise[:n_out]
        holder['handle'] = _GenHandle() if need_grad else None
        holder['n_out'] = n_out
        ctx.handle = holder['handle']
        ctx.cfg = dict(cfg, noise=None)
        ctx.rows = (r0, r1, B)
        if need_grad:
            if (r0, r1) == (0, B):
                ctx.n = n_out
                ctx.save_for_backward(ctxt, tok0, emb_W, w_ih, w_hh_c, W_out, hs_all, gates, cs, ybuf, stats, tok, val_in,
                                      val_out)
            else:
                # keep only the rows that will be differentiated, as dense [.., r1-r0, ..] tensors
                ctx.n = n_g = max(1, min(int(gl_host[r0:r1].max()), ran))
                sl = lambda t_: t_[:, r0:r1].contiguous()
                ctx.save_for_backward(ctxt[r0:r1].contiguous(), tok0[r0:r1].contiguous(), emb_W, w_ih, w_hh_c, W_out,
                                      sl(hs_all[:n_g + 1]), sl(gates[:n_g]), sl(cs[:n_g]), ybuf, sl(stats[:n_g]),
                                      sl(tok[:n_g]), sl(val_in[:n_g]), sl(val_out[:n_g]))
        tok_ret = tok[:n_out].clone()
        ctx.mark_non_differentiable(tok_ret, gen_lens)
        return val_out[:n_out].clone(), tok_ret, gen_lens

    @staticmethod
    def backward(ctx, _dval, _dtok, _dlens):
        from . import _lib
        lib = _lib.load()
        (ctxt, tok0, emb_W, w_ih, w_hh, W_out, hs_all, gates, cs, ybuf, stats, tok, val_in, val_out) = ctx.saved_tensors
        cfg, n, handle = ctx.cfg, ctx.n, ctx.handle
        r0, r1, B_all = ctx.rows
        dev = ctxt.device
        B, Hc = ctxt.shape
        H = w_hh.shape[1]
        V, E = emb_W.shape
        s = ops._s()
        # d(sample) from every consumer, contracted over its embedding matrix, time-batched
        S2 = ops.padded_rows(n * B, V, dev)             # rows 16-byte aligned for the time-batched GEMMs
        S = S2.unflatten(0, (n, B))
        first = True
        contribs = [(demb[:n].reshape(n * B, -1), W) for demb, W in handle.contribs]
        # Consumers that embedded the samples with THIS decoder's embedding matrix (the reconstruction encoder): their
        # d(embedding) joins the decoder's own d(embedding input of step t+1) before the contraction with emb_W^T in the
        # reverse loop -- (a + b) E^T instead of a E^T + b E^T -- so the time-batched [n*B, V] x K product below loses
        # their K columns (131 GFLOP of 262 at cfg 3).  Not with the GradFilter, which looks at S before the loop.
        fold = None
        if FOLD_SAME_EMBEDDING and not handle.grad_filter:
            same = [d for d, W in contribs if W.data_ptr() == emb_W.data_ptr() and W.shape == emb_W.shape and W.is_contiguous()]
            if same:
                contribs = [(d, W) for d, W in contribs if not (W.data_ptr() == emb_W.data_ptr() and W.shape == emb_W.shape
                                                                  and W.is_contiguous())]
                fold = same[0] if len(same) == 1 else torch.stack(same).sum(0)
                fold = fold.contiguous().view(n, B, E)
        if len(contribs) > 1 and all(W.shape[0] == V for _, W in contribs):
            # sum_i demb_i W_i^T = [demb_1 | demb_2 | ..] [W_1 | W_2 | ..]^T: one product over the concatenated
            # embedding dimensions instead of one per consumer, each re-reading and re-writing the [n*B, V] buffer
            contribs = [(torch.cat([d for d, _ in contribs], dim=1), torch.cat([W for _, W in contribs], dim=1))]
        for demb2, W in contribs:
            gemm(demb2, W, transB=True, out=S2, beta=0 if first else 1)
            first = False
        for dd in handle.dense_grads:
            if first:
                S.copy_(dd[:n])
            else:
                S.add_(dd[:n])
            first = False
        if first:
            S.zero_()
        handle.contribs, handle.dense_grads = [], []
        if handle.grad_filter:
            _, top = S.abs().sum(dim=-1).topk(1, dim=0)                       # [1,B]: one time step per column
            S.mul_(torch.zeros((n, B), device=dev).scatter_(0, top, 1.0).unsqueeze(-1))
        w_ic = w_ih[:, :Hc]
        w_ie = w_ih[:, Hc:].contiguous()
        Wcat = torch.cat([w_hh, w_ie], dim=1).contiguous()             # [4H, H+E]
        # dlogits of every step (separate from S so that several workgroups can share a row in the Gumbel backward)
        D2 = ops.padded_rows(n * B, V, dev)
        D = D2.unflatten(0, (n, B))
        coef = float(cfg['scale']) / float(cfg['tau'])
        dc = torch.zeros((B, H), device=dev)
        dG = torch.empty((n, B, 4 * H), device=dev)
        # rec[t] = dgates_t * [W_hh | W_ie]: columns [0,H) = dh flowing into step t-1, [H,H+E) = d(emb input of step t)
        rec = torch.empty((n, B, H + E), device=dev)
        ks_v = max(1, min(16, ops._auto_ksplit(B, H, V)))
        ks_r = max(1, ops._auto_ksplit(B, H + E, 4 * H))
        dh_slabs = torch.empty((ks_v, B, H), device=dev)
        rec_slabs = torch.empty((ks_r, B, H + E), device=dev) if ks_r > 1 else None
        if _native_loops():
            import ctypes
            ba = _lib.GenBwdArgs()
            emb_c, Wo_c = ops._rowmajor(emb_W), ops._rowmajor(W_out)
            ba.n, ba.B, ba.H, ba.E, ba.V = n, B, H, E, V
            ba.S, ba.lds, ba.D, ba.ldd = ops._p(S2), ops._ld(S2), ops._p(D2), ops._ld(D2)
            ba.y, ba.ldy, ba.stats = ops._p(ybuf), V, ops._p(stats)
            ba.coef, ba.inv_tau = coef, 1.0 / float(cfg['tau'])
            assert ops._ld(emb_c) == E and ybuf.is_contiguous() and stats.is_contiguous() and gates.is_contiguous() and cs.is_contiguous()
            ba.emb_W, ba.W_out, ba.ldw, ba.Wcat = ops._p(emb_c), ops._p(Wo_c), ops._ld(Wo_c), ops._p(Wcat)
            ba.gates, ba.cs, ba.dc, ba.dG, ba.rec = ops._p(gates), ops._p(cs), ops._p(dc), ops._p(dG), ops._p(rec)
            ba.dh_slabs, ba.ks_v, ba.rec_slabs, ba.ks_r = ops._p(dh_slabs), ks_v, ops._p(rec_slabs), ks_r
            abuf = torch.empty((B, E), device=dev) if fold is not None else None
            ba.fold, ba.abuf = ops._p(fold), ops._p(abuf)
            _lib.check(lib.a4nt_gen_loop_bwd(ctypes.byref(ba), s), 'a4nt_gen_loop_bwd')
        else:
            for t in range(n - 1, -1, -1):
                if fold is not None:
                    gemm(fold[t] if t == n - 1 else rec[t + 1][:, H:] + fold[t], emb_W, transB=True, out=S[t], beta=1)
                elif t < n - 1:
                    gemm(rec[t + 1][:, H:], emb_W, transB=True, out=S[t], beta=1)      # += d(emb_{t+1}) * E^T
                ops.gumbel_bwd_rows(S[t], ybuf[t], stats[t], coef, 1.0 / float(cfg['tau']), out=D[t])   # dlogits_t
                gemm(D[t], W_out, transB=True, out=dh_slabs if ks_v > 1 else dh_slabs[0], ksplit=ks_v)
                _lib.check(lib.a4nt_lstm_cell_bwd(B, H, ops._p(gates[t]), ops._p(cs[t]), ops._p(cs[t - 1] if t > 0 else None),
                                                  ops._p(dh_slabs), ks_v, B * H,
                                                  ops._p(rec[t + 1]) if t < n - 1 else None, H + E,
                                                  ops._p(dc), ops._p(dG[t]), s), 'a4nt_lstm_cell_bwd')
                if ks_r > 1:
                    gemm(dG[t], Wcat, out=rec_slabs, ksplit=ks_r)
                    lib_rc = lib.a4nt_slab_reduce(ks_r, B * (H + E), B, H + E, ops._p(rec_slabs), H + E, 0, None, 0,
                                                  ops._p(rec[t]), H + E, s)
                    _lib.check(lib_rc, 'a4nt_slab_reduce')
                else:
                    gemm(dG[t], Wcat, out=rec[t])
        demb_all = rec[:, :, H:].contiguous()
        dG2 = dG.view(n * B, 4 * H)
        d_W_out, d_b_out = ops.gemm_tn_colsum(hs_all[1:n + 1].reshape(n * B, H), D2)
        d_w_hh = gemm_auto(dG2, hs_all[:n].reshape(n * B, H), transA=True)
        # decoder inputs: step 0 = E[x0] (nn.Embedding, padding row 0 has no gradient);
        # step t>0 = val_in[t-1]*E[tok[t-1]] (a dense mm in the reference: no padding semantics, and its
        # weight gradient sees the .data-zeroed sample, i.e. val_out, char_translator.py:381-385)
        tok_prev = torch.cat([tok0.view(1, B), tok[:n - 1]])
        val_prev_fwd = torch.cat([torch.ones((1, B), device=dev), val_in[:n - 1]])
        emb_in = embed_fwd(tok_prev, emb_W, val_prev_fwd.contiguous())
        d_w_ie = gemm_auto(dG2, emb_in.view(n * B, E), transA=True)
        sumG = colsum(dG.view(n, B * 4 * H)).view(B, 4 * H)
        d_w_ic = gemm(sumG, ctxt, transA=True)
        d_ctxt = gemm(sumG, w_ic)
        d_b = colsum(sumG)                       # = colsum(dG2): the [B, 4H] sums over time are there already
        d_w_ih = torch.cat([d_w_ic, d_w_ie], dim=1)
        d_emb = torch.zeros_like(emb_W)
        embed_bwd(tok0, demb_all[0], d_emb, None, 0)
        if n > 1:
            embed_bwd(tok[:n - 1], demb_all[1:], d_emb, val_out[:n - 1].contiguous(), -1)
        if B != B_all:
            full = torch.zeros((B_all, Hc), device=dev)
            full[r0:r1] = d_ctxt
            d_ctxt = full
        return d_ctxt, None, d_emb, d_w_ih, d_w_hh, d_b, d_b, d_W_out, d_b_out, None, None


def gen_decode(ctxt, tok0, emb_W, w_ih, w_hh, b_ih, b_hh, W_out, b_out, cfg):
    """Returns (SoftSamples or None, gen_lens [B] int32, tok [n,B] int64)."""
    holder = {}
    val_out, tok, gen_lens = _GenDecode.apply(ctxt, tok0, emb_W, w_ih, w_hh, b_ih, b_hh, W_out, b_out, cfg, holder)
    samples = None
    if not cfg['greedy']:
        origin = (holder['handle'], 0, None) if holder['handle'] is not None else None
        samples = SoftSamples(tok, val_out, emb_W.shape[0], origin)
    ops.attach_host_lens(gen_lens, holder['lens_host'])
    return samples, gen_lens, tok
