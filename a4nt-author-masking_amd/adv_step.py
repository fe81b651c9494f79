"""The adversarial training step over the HIP-backed modules: Py3 restatement of
train_adversarial_translator.py:118-186 (adv_forward_pass), :433-547 (discriminator update),
:553-735 (generator update: cycle_loss_type in {None, 'ml', 'enc', 'noncyc_enc', 'bow'}, feature matching,
ml_update), :25-38 (GradFilter), :306-312 (optimizers), plus the batch
data-parallel gradient exchange the reference does not have (one process per GPU, RCCL all-reduce).

Differences from the reference driver are structural, not numerical:
  * samples travel between generator, discriminator and the reconstruction pass as SoftSamples
    (token, value) pairs; the [T,B,V] tensor and its copies (cat / index_select / detach,
    :139-147) are never built;
  * the reconstruction loss uses the fused projection + NLL (forward_mltrain_loss);
  * gen_lens is read back once per generation (for the length sort), not once per step.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .gen_decode import SoftSamples


def adv_forward_pass(modelGen, modelEval, inps, lens, end_c=0, backprop_for='all', maxlen=100, auths=None,
                     cycle_compute=False, append_symb=None, temp=0.5, cycle_loss_type='ml', targs=None,
                     real_batch=None, before_disc=None, gen=None, grad_filter=False):
    """Returns (eval_logits [B,C] in input order, enc_out, samples (SoftSamples, generation order), gen_lens,
    char_outs[, cycle loss][, (real logits, real enc_out)]).

    real_batch = (ids, lens): also classify a batch of real sentences; when the classifier offers
    forward_classify_multi the generated and the real batch share one pass (train_adversarial_translator.py
    :448 and :463 are two separate forward_classify calls in the reference; rows are independent)."""
    if gen is not None:
        samples, gen_lens, char_outs = gen          # generated beforehand (fused_update)
    else:
        if backprop_for in ('eval', 'None'):
            modelGen.eval()
        else:
            modelGen.train()
        samples, gen_lens, char_outs = modelGen.forward_advers_gen(
            inps, lens, soft_samples=True, end_c=end_c, n_max=maxlen, temp=temp, auths=1 - auths, compact=True)
    ml_cycle = cycle_compute and backprop_for != 'eval' and cycle_loss_type == 'ml'
    # (only when paired() really interleaves: run one after the other, the reconstruction pass would launch its
    # forward sweeps, two workgroups per CU, beside the collective that is still in flight)
    late_disc = before_disc is not None and ml_cycle and real_batch is None and ops.pairing_available()
    if before_disc is not None and not late_disc:
        before_disc()                   # e.g. the deferred discriminator step: the generation above did not need it
    if grad_filter:
        samples.set_grad_filter()
    gl_host = ops.host_lens(gen_lens)
    assert int(gl_host.min()) > 0
    len_sorted, sort_idx = gl_host.sort(dim=0, descending=True, stable=True)
    _, rev_sort_idx = sort_idx.sort(dim=0)
    dev = samples.val.device
    sort_dev, rev_dev = ops.h2d(sort_idx, dev), ops.h2d(rev_sort_idx, dev)
    samples_srt = samples.index_select_batch(sort_dev)
    if backprop_for == 'eval':
        samples_srt = samples_srt.detach()
    drop = backprop_for == 'eval'
    real_out = None
    cyc = None
    if ml_cycle and real_batch is None:
        # The discriminator over the generated batch and the reconstruction pass over the same batch do not use
        # each other's results: their LSTM layers are launched pairwise (ops.paired), forward and backward.
        reverse_inp = samples_srt.prepend_token(_start_id(append_symb))
        f_disc = lambda: modelEval.forward_classify(samples_srt, adv_inp=True, lens=len_sorted.tolist(), drop=drop)
        f_cyc = lambda: modelGen.forward_mltrain_loss(reverse_inp, len_sorted.tolist(), inps, lens, targs, auths=auths,
                                                      adv_inp=True, sort_enc=rev_dev)
        if late_disc:
            # A deferred discriminator step (its gradient exchange still in flight): the reconstruction pass goes
            # first and queues everything in front of its first sweep (sample embedding, input-gate product) before
            # the discriminator branch waits for the exchange and steps -- that work runs beside the all-reduce
            def f_disc_late():
                before_disc()
                return f_disc()
            cyc, eval_out = ops.paired(f_cyc, f_disc_late)
        else:
            eval_out, cyc = ops.paired(f_disc, f_cyc)
    elif real_batch is not None and hasattr(modelEval, 'forward_classify_multi'):
        eval_out, real_out = modelEval.forward_classify_multi(
            [(samples_srt, len_sorted.tolist(), True), (real_batch[0], real_batch[1], False)], drop=drop)
    else:
        eval_out = modelEval.forward_classify(samples_srt, adv_inp=True, lens=len_sorted.tolist(), drop=drop)
        if real_batch is not None:
            real_out = modelEval.forward_classify(real_batch[0], lens=real_batch[1])
    logits = eval_out[0].index_select(0, rev_dev)
    out = (logits, eval_out[1], samples, gen_lens, char_outs)
    if cycle_compute and backprop_for != 'eval':
        if cyc is not None:
            out = out + (cyc,)
        elif cycle_loss_type != 'ml':
            reverse_inp = samples_srt.prepend_token(_start_id(append_symb))
            # reverse translation: a second, differentiable free-running generation from the soft samples (:165-174)
            rev, rev_lens, rev_chars = modelGen.forward_advers_gen(
                reverse_inp, len_sorted.tolist(), soft_samples=True, end_c=end_c, n_max=maxlen, temp=temp, auths=auths,
                adv_inp=True, compact=True)
            if grad_filter:
                rev.set_grad_filter()
            out = out + (rev.index_select_batch(rev_dev), rev_lens.index_select(0, rev_dev),
                         [rc.index_select(0, rev_dev) for rc in rev_chars])
        else:
            reverse_inp = samples_srt.prepend_token(_start_id(append_symb))
            cyc = modelGen.forward_mltrain_loss(reverse_inp, len_sorted.tolist(), inps, lens, targs, auths=auths,
                                                adv_inp=True, sort_enc=rev_dev)
            out = out + (cyc,)
    if real_batch is not None:
        out = out + (real_out,)
    return out


def _start_id(append_symb):
    """append_symb: the reference's one-hot [1,B,V] start row (train_adversarial_translator.py:158), or the id."""
    return int(append_symb) if not torch.is_tensor(append_symb) else int(append_symb[0, 0].argmax())


def _sorted_encode(encoder, samples, lens_dev, start_id):
    """Semantic encoding of generated samples: sort by length, prepend the start row, encode, undo the sort
    (train_adversarial_translator.py:631-638, 651-658)."""
    gl = lens_dev.cpu()
    len_sorted, sort_idx = gl.sort(dim=0, descending=True, stable=True)
    _, rev_idx = sort_idx.sort(dim=0)
    dev = samples.val.device
    enc_inp = samples.index_select_batch(ops.h2d(sort_idx, dev)).prepend_token(start_id)
    rev_enc = encoder.forward_encode(enc_inp, len_sorted.tolist(), adv_inp=True)
    return rev_enc.index_select(0, ops.h2d(rev_idx, dev))


def gan_generator_loss(logits, targ_aid, wasserstein=False, maximize_entropy=0, sample_weight=None):
    """The generator's GAN loss on the discriminator logits [B,C] (train_adversarial_translator.py:689-716):
    5*BCE(fake, 1) [per-sample weights], the entropy (1, 2) / margin (3) variants, or -E_fake."""
    gen_aid_out = logits[:, targ_aid]
    if wasserstein:
        return -gen_aid_out.mean()
    if not maximize_entropy:
        loss_aid = ops.bce_with_logits_col(logits, targ_aid, 1.0, sample_weight)
    elif maximize_entropy == 1:
        p = torch.sigmoid(gen_aid_out)
        loss_aid = 4. * (p * torch.log(p) + (1. - p) * torch.log(1. - p)).mean()
    elif maximize_entropy == 2:
        p = F.softmax(logits, dim=1)[:, targ_aid]
        loss_aid = 4. * ((1. - p) * torch.log(1.2 * (1. - p))).mean()
    else:
        p = F.softmax(logits, dim=1)[:, targ_aid]
        slp, b1, b2 = 0.7, 0.6, 0.7
        pdet = p.detach()
        loss_aid = (-torch.log(p) * (pdet < b1).float() + float(-math.log(b1)) * (pdet >= b1).float() +
                    slp * (p - b2) * (pdet >= b2).float()).mean()
    return 5 * loss_aid


def _language_model_ce(lang_model, lang_map, samples, lens_dev, start_id):
    """CE of the language model's next-symbol scores against the generated sentence itself (:669-678)."""
    gl = lens_dev.cpu()
    len_sorted, sort_idx = gl.sort(dim=0, descending=True, stable=True)
    dev = samples.val.device
    enc_inp = samples.index_select_batch(ops.h2d(sort_idx, dev)).prepend_token(start_id)
    from .gen_decode import mapped
    ll = len_sorted.tolist()
    lm_inp = mapped(enc_inp, lang_map)                   # rows = one_hot(tok) * val @ lang_map, never materialised
    # target of position t = arg max of the mapped row t+1 (:677); val > 0, so that is the map row's arg max
    targ_ids = lang_map.argmax(dim=1)[enc_inp.tok[1:]]
    lang_prob, _ = lang_model.forward_mltrain(lm_inp, ll, lm_inp, ll, adv_targ=True)
    tp = torch.nn.utils.rnn.pack_padded_sequence(targ_ids, ll)
    return F.cross_entropy(torch.nn.utils.rnn.pack_padded_sequence(lang_prob, ll)[0], tp[0])


def _cycle_loss_fn(kind):
    """train_adversarial_translator.py:324: CosineEmbeddingLoss(margin=0.1) against ones, or L1."""
    if kind == 'cosine':
        return lambda a, b: F.cosine_embedding_loss(a, b, torch.ones(a.shape[0], device=a.device), margin=0.1)
    return F.l1_loss


def make_optimizers(modelGen, modelEval, lr_gen=1e-4, lr_eval=1e-3, decay_rate=0.99, smooth_eps=1e-8):
    """train_adversarial_translator.py:306-312."""
    optimGen = ops.RMSprop(modelGen.parameters(), lr=lr_gen, alpha=decay_rate, eps=smooth_eps)
    optimEval = ops.RMSprop(
        [{'params': [p for n, p in modelEval.named_parameters() if n != 'decoder_W']},
         {'params': [modelEval.decoder_W], 'weight_decay': 0.0}],
        lr=lr_eval, alpha=decay_rate, eps=smooth_eps)
    return optimGen, optimEval


def _set_requires_grad(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


class GradAllReducer(object):
    """Batch data parallelism: average the gradients of `params` over all ranks (RCCL all-reduce on
    GPU, gloo on CPU in the tests).

    Gradients travel in a few large flat buckets (xGMI links are per-peer: few large collectives beat
    many small ones).  With `overlap` (default) the exchange runs under the backward pass: `arm()`
    before backward() installs post-accumulate hooks, a bucket's all-reduce is issued (async, on RCCL's
    own stream) as soon as the last gradient of that bucket has been accumulated, and `wait()` only
    has the tail to wait for.  Buckets are formed in the order the gradients became ready in the first
    armed backward (rank 0's order is broadcast so that every rank uses the same layout) and are always
    issued in index order, so every rank enqueues the same sequence of collectives.  Parameters whose
    gradient never arrives in a backward (frozen, or unused by that pass) are skipped; that set is a
    property of the graph and therefore the same on every rank.
    With world_size 1 (or no process group) every method is a no-op."""

    def __init__(self, params, bucket_mb=32.0, group=None, overlap=None, force=False, before_wait=None):
        import os
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        # force: run the collectives even in a group of one rank (exercises the RCCL path on a single GPU)
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or bool(force))
        self.world = dist.get_world_size(group) if self.on else 1
        # gloo (the CPU stand-in for RCCL in the tests) gets host tensors: its own staging of device tensors has been
        # seen to stall when the GPU is busy (two ranks sharing one card), and it is not what this class is testing
        self.host_staged = self.on and dist.get_backend(group) == 'gloo'
        # called at the top of wait() when an exchange is pending (tests pass ops.check_timeouts here: a persistent
        # sweep of this rank that gave up waiting for a peer workgroup is then reported BEFORE the rank blocks in the
        # collective, so a co-residency stall on a shared GPU cannot pass as a stall of the collective)
        self.before_wait = before_wait
        self.params = [p for p in params]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        if overlap is None:
            overlap = os.environ.get('A4NT_DP_OVERLAP', '1') != '0'
        self.overlap = bool(overlap) and self.on
        self._pending = []
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._hooks = None
        self._armed = False
        self._order = None          # parameter indices in gradient-ready order (learnt in the first armed backward)
        self._seen = []
        self._buckets_fixed = None  # list of lists of parameter indices
        self._bucket_of = {}
        self._ready = []
        self._next = 0
        self._flat, self._views = [], {}
        self._was_none = set()

    # -- bucket layout --------------------------------------------------------------------------
    def _make_buckets(self, order):
        buckets, cur, n = [], [], 0
        for i in order:
            cur.append(i)
            n += self.params[i].numel()
            if n >= self.bucket_elems:
                buckets.append(cur)
                cur, n = [], 0
        if cur:
            buckets.append(cur)
        return buckets

    def _set_layout(self, order):
        """Fix the bucket layout: every bucket is ONE persistent flat tensor and every parameter of it owns a fixed
        slice (`_views`).  Gradients are accumulated straight into those slices (arm() makes them the parameters'
        `.grad`), so a step neither concatenates a bucket nor copies the averaged values back."""
        self._order = [int(i) for i in order]
        self._buckets_fixed = self._make_buckets(self._order)
        self._bucket_of = {i: b for b, idxs in enumerate(self._buckets_fixed) for i in idxs}
        self._flat, self._views = [], {}
        for idxs in self._buckets_fixed:
            p0 = self.params[idxs[0]]
            flat = torch.zeros(sum(self.params[i].numel() for i in idxs), device=p0.device, dtype=p0.dtype)
            off = 0
            for i in idxs:
                n = self.params[i].numel()
                self._views[i] = flat[off:off + n].view(self.params[i].shape)
                off += n
            self._flat.append(flat)

    def _fix_layout(self):
        """After the first armed backward: adopt rank 0's ready order on every rank."""
        seen = self._seen + [i for i in range(len(self.params)) if i not in set(self._seen)]
        dev = self.params[0].device
        order = torch.tensor(seen, dtype=torch.int64, device='cpu' if self.host_staged else dev)
        self.dist.broadcast(order, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                            group=self.group)
        self._set_layout(order.tolist())

    def _is_view(self, i):
        g, v = self.params[i].grad, self._views[i]
        return g is not None and g.data_ptr() == v.data_ptr() and g.shape == v.shape and g.is_contiguous()

    def _adopt(self, b):
        """Make the gradients of bucket b live in its flat tensor.  A gradient that exists elsewhere (accumulated by
        an earlier backward, or produced before the layout was known) is copied in once; a parameter without a
        gradient gets a zeroed slice and is remembered, so that it is handed back as None if this pass does not
        touch it either (optimizers skip None gradients: a parameter a pass does not use must stay untouched)."""
        idxs = self._buckets_fixed[b]
        if all(self.params[i].grad is None for i in idxs):
            self._flat[b].zero_()
            for i in idxs:
                self.params[i].grad = self._views[i]
                self._was_none.add(i)
            return
        for i in idxs:
            p, v = self.params[i], self._views[i]
            if p.grad is None:
                v.zero_()
                self._was_none.add(i)
            elif not self._is_view(i):
                v.copy_(p.grad)
            p.grad = v

    # -- overlapped path ------------------------------------------------------------------------
    def arm(self):
        """Call right before backward().  Installs the hooks once; (re)starts the per-step bookkeeping."""
        if not self.overlap:
            return
        if self._hooks is None:
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._armed = True
        self._seen = []
        self._next = 0
        self._was_none = set()
        if self._buckets_fixed is not None:
            self._ready = [0] * len(self._buckets_fixed)
            for b in range(len(self._buckets_fixed)):
                self._adopt(b)               # backward() now accumulates in place, into the flat buckets

    def _on_grad(self, p):
        if not self._armed:
            return
        i = self._index[id(p)]
        self._seen.append(i)
        if self._buckets_fixed is None:
            return                       # first step: only learn the order; everything is reduced in wait()
        b = self._bucket_of[i]
        self._ready[b] += 1
        # issue every complete bucket, strictly in index order
        while self._next < len(self._buckets_fixed) and self._ready[self._next] == len(self._buckets_fixed[self._next]):
            self._issue(self._next)
            self._next += 1

    def _issue(self, b):
        flat = self._flat[b]
        wire = flat.cpu() if (self.host_staged and flat.is_cuda) else flat
        work = self.dist.all_reduce(wire, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, wire))
        ops.COLLECTIVES_IN_FLIGHT += 1       # sweeps queued from here to wait() may run beside the collective's kernels

    # -- common ---------------------------------------------------------------------------------
    def launch(self):
        """Start the all-reduce of everything not issued yet (async); call after backward()."""
        if not self.on:
            return
        if self.overlap and self._armed:
            self._armed = False
            if self._buckets_fixed is None:
                self._fix_layout()           # first step: the gradients exist as autograd made them
                self._seen_set = set(self._seen)
                self._was_none = set()
                for b in range(len(self._buckets_fixed)):
                    self._adopt(b)
                self._next = 0
            for b in range(self._next, len(self._buckets_fixed)):
                self._issue(b)
            self._next = len(self._buckets_fixed)
            return
        # exchange after backward: parameter order, gradients copied into the flat buckets once
        if self._buckets_fixed is None:
            self._set_layout(range(len(self.params)))
        self._was_none = set()
        self._seen = [i for i, p in enumerate(self.params) if p.grad is not None]
        for b in range(len(self._buckets_fixed)):
            self._adopt(b)
            self._issue(b)

    def wait(self):
        """Finish the exchange: the averaged gradients are in place (the parameters' .grad are slices of the flat
        buckets)."""
        if self._pending and self.before_wait is not None:
            self.before_wait()
        for work, flat, wire in self._pending:
            work.wait()
            if wire is not flat:
                flat.copy_(wire)
            flat.div_(self.world)
        if self._pending:
            # parameters this pass never produced a gradient for (their slices travelled as zeros on every rank)
            for i in self._was_none - set(self._seen):
                self.params[i].grad = None
        ops.COLLECTIVES_IN_FLIGHT -= len(self._pending)
        self._pending = []

    def all_reduce_scalars(self, t):
        """Sum a small tensor over ranks (the err_a1/err_a2 counters that steer the D loop,
        train_adversarial_translator.py:438,745-746, must agree on every rank)."""
        if self.on:
            if self.host_staged and t.is_cuda:
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def token_weight(self, n_local):
        """Factor for a per-token mean loss so that the rank-averaged gradient equals the gradient of the
        mean over the GLOBAL token count (ragged batches: ranks hold different numbers of valid tokens)."""
        if not self.on:
            return 1.0
        if self.host_staged:
            t = torch.tensor(float(n_local))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return (float(n_local) * self.world) / float(t)
        t = ops.h2d(float(n_local), self.params[0].device, torch.float32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return (float(n_local) * self.world) / t          # stays on the device: no host sync


def d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=True,
             reducer=None, defer=False, gen=None, wasserstein=False, fisher_gan=False, weight_penalty=0.0, alpha=None,
             maximize_entropy=0):
    """One discriminator update (:433-547): BCE(real, 1) + BCE(fake, 0) on column 1 - c_aid, RMSprop, no clipping.
    maximize_entropy != 0: two-class CE on both logits (:490-493).  wasserstein + fisher_gan: Fisher-GAN critic,
    E_real - E_fake with the variance constraint and its multiplier `alpha` (1-element tensor with requires_grad,
    updated in place, :498-514).  A discriminator with a compression layer adds the mean L1 norm of its
    compression rows (:495-496).

    defer=True: return with the gradient exchange in flight and the optimizer step not yet applied;
    result['finish']() completes both.  g_update(before_disc=finish) calls it right before the
    generator update first touches the discriminator, so the exchange and the step's host work overlap
    the generator's free-running generation (which does not read the discriminator)."""
    _set_requires_grad(modelEval, True)
    if gen is None:
        _set_requires_grad(modelGen, False)
    inps, targs, auths, lens = src_batch
    _, gt_targs, _, gt_lens = tgt_batch
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, backprop_for='eval', maxlen=maxlen,
                            auths=auths, temp=temp, real_batch=(gt_targs, gt_lens), gen=gen)
    eval_out_gt = outs[-1]
    optimEval.zero_grad()
    targ_aid = 1 - c_aid
    real_aid_out = eval_out_gt[0][:, targ_aid]
    gen_aid_out = outs[0][:, targ_aid]
    extra = {}
    if reducer is not None:
        reducer.arm()
    if not wasserstein:
        if maximize_entropy == 0:
            loss = (ops.bce_with_logits_col(eval_out_gt[0], targ_aid, 1.0) +
                    ops.bce_with_logits_col(outs[0], targ_aid, 0.0))
        else:
            n, dev = real_aid_out.shape[0], real_aid_out.device
            loss = (F.cross_entropy(eval_out_gt[0], torch.full((n,), 1 - c_aid, dtype=torch.long, device=dev)) +
                    F.cross_entropy(outs[0], torch.full((n,), 1 - targ_aid, dtype=torch.long, device=dev)))
        if getattr(modelEval, 'compression_layer', 0):
            loss = loss + modelEval.compression_W.weight.norm(p=1, dim=1).mean()
        loss.backward()
    else:
        if not fisher_gan or alpha is None:
            raise NotImplementedError('wasserstein needs fisher_gan and alpha (the reference\'s plain-WGAN branch '
                                      'uses an undefined name, train_adversarial_translator.py:521)')
        loss_aid = real_aid_out.mean() - gen_aid_out.mean()
        constraint = 1 - (0.5 * (real_aid_out ** 2).mean() + 0.5 * (gen_aid_out ** 2).mean())
        loss = (loss_aid + alpha * constraint - (weight_penalty / 2.) * (constraint ** 2)).sum()
        (-loss).backward()                       # the critic ascends (backward(mone), :512)
        with torch.no_grad():
            alpha += weight_penalty * alpha.grad
            alpha.grad.zero_()
        extra = {'loss_aid': loss_aid.detach(), 'constraint': constraint.detach(), 'alpha': alpha.detach().clone()}
    if reducer is not None:
        reducer.launch()

    def finish():
        if reducer is not None:
            reducer.wait()
        if step:
            optimEval.step()

    res = {'loss': loss.detach(), 'fake_logits': outs[0].detach(), 'real_logits': eval_out_gt[0].detach(),
           'gen_lens': outs[3], 'char_outs': outs[4]}
    res.update(extra)
    if defer:
        res['finish'] = finish
    else:
        finish()
    return res


def g_update(modelGen, modelEval, optimGen, batch, c_aid, end_c, maxlen, temp, append_symb, cycle_loss_type='ml',
             cycle_loss_w=1.0, grad_clip=10.0, step=True, reducer=None, before_disc=None, gen=None,
             modelGenEncoder=None, cycle_loss_func='l1', grad_filter=False, feature_matching=None, weigh_feat_match=0,
             ml_update=0, tgt_batch=None, wasserstein=False, maximize_entropy=0, weigh_difficult=0., language_loss=0.,
             lang_model=None, lang_map=None):
    """One generator update (:553-735): 5*BCE(fake, 1) + cycle loss [+ feature matching], clip_grad_norm, RMSprop.

    cycle_loss_type: 'ml' (reconstruction CE through the teacher-forced pass), 'enc' (semantic encoder of the
    input vs. of the reverse translation), 'noncyc_enc' (vs. of the generated sentence), 'bow' (bag of words of
    the input vs. the time-summed reverse samples) or None.  modelGenEncoder: frozen CharTranslator(encoder_only=True)
    for the two encoder losses; cycle_loss_func 'l1' | 'cosine'.  feature_matching / ml_update use tgt_batch, a
    batch of the target author (:572-591).  grad_filter: the driver's GradFilter on both sample tensors.
    GAN loss variants: wasserstein (-E_fake), maximize_entropy 1/2/3 and weigh_difficult (gan_generator_loss).
    language_loss: weight of the CE of a frozen decoder-only language model `lang_model` on the generated
    sentence (:668-683); lang_map [V, V_lm] maps this vocabulary onto the language model's (the samples stay
    compact: the language model embeds them with lang_map @ its embedding matrix).
    before_disc: callable run after the generation forward and before the discriminator is used
    (the deferred tail of d_update)."""
    _set_requires_grad(modelEval, False)
    if gen is None:
        _set_requires_grad(modelGen, True)
        optimGen.zero_grad()
    modelGen.train()
    inps, targs, auths, lens = batch
    ml_loss, targ_mean_vec = None, None
    if feature_matching or ml_update:
        gt_inps, gt_targs, gt_auths, gt_lens = tgt_batch
        if before_disc is not None:
            before_disc()
            before_disc = None
        if feature_matching:
            eval_out_gt = modelEval.forward_classify(gt_targs, lens=gt_lens, drop=False)
            if weigh_feat_match:
                w = -F.log_softmax(eval_out_gt[0], dim=1)[:, 1 - c_aid]
                w = w / w.sum()
                targ_mean_vec = (eval_out_gt[1] * w[:, None]).sum(dim=0).detach()
            else:
                targ_mean_vec = eval_out_gt[1].mean(dim=0).detach()
        if ml_update:
            ml_loss = ml_update * modelGen.forward_mltrain_loss(gt_inps, gt_lens, gt_inps, gt_lens, gt_targs,
                                                                auths=gt_auths)
            ml_loss.backward()
    sample_weight = None
    if weigh_difficult > 0.:
        if before_disc is not None:
            before_disc()
            before_disc = None
        with torch.no_grad():
            eval_out_inp = modelEval.forward_classify(targs, lens=lens, drop=False)
        sample_weight = -F.log_softmax(eval_out_inp[0], dim=1)[:, 1 - c_aid]
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, maxlen=maxlen, auths=auths,
                            cycle_compute=(cycle_loss_type is not None and cycle_loss_type != 'noncyc_enc'),
                            append_symb=append_symb, temp=temp, cycle_loss_type=cycle_loss_type, targs=targs,
                            before_disc=before_disc, gen=gen, grad_filter=grad_filter)
    dev = outs[0].device
    fm_loss = torch.zeros((), device=dev)
    if feature_matching:
        fm_loss = feature_matching * F.l1_loss(outs[1].mean(dim=0), targ_mean_vec)
    loss_fn = _cycle_loss_fn(cycle_loss_func)
    rev_out = None
    if cycle_loss_type == 'ml':
        w_tok = reducer.token_weight(sum(int(l) for l in lens)) if reducer is not None else 1.0
        cyc_loss = cycle_loss_w * w_tok * outs[-1]
    elif cycle_loss_type == 'bow':
        rev, rev_lens, rev_chars = outs[-3], outs[-2], outs[-1]
        cyc_targ = torch.zeros(targs.shape[1], rev.vocab, device=dev).scatter_add_(
            1, ops.h2d(targs.t().contiguous(), dev), torch.ones(targs.shape[::-1], device=dev))
        cyc_targ[:, 0] = 0
        cyc_loss = cycle_loss_w * loss_fn(rev.to_dense().sum(dim=0), cyc_targ)
        rev_out = (rev_lens, rev_chars)
    elif cycle_loss_type == 'noncyc_enc':
        with torch.no_grad():
            enc_gt = modelGenEncoder.forward_encode(inps, lens)
        rev_enc = _sorted_encode(modelGenEncoder, outs[2], outs[3], _start_id(append_symb))
        cyc_loss = cycle_loss_w * loss_fn(rev_enc, enc_gt)
    elif cycle_loss_type == 'enc':
        with torch.no_grad():
            enc_gt = modelGenEncoder.forward_encode(inps, lens)
        rev, rev_lens, rev_chars = outs[-3], outs[-2], outs[-1]
        rev_enc = _sorted_encode(modelGenEncoder, rev, rev_lens, _start_id(append_symb))
        cyc_loss = cycle_loss_w * loss_fn(rev_enc, enc_gt)
        rev_out = (rev_lens, rev_chars)
    elif cycle_loss_type is None:
        cyc_loss = torch.zeros((), device=dev)
    else:
        raise NotImplementedError('cycle_loss_type %r' % (cycle_loss_type,))
    lang_loss = torch.zeros((), device=dev)
    if language_loss:
        lang_loss = language_loss * _language_model_ce(lang_model, lang_map, outs[2], outs[3], _start_id(append_symb))
        if float(lang_loss.detach()) > 20:               # the reference drops an exploding language loss (:679-681)
            lang_loss = torch.zeros((), device=dev)
    targ_aid = 1 - c_aid
    gen_aid_out = outs[0][:, targ_aid]
    loss_gan = gan_generator_loss(outs[0], targ_aid, wasserstein, maximize_entropy, sample_weight)
    err = (gen_aid_out.detach() > 0.).float().mean()
    loss_tot = loss_gan + cyc_loss + fm_loss + lang_loss
    if reducer is not None:
        reducer.arm()
    loss_tot.backward()
    if reducer is not None:
        reducer.launch()
        reducer.wait()
    gnorm = ops.clip_grad_norm_(modelGen.parameters(), grad_clip)
    if step:
        optimGen.step()
    res = {'loss': loss_tot.detach(), 'loss_gan': loss_gan.detach(), 'cyc_loss': cyc_loss.detach(),
           'fm_loss': fm_loss.detach(), 'lang_loss': lang_loss.detach(), 'grad_norm': gnorm.detach(), 'err': err, 'gen_lens': outs[3],
           'char_outs': outs[4], 'fake_logits': outs[0].detach()}
    if ml_loss is not None:
        res['ml_loss'] = ml_loss.detach()
    if rev_out is not None:
        res['rev_gen_lens'], res['rev_char_outs'] = rev_out
    return res


def fused_update(modelGen, modelEval, optimGen, optimEval, src_batch, tgt_batch, gen_batch, c_aid, end_c, maxlen, temp,
                 append_symb, cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, step=True, reducers=(None, None)):
    """d_update(src_batch, tgt_batch) followed by g_update(gen_batch), with ONE free-running generation for the
    2B source sentences of both updates.

    The generator's weights are the same in both generations of the reference's loop (the discriminator
    update between them does not touch the generator, train_adversarial_translator.py:433-547) and
    forward_advers_gen has no dropout, so generating the two batches in one batch of 2B rows is the same
    computation; per decoding step the kernels then work on 2B rows instead of being launched twice.
    Only the generator update's rows keep what BPTT needs (grad_rows).  Returns (d result, g result)."""
    s_inps, _, s_auths, s_lens = src_batch
    g_inps, _, g_auths, g_lens = gen_batch
    B = s_inps.shape[1]
    assert g_inps.shape == s_inps.shape
    from . import ops
    if not ops.lstm_rows_fit(2 * B, ops.padded_hidden(modelGen.dec_hidden_size)):
        # 2B rows exceed one persistent sweep at this hidden size: two passes of B rows, as the reference does
        rd = d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=step,
                      reducer=reducers[1], defer=True)
        rg = g_update(modelGen, modelEval, optimGen, gen_batch, c_aid, end_c, maxlen, temp, append_symb,
                      cycle_loss_type=cycle_loss_type, cycle_loss_w=cycle_loss_w, grad_clip=grad_clip, step=step,
                      reducer=reducers[0], before_disc=rd.pop('finish'))
        return rd, rg
    _set_requires_grad(modelGen, True)
    optimGen.zero_grad()
    modelGen.train()
    samples, gen_lens, char_outs = modelGen.forward_advers_gen(
        torch.cat([s_inps, g_inps], dim=1), list(s_lens) + list(g_lens), soft_samples=True, end_c=end_c, n_max=maxlen,
        temp=temp, auths=1 - torch.cat([s_auths, g_auths]), compact=True, grad_rows=(B, 2 * B))
    gl = ops.host_lens(gen_lens)
    n_d, n_g = int(gl[:B].max()), int(gl[B:].max())
    tok = torch.stack(list(char_outs), 0) if not torch.is_tensor(char_outs) else char_outs
    lens_d = ops.attach_host_lens(gen_lens[:B], gl[:B])
    lens_g = ops.attach_host_lens(gen_lens[B:], gl[B:])
    gen_d = (samples.rows(0, B, n_d, differentiable=False), lens_d, list(tok[:n_d, :B].unbind(0)))
    gen_g = (samples.rows(B, 2 * B, n_g), lens_g, list(tok[:n_g, B:].unbind(0)))
    # Data parallel: the discriminator's exchange and optimizer step are deferred into the generator update, which
    # runs them right before it first reads the discriminator -- behind the reconstruction pass's first products
    # (adv_forward_pass, late_disc).  Single process: nothing is in flight, the step is finished here as before.
    defer = reducers[1] is not None and reducers[1].on
    rd = d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=step,
                  reducer=reducers[1], gen=gen_d, defer=defer)
    rg = g_update(modelGen, modelEval, optimGen, gen_batch, c_aid, end_c, maxlen, temp, append_symb,
                  cycle_loss_type=cycle_loss_type, cycle_loss_w=cycle_loss_w, grad_clip=grad_clip, step=step,
                  reducer=reducers[0], gen=gen_g, before_disc=rd.pop('finish') if defer else None)
    return rd, rg


def synthetic_batch(gen, T, B, V, lens=None, start=2, end=1):
    """Synthetic (inps, targs, lens) with the output contract of the reference's
    DataProvider.prepare_data (utils/data_provider.py:310-348): ids ~ randint(3, V], row 0 = START,
    targets = inputs shifted by one with END appended, lengths sorted descending."""
    if lens is None:
        lens = [T] * B
    inps = torch.zeros(T, B, dtype=torch.long)
    targs = torch.zeros(T, B, dtype=torch.long)
    for b, L in enumerate(lens):
        body = torch.randint(3, V + 1, (L,), generator=gen)
        inps[0, b] = start
        inps[1:L, b] = body[:L - 1]
        targs[:L - 1, b] = body[:L - 1]
        targs[L - 1, b] = end
    return inps, targs, list(lens)
