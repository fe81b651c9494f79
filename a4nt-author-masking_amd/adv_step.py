"""The adversarial training step over the HIP-backed modules: Py3 restatement of
train_adversarial_translator.py:118-186 (adv_forward_pass), :433-547 (discriminator update),
:553-735 (generator update, cycle_loss_type in {None, 'ml'}), :306-312 (optimizers), plus the batch
data-parallel gradient exchange the reference does not have (one process per GPU, RCCL all-reduce).

Differences from the reference driver are structural, not numerical:
  * samples travel between generator, discriminator and the reconstruction pass as SoftSamples
    (token, value) pairs; the [T,B,V] tensor and its copies (cat / index_select / detach,
    :139-147) are never built;
  * the reconstruction loss uses the fused projection + NLL (forward_mltrain_loss);
  * gen_lens is read back once per generation (for the length sort), not once per step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gen_decode import SoftSamples


def adv_forward_pass(modelGen, modelEval, inps, lens, end_c=0, backprop_for='all', maxlen=100, auths=None,
                     cycle_compute=False, append_symb=None, temp=0.5, cycle_loss_type='ml', targs=None,
                     real_batch=None):
    """Returns (eval_logits [B,C] in input order, enc_out, samples (SoftSamples, generation order), gen_lens,
    char_outs[, cycle loss][, (real logits, real enc_out)]).

    real_batch = (ids, lens): also classify a batch of real sentences; when the classifier offers
    forward_classify_multi the generated and the real batch share one pass (train_adversarial_translator.py
    :448 and :463 are two separate forward_classify calls in the reference; rows are independent)."""
    if backprop_for in ('eval', 'None'):
        modelGen.eval()
    else:
        modelGen.train()
    samples, gen_lens, char_outs = modelGen.forward_advers_gen(
        inps, lens, soft_samples=True, end_c=end_c, n_max=maxlen, temp=temp, auths=1 - auths, compact=True)
    gl_host = gen_lens.cpu()
    assert int(gl_host.min()) > 0
    len_sorted, sort_idx = gl_host.sort(dim=0, descending=True, stable=True)
    _, rev_sort_idx = sort_idx.sort(dim=0)
    dev = samples.val.device
    sort_dev, rev_dev = sort_idx.to(dev), rev_sort_idx.to(dev)
    samples_srt = samples.index_select_batch(sort_dev)
    if backprop_for == 'eval':
        samples_srt = samples_srt.detach()
    drop = backprop_for == 'eval'
    real_out = None
    if real_batch is not None and hasattr(modelEval, 'forward_classify_multi'):
        eval_out, real_out = modelEval.forward_classify_multi(
            [(samples_srt, len_sorted.tolist(), True), (real_batch[0], real_batch[1], False)], drop=drop)
    else:
        eval_out = modelEval.forward_classify(samples_srt, adv_inp=True, lens=len_sorted.tolist(), drop=drop)
        if real_batch is not None:
            real_out = modelEval.forward_classify(real_batch[0], lens=real_batch[1])
    logits = eval_out[0].index_select(0, rev_dev)
    out = (logits, eval_out[1], samples, gen_lens, char_outs)
    if cycle_compute and backprop_for != 'eval':
        if cycle_loss_type != 'ml':
            raise NotImplementedError('cycle_loss_type %r is a section-8(f) "next" row' % cycle_loss_type)
        start_id = int(append_symb) if not torch.is_tensor(append_symb) else int(append_symb[0, 0].argmax())
        reverse_inp = samples_srt.prepend_token(start_id)
        cyc = modelGen.forward_mltrain_loss(reverse_inp, len_sorted.tolist(), inps, lens, targs, auths=auths,
                                            adv_inp=True, sort_enc=rev_dev)
        out = out + (cyc,)
    if real_batch is not None:
        out = out + (real_out,)
    return out


def make_optimizers(modelGen, modelEval, lr_gen=1e-4, lr_eval=1e-3, decay_rate=0.99, smooth_eps=1e-8):
    """train_adversarial_translator.py:306-312."""
    optimGen = torch.optim.RMSprop(modelGen.parameters(), lr=lr_gen, alpha=decay_rate, eps=smooth_eps)
    optimEval = torch.optim.RMSprop(
        [{'params': [p for n, p in modelEval.named_parameters() if n != 'decoder_W']},
         {'params': [modelEval.decoder_W], 'weight_decay': 0.0}],
        lr=lr_eval, alpha=decay_rate, eps=smooth_eps)
    return optimGen, optimEval


def _set_requires_grad(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


class GradAllReducer(object):
    """Batch data parallelism: average the gradients of `params` over all ranks (RCCL all-reduce on
    GPU, gloo on CPU in the tests).  Gradients are flattened into a few large buckets (xGMI links
    are per-peer: few large collectives beat many small ones) and reduced asynchronously, so the
    caller can overlap the exchange with independent work and `wait()` before the optimizer step.
    With world_size 1 (or no process group) it is a no-op."""

    def __init__(self, params, bucket_mb=32.0, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.on else 1
        self.params = [p for p in params]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        self._pending = []

    def _buckets(self):
        cur, n = [], 0
        for p in self.params:
            if p.grad is None:
                continue
            cur.append(p)
            n += p.numel()
            if n >= self.bucket_elems:
                yield cur
                cur, n = [], 0
        if cur:
            yield cur

    def launch(self):
        """Start the all-reduce of every bucket (async); call after backward()."""
        if not self.on:
            return
        for bucket in self._buckets():
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, flat, bucket))

    def wait(self):
        """Finish the exchange and write the averaged gradients back."""
        for work, flat, bucket in self._pending:
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._pending = []

    def all_reduce_scalars(self, t):
        """Sum a small tensor over ranks (the err_a1/err_a2 counters that steer the D loop,
        train_adversarial_translator.py:438,745-746, must agree on every rank)."""
        if self.on:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


def d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=True,
             reducer=None):
    """One discriminator update (:433-547): BCE(real, 1) + BCE(fake, 0) on column 1 - c_aid, RMSprop, no clipping."""
    _set_requires_grad(modelEval, True)
    _set_requires_grad(modelGen, False)
    inps, targs, auths, lens = src_batch
    _, gt_targs, _, gt_lens = tgt_batch
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, backprop_for='eval', maxlen=maxlen,
                            auths=auths, temp=temp, real_batch=(gt_targs, gt_lens))
    eval_out_gt = outs[-1]
    optimEval.zero_grad()
    targ_aid = 1 - c_aid
    real_aid_out = eval_out_gt[0][:, targ_aid]
    gen_aid_out = outs[0][:, targ_aid]
    loss = (F.binary_cross_entropy_with_logits(real_aid_out, torch.ones_like(real_aid_out)) +
            F.binary_cross_entropy_with_logits(gen_aid_out, torch.zeros_like(gen_aid_out)))
    loss.backward()
    if reducer is not None:
        reducer.launch()
        reducer.wait()
    if step:
        optimEval.step()
    return {'loss': loss.detach(), 'fake_logits': outs[0].detach(), 'real_logits': eval_out_gt[0].detach(),
            'gen_lens': outs[3], 'char_outs': outs[4]}


def g_update(modelGen, modelEval, optimGen, batch, c_aid, end_c, maxlen, temp, append_symb, cycle_loss_type='ml',
             cycle_loss_w=1.0, grad_clip=10.0, step=True, reducer=None):
    """One generator update (:553-735): 5*BCE(fake, 1) + w * CE reconstruction, clip_grad_norm, RMSprop."""
    _set_requires_grad(modelEval, False)
    _set_requires_grad(modelGen, True)
    optimGen.zero_grad()
    modelGen.train()
    inps, targs, auths, lens = batch
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, maxlen=maxlen, auths=auths,
                            cycle_compute=(cycle_loss_type is not None), append_symb=append_symb, temp=temp,
                            cycle_loss_type=cycle_loss_type, targs=targs)
    if cycle_loss_type == 'ml':
        cyc_loss = cycle_loss_w * outs[-1]
    else:
        cyc_loss = torch.zeros((), device=outs[0].device)
    targ_aid = 1 - c_aid
    gen_aid_out = outs[0][:, targ_aid]
    loss_gan = 5.0 * F.binary_cross_entropy_with_logits(gen_aid_out, torch.ones_like(gen_aid_out))
    err = (gen_aid_out.detach() > 0.).float().mean()
    loss_tot = loss_gan + cyc_loss
    loss_tot.backward()
    if reducer is not None:
        reducer.launch()
        reducer.wait()
    gnorm = torch.nn.utils.clip_grad_norm_(modelGen.parameters(), grad_clip)
    if step:
        optimGen.step()
    return {'loss': loss_tot.detach(), 'loss_gan': loss_gan.detach(), 'cyc_loss': cyc_loss.detach(),
            'grad_norm': gnorm.detach(), 'err': err, 'gen_lens': outs[3], 'char_outs': outs[4],
            'fake_logits': outs[0].detach()}


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
