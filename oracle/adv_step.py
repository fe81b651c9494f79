"""Oracle (test infrastructure): restatement of the adversarial step of the reference driver.

The reference's ``train_adversarial_translator.py`` is Python-2 only (print statements) and cannot
be imported; this file restates, in Python 3, the parts SURVEY.md section 8 (a12, a13) puts on the
hot path:

  * adv_forward_pass     train_adversarial_translator.py:118-186
  * d_update             train_adversarial_translator.py:433-547   (non-WGAN, maximize_entropy == 0)
  * g_update             train_adversarial_translator.py:553-735   (cycle_loss_type in {None, 'ml', 'enc',
                         'noncyc_enc', 'bow'}, feature matching :572-585,609-613, ml_update :587-591)
  * GradFilter           train_adversarial_translator.py:25-38
  * make_optimizers      train_adversarial_translator.py:306-312

The functions are model-agnostic: they only use the nn.Module API of section 8(b), so the golden
fixtures are produced by running THIS glue over the REFERENCE's own model classes
(oracle/gen_golden.py) and the tests run it over the oracle's and the HIP-backed classes.
"""
import inspect
import math

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence


def _accepts_drop(model):
    return 'drop' in inspect.signature(model.forward_classify).parameters


class GradFilter(torch.autograd.Function):
    """Identity whose backward keeps the gradient of ONE time step per batch column, the one with the largest
    absolute sum over the vocabulary (train_adversarial_translator.py:25-38; the reference stores topk = 1
    whatever its argument is, :28)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        n_time, b_sz = grad_output.shape[:2]
        _, topkidx = grad_output.abs().sum(dim=-1).topk(1, dim=0)
        mask = torch.zeros(n_time, b_sz, device=grad_output.device).scatter_(0, topkidx, 1.)
        return grad_output * mask.unsqueeze(-1)


def adv_forward_pass(modelGen, modelEval, inps, lens, end_c=0, backprop_for='all', maxlen=100, auths=None,
                     cycle_compute=False, append_symb=None, temp=0.5, cycle_loss_type='enc', grad_filter=False):
    if backprop_for in ('eval', 'None'):
        modelGen.eval()
    else:
        modelGen.train()
    gen_samples, gen_lens, char_outs = modelGen.forward_advers_gen(
        inps, lens, soft_samples=True, end_c=end_c, n_max=maxlen, temp=temp, auths=1 - auths)
    assert not bool((gen_lens <= 0).any())
    # sort by generated length, longest first, so the samples can be packed (:134-144)
    len_sorted, sort_idx = gen_lens.sort(dim=0, descending=True)
    _, rev_sort_idx = sort_idx.sort(dim=0)
    gen_samples_tensor = torch.stack(list(gen_samples), dim=0)
    if grad_filter:
        gen_samples_tensor = GradFilter.apply(gen_samples_tensor)
    gen_samples_srt = gen_samples_tensor.index_select(1, sort_idx)
    if backprop_for == 'eval':
        gen_samples_srt = gen_samples_srt.detach()
    kw = {'drop': backprop_for == 'eval'} if _accepts_drop(modelEval) else {}
    eval_out_gen = modelEval.forward_classify(gen_samples_srt, adv_inp=True, lens=len_sorted.tolist(), **kw)
    eval_out_unsorted = eval_out_gen[0].index_select(0, rev_sort_idx)

    if cycle_compute and backprop_for != 'eval':
        reverse_inp = torch.cat([append_symb, gen_samples_srt])
        if cycle_loss_type != 'ml':
            rev_s, rev_lens, rev_chars = modelGen.forward_advers_gen(
                reverse_inp, len_sorted.tolist(), soft_samples=True, end_c=end_c, n_max=maxlen, temp=temp,
                auths=auths, adv_inp=True)
            rev_s = torch.stack(list(rev_s), dim=0).index_select(1, rev_sort_idx)
            if grad_filter:
                rev_s = GradFilter.apply(rev_s)
            rev_lens = rev_lens.index_select(0, rev_sort_idx)
            rev_chars = [rc.index_select(0, rev_sort_idx) for rc in rev_chars]
            samples_out = (gen_samples_tensor, gen_lens, char_outs, rev_s, rev_lens, rev_chars)
        else:
            rev_ml_out, _ = modelGen.forward_mltrain(reverse_inp, len_sorted.tolist(), inps, lens, auths=auths,
                                                     adv_inp=True, sort_enc=rev_sort_idx)
            samples_out = (gen_samples_tensor, gen_lens, char_outs, rev_ml_out)
    else:
        samples_out = (gen_samples_tensor, gen_lens, char_outs)
    return (eval_out_unsorted,) + tuple(eval_out_gen[1:]) + samples_out


def eval_forward_pass(modelGen, modelEval, inps, lens, end_c=0, maxlen=100, auths=None, cycle_compute=True,
                      append_symb=None):
    """The translation pass of the evaluation driver (eval_gantranslate.py:19-53): greedy translation to
    `auths`, classifier probabilities of the translated ids, and the greedy reverse translation of those ids.
    append_symb: [1,1] LongTensor holding the start id."""
    modelGen.eval()
    modelEval.eval()
    b_sz = len(lens)
    with torch.no_grad():
        _, gen_lens, char_outs = modelGen.forward_advers_gen(inps, lens, end_c=end_c, n_max=maxlen, auths=auths,
                                                             soft_samples=False)
        len_sorted, sort_idx = gen_lens.sort(dim=0, descending=True)
        _, rev_sort_idx = sort_idx.sort(dim=0)
        eval_inp = torch.stack(list(char_outs), 0).index_select(1, sort_idx)
        probs = modelEval.forward_classify(eval_inp, lens=len_sorted.tolist(), compute_softmax=True)[0]
        probs = probs.detach().index_select(0, rev_sort_idx)
        if not cycle_compute:
            return probs, char_outs, gen_lens
        reverse_inp = torch.cat([append_symb.to(eval_inp.device).repeat(1, b_sz), eval_inp], dim=0)
        _, rev_lens, rev_chars = modelGen.forward_advers_gen(reverse_inp, len_sorted.tolist(), end_c=end_c,
                                                             n_max=maxlen, auths=1 - auths, soft_samples=False)
        rev_chars = [rc.index_select(0, rev_sort_idx) for rc in rev_chars]
    return probs, char_outs, gen_lens, rev_chars, rev_lens


class AdvState(object):
    """Scalars the reference loop carries between iterations (:351-382)."""

    def __init__(self):
        self.err_a = [1.0, 1.0]
        self.accum_err_eval = [0.0, 0.0]
        self.accum_count_gen = [0.0, 0.0]
        self.skip_first = 40
        self.it = 0


def make_optimizers(modelGen, modelEval, lr_gen=1e-4, lr_eval=1e-3, decay_rate=0.99, smooth_eps=1e-8):
    optimGen = torch.optim.RMSprop(modelGen.parameters(), lr=lr_gen, alpha=decay_rate, eps=smooth_eps)
    optimEval = torch.optim.RMSprop(
        [{'params': [p for n, p in modelEval.named_parameters() if n != 'decoder_W']},
         {'params': [modelEval.decoder_W], 'weight_decay': 0.0}],
        lr=lr_eval, alpha=decay_rate, eps=smooth_eps)
    return optimGen, optimEval


def _set_requires_grad(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


def d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=True,
             wasserstein=False, fisher_gan=False, weight_penalty=0.0, alpha=None, maximize_entropy=0):
    """One discriminator update (:433-547).  src_batch / tgt_batch = (inps, targs, auths, lens).
    Default: BCE(real, 1) + BCE(fake, 0) on the target author's logit.  maximize_entropy != 0: two-class CE on
    both logits (:490-493).  wasserstein + fisher_gan: Fisher-GAN critic, E_real - E_fake with the variance
    constraint and its Lagrange multiplier `alpha` (a 1-element tensor with requires_grad, updated in place,
    :498-514; the reference's plain-WGAN branch references an undefined name, :521, and is not restated).
    A discriminator with a compression layer adds the mean L1 norm of its rows (:495-496)."""
    _set_requires_grad(modelEval, True)
    _set_requires_grad(modelGen, False)
    inps, targs, auths, lens = src_batch
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, backprop_for='eval', maxlen=maxlen,
                            auths=auths, temp=temp)
    _, gt_targs, _, gt_lens = tgt_batch
    eval_out_gt = modelEval.forward_classify(gt_targs, lens=gt_lens)
    optimEval.zero_grad()
    targ_aid = 1 - c_aid
    real_aid_out = eval_out_gt[0][:, targ_aid]
    gen_aid_out = outs[0][:, targ_aid]
    extra = {}
    if not wasserstein:
        if maximize_entropy == 0:
            bce = nn.BCEWithLogitsLoss()
            loss = bce(real_aid_out, torch.ones_like(real_aid_out)) + bce(gen_aid_out, torch.zeros_like(gen_aid_out))
        else:
            n = real_aid_out.shape[0]
            dev = real_aid_out.device
            loss = (nn.functional.cross_entropy(eval_out_gt[0], torch.full((n,), 1 - c_aid, dtype=torch.long, device=dev)) +
                    nn.functional.cross_entropy(outs[0], torch.full((n,), 1 - targ_aid, dtype=torch.long, device=dev)))
        if getattr(modelEval, 'compression_layer', 0):
            loss = loss + modelEval.compression_W.weight.norm(p=1, dim=1).mean()
        loss.backward()
    else:
        assert fisher_gan and alpha is not None
        loss_aid = real_aid_out.mean() - gen_aid_out.mean()
        constraint = 1 - (0.5 * (real_aid_out ** 2).mean() + 0.5 * (gen_aid_out ** 2).mean())
        loss_constraint = alpha * constraint - (weight_penalty / 2.) * (constraint ** 2)
        loss = (loss_aid + loss_constraint).sum()
        (-loss).backward()                       # the critic ascends (backward(mone), :512)
        with torch.no_grad():
            alpha += weight_penalty * alpha.grad
            alpha.grad.zero_()
        extra = {'loss_aid': loss_aid.detach(), 'constraint': constraint.detach(), 'alpha': alpha.detach().clone()}
    if step:
        optimEval.step()
    res = {'loss': loss.detach(), 'fake_logits': outs[0].detach(), 'real_logits': eval_out_gt[0].detach(),
           'gen_lens': outs[3], 'char_outs': outs[4]}
    res.update(extra)
    return res


def gan_generator_loss(logits, targ_aid, wasserstein=False, maximize_entropy=0, sample_weight=None):
    """train_adversarial_translator.py:689-716."""
    gen_aid_out = logits[:, targ_aid]
    if wasserstein:
        return -gen_aid_out.mean()
    if not maximize_entropy:
        loss_aid = nn.functional.binary_cross_entropy_with_logits(gen_aid_out, torch.ones_like(gen_aid_out),
                                                                  weight=sample_weight)
    elif maximize_entropy == 1:
        p = torch.sigmoid(gen_aid_out)
        loss_aid = 4. * (p * torch.log(p) + (1. - p) * torch.log(1. - p)).mean()
    elif maximize_entropy == 2:
        p = nn.functional.softmax(logits, dim=1)[:, targ_aid]
        loss_aid = 4. * ((1. - p) * torch.log(1.2 * (1. - p))).mean()
    else:
        p = nn.functional.softmax(logits, dim=1)[:, targ_aid]
        slp, b1, b2 = 0.7, 0.6, 0.7
        pdet = p.detach()
        loss_aid = (-torch.log(p) * (pdet < b1).float() + float(-math.log(b1)) * (pdet >= b1).float() +
                    slp * (p - b2) * (pdet >= b2).float()).mean()
    return 5 * loss_aid


def _cycle_loss_fn(kind):
    """train_adversarial_translator.py:324."""
    if kind == 'cosine':
        cos = nn.CosineEmbeddingLoss(margin=0.1)
        return lambda a, b: cos(a, b, torch.ones(a.shape[0], device=a.device))
    return nn.L1Loss()


def g_update(modelGen, modelEval, optimGen, batch, c_aid, end_c, maxlen, temp, append_symb,
             cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, step=True, modelGenEncoder=None,
             cycle_loss_func='l1', grad_filter=False, feature_matching=None, weigh_feat_match=0, ml_update=0,
             tgt_batch=None, wasserstein=False, maximize_entropy=0, weigh_difficult=0., language_loss=0.,
             lang_model=None, lang_map=None):
    """One generator update (:553-735).  tgt_batch: a batch of the target author (feature matching / ml_update).
    GAN loss: 5*BCE(fake, 1) (optionally weighted per sample by the discriminator's difficulty on the input,
    :593-595, :693-694), one of the three entropy / margin variants (:697-710), or -E_fake (:712-716)."""
    _set_requires_grad(modelEval, False)
    _set_requires_grad(modelGen, True)
    optimGen.zero_grad()
    modelGen.train()
    inps, targs, auths, lens = batch
    ml_loss = None
    if feature_matching or ml_update:
        gt_inps, gt_targs, gt_auths, gt_lens = tgt_batch
        if feature_matching:
            kw = {'drop': False} if _accepts_drop(modelEval) else {}
            eval_out_gt = modelEval.forward_classify(gt_targs, lens=gt_lens, **kw)
            if weigh_feat_match:
                w = -nn.functional.log_softmax(eval_out_gt[0], dim=1)[:, 1 - c_aid]
                w = w / w.sum()
                targ_mean_vec = (eval_out_gt[1] * w[:, None]).sum(dim=0).detach()
            else:
                targ_mean_vec = eval_out_gt[1].mean(dim=0).detach()
        if ml_update:
            ml_out, _ = modelGen.forward_mltrain(gt_inps, gt_lens, gt_inps, gt_lens, auths=gt_auths)
            ml_targ = pack_padded_sequence(gt_targs.to(ml_out.device), gt_lens)
            ml_loss = ml_update * nn.functional.cross_entropy(pack_padded_sequence(ml_out, gt_lens)[0], ml_targ[0])
            ml_loss.backward()
    sample_weight = None
    if weigh_difficult > 0.:
        kw = {'drop': False} if _accepts_drop(modelEval) else {}
        eval_out_inp = modelEval.forward_classify(targs, lens=lens, **kw)
        sample_weight = (-nn.functional.log_softmax(eval_out_inp[0], dim=1)[:, 1 - c_aid]).detach()
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, maxlen=maxlen, auths=auths,
                            cycle_compute=(cycle_loss_type is not None and cycle_loss_type != 'noncyc_enc'),
                            append_symb=append_symb, temp=temp, cycle_loss_type=cycle_loss_type, grad_filter=grad_filter)
    dev = outs[0].device
    fm_loss = torch.zeros((), device=dev)
    if feature_matching:
        fm_loss = feature_matching * nn.functional.l1_loss(outs[1].mean(dim=0), targ_mean_vec)
    loss_fn = _cycle_loss_fn(cycle_loss_func)
    rev_lens_out, rev_chars_out = None, None
    if cycle_loss_type == 'ml':
        rev_ml = outs[-1]
        targ_p = pack_padded_sequence(targs.to(dev), lens)
        cyc_loss = cycle_loss_w * nn.functional.cross_entropy(pack_padded_sequence(rev_ml, lens)[0], targ_p[0])
    elif cycle_loss_type == 'bow':
        # bag of words of the input sentence against the time-summed reverse samples (:617-625)
        b_sz = targs.shape[1]
        cyc_targ = torch.zeros(b_sz, append_symb.shape[-1], device=dev).scatter_add_(
            1, targs.t().to(dev), torch.ones(targs.shape[::-1], device=dev))
        cyc_targ[:, 0] = 0
        cyc_loss = cycle_loss_w * loss_fn(outs[-3].sum(dim=0), cyc_targ)
        rev_lens_out, rev_chars_out = outs[-2], outs[-1]
    elif cycle_loss_type == 'noncyc_enc':
        # semantic encoder of the input vs. of the generated sentence (:626-643)
        enc_inp_gt = modelGenEncoder.forward_encode(inps, lens)
        gen_samples, gen_lens = outs[2], outs[3]
        len_sorted, sort_idx = gen_lens.sort(dim=0, descending=True)
        _, rev_sort_idx = sort_idx.sort(dim=0)
        enc_inp = torch.cat([append_symb, gen_samples.index_select(1, sort_idx)])
        rev_enc = modelGenEncoder.forward_encode(enc_inp, len_sorted.tolist(), adv_inp=True)
        cyc_loss = cycle_loss_w * loss_fn(rev_enc.index_select(0, rev_sort_idx), enc_inp_gt.detach())
    elif cycle_loss_type == 'enc':
        # semantic encoder of the input vs. of the reverse translation of the generated sentence (:644-660)
        enc_inp_gt = modelGenEncoder.forward_encode(inps, lens)
        rev_samples, rev_gen_lens = outs[-3], outs[-2]
        len_sorted, sort_idx = rev_gen_lens.sort(dim=0, descending=True)
        _, rev_sort_idx = sort_idx.sort(dim=0)
        enc_inp = torch.cat([append_symb, rev_samples.index_select(1, sort_idx)])
        rev_enc = modelGenEncoder.forward_encode(enc_inp, len_sorted.tolist(), adv_inp=True)
        cyc_loss = cycle_loss_w * loss_fn(rev_enc.index_select(0, rev_sort_idx), enc_inp_gt.detach())
        rev_lens_out, rev_chars_out = outs[-2], outs[-1]
    elif cycle_loss_type is None:
        cyc_loss = torch.zeros((), device=dev)
    else:
        raise NotImplementedError('oracle: cycle_loss_type %r' % cycle_loss_type)
    lang_loss = torch.zeros((), device=dev)
    if language_loss:
        # a frozen decoder-only language model of the target author scores the generated sentence (:668-683);
        # lang_map [V, V_lm] maps this vocabulary onto the language model's
        gen_samples, gen_lens = outs[2], outs[3]
        len_sorted, sort_idx = gen_lens.sort(dim=0, descending=True)
        enc_inp = torch.cat([append_symb, gen_samples.index_select(1, sort_idx)])
        n_steps, b_sz = enc_inp.shape[0], enc_inp.shape[1]
        lm_inp = enc_inp.view(n_steps * b_sz, -1).mm(lang_map).view(n_steps, b_sz, -1)
        ll = len_sorted.tolist()
        lang_prob, _ = lang_model.forward_mltrain(lm_inp, ll, lm_inp, ll, adv_targ=True)
        lm_targ = pack_padded_sequence(lm_inp.detach()[1:].max(dim=-1)[1], ll)
        lang_loss = language_loss * nn.functional.cross_entropy(pack_padded_sequence(lang_prob, ll)[0], lm_targ[0])
        if float(lang_loss.detach()) > 20:
            lang_loss = torch.zeros((), device=dev)
    targ_aid = 1 - c_aid
    gen_aid_out = outs[0][:, targ_aid]
    loss_gan = gan_generator_loss(outs[0], targ_aid, wasserstein, maximize_entropy, sample_weight)
    err = (gen_aid_out.detach() > 0.).float().mean()
    loss_tot = loss_gan + cyc_loss + fm_loss + lang_loss
    loss_tot.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(modelGen.parameters(), grad_clip)
    if step:
        optimGen.step()
    res = {'loss': loss_tot.detach(), 'loss_gan': loss_gan.detach(), 'cyc_loss': cyc_loss.detach(),
           'fm_loss': fm_loss.detach(), 'lang_loss': lang_loss.detach(), 'grad_norm': gnorm.detach() if torch.is_tensor(gnorm) else torch.tensor(gnorm),
           'err': err, 'gen_lens': outs[3], 'char_outs': outs[4], 'fake_logits': outs[0].detach()}
    if ml_loss is not None:
        res['ml_loss'] = ml_loss.detach()
    if rev_lens_out is not None:
        res['rev_gen_lens'], res['rev_char_outs'] = rev_lens_out, rev_chars_out
    return res
