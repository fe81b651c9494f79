"""Oracle (test infrastructure): restatement of the adversarial step of the reference driver.

The reference's ``train_adversarial_translator.py`` is Python-2 only (print statements) and cannot
be imported; this file restates, in Python 3, the parts SURVEY.md section 8 (a12, a13) puts on the
hot path:

  * adv_forward_pass     train_adversarial_translator.py:118-186
  * d_update             train_adversarial_translator.py:433-547   (non-WGAN, maximize_entropy == 0)
  * g_update             train_adversarial_translator.py:553-735   (cycle_loss_type in {None, 'ml'})
  * make_optimizers      train_adversarial_translator.py:306-312

The functions are model-agnostic: they only use the nn.Module API of section 8(b), so the golden
fixtures are produced by running THIS glue over the REFERENCE's own model classes
(oracle/gen_golden.py) and the tests run it over the oracle's and the HIP-backed classes.
"""
import inspect

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence


def _accepts_drop(model):
    return 'drop' in inspect.signature(model.forward_classify).parameters


def adv_forward_pass(modelGen, modelEval, inps, lens, end_c=0, backprop_for='all', maxlen=100, auths=None,
                     cycle_compute=False, append_symb=None, temp=0.5, cycle_loss_type='enc'):
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


def d_update(modelGen, modelEval, optimEval, src_batch, tgt_batch, c_aid, end_c, maxlen, temp, step=True):
    """One discriminator update (:433-547).  src_batch / tgt_batch = (inps, targs, auths, lens)."""
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
    bce = nn.BCEWithLogitsLoss()
    loss = bce(real_aid_out, torch.ones_like(real_aid_out)) + bce(gen_aid_out, torch.zeros_like(gen_aid_out))
    loss.backward()
    if step:
        optimEval.step()
    return {'loss': loss.detach(), 'fake_logits': outs[0].detach(), 'real_logits': eval_out_gt[0].detach(),
            'gen_lens': outs[3], 'char_outs': outs[4]}


def g_update(modelGen, modelEval, optimGen, batch, c_aid, end_c, maxlen, temp, append_symb,
             cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, step=True):
    """One generator update (:553-735)."""
    _set_requires_grad(modelEval, False)
    _set_requires_grad(modelGen, True)
    optimGen.zero_grad()
    modelGen.train()
    inps, targs, auths, lens = batch
    outs = adv_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, maxlen=maxlen, auths=auths,
                            cycle_compute=(cycle_loss_type is not None), append_symb=append_symb, temp=temp,
                            cycle_loss_type=cycle_loss_type)
    dev = outs[0].device
    if cycle_loss_type == 'ml':
        rev_ml = outs[-1]
        targ_p = pack_padded_sequence(targs.to(dev), lens)
        cyc_loss = cycle_loss_w * nn.functional.cross_entropy(pack_padded_sequence(rev_ml, lens)[0], targ_p[0])
    elif cycle_loss_type is None:
        cyc_loss = torch.zeros((), device=dev)
    else:
        raise NotImplementedError('oracle: cycle_loss_type %r is a section-8(f) "next" row' % cycle_loss_type)
    targ_aid = 1 - c_aid
    gen_aid_out = outs[0][:, targ_aid]
    loss_gan = 5.0 * nn.functional.binary_cross_entropy_with_logits(gen_aid_out, torch.ones_like(gen_aid_out))
    err = (gen_aid_out.detach() > 0.).float().mean()
    loss_tot = loss_gan + cyc_loss
    loss_tot.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(modelGen.parameters(), grad_clip)
    if step:
        optimGen.step()
    return {'loss': loss_tot.detach(), 'loss_gan': loss_gan.detach(), 'cyc_loss': cyc_loss.detach(),
            'grad_norm': gnorm.detach() if torch.is_tensor(gnorm) else torch.tensor(gnorm), 'err': err,
            'gen_lens': outs[3], 'char_outs': outs[4], 'fake_logits': outs[0].detach()}
