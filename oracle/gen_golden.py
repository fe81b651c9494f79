#!/usr/bin/env python3
"""Oracle (test infrastructure): generate tests/golden/*.npz FROM THE REFERENCE'S OWN MODEL FILES.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

It imports /root/reference/models/{char_translator,char_lstm,char_cnn,model_utils}.py unmodified
through the harness-side shim of SURVEY.md Appendix D (Py2 ``xrange``, ``.cuda()`` -> identity,
``torch.cuda.*Tensor`` -> CPU tensor types), pins the Gumbel noise by replacing
``char_translator.sample_gumbel`` with a function that pops pre-generated uniforms and applies the
reference formula, runs the modules on random-init weights and synthetic ids, and stores
inputs + state_dicts + outputs + gradients as small .npz files.  The step-level fixture drives
the reference models with oracle/adv_step.py (the reference driver is Py2-only).

Nothing of the reference's source text is written to the fixtures: they hold arrays and the
JSON of the params dict only.
"""
import builtins
import json
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def install_shim():
    builtins.xrange = range
    nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.ByteTensor = torch.BoolTensor
    torch.cuda.LongTensor = torch.LongTensor
    sys.path.insert(0, '/root/reference/models')
    import model_utils
    ppu = model_utils.pad_packed_sequence
    model_utils.pad_packed_sequence = lambda p: (lambda o, l: (o, l.tolist()))(*ppu(p))
    import char_translator, char_lstm, char_cnn
    return char_translator, char_lstm, char_cnn, model_utils


class Noise(object):
    """Queue of uniforms; generated on demand with a private generator and recorded."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.drawn = []
        self.replay = None

    def sample_gumbel(self, x):
        if self.replay is not None:
            u = self.replay.pop(0)
        else:
            u = torch.rand(*x.shape, generator=self.gen)
            self.drawn.append(u.clone())
        return -torch.log(-torch.log(u + 1e-20) + 1e-20)


def sd_np(model, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model, prefix):
    out = {}
    for n, p in model.named_parameters():
        out[prefix + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    return out


def save(name, params, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    arrays['params_json'] = np.array(json.dumps(params))
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %-32s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.))


# dims: H multiple of 64 (the HIP LSTM kernels tile hidden units by 16 and K by 64), odd V,
# E and batch sizes that are not multiples of anything.
V0, E0, H0, A0 = 50, 24, 64, 10


def gen_params(**over):
    p = dict(vocabulary_size=V0, embedding_size=E0, enc_hidden_size=H0, dec_hidden_size=H0,
             enc_hidden_depth=1, dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=A0,
             num_output_layers=2, batch_size=999, gumbel_hard=True, softmax_scale=3.0, maxpoolrnn=0,
             drop_prob_emb=0.0, drop_prob_encoder=0.0, drop_prob_decoder=0.0)
    p.update(over)
    return p


def eval_params(**over):
    p = dict(vocabulary_size=V0, embedding_size=E0, hidden_size=H0, hidden_depth=2, en_residual_conn=0,
             num_output_layers=2, mode='classifier', maxpoolrnn=0, bidir=False,
             drop_prob_encoder=0.0, drop_prob_decoder=0.0)
    p.update(over)
    return p


def synth_batch(gen, T, B, V, lens=None, start=2, end=1):
    """ids ~ randint(3,V); row 0 = START; targets = ids shifted with END appended."""
    if lens is None:
        lens = sorted([int(x) for x in torch.randint(3, T + 1, (B,), generator=gen)], reverse=True)
        lens[0] = T
    inps = torch.zeros(T, B, dtype=torch.long)
    targs = torch.zeros(T, B, dtype=torch.long)
    for b, L in enumerate(lens):
        body = torch.randint(3, V + 1, (L,), generator=gen)
        inps[0, b] = start
        inps[1:L, b] = body[:L - 1]
        targs[:L - 1, b] = body[:L - 1]
        targs[L - 1, b] = end
    return inps, targs, lens


def scale_weights(model, factor_W=200.0):
    """decoder_W init is +-1e-3 (logits nearly flat).  Scale it so that soft-max / argmax paths
    see well-separated logits; this is 'identical weights' for both sides, so parity is unaffected."""
    with torch.no_grad():
        model.decoder_W.mul_(factor_W)
        if hasattr(model, 'decoder_W_2'):
            model.decoder_W_2.mul_(factor_W)
        if hasattr(model, 'gumbel_W'):
            model.gumbel_W.mul_(100.0)          # so that the learnt temperature differs between rows


# ------------------------------------------------------------------------------------------
def fx_gumbel(ct):
    g = torch.Generator().manual_seed(11)
    for hard in (0, 1):
        x = torch.randn(6, 37, generator=g) * 2.0
        x.requires_grad_(True)
        noise = Noise(5 + hard)
        ct.sample_gumbel = noise.sample_gumbel
        y = ct.gumbel_softmax_sample(x, tau=0.5, hard=bool(hard))
        up = torch.randn(6, 37, generator=g)
        (y * up).sum().backward()
        save('gumbel_hard%d' % hard, {'tau': 0.5, 'hard': hard}, x=x, u=noise.drawn[0], y=y, upstream=up, dx=x.grad)


def fx_translator_mltrain(ct):
    cases = {
        'mltrain_plain': dict(),
        'mltrain_res2': dict(en_residual_conn=1, enc_hidden_depth=2, dec_hidden_depth=2),
        'mltrain_deep2': dict(enc_hidden_depth=2, dec_hidden_depth=2),
        'mltrain_maxpool': dict(maxpoolrnn=1),
    }
    for name, over in cases.items():
        g = torch.Generator().manual_seed(21)
        torch.manual_seed(3)
        p = gen_params(**over)
        m = ct.CharTranslator(p)
        scale_weights(m, 100.0)
        m.train()
        T, B = 7, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        auths = torch.LongTensor([1] * B)
        scores, _ = m.forward_mltrain(inps, lens, inps, lens, auths=auths)
        from torch.nn.utils.rnn import pack_padded_sequence as pps
        loss = nn.functional.cross_entropy(pps(scores, lens)[0], pps(targs, lens)[0])
        loss.backward()
        arrs = dict(inps=inps, targs=targs, lens=np.array(lens), auths=auths, scores=scores, loss=loss)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        save(name, p, **arrs)

    # soft (adv_inp) encoder input + sort_enc permutation, as the 'ml' cycle pass uses it
    g = torch.Generator().manual_seed(22)
    torch.manual_seed(4)
    p = gen_params()
    m = ct.CharTranslator(p)
    scale_weights(m, 100.0)
    m.train()
    T, B = 6, 5
    inps, targs, lens = synth_batch(g, T, B, V0)
    soft_lens = [7, 6, 6, 4, 2]
    soft = torch.softmax(torch.randn(7, B, V0 + 1, generator=g) * 3, dim=-1)
    for b, L in enumerate(soft_lens):
        soft[L:, b] = 0
    soft.requires_grad_(True)
    perm = torch.LongTensor([3, 0, 4, 1, 2])
    auths = torch.LongTensor([0] * B)
    scores, _ = m.forward_mltrain(soft, soft_lens, inps, lens, auths=auths, adv_inp=True, sort_enc=perm)
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    loss = nn.functional.cross_entropy(pps(scores, lens)[0], pps(targs, lens)[0])
    loss.backward()
    arrs = dict(inps=inps, targs=targs, lens=np.array(lens), auths=auths, soft=soft.detach(),
                soft_lens=np.array(soft_lens), sort_enc=perm, scores=scores, loss=loss, dsoft=soft.grad)
    arrs.update(sd_np(m, 'sd.'))
    arrs.update(grads_np(m, 'grad.'))
    save('mltrain_soft_sortenc', p, **arrs)


def fx_advers_gen(ct):
    for name, over, T, B in (('gen_soft_plain', dict(), 7, 5), ('gen_soft_res2',
                             dict(en_residual_conn=1, enc_hidden_depth=2, dec_hidden_depth=2), 6, 4)):
        g = torch.Generator().manual_seed(31)
        torch.manual_seed(5)
        p = gen_params(**over)
        m = ct.CharTranslator(p)
        scale_weights(m, 300.0)
        m.train()
        inps, targs, lens = synth_batch(g, T, B, V0)
        auths = torch.LongTensor([1] * B)
        n_max = T + 2
        # pass 1: no end symbol, to find a token that terminates some rows early
        noise = Noise(77)
        ct.sample_gumbel = noise.sample_gumbel
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=-1, auths=auths)
        end_c = int(co[2][1])
        # pass 2: same noise, early termination for some rows
        noise.replay = [u.clone() for u in noise.drawn]
        m.zero_grad()
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=end_c, auths=auths)
        samp = torch.stack(list(s), 0)
        R = torch.randn(samp.shape, generator=g)
        loss = (samp * R).sum()
        loss.backward()
        arrs = dict(inps=inps, lens=np.array(lens), auths=auths, n_max=n_max, end_c=end_c, temp=0.5,
                    samp=samp, gen_lens=gl, char_outs=torch.stack(list(co), 0), R=R, loss=loss)
        for i, u in enumerate(noise.drawn):
            arrs['u%d' % i] = u
        arrs['n_noise'] = len(noise.drawn)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        save(name, p, **arrs)

    # greedy decoding, with top-1/top-2 logit margins so exactness is asserted where meaningful
    g = torch.Generator().manual_seed(32)
    torch.manual_seed(6)
    p = gen_params()
    m = ct.CharTranslator(p)
    scale_weights(m, 300.0)
    m.eval()
    T, B = 7, 5
    inps, targs, lens = synth_batch(g, T, B, V0)
    auths = torch.LongTensor([0] * B)
    with torch.no_grad():
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=False, n_max=10, end_c=-1, auths=auths)
        end_c = int(co[3][2])
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=False, n_max=10, end_c=end_c, auths=auths)
    arrs = dict(inps=inps, lens=np.array(lens), auths=auths, n_max=10, end_c=end_c, gen_lens=gl,
                char_outs=torch.stack(list(co), 0))
    arrs.update(sd_np(m, 'sd.'))
    save('gen_greedy', p, **arrs)


def fx_translator_options(ct):
    """Constructor options off the default path: split generators (second decoder for author 1), learnt
    Gumbel temperature, soft (non straight-through) samples with n_samples = 2, soft encoder input,
    decoder-only translator."""
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    # teacher forced, second decoder / no encoder
    for name, over, a in (('mltrain_split', dict(split_generators=1), 1),
                          ('mltrain_noenc', dict(no_encoder=1, pad_auth_vec=0), 0)):
        g = torch.Generator().manual_seed(23)
        torch.manual_seed(12)
        p = gen_params(**over)
        m = ct.CharTranslator(p)
        scale_weights(m, 100.0)
        m.train()
        T, B = 7, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        auths = torch.LongTensor([a] * B)
        scores, _ = m.forward_mltrain(inps, lens, inps, lens, auths=auths)
        loss = nn.functional.cross_entropy(pps(scores, lens)[0], pps(targs, lens)[0])
        loss.backward()
        arrs = dict(inps=inps, targs=targs, lens=np.array(lens), auths=auths, scores=scores, loss=loss)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        save(name, p, **arrs)

    # free running
    cases = (('gen_soft_split', dict(split_generators=1), dict(), 1),
             ('gen_soft_learntau', dict(learn_gumbel=True), dict(), 1),
             ('gen_soft_nsamp2_st0', dict(gumbel_hard=False), dict(n_samples=2), 0),
             ('gen_soft_advinp', dict(), dict(adv_inp=True), 1))
    for name, over, kw, a in cases:
        g = torch.Generator().manual_seed(33)
        torch.manual_seed(13)
        p = gen_params(**over)
        m = ct.CharTranslator(p)
        scale_weights(m, 300.0)
        m.train()
        T, B = 6, 4
        inps, targs, lens = synth_batch(g, T, B, V0)
        auths = torch.LongTensor([a] * B)
        x = inps
        if kw.get('adv_inp'):
            x = torch.softmax(torch.randn(T, B, V0 + 1, generator=g) * 3, dim=-1)
            for b, L in enumerate(lens):
                x[L:, b] = 0
            x.requires_grad_(True)
        n_max = T + 2
        noise = Noise(79)
        ct.sample_gumbel = noise.sample_gumbel
        s, gl, co = m.forward_advers_gen(x, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=-1, auths=auths, **kw)
        end_c = int(co[2][1])
        noise.replay = [u.clone() for u in noise.drawn]
        m.zero_grad()
        if kw.get('adv_inp'):
            x.grad = None
        s, gl, co = m.forward_advers_gen(x, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=end_c, auths=auths, **kw)
        samp = torch.stack(list(s), 0)
        R = torch.randn(samp.shape, generator=g)
        loss = (samp * R).sum()
        loss.backward()
        arrs = dict(inps=inps, lens=np.array(lens), auths=auths, n_max=n_max, end_c=end_c, temp=0.5,
                    samp=samp, gen_lens=gl, char_outs=torch.stack(list(co), 0), R=R, loss=loss,
                    n_samples=kw.get('n_samples', 1))
        if kw.get('adv_inp'):
            arrs['soft'] = x.detach()
            arrs['dsoft'] = x.grad
        for i, u in enumerate(noise.drawn):
            arrs['u%d' % i] = u
        arrs['n_noise'] = len(noise.drawn)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        save(name, p, **arrs)


CLF_LSTM_CASES = (('clf_lstm_mp0', dict()), ('clf_lstm_mp1', dict(maxpoolrnn=1)),
                  ('clf_lstm_mp2', dict(maxpoolrnn=2)), ('clf_lstm_res', dict(en_residual_conn=1)))
# maxpoolrnn=3 concatenates the max over time with the last state of EVERY layer while decoder_W is sized for
# 2H (char_lstm.py:65,309): it only type-checks with one layer
CLF_LSTM_CASES_R2 = (('clf_lstm_mp3', dict(maxpoolrnn=3, hidden_depth=1)),)


def fx_charlstm(cl, cases=CLF_LSTM_CASES):
    from torch.nn.utils.rnn import pack_padded_sequence as pps  # noqa: F401
    for name, over in cases:
        g = torch.Generator().manual_seed(41)
        torch.manual_seed(7)
        p = eval_params(**over)
        m = cl.CharLstm(p)
        m.train()
        T, B = 7, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        labels = torch.LongTensor([0, 1, 1, 0, 1])
        logits, feat = m.forward_classify(targs, lens=lens)
        loss = nn.functional.cross_entropy(logits, labels)
        loss.backward()
        arrs = dict(x=targs, lens=np.array(lens), labels=labels, logits=logits, feat=feat, loss=loss)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        # soft input path on the same weights
        m.zero_grad()
        soft = torch.softmax(torch.randn(T, B, V0 + 1, generator=g) * 3, dim=-1)
        for b, L in enumerate(lens):
            soft[L:, b] = 0
        soft.requires_grad_(True)
        logits_s, feat_s = m.forward_classify(soft, adv_inp=True, lens=lens, drop=False)
        loss_s = nn.functional.cross_entropy(logits_s, labels)
        loss_s.backward()
        arrs.update(dict(soft=soft.detach(), logits_soft=logits_s, feat_soft=feat_s, loss_soft=loss_s,
                         dsoft=soft.grad))
        arrs.update(grads_np(m, 'gradsoft.'))
        save(name, p, **arrs)


def fx_charlstm_heads(cl):
    """CharLstm options off the default path: bidirectional LSTM (with the only pooling it type-checks with),
    MLP head, CNN-over-hidden head, generic real/fake head, compression layer; generative multi-head LM mode."""
    cases = (('clf_lstm_bidir_mp2', dict(bidir=True, maxpoolrnn=2)),
             ('clf_lstm_mlp', dict(decoder_mlp=20)),
             ('clf_lstm_cnnhead', dict(decoder_cnn=1, decoder_cnn_ks=[2, 3], decoder_cnn_nfilt=12, hidden_depth=1)),
             ('clf_lstm_generic', dict(generic_classifier=True, maxpoolrnn=1)),
             ('clf_lstm_compress', dict(compression_layer=8)))
    for name, over in cases:
        g = torch.Generator().manual_seed(43)
        torch.manual_seed(15)
        p = eval_params(**over)
        m = cl.CharLstm(p)
        if hasattr(m, 'generic_W'):
            with torch.no_grad():
                m.generic_W.mul_(0.02)       # init is +-sqrt(fan_in): keep the sigmoid off saturation
        m.train()
        T, B = 7, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        labels = torch.LongTensor([0, 1, 1, 0, 1])

        def run(x, **kw):
            out = m.forward_classify(x, lens=lens, **kw)
            loss = nn.functional.cross_entropy(out[0], labels)
            if len(out) == 3:
                loss = loss + nn.functional.binary_cross_entropy(out[2].view(-1), labels.float())
            loss.backward()
            return out, loss

        out, loss = run(targs)
        arrs = dict(x=targs, lens=np.array(lens), labels=labels, logits=out[0], loss=loss)
        if len(out) == 3:
            arrs['generic'] = out[2]
        else:
            arrs['feat'] = out[1]
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        m.zero_grad()
        soft = torch.softmax(torch.randn(T, B, V0 + 1, generator=g) * 3, dim=-1)
        for b, L in enumerate(lens):
            soft[L:, b] = 0
        soft.requires_grad_(True)
        out_s, loss_s = run(soft, adv_inp=True, drop=False)
        arrs.update(dict(soft=soft.detach(), logits_soft=out_s[0], loss_soft=loss_s, dsoft=soft.grad))
        arrs.update(grads_np(m, 'gradsoft.'))
        save(name, p, **arrs)

    # generative mode: one output layer per author
    g = torch.Generator().manual_seed(44)
    torch.manual_seed(16)
    p = eval_params(mode='generative')
    m = cl.CharLstm(p)
    with torch.no_grad():
        m.decoder_W.mul_(3.0)
    m.train()
    T, B = 7, 5
    inps, targs, lens = synth_batch(g, T, B, V0)
    heads = torch.LongTensor([0, 1, 1, 0, 1])
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    scores, _ = m.forward(inps, lens, None, heads)
    loss = nn.functional.cross_entropy(pps(scores, lens)[0], pps(targs, lens)[0])
    loss.backward()
    arrs = dict(inps=inps, targs=targs, lens=np.array(lens), heads=heads, scores=scores, loss=loss)
    arrs.update(sd_np(m, 'sd.'))
    arrs.update(grads_np(m, 'grad.'))
    m.eval()
    seq = inps[:5, :1].contiguous()
    probs, _ = m.forward_eval(seq, None)
    # CharLstm.forward_gen feeds a 2-D embedding to nn.LSTM (char_lstm.py:255-258), which torch >= 1.11 reads as
    # an unbatched sequence and rejects together with the 3-D state: it cannot be run here, so it is not pinned
    arrs.update(dict(seq=seq, eval_probs=probs))
    save('lm_multihead', p, **arrs)


def fx_charcnn(cc):
    for name, over in (('clf_cnn_234', dict(decoder_cnn_ks=[2, 3, 4])),
                       ('clf_cnn_43', dict(decoder_cnn_ks=[4, 3])),
                       ('clf_cnn_3layer_res', dict(decoder_cnn_ks=[3, 5], decoder_cnn_nlayers=3,
                                                   decoder_cnn_residual=1))):
        g = torch.Generator().manual_seed(51)
        torch.manual_seed(8)
        p = dict(vocabulary_size=V0, embedding_size=E0, num_output_layers=2, decoder_cnn_nfilt=12,
                 drop_prob_encoder=0.0, drop_prob_decoder=0.0, modeltype='cnn')
        p.update(over)
        m = cc.CharCNN(p)
        with torch.no_grad():
            for layer in m.decoder_cnn:
                for conv in layer:
                    conv.weight.mul_(30.0)
                    conv.bias.uniform_(-0.1, 0.1)
        m.train()
        T, B = 9, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        labels = torch.LongTensor([0, 1, 1, 0, 1])
        logits, feat = m.forward_classify(targs, lens=lens)
        loss = nn.functional.cross_entropy(logits, labels)
        loss.backward()
        arrs = dict(x=targs, lens=np.array(lens), labels=labels, logits=logits, feat=feat, loss=loss)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        m.zero_grad()
        soft = torch.softmax(torch.randn(T, B, V0 + 1, generator=g) * 3, dim=-1)
        soft.requires_grad_(True)
        logits_s, feat_s = m.forward_classify(soft, adv_inp=True, lens=lens)
        loss_s = nn.functional.cross_entropy(logits_s, labels)
        loss_s.backward()
        arrs.update(dict(soft=soft.detach(), logits_soft=logits_s, feat_soft=feat_s, loss_soft=loss_s,
                         dsoft=soft.grad))
        arrs.update(grads_np(m, 'gradsoft.'))
        save(name, p, **arrs)


def fx_adv_step(ct, cl, cc):
    from oracle import adv_step
    for name, clf in (('adv_step_lstm', 'lstm'), ('adv_step_cnn', 'cnn')):
        g = torch.Generator().manual_seed(61)
        torch.manual_seed(9)
        gp = gen_params()
        modelGen = ct.CharTranslator(gp)
        scale_weights(modelGen, 300.0)
        if clf == 'lstm':
            ep = eval_params()
            modelEval = cl.CharLstm(ep)
        else:
            ep = dict(vocabulary_size=V0, embedding_size=E0, num_output_layers=2, decoder_cnn_nfilt=12,
                      decoder_cnn_ks=[2, 3, 4], drop_prob_encoder=0.0, drop_prob_decoder=0.0, modeltype='cnn')
            modelEval = cc.CharCNN(ep)
            with torch.no_grad():
                for layer in modelEval.decoder_cnn:
                    for conv in layer:
                        conv.weight.mul_(30.0)
        modelGen.train()
        modelEval.train()
        optG, optE = adv_step.make_optimizers(modelGen, modelEval)
        T, B = 7, 6
        src = synth_batch(g, T, B, V0)
        tgt = synth_batch(g, T, B, V0)
        gb = synth_batch(g, T, B, V0)
        c_aid = 0
        mk = lambda b, a: (b[0], b[1], torch.LongTensor([a] * B), b[2])
        end_c, maxlen, temp = 1, T, 0.5
        append = torch.zeros(1, B, V0 + 1)
        append[:, :, 2] = 1.0
        arrs = {}
        arrs.update(sd_np(modelGen, 'sdG0.'))
        arrs.update(sd_np(modelEval, 'sdE0.'))
        noise = Noise(88)
        ct.sample_gumbel = noise.sample_gumbel
        # --- D update
        rd = adv_step.d_update(modelGen, modelEval, optE, mk(src, c_aid), mk(tgt, 1 - c_aid), c_aid, end_c,
                               maxlen, temp, step=False)
        arrs.update(grads_np(modelEval, 'gradE.'))
        optE.step()
        arrs.update(sd_np(modelEval, 'sdE1.'))
        n_noise_d = len(noise.drawn)
        # --- G update (ml cycle)
        rg = adv_step.g_update(modelGen, modelEval, optG, mk(gb, c_aid), c_aid, end_c, maxlen, temp, append,
                               cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, step=False)
        arrs.update(grads_np(modelGen, 'gradG.'))       # after clipping (in place), as the optimizer sees them
        optG.step()
        arrs.update(sd_np(modelGen, 'sdG1.'))
        for k, v in rd.items():
            arrs['d.' + k] = torch.stack(list(v), 0) if isinstance(v, list) else v
        for k, v in rg.items():
            arrs['g.' + k] = torch.stack(list(v), 0) if isinstance(v, list) else v
        for nm, b in (('src', src), ('tgt', tgt), ('gb', gb)):
            arrs[nm + '.inps'], arrs[nm + '.targs'], arrs[nm + '.lens'] = b[0], b[1], np.array(b[2])
        for i, u in enumerate(noise.drawn):
            arrs['u%d' % i] = u
        arrs.update(dict(n_noise=len(noise.drawn), n_noise_d=n_noise_d, c_aid=c_aid, end_c=end_c, maxlen=maxlen,
                         temp=temp, eval_params_json=np.array(json.dumps(ep))))
        save(name, gp, **arrs)


def fx_gstep_variants(ct, cl):
    """Generator update with the other cycle / semantic losses of the reference driver:
    'noncyc_enc', 'enc' (second, soft-input generation; cosine loss; GradFilter), 'bow' (+ feature matching,
    weighted, + ml_update).  Reference model classes driven by oracle/adv_step.py."""
    from oracle import adv_step
    cases = (('gstep_noncyc', dict(cycle_loss_type='noncyc_enc', cycle_loss_func='l1')),
             ('gstep_enc_cos_gf', dict(cycle_loss_type='enc', cycle_loss_func='cosine', grad_filter=True)),
             ('gstep_bow_fm_ml', dict(cycle_loss_type='bow', feature_matching=0.5, weigh_feat_match=1, ml_update=0.7)))
    for name, kw in cases:
        g = torch.Generator().manual_seed(71)
        torch.manual_seed(14)
        gp = gen_params()
        modelGen = ct.CharTranslator(gp)
        scale_weights(modelGen, 300.0)
        ep = eval_params()
        modelEval = cl.CharLstm(ep)
        enc = None
        if kw['cycle_loss_type'] in ('enc', 'noncyc_enc'):
            enc = ct.CharTranslator(gp, encoder_only=True)
            st = enc.state_dict()
            for k, v in modelGen.state_dict().items():
                if k in st:
                    st[k] = v.clone()
            enc.load_state_dict(st)
            enc.train()
            for p_ in enc.parameters():
                p_.requires_grad = False
        modelGen.train()
        modelEval.train()
        optG, optE = adv_step.make_optimizers(modelGen, modelEval)
        T, B = 7, 6
        gb = synth_batch(g, T, B, V0)
        tgt = synth_batch(g, T, B, V0)
        c_aid = 0
        mk = lambda b, a: (b[0], b[1], torch.LongTensor([a] * B), b[2])
        end_c, maxlen, temp = 1, T, 0.5
        append = torch.zeros(1, B, V0 + 1)
        append[:, :, 2] = 1.0
        arrs = {}
        arrs.update(sd_np(modelGen, 'sdG0.'))
        arrs.update(sd_np(modelEval, 'sdE0.'))
        noise = Noise(91)
        ct.sample_gumbel = noise.sample_gumbel
        rg = adv_step.g_update(modelGen, modelEval, optG, mk(gb, c_aid), c_aid, end_c, maxlen, temp, append,
                               cycle_loss_w=1.0, grad_clip=10.0, step=False, modelGenEncoder=enc,
                               tgt_batch=mk(tgt, 1 - c_aid), **kw)
        arrs.update(grads_np(modelGen, 'gradG.'))
        for k, v in rg.items():
            arrs['g.' + k] = torch.stack(list(v), 0) if isinstance(v, list) else v
        for nm, b in (('gb', gb), ('tgt', tgt)):
            arrs[nm + '.inps'], arrs[nm + '.targs'], arrs[nm + '.lens'] = b[0], b[1], np.array(b[2])
        for i, u in enumerate(noise.drawn):
            arrs['u%d' % i] = u
        arrs.update(dict(n_noise=len(noise.drawn), c_aid=c_aid, end_c=end_c, maxlen=maxlen, temp=temp,
                         eval_params_json=np.array(json.dumps(ep)), step_kwargs_json=np.array(json.dumps(kw))))
        save(name, gp, **arrs)


def fx_loss_variants(ct, cl):
    """The driver's other GAN losses: Fisher-GAN critic update, two-class CE discriminator loss with a
    compression-layer discriminator, Wasserstein / margin / difficulty-weighted generator losses."""
    from oracle import adv_step
    cases = (('dstep_fisher', 'd', dict(), dict(wasserstein=True, fisher_gan=True, weight_penalty=0.5)),
             ('dstep_maxent_comp', 'd', dict(compression_layer=8), dict(maximize_entropy=1)),
             ('gstep_wgan', 'g', dict(), dict(wasserstein=True)),
             ('gstep_wd', 'g', dict(), dict(weigh_difficult=1.0)),
             ('gstep_maxent3', 'g', dict(), dict(maximize_entropy=3)))
    for name, which, ep_over, kw in cases:
        g = torch.Generator().manual_seed(81)
        torch.manual_seed(17)
        gp = gen_params()
        modelGen = ct.CharTranslator(gp)
        scale_weights(modelGen, 300.0)
        ep = eval_params(**ep_over)
        modelEval = cl.CharLstm(ep)
        modelGen.train()
        modelEval.train()
        optG, optE = adv_step.make_optimizers(modelGen, modelEval)
        T, B = 7, 6
        src = synth_batch(g, T, B, V0)
        tgt = synth_batch(g, T, B, V0)
        c_aid = 1
        mk = lambda b, a: (b[0], b[1], torch.LongTensor([a] * B), b[2])
        end_c, maxlen, temp = 1, T, 0.5
        append = torch.zeros(1, B, V0 + 1)
        append[:, :, 2] = 1.0
        arrs = {}
        arrs.update(sd_np(modelGen, 'sdG0.'))
        arrs.update(sd_np(modelEval, 'sdE0.'))
        noise = Noise(93)
        ct.sample_gumbel = noise.sample_gumbel
        if which == 'd':
            kw2 = dict(kw)
            if kw.get('fisher_gan'):
                kw2['alpha'] = torch.full((1,), 0.3, requires_grad=True)
            r = adv_step.d_update(modelGen, modelEval, optE, mk(src, c_aid), mk(tgt, 1 - c_aid), c_aid, end_c, maxlen,
                                  temp, step=False, **kw2)
            arrs.update(grads_np(modelEval, 'gradE.'))
        else:
            r = adv_step.g_update(modelGen, modelEval, optG, mk(src, c_aid), c_aid, end_c, maxlen, temp, append,
                                  cycle_loss_type=None, grad_clip=10.0, step=False, **kw)
            arrs.update(grads_np(modelGen, 'gradG.'))
        for k, v in r.items():
            arrs['r.' + k] = torch.stack(list(v), 0) if isinstance(v, list) else v
        for nm, b in (('src', src), ('tgt', tgt)):
            arrs[nm + '.inps'], arrs[nm + '.targs'], arrs[nm + '.lens'] = b[0], b[1], np.array(b[2])
        for i, u in enumerate(noise.drawn):
            arrs['u%d' % i] = u
        arrs.update(dict(n_noise=len(noise.drawn), c_aid=c_aid, end_c=end_c, maxlen=maxlen, temp=temp, which=which,
                         eval_params_json=np.array(json.dumps(ep)), step_kwargs_json=np.array(json.dumps(kw))))
        save(name, gp, **arrs)


def fx_language_loss(ct, cl):
    """adv_targ (soft target rows into a decoder-only translator) and the generator update's language-model loss."""
    from oracle import adv_step
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    # (1) forward_mltrain(adv_targ=True) on a decoder-only model
    g = torch.Generator().manual_seed(97)
    torch.manual_seed(19)
    V_lm = 30
    lp = gen_params(no_encoder=1, pad_auth_vec=0, vocabulary_size=V_lm)
    lm = ct.CharTranslator(lp)
    scale_weights(lm, 100.0)
    lm.train()
    T, B = 7, 5
    lens = [7, 6, 6, 4, 3]
    soft = torch.softmax(torch.randn(T, B, V_lm + 1, generator=g) * 3, dim=-1)
    soft.requires_grad_(True)
    scores, _ = lm.forward_mltrain(soft, lens, soft, lens, adv_targ=True)
    targ_ids = torch.randint(1, V_lm + 1, (T, B), generator=g)
    loss = nn.functional.cross_entropy(pps(scores, lens)[0], pps(targ_ids, lens)[0])
    loss.backward()
    arrs = dict(soft=soft.detach(), lens=np.array(lens), targ_ids=targ_ids, scores=scores, loss=loss, dsoft=soft.grad)
    arrs.update(sd_np(lm, 'sd.'))
    arrs.update(grads_np(lm, 'grad.'))
    save('mltrain_advtarg_noenc', lp, **arrs)

    # (2) generator update with the language-model loss (vocabulary map: every id i -> i % (V_lm+1))
    g = torch.Generator().manual_seed(98)
    torch.manual_seed(20)
    gp = gen_params()
    modelGen = ct.CharTranslator(gp)
    scale_weights(modelGen, 300.0)
    ep = eval_params()
    modelEval = cl.CharLstm(ep)
    lm = ct.CharTranslator(lp)
    scale_weights(lm, 100.0)
    lm.train()
    for p_ in lm.parameters():
        p_.requires_grad = False
    lang_map = torch.zeros(V0 + 1, V_lm + 1)
    lang_map[torch.arange(V0 + 1), torch.arange(V0 + 1) % (V_lm + 1)] = 1.
    modelGen.train()
    modelEval.train()
    optG, optE = adv_step.make_optimizers(modelGen, modelEval)
    T, B = 7, 6
    gb = synth_batch(g, T, B, V0)
    c_aid = 0
    mk = lambda b, a: (b[0], b[1], torch.LongTensor([a] * B), b[2])
    append = torch.zeros(1, B, V0 + 1)
    append[:, :, 2] = 1.0
    arrs = {}
    arrs.update(sd_np(modelGen, 'sdG0.'))
    arrs.update(sd_np(modelEval, 'sdE0.'))
    arrs.update(sd_np(lm, 'sdLM.'))
    noise = Noise(99)
    ct.sample_gumbel = noise.sample_gumbel
    rg = adv_step.g_update(modelGen, modelEval, optG, mk(gb, c_aid), c_aid, 1, T, 0.5, append, cycle_loss_type=None,
                           grad_clip=10.0, step=False, language_loss=0.8, lang_model=lm, lang_map=lang_map)
    arrs.update(grads_np(modelGen, 'gradG.'))
    for k, v in rg.items():
        arrs['g.' + k] = torch.stack(list(v), 0) if isinstance(v, list) else v
    arrs['gb.inps'], arrs['gb.targs'], arrs['gb.lens'] = gb[0], gb[1], np.array(gb[2])
    for i, u in enumerate(noise.drawn):
        arrs['u%d' % i] = u
    arrs.update(dict(n_noise=len(noise.drawn), c_aid=c_aid, end_c=1, maxlen=T, temp=0.5, lang_map=lang_map,
                     eval_params_json=np.array(json.dumps(ep)), lm_params_json=np.array(json.dumps(lp))))
    save('gstep_langloss', gp, **arrs)


def fx_eval_translate(ct, cl):
    """Evaluation-time translation pass (eval_gantranslate.py:19-53): greedy translation, classifier
    probabilities on the translated ids, greedy reverse translation."""
    from oracle import adv_step
    g = torch.Generator().manual_seed(95)
    torch.manual_seed(18)
    gp = gen_params()
    modelGen = ct.CharTranslator(gp)
    scale_weights(modelGen, 300.0)
    ep = eval_params()
    modelEval = cl.CharLstm(ep)
    T, B = 7, 5
    inps, targs, lens = synth_batch(g, T, B, V0)
    auths = torch.LongTensor([1] * B)
    append = torch.LongTensor([[2]])
    modelGen.eval()
    with torch.no_grad():
        _, _, co = modelGen.forward_advers_gen(inps, lens, soft_samples=False, n_max=9, end_c=-1, auths=auths)
    end_c = int(co[3][2])
    out = adv_step.eval_forward_pass(modelGen, modelEval, inps, lens, end_c=end_c, maxlen=9, auths=auths,
                                     cycle_compute=True, append_symb=append)
    arrs = dict(inps=inps, lens=np.array(lens), auths=auths, end_c=end_c, maxlen=9, probs=out[0],
                char_outs=torch.stack(list(out[1]), 0), gen_lens=out[2], rev_char_outs=torch.stack(list(out[3]), 0),
                rev_gen_lens=out[4], eval_params_json=np.array(json.dumps(ep)))
    arrs.update(sd_np(modelGen, 'sdG.'))
    arrs.update(sd_np(modelEval, 'sdE.'))
    save('eval_translate', gp, **arrs)


def fx_encode(ct):
    """CharTranslator.forward_encode (char_translator.py:288-312) with and without encoder_mean_vec, hard ids and
    soft (adv_inp) rows, gradients of a random projection of the context vector."""
    for name, over in (('encode_meanvec', dict(encoder_mean_vec=1)), ('encode_last', dict())):
        g = torch.Generator().manual_seed(61)
        torch.manual_seed(19)
        p = gen_params(**over)
        m = ct.CharTranslator(p)
        m.train()
        T, B = 7, 5
        inps, targs, lens = synth_batch(g, T, B, V0)
        ctxt = m.forward_encode(inps, lens)
        R = torch.randn(ctxt.shape, generator=g)
        (ctxt * R).sum().backward()
        arrs = dict(inps=inps, lens=np.array(lens), ctxt=ctxt, R=R)
        arrs.update(sd_np(m, 'sd.'))
        arrs.update(grads_np(m, 'grad.'))
        m.zero_grad()
        soft = torch.softmax(torch.randn(T, B, V0 + 1, generator=g) * 3, dim=-1)
        for b, L in enumerate(lens):
            soft[L:, b] = 0
        soft.requires_grad_(True)
        ctxt_s = m.forward_encode(soft, lens, adv_inp=True)
        (ctxt_s * R).sum().backward()
        arrs.update(dict(soft=soft.detach(), ctxt_soft=ctxt_s, dsoft=soft.grad))
        arrs.update(grads_np(m, 'gradsoft.'))
        save(name, p, **arrs)


def fx_init_full(ct, cl, cc):
    """Every initial weight of the three model classes under torch.manual_seed (a3 / a8): the product's own
    constructors must draw the same numbers in the same order."""
    torch.manual_seed(20)
    gp = gen_params(enc_hidden_depth=2, dec_hidden_depth=2, learn_gumbel=True, split_generators=1)
    ep = eval_params(decoder_mlp=20, generic_classifier=True, maxpoolrnn=1)
    cp = dict(vocabulary_size=V0, embedding_size=E0, num_output_layers=2, decoder_cnn_nfilt=12, modeltype='cnn',
              decoder_cnn_ks=[2, 3, 4])
    m = ct.CharTranslator(gp)
    e = cl.CharLstm(ep)
    c = cc.CharCNN(cp)
    gr = gen_params(en_residual_conn=1, enc_hidden_depth=2, dec_hidden_depth=2)
    er = eval_params(en_residual_conn=1)
    mr = ct.CharTranslator(gr)
    er_m = cl.CharLstm(er)
    arrs = dict(eval_params_json=np.array(json.dumps(ep)), cnn_params_json=np.array(json.dumps(cp)),
                gen_res_params_json=np.array(json.dumps(gr)), eval_res_params_json=np.array(json.dumps(er)), seed=20)
    for pre, mod in (('sdG.', m), ('sdE.', e), ('sdC.', c), ('sdGr.', mr), ('sdEr.', er_m)):
        arrs.update(sd_np(mod, pre))
    save('init_full', gp, **arrs)


def fx_init(ct, cl):
    torch.manual_seed(10)
    m = ct.CharTranslator(gen_params())
    e = cl.CharLstm(eval_params())
    save('init_bias', gen_params(),
         g_enc_bias_ih=m.enc_rec_layers.bias_ih_l0.detach(), g_dec_bias_hh=m.dec_rec_layers.bias_hh_l0.detach(),
         e_bias_ih_l1=e.rec_layers.bias_ih_l1.detach(),
         g_decoder_W_absmax=m.decoder_W.detach().abs().max(), e_decoder_W_absmax=e.decoder_W.detach().abs().max())


if __name__ == '__main__':
    ct, cl, cc, mu = install_shim()
    if sys.argv[1:] == ['options']:           # only the fixtures added after the first set
        fx_translator_options(ct)
        sys.exit(0)
    if sys.argv[1:] == ['gstep']:
        fx_gstep_variants(ct, cl)
        sys.exit(0)
    if sys.argv[1:] == ['lang']:
        fx_language_loss(ct, cl)
        sys.exit(0)
    if sys.argv[1:] == ['evaltr']:
        fx_eval_translate(ct, cl)
        sys.exit(0)
    if sys.argv[1:] == ['losses']:
        fx_loss_variants(ct, cl)
        sys.exit(0)
    if sys.argv[1:] == ['heads']:
        fx_charlstm_heads(cl)
        sys.exit(0)
    if sys.argv[1:] == ['round2']:            # fixtures added in round 2
        fx_charlstm(cl, CLF_LSTM_CASES_R2)
        fx_encode(ct)
        fx_init_full(ct, cl, cc)
        sys.exit(0)
    fx_translator_options(ct)
    fx_gstep_variants(ct, cl)
    fx_loss_variants(ct, cl)
    fx_eval_translate(ct, cl)
    fx_language_loss(ct, cl)
    fx_gumbel(ct)
    fx_translator_mltrain(ct)
    fx_advers_gen(ct)
    fx_charlstm(cl)
    fx_charlstm_heads(cl)
    fx_charcnn(cc)
    fx_adv_step(ct, cl, cc)
    fx_init(ct, cl)
    fx_charlstm(cl, CLF_LSTM_CASES_R2)
    fx_encode(ct)
    fx_init_full(ct, cl, cc)
