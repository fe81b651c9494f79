"""Scenario runners shared by the oracle-vs-golden tests (CPU) and the HIP-vs-golden tests (GPU).

Each runner takes an ``impl`` namespace exposing CharTranslator / CharLstm / CharCNN (the oracle
package or the HIP-backed package) and a device, replays one golden fixture and asserts parity.
"""
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence as pps

from golden_util import Fixture, assert_close, assert_grads_close

MLTRAIN = ['mltrain_plain', 'mltrain_res2', 'mltrain_deep2', 'mltrain_maxpool', 'mltrain_split', 'mltrain_noenc']
GEN_SOFT = ['gen_soft_plain', 'gen_soft_res2', 'gen_soft_split', 'gen_soft_learntau', 'gen_soft_nsamp2_st0',
            'gen_soft_advinp']
CLF_LSTM = ['clf_lstm_mp0', 'clf_lstm_mp1', 'clf_lstm_mp2', 'clf_lstm_res', 'clf_lstm_mp3']
ENCODE = ['encode_meanvec', 'encode_last']
CLF_CNN = ['clf_cnn_234', 'clf_cnn_43', 'clf_cnn_3layer_res']
ADV_STEP = ['adv_step_lstm', 'adv_step_cnn']
CLF_HEADS = ['clf_lstm_bidir_mp2', 'clf_lstm_mlp', 'clf_lstm_cnnhead', 'clf_lstm_generic', 'clf_lstm_compress']
LOSS_VARIANTS = ['dstep_fisher', 'dstep_maxent_comp', 'gstep_wgan', 'gstep_wd', 'gstep_maxent3']
GSTEP = ['gstep_noncyc', 'gstep_enc_cos_gf', 'gstep_bow_fm_ml']


def _load(model, sd, device):
    model.to(device)
    missing = model.load_state_dict({k: v.to(device) for k, v in sd.items()})
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    return model


def _packed_ce(scores, targs, lens):
    return F.cross_entropy(pps(scores, lens)[0], pps(targs.to(scores.device), lens)[0])


def run_gumbel(impl, device, name, tol):
    fx = Fixture(name)
    x = fx.t('x', device).requires_grad_(True)
    y = impl.gumbel_softmax_sample(x, float(fx.params['tau']), bool(fx.params['hard']),
                                   impl.NoiseSource(queue=[fx.t('u', device)]))
    (y * fx.t('upstream', device)).sum().backward()
    assert_close(y, fx.t('y'), tol, name + ' y')
    assert_close(x.grad, fx.t('dx'), 5 * tol, name + ' dx')


def run_mltrain(impl, device, name, tol_out, tol_grad):
    fx = Fixture(name)
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    scores, _ = m.forward_mltrain(fx.t('inps'), lens, fx.t('inps'), lens, auths=fx.t('auths'))
    loss = _packed_ce(scores, fx.t('targs'), lens)
    loss.backward()
    # padded positions hold decoder_b in both implementations; compare everything
    assert_close(scores, fx.t('scores'), tol_out, name + ' scores')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name)


def run_mltrain_soft(impl, device, tol_out, tol_grad):
    name = 'mltrain_soft_sortenc'
    fx = Fixture(name)
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    soft = fx.t('soft', device).requires_grad_(True)
    scores, _ = m.forward_mltrain(soft, fx.np('soft_lens').tolist(), fx.t('inps'), lens, auths=fx.t('auths'),
                                  adv_inp=True, sort_enc=fx.t('sort_enc', device))
    loss = _packed_ce(scores, fx.t('targs'), lens)
    loss.backward()
    assert_close(scores, fx.t('scores'), tol_out, name + ' scores')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_close(soft.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name)


def run_gen_soft(impl, device, name, tol_out, tol_grad):
    fx = Fixture(name)
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).train()
    m.set_noise(fx.noise_list(device))
    lens = fx.np('lens').tolist()
    kw, x = {}, fx.t('inps')
    if 'n_samples' in fx and int(fx.scalar('n_samples')) != 1:
        kw['n_samples'] = int(fx.scalar('n_samples'))
    if 'soft' in fx:                        # soft (adv_inp) encoder input
        x = fx.t('soft', device).requires_grad_(True)
        kw['adv_inp'] = True
    s, gl, co = m.forward_advers_gen(x, lens, soft_samples=True, temp=float(fx.scalar('temp')),
                                     n_max=int(fx.scalar('n_max')), end_c=int(fx.scalar('end_c')),
                                     auths=fx.t('auths'), **kw)
    samp = torch.stack(list(s), 0)
    chars = torch.stack(list(co), 0)
    assert samp.shape == fx.t('samp').shape, (samp.shape, fx.t('samp').shape)
    assert torch.equal(chars.cpu().long(), fx.t('char_outs').long()), name + ' sampled tokens differ'
    assert torch.equal(gl.cpu().long(), fx.t('gen_lens').long()), name + ' gen_lens differ'
    loss = (samp * fx.t('R', device)).sum()
    loss.backward()
    assert_close(samp, fx.t('samp'), tol_out, name + ' samples')
    assert_close(loss, fx.t('loss'), 10 * tol_out, name + ' loss')
    if 'dsoft' in fx:
        assert_close(x.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name)


def run_encode(impl, device, name, tol_out, tol_grad):
    """CharTranslator.forward_encode: hard ids, then soft (adv_inp) rows on the same weights."""
    fx = Fixture(name)
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    R = fx.t('R', device)
    ctxt = m.forward_encode(fx.t('inps'), lens)
    (ctxt * R).sum().backward()
    assert_close(ctxt, fx.t('ctxt'), tol_out, name + ' ctxt')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name + ' hard')
    m.zero_grad()
    soft = fx.t('soft', device).requires_grad_(True)
    ctxt = m.forward_encode(soft, lens, adv_inp=True)
    (ctxt * R).sum().backward()
    assert_close(ctxt, fx.t('ctxt_soft'), tol_out, name + ' ctxt(soft)')
    assert_close(soft.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('gradsoft.'), tol_grad, name + ' soft')


def run_init(impl, what='module', extra=None):
    """Constructors under torch.manual_seed draw the reference's initial weights: same values, same RNG order
    (models/char_translator.py:121-161, models/char_lstm.py:100-134, models/char_cnn.py:44-53)."""
    fx = Fixture('init_full')
    torch.manual_seed(int(fx.scalar('seed')))
    P = lambda d: dict(d, **(extra or {}))
    mods = [('sdG.', impl.CharTranslator(P(fx.params))), ('sdE.', impl.CharLstm(P(fx.json('eval_params_json')))),
            ('sdC.', impl.CharCNN(P(fx.json('cnn_params_json')))),
            ('sdGr.', impl.CharTranslator(P(fx.json('gen_res_params_json')))),
            ('sdEr.', impl.CharLstm(P(fx.json('eval_res_params_json'))))]
    for pre, m in mods:
        want, got = fx.group(pre), m.state_dict()
        assert sorted(want) == sorted(got), (pre, sorted(set(want) ^ set(got)))
        for k, v in want.items():
            assert got[k].shape == v.shape, (pre + k, tuple(got[k].shape), tuple(v.shape))
            assert torch.equal(got[k].detach().cpu(), v), '%s %s%s: initial values differ from the reference under the same seed' % (what, pre, k)


def run_gen_greedy(impl, device):
    fx = Fixture('gen_greedy')
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).eval()
    with torch.no_grad():
        s, gl, co = m.forward_advers_gen(fx.t('inps'), fx.np('lens').tolist(), soft_samples=False,
                                         n_max=int(fx.scalar('n_max')), end_c=int(fx.scalar('end_c')),
                                         auths=fx.t('auths'))
    assert len(s) == 0
    chars = torch.stack(list(co), 0)
    assert torch.equal(chars.cpu().long(), fx.t('char_outs').long()), 'greedy tokens differ'
    assert torch.equal(gl.cpu().long(), fx.t('gen_lens').long()), 'greedy gen_lens differ'


def _clf(impl, fx):
    if fx.params.get('modeltype', 'lstm') == 'lstm':
        return impl.CharLstm(fx.params)
    return impl.CharCNN(fx.params)


def run_classifier(impl, device, name, tol_out, tol_grad):
    fx = Fixture(name)
    m = _load(_clf(impl, fx), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    labels = fx.t('labels', device)
    logits, feat = m.forward_classify(fx.t('x'), lens=lens)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    assert_close(logits, fx.t('logits'), tol_out, name + ' logits')
    assert_close(feat, fx.t('feat'), tol_out, name + ' feat')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name + ' hard')
    m.zero_grad()
    soft = fx.t('soft', device).requires_grad_(True)
    logits, feat = m.forward_classify(soft, adv_inp=True, lens=lens, drop=False)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    assert_close(logits, fx.t('logits_soft'), tol_out, name + ' logits(soft)')
    assert_close(loss, fx.t('loss_soft'), tol_out, name + ' loss(soft)')
    assert_close(soft.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('gradsoft.'), tol_grad, name + ' soft')


def run_adv_step(impl, device, name, tol_out, tol_grad, step_mod=None, tol_weights=None):
    """D update then G update ('ml' cycle) on the fixture's weights, inputs and noise."""
    fx = Fixture(name)
    step_mod = step_mod or impl
    # RMSprop's first step is ~ lr * 10 * sign(g): it amplifies rounding noise on near-zero gradient
    # elements, so the post-step weights get their own (looser) tolerance on the HIP path.
    tol_weights = tol_weights if tol_weights is not None else tol_grad
    ep = fx.json('eval_params_json')
    G = _load(impl.CharTranslator(fx.params), fx.group('sdG0.'), device).train()
    E = _load(impl.CharLstm(ep) if ep.get('modeltype', 'lstm') == 'lstm' else impl.CharCNN(ep),
              fx.group('sdE0.'), device).train()
    optG, optE = step_mod.make_optimizers(G, E)
    G.set_noise(fx.noise_list(device))
    B = fx.t('src.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    end_c, maxlen, temp = int(fx.scalar('end_c')), int(fx.scalar('maxlen')), float(fx.scalar('temp'))
    mk = lambda nm, a: (fx.t(nm + '.inps'), fx.t(nm + '.targs'), torch.LongTensor([a] * B),
                        fx.np(nm + '.lens').tolist())
    rd = step_mod.d_update(G, E, optE, mk('src', c_aid), mk('tgt', 1 - c_aid), c_aid, end_c, maxlen, temp,
                           step=False)
    assert torch.equal(torch.stack(list(rd['char_outs']), 0).cpu().long(), fx.t('d.char_outs').long())
    assert_close(rd['fake_logits'], fx.t('d.fake_logits'), tol_out, name + ' D fake logits')
    assert_close(rd['real_logits'], fx.t('d.real_logits'), tol_out, name + ' D real logits')
    assert_close(rd['loss'], fx.t('d.loss'), tol_out, name + ' D loss')
    assert_grads_close(E.named_parameters(), fx.group('gradE.'), tol_grad, name + ' D')
    optE.step()
    for k, v in fx.group('sdE1.').items():
        assert_close(E.state_dict()[k], v, tol_weights, name + ' D weights after step: ' + k)
    # continue from the fixture's exact post-step discriminator so G errors do not compound D ones
    E.load_state_dict({k: v.to(device) for k, v in fx.group('sdE1.').items()})
    append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device=device)
    append[:, :, 2] = 1.0
    rg = step_mod.g_update(G, E, optG, mk('gb', c_aid), c_aid, end_c, maxlen, temp, append,
                           cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, step=False)
    assert torch.equal(torch.stack(list(rg['char_outs']), 0).cpu().long(), fx.t('g.char_outs').long())
    assert torch.equal(rg['gen_lens'].cpu().long(), fx.t('g.gen_lens').long())
    assert_close(rg['loss_gan'], fx.t('g.loss_gan'), tol_out, name + ' G gan loss')
    assert_close(rg['cyc_loss'], fx.t('g.cyc_loss'), tol_out, name + ' G cycle loss')
    assert_close(rg['grad_norm'], fx.t('g.grad_norm'), tol_grad, name + ' G grad norm')
    assert_grads_close(G.named_parameters(), fx.group('gradG.'), tol_grad, name + ' G')
    optG.step()
    for k, v in fx.group('sdG1.').items():
        assert_close(G.state_dict()[k], v, tol_weights, name + ' G weights after step: ' + k)


def run_gstep_variant(impl, device, name, tol_out, tol_grad, step_mod=None):
    """Generator update with the 'noncyc_enc' / 'enc' / 'bow' cycle losses, feature matching, ml_update and
    GradFilter (train_adversarial_translator.py:572-591, 609-660, 25-38) on the fixture's weights and noise."""
    fx = Fixture(name)
    step_mod = step_mod or impl
    kw = fx.json('step_kwargs_json')
    ep = fx.json('eval_params_json')
    G = _load(impl.CharTranslator(fx.params), fx.group('sdG0.'), device).train()
    E = _load(impl.CharLstm(ep), fx.group('sdE0.'), device).train()
    enc = None
    if kw['cycle_loss_type'] in ('enc', 'noncyc_enc'):
        enc = impl.CharTranslator(fx.params, encoder_only=True).to(device)
        st = enc.state_dict()
        for k, v in G.state_dict().items():
            if k in st:
                st[k] = v.clone()
        enc.load_state_dict(st)
        enc.train()
        for p in enc.parameters():
            p.requires_grad = False
    optG, _ = step_mod.make_optimizers(G, E)
    G.set_noise(fx.noise_list(device))
    B = fx.t('gb.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    end_c, maxlen, temp = int(fx.scalar('end_c')), int(fx.scalar('maxlen')), float(fx.scalar('temp'))
    mk = lambda nm, a: (fx.t(nm + '.inps'), fx.t(nm + '.targs'), torch.LongTensor([a] * B), fx.np(nm + '.lens').tolist())
    append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device=device)
    append[:, :, 2] = 1.0
    rg = step_mod.g_update(G, E, optG, mk('gb', c_aid), c_aid, end_c, maxlen, temp, append, cycle_loss_w=1.0,
                           grad_clip=10.0, step=False, modelGenEncoder=enc, tgt_batch=mk('tgt', 1 - c_aid), **kw)
    assert torch.equal(torch.stack(list(rg['char_outs']), 0).cpu().long(), fx.t('g.char_outs').long())
    assert torch.equal(rg['gen_lens'].cpu().long(), fx.t('g.gen_lens').long())
    if 'g.rev_char_outs' in fx:
        assert torch.equal(torch.stack(list(rg['rev_char_outs']), 0).cpu().long(), fx.t('g.rev_char_outs').long())
        assert torch.equal(rg['rev_gen_lens'].cpu().long(), fx.t('g.rev_gen_lens').long())
    for k in ('loss_gan', 'cyc_loss', 'fm_loss', 'loss'):
        if float(fx.t('g.' + k).abs().max()) > 0:
            assert_close(rg[k], fx.t('g.' + k), tol_out, name + ' ' + k)
    if 'g.ml_loss' in fx:
        assert_close(rg['ml_loss'], fx.t('g.ml_loss'), tol_out, name + ' ml_loss')
    assert_close(rg['grad_norm'], fx.t('g.grad_norm'), tol_grad, name + ' grad norm')
    assert_grads_close(G.named_parameters(), fx.group('gradG.'), tol_grad, name + ' G')


def run_classifier_heads(impl, device, name, tol_out, tol_grad):
    """CharLstm with the optional heads / bidirectional LSTM / compression layer: hard and soft input."""
    fx = Fixture(name)
    m = _load(impl.CharLstm(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    labels = fx.t('labels', device)

    def run(x, **kw):
        out = m.forward_classify(x, lens=lens, **kw)
        loss = F.cross_entropy(out[0], labels)
        if len(out) == 3:
            loss = loss + F.binary_cross_entropy(out[2].view(-1), labels.float())
        loss.backward()
        return out, loss

    out, loss = run(fx.t('x'))
    assert_close(out[0], fx.t('logits'), tol_out, name + ' logits')
    if 'generic' in fx:
        assert len(out) == 3
        assert_close(out[2], fx.t('generic'), tol_out, name + ' generic head')
    else:
        assert_close(out[1], fx.t('feat'), tol_out, name + ' feat')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name + ' hard')
    m.zero_grad()
    soft = fx.t('soft', device).requires_grad_(True)
    out, loss = run(soft, adv_inp=True, drop=False)
    assert_close(out[0], fx.t('logits_soft'), tol_out, name + ' logits(soft)')
    assert_close(loss, fx.t('loss_soft'), tol_out, name + ' loss(soft)')
    assert_close(soft.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('gradsoft.'), tol_grad, name + ' soft')


def run_lm_multihead(impl, device, tol_out, tol_grad):
    """Generative mode of CharLstm (char_lstm.py:166-222): per-row author heads, and every head on one sequence."""
    name = 'lm_multihead'
    fx = Fixture(name)
    m = _load(impl.CharLstm(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    scores, _ = m.forward(fx.t('inps'), lens, None, fx.t('heads'))
    loss = _packed_ce(scores, fx.t('targs'), lens)
    loss.backward()
    assert_close(scores, fx.t('scores'), tol_out, name + ' scores')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name)
    m.eval()
    probs, _ = m.forward_eval(fx.t('seq'), None)
    assert_close(probs, fx.t('eval_probs'), tol_out, name + ' forward_eval')
    # forward_gen is not pinned by the reference (it cannot run under this torch); check it against forward_eval:
    # greedy continuation = arg max of the target author's distribution after the same prefix
    gen = m.forward_gen(fx.t('seq'), None, torch.LongTensor([1]), n_max=3)
    seq = fx.t('seq')
    for i, c in enumerate(gen):
        p, _ = m.forward_eval(seq, None, compute_softmax=False)
        assert int(c.view(-1)[0]) == int(p[-1, 1].argmax())
        seq = torch.cat([seq, c.view(1, 1).cpu()], 0)


def run_loss_variant(impl, device, name, tol_out, tol_grad, step_mod=None):
    """One discriminator or generator update with the driver's other GAN losses (Fisher-GAN critic, two-class
    CE + compression regulariser, Wasserstein / margin / difficulty-weighted generator loss)."""
    fx = Fixture(name)
    step_mod = step_mod or impl
    kw = fx.json('step_kwargs_json')
    ep = fx.json('eval_params_json')
    G = _load(impl.CharTranslator(fx.params), fx.group('sdG0.'), device).train()
    E = _load(impl.CharLstm(ep), fx.group('sdE0.'), device).train()
    optG, optE = step_mod.make_optimizers(G, E)
    G.set_noise(fx.noise_list(device))
    B = fx.t('src.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    end_c, maxlen, temp = int(fx.scalar('end_c')), int(fx.scalar('maxlen')), float(fx.scalar('temp'))
    mk = lambda nm, a: (fx.t(nm + '.inps'), fx.t(nm + '.targs'), torch.LongTensor([a] * B), fx.np(nm + '.lens').tolist())
    if str(fx.np('which')) == 'd':
        if kw.get('fisher_gan'):
            kw['alpha'] = torch.full((1,), 0.3, device=device, requires_grad=True)
        r = step_mod.d_update(G, E, optE, mk('src', c_aid), mk('tgt', 1 - c_aid), c_aid, end_c, maxlen, temp,
                              step=False, **kw)
        assert_close(r['loss'], fx.t('r.loss'), tol_out, name + ' loss')
        for k in ('loss_aid', 'constraint', 'alpha'):
            if 'r.' + k in fx:
                assert_close(r[k], fx.t('r.' + k), tol_out, name + ' ' + k)
        assert_grads_close(E.named_parameters(), fx.group('gradE.'), tol_grad, name + ' D')
    else:
        append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device=device)
        append[:, :, 2] = 1.0
        r = step_mod.g_update(G, E, optG, mk('src', c_aid), c_aid, end_c, maxlen, temp, append, cycle_loss_type=None,
                              grad_clip=10.0, step=False, **kw)
        assert torch.equal(torch.stack(list(r['char_outs']), 0).cpu().long(), fx.t('r.char_outs').long())
        assert_close(r['loss_gan'], fx.t('r.loss_gan'), tol_out, name + ' gan loss')
        assert_close(r['grad_norm'], fx.t('r.grad_norm'), tol_grad, name + ' grad norm')
        assert_grads_close(G.named_parameters(), fx.group('gradG.'), tol_grad, name + ' G')


def run_eval_translate(impl, device, fwd, tol_out):
    """Evaluation-time translation pass: greedy tokens, lengths and reverse translation exact, classifier
    probabilities within tolerance.  fwd: the eval_forward_pass restatement under test."""
    fx = Fixture('eval_translate')
    ep = fx.json('eval_params_json')
    G = _load(impl.CharTranslator(fx.params), fx.group('sdG.'), device)
    E = _load(impl.CharLstm(ep), fx.group('sdE.'), device)
    out = fwd(G, E, fx.t('inps'), fx.np('lens').tolist(), end_c=int(fx.scalar('end_c')), maxlen=int(fx.scalar('maxlen')),
              auths=fx.t('auths'), cycle_compute=True, append_symb=torch.LongTensor([[2]]))
    assert torch.equal(torch.stack(list(out[1]), 0).cpu().long(), fx.t('char_outs').long())
    assert torch.equal(out[2].cpu().long(), fx.t('gen_lens').long())
    assert torch.equal(torch.stack(list(out[3]), 0).cpu().long(), fx.t('rev_char_outs').long())
    assert torch.equal(out[4].cpu().long(), fx.t('rev_gen_lens').long())
    assert_close(out[0], fx.t('probs'), tol_out, 'eval_translate classifier probabilities')


def run_mltrain_advtarg(impl, device, tol_out, tol_grad):
    """forward_mltrain(adv_targ=True): soft rows as the decoder's input on a decoder-only translator."""
    name = 'mltrain_advtarg_noenc'
    fx = Fixture(name)
    m = _load(impl.CharTranslator(fx.params), fx.group('sd.'), device).train()
    lens = fx.np('lens').tolist()
    soft = fx.t('soft', device).requires_grad_(True)
    scores, _ = m.forward_mltrain(soft, lens, soft, lens, adv_targ=True)
    loss = _packed_ce(scores, fx.t('targ_ids'), lens)
    loss.backward()
    assert_close(scores, fx.t('scores'), tol_out, name + ' scores')
    assert_close(loss, fx.t('loss'), tol_out, name + ' loss')
    assert_close(soft.grad, fx.t('dsoft'), tol_grad, name + ' dsoft')
    assert_grads_close(m.named_parameters(), fx.group('grad.'), tol_grad, name)


def run_gstep_langloss(impl, device, tol_out, tol_grad, step_mod=None):
    """Generator update with the language-model loss (train_adversarial_translator.py:668-683)."""
    name = 'gstep_langloss'
    fx = Fixture(name)
    step_mod = step_mod or impl
    ep = fx.json('eval_params_json')
    G = _load(impl.CharTranslator(fx.params), fx.group('sdG0.'), device).train()
    E = _load(impl.CharLstm(ep), fx.group('sdE0.'), device).train()
    LM = _load(impl.CharTranslator(fx.json('lm_params_json')), fx.group('sdLM.'), device).train()
    for p in LM.parameters():
        p.requires_grad = False
    optG, _ = step_mod.make_optimizers(G, E)
    G.set_noise(fx.noise_list(device))
    B = fx.t('gb.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    batch = (fx.t('gb.inps'), fx.t('gb.targs'), torch.LongTensor([c_aid] * B), fx.np('gb.lens').tolist())
    append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device=device)
    append[:, :, 2] = 1.0
    rg = step_mod.g_update(G, E, optG, batch, c_aid, int(fx.scalar('end_c')), int(fx.scalar('maxlen')),
                           float(fx.scalar('temp')), append, cycle_loss_type=None, grad_clip=10.0, step=False,
                           language_loss=0.8, lang_model=LM, lang_map=fx.t('lang_map', device))
    assert torch.equal(torch.stack(list(rg['char_outs']), 0).cpu().long(), fx.t('g.char_outs').long())
    assert float(fx.t('g.lang_loss')) > 0
    assert_close(rg['lang_loss'], fx.t('g.lang_loss'), tol_out, name + ' language loss')
    assert_close(rg['loss_gan'], fx.t('g.loss_gan'), tol_out, name + ' gan loss')
    assert_close(rg['grad_norm'], fx.t('g.grad_norm'), tol_grad, name + ' grad norm')
    assert_grads_close(G.named_parameters(), fx.group('gradG.'), tol_grad, name + ' G')
