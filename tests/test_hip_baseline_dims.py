"""GPU: parity at BASELINE.json's real dimensions (H / V / T of each config, reduced batch where the oracle would
take too long), HIP path against the oracle COMPUTED IN THE TEST on the box's host cores, on the same random-init
weights, synthetic inputs and injected Gumbel uniforms (dropout 0).

Stated tolerance at config dims: max-normalised error <= 1e-5 on logits / scores / losses,
<= 1e-4 on every parameter gradient (T = 100-200: longer fp32 recurrences than the fixtures'); sampled and greedy
tokens and generated lengths exact.  Every exact-token claim prints the smallest top-1 / top-2 gap the oracle saw
while deciding those tokens, so the claim can be read against the fp32 tolerance.

  cfg 1  CharLstm 2x512, V=10001, T=100, B=64: forward_classify + CE + backward (train.py:94-139)
  cfg 2  CharTranslator enc/dec 1x512, V=10001, T=100, B=64: forward_mltrain + packed CE + backward
         (train_translator.py:93-125), and the fused projection+NLL entry on the same batch
  cfg 3  full adversarial step (D update + G update with 'ml' reconstruction), H=E=512, V=10001, T=100, B=16,
         ragged and all-T lengths (train_adversarial_translator.py:433-735); greedy decode of the same model
  cfg 5  stress shape: translator 1x1024, V=30001, T=200, CharCNN discriminator, B=8; and the same 8 sentences as
         rows 0-7 of a B=256 batch (the benchmark's geometry: 64 rows per sweep group) against the 8-row run
The oracle itself is pinned to the reference by tests/test_oracle_golden.py.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence as pps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import assert_close, assert_grads_close, rel_err

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_GRAD = 1e-5, 1e-4


def _gen_params(V, E, H, B):
    return dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
                dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=10, num_output_layers=2, batch_size=B,
                gumbel_hard=True, softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0., drop_prob_encoder=0.,
                drop_prob_decoder=0.)


def _lstm_params(V, E):
    return dict(vocabulary_size=V, embedding_size=E, hidden_size=512, hidden_depth=2, en_residual_conn=0,
                num_output_layers=2, mode='classifier', maxpoolrnn=0, bidir=False, drop_prob_encoder=0.,
                drop_prob_decoder=0.)


def _cnn_params(V, E):
    return dict(vocabulary_size=V, embedding_size=E, num_output_layers=2, modeltype='cnn', drop_prob_encoder=0.,
                drop_prob_decoder=0., decoder_cnn_ks=[2, 3, 4], decoder_cnn_nfilt=100, decoder_cnn_nlayers=1)


def _ragged(g, B, T, lo=4):
    lens = sorted([int(v) for v in torch.randint(lo, T + 1, (B,), generator=g)], reverse=True)
    lens[0] = T
    return lens


def _pair(cls, params, seed, sharpen=None):
    import a4nt_amd
    import oracle
    torch.manual_seed(seed)
    ref = getattr(oracle, cls)(params).train()
    if sharpen:
        with torch.no_grad():
            ref.decoder_W.mul_(sharpen)
    hip = getattr(a4nt_amd, cls)(params)
    res = hip.load_state_dict(ref.state_dict())
    assert not res.missing_keys and not res.unexpected_keys
    return ref, hip.cuda().train()


def _grads(m):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}


def _assert_grad_dicts(got, ref, tol, what):
    worst = (0.0, None)
    for k, r in ref.items():
        if float(r.abs().max()) == 0.0:
            assert float(got[k].abs().max()) <= 1e-12, '%s grad %s should be zero' % (what, k)
            continue
        e = rel_err(got[k], r)
        if e > worst[0]:
            worst = (e, k)
    assert worst[0] <= tol, '%s: worst grad %s max-normalised error %.3e > %.1e' % (what, worst[1], worst[0], tol)
    return worst[0]


@pytest.fixture(scope='module')
def env():
    import a4nt_amd
    import oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return a4nt_amd, oracle


@pytest.mark.timeout(600)
@pytest.mark.parametrize('ragged', [True, False])
def test_cfg1_classifier_full_size(env, ragged):
    impl, oracle = env
    V, E, T, B = 10000, 512, 100, 64
    ref, hip = _pair('CharLstm', _lstm_params(V, E), 21)
    g = torch.Generator().manual_seed(31)
    lens = _ragged(g, B, T) if ragged else [T] * B
    _, x, lens = impl.synthetic_batch(g, T, B, V, lens)
    labels = torch.arange(B) % 2
    out = {}
    import copy
    ref64 = copy.deepcopy(ref).double()          # the same oracle in fp64: what fp32 rounding alone costs at this size
    for tag, m in (('ref', ref), ('hip', hip), ('f64', ref64)):
        logits, feat = m.forward_classify(x, lens=lens)
        loss = F.cross_entropy(logits, labels.to(logits.device))
        loss.backward()
        out[tag] = (logits.detach().cpu(), feat.detach().cpu(), loss.detach().cpu())
    impl.check_timeouts()
    floor = rel_err(out['ref'][1], out['f64'][1])
    print('cfg1 %s: logits err %.2e, loss err %.2e, features err %.2e (oracle fp32 vs fp64 on the features: %.2e; HIP vs '
          'fp64: %.2e)' % ('ragged' if ragged else 'all-T', rel_err(out['hip'][0], out['ref'][0]),
                          rel_err(out['hip'][2], out['ref'][2]), rel_err(out['hip'][1], out['ref'][1]), floor,
                          rel_err(out['hip'][1], out['f64'][1])))
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'cfg1 logits')
    # the feature vector is the last hidden state after 2 x 100 recurrent steps: the oracle's own fp32 arithmetic is
    # 1.1e-5 away from its fp64 run here, so 1e-5 is below the fp32 noise floor for this one quantity; the HIP path
    # must be as close to the exact (fp64) features as torch's CPU fp32 is, within a factor of 2
    assert rel_err(out['hip'][1], out['f64'][1]) <= max(TOL_OUT, 2.0 * floor), 'cfg1 features vs fp64'
    assert_close(out['hip'][1], out['ref'][1], max(TOL_OUT, 3.0 * floor), 'cfg1 features')
    assert_close(out['hip'][2], out['ref'][2], TOL_OUT, 'cfg1 loss')
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'cfg1')


@pytest.mark.timeout(600)
def test_cfg2_teacher_forced_full_size(env):
    impl, oracle = env
    V, E, H, T, B = 10000, 512, 512, 100, 64
    ref, hip = _pair('CharTranslator', _gen_params(V, E, H, B), 22)
    g = torch.Generator().manual_seed(32)
    inps, targs, lens = impl.synthetic_batch(g, T, B, V, _ragged(g, B, T))
    auths = torch.arange(B) % 2
    out = {}
    for tag, m in (('ref', ref), ('hip', hip)):
        scores, _ = m.forward_mltrain(inps, lens, inps, lens, auths=auths)
        loss = F.cross_entropy(pps(scores, lens)[0], pps(targs.to(scores.device), lens)[0])
        loss.backward()
        out[tag] = (scores.detach().cpu(), loss.detach().cpu())
    impl.check_timeouts()
    print('cfg2: scores err %.2e, loss err %.2e' % (rel_err(out['hip'][0], out['ref'][0]), rel_err(out['hip'][1], out['ref'][1])))
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'cfg2 scores')
    assert_close(out['hip'][1], out['ref'][1], TOL_OUT, 'cfg2 loss')
    ref_grads = _grads(ref)
    assert_grads_close(hip.named_parameters(), ref_grads, TOL_GRAD, 'cfg2')
    # the fused projection + log-softmax/NLL entry the benchmark's step uses, same batch
    hip.zero_grad()
    loss = hip.forward_mltrain_loss(inps, lens, inps, lens, targs, auths=auths)
    loss.backward()
    impl.check_timeouts()
    assert_close(loss, out['ref'][1], TOL_OUT, 'cfg2 fused loss')
    assert_grads_close(hip.named_parameters(), ref_grads, TOL_GRAD, 'cfg2 fused')


def _adv_step_case(env, V, E, H, T, B, disc, lens_kind, seed, sharpen):
    impl, oracle = env
    import oracle.models as om
    gp = _gen_params(V, E, H, B)
    ep = _cnn_params(V, E) if disc == 'cnn' else _lstm_params(V, E)
    Gc, Gg = _pair('CharTranslator', gp, seed, sharpen)
    Dc, Dg = _pair('CharCNN' if disc == 'cnn' else 'CharLstm', ep, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    mk_lens = (lambda: _ragged(g, B, T, lo=T // 4)) if lens_kind == 'ragged' else (lambda: [T] * B)
    mk = lambda a: (lambda b: (b[0], b[1], torch.full((B,), a, dtype=torch.long), b[2]))(
        impl.synthetic_batch(g, T, B, V, mk_lens()))
    src, tgt, gb = mk(0), mk(1), mk(0)
    noise = [torch.rand(B, V + 1, generator=g) for _ in range(2 * T)]
    res = {}
    om.MARGIN_LOG = []
    try:
        for name, mod, Gm, Dm, dev in (('cpu', oracle, Gc, Dc, 'cpu'), ('hip', impl, Gg, Dg, 'cuda')):
            optG, optD = mod.make_optimizers(Gm, Dm)
            Gm.set_noise([u.to(dev) for u in noise])
            append = torch.zeros(1, B, V + 1, device=dev)
            append[:, :, 2] = 1.0
            rd = mod.d_update(Gm, Dm, optD, src, tgt, 0, 1, T, 0.5, step=False)
            gD = {k: v.detach().clone().cpu() for k, v in _grads(Dm).items()}
            # grad_clip far above the norm: the gradients stay as backward() left them.  (torch's CPU fp32 vector norm
            # is itself 2.7e-4 low on the 5M-element decoder_W gradient -- sequential accumulation absorbs the small
            # squares; measured against fp64 -- so a clipped comparison would test that, not the kernels.  Clipping
            # at 10 is covered by the step fixtures of tests/test_hip_golden.py.)
            rg = mod.g_update(Gm, Dm, optG, gb, 0, 1, T, 0.5, append, cycle_loss_type='ml', cycle_loss_w=1.0,
                              grad_clip=1e9, step=False)
            res[name] = (rd, rg, gD, {k: v.detach().clone().cpu() for k, v in _grads(Gm).items()})
        margins = torch.cat(om.MARGIN_LOG)
    finally:
        om.MARGIN_LOG = None
    impl.check_timeouts()
    (rdc, rgc, gDc, gGc), (rdh, rgh, gDh, gGh) = res['cpu'], res['hip']
    tag = 'H=%d V=%d T=%d B=%d %s D, %s lens' % (H, V + 1, T, B, disc, lens_kind)
    print('%s: %d sampled tokens, smallest top-1/top-2 gap of (log_softmax(x)+g)/tau in the oracle %.3e; '
          'G generated lengths %d..%d' % (tag, margins.numel(), float(margins.min()), int(rgc['gen_lens'].min()),
                                          int(rgc['gen_lens'].max())))
    for which, a, b in (('D', rdh, rdc), ('G', rgh, rgc)):
        ta, tb = torch.stack(list(a['char_outs'])).cpu().long(), torch.stack(list(b['char_outs'])).long()
        assert ta.shape == tb.shape, '%s %s update: generated %s steps, oracle %s' % (tag, which, ta.shape, tb.shape)
        assert torch.equal(ta, tb), '%s %s update: %d of %d sampled tokens differ' % (tag, which, int((ta != tb).sum()), ta.numel())
    assert torch.equal(rgh['gen_lens'].cpu().long(), rgc['gen_lens'].long())
    assert_close(rdh['fake_logits'], rdc['fake_logits'], TOL_OUT, tag + ' D fake logits')
    assert_close(rdh['real_logits'], rdc['real_logits'], TOL_OUT, tag + ' D real logits')
    assert_close(rdh['loss'], rdc['loss'], TOL_OUT, tag + ' D loss')
    for k in ('loss_gan', 'cyc_loss'):
        assert_close(rgh[k], rgc[k], TOL_OUT, tag + ' G ' + k)
    # global gradient norm against the fp64 norm of the oracle's gradients (see above)
    norm64 = torch.sqrt(sum((v.double() ** 2).sum() for v in gGc.values()))
    assert_close(rgh['grad_norm'].double().cpu(), norm64, TOL_GRAD, tag + ' G grad norm')
    _assert_grad_dicts(gDh, gDc, TOL_GRAD, tag + ' D grads')
    worst = _assert_grad_dicts(gGh, gGc, TOL_GRAD, tag + ' G grads')
    print('%s: D loss err %.2e, G cycle loss err %.2e, worst G grad err %.2e' % (
        tag, rel_err(rdh['loss'], rdc['loss']), rel_err(rgh['cyc_loss'], rgc['cyc_loss']), worst))
    return (Gc, Gg)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('lens_kind', ['ragged', 'all-T'])
def test_cfg3_adversarial_step_full_dims(env, lens_kind):
    """cfg 3 at H=E=512, V=10001, T=100 with 16 sentences: D update then G update, then greedy decoding."""
    impl, oracle = env
    import oracle.models as om
    V, E, H, T, B = 10000, 512, 512, 100, 16
    Gc, Gg = _adv_step_case(env, V, E, H, T, B, 'lstm', lens_kind, 41 if lens_kind == 'ragged' else 51, sharpen=30.0)
    # greedy decoding (eval_gantranslate.py:25): bit-exact tokens
    g = torch.Generator().manual_seed(77)
    inps, _, lens = impl.synthetic_batch(g, T, B, V, _ragged(g, B, T, lo=T // 4))
    auths = torch.ones(B, dtype=torch.long)
    om.MARGIN_LOG = []
    try:
        outs = []
        for m in (Gc, Gg):
            m.eval()
            with torch.no_grad():
                s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=False, n_max=T, end_c=1, auths=auths)
            outs.append((torch.stack(list(co)).cpu().long(), gl.cpu().long()))
        margins = torch.cat(om.MARGIN_LOG)
    finally:
        om.MARGIN_LOG = None
    impl.check_timeouts()
    print('cfg3 greedy: %d tokens, smallest top-1/top-2 logit gap in the oracle %.3e (logit scale %.3e)' % (
        margins.numel(), float(margins.min()), float(Gc.decoder_W.abs().max())))
    assert torch.equal(outs[0][0], outs[1][0]), 'greedy tokens differ: %d of %d' % (int((outs[0][0] != outs[1][0]).sum()), outs[0][0].numel())
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.timeout(1200)
def test_cfg5_shape_adversarial_step(env):
    """cfg 5's shape (translator 1x1024, V=30001, T=200, CharCNN discriminator) with 8 sentences."""
    _adv_step_case(env, 30000, 512, 1024, 200, 8, 'cnn', 'ragged', 61, sharpen=30.0)


def _cfg5_inputs(impl, B=8, seed=61):
    """The sentences and the Gumbel uniforms of test_cfg5_shape_adversarial_step's generator update (same generator
    stream as _adv_step_case: three batches, then 2T noise tensors of which the G update uses the last T)."""
    V, T = 30000, 200
    g = torch.Generator().manual_seed(seed + 2)
    mk = lambda a: (lambda b: (b[0], b[1], torch.full((B,), a, dtype=torch.long), b[2]))(
        impl.synthetic_batch(g, T, B, V, _ragged(g, B, T, lo=T // 4)))
    src, tgt, gb = mk(0), mk(1), mk(0)
    noise = [torch.rand(B, V + 1, generator=g) for _ in range(2 * T)]
    return gb, noise[T:]


@pytest.mark.timeout(900)
def test_cfg5_production_geometry_rows(env):
    """cfg 5 at its BENCHMARK geometry.  H=1024 with B=256 runs the 64-rows-per-group sweeps (lstm_fwd_kernel<4,16>
    and the 64-row backward) where the oracle-pinned 8-sentence case above runs the 16-row kernels.  Rows are
    independent, so: the 8 oracle-pinned sentences (same weights, ids, lengths and Gumbel uniforms as
    test_cfg5_shape_adversarial_step's generator update) are rows 0-7 of a 256-row batch whose other rows are
    random sentences; a loss over rows 0-7 only (GAN loss of the CharCNN discriminator on the sampled sentences +
    teacher-forced CE of the translator) must give the same sampled tokens (exact), the same per-row logits / CE
    (1e-5) and the same parameter gradients (1e-5 of each gradient's max: the other rows contribute exact zeros)
    as the 8-row run.  Both runs are the HIP path; the 8-row run is what the test above compares with the oracle."""
    impl, oracle = env
    V, E, H, T, R, BB = 30000, 512, 1024, 200, 8, 256
    torch.manual_seed(61)
    G = impl.CharTranslator(_gen_params(V, E, H, R)).cuda().train()
    with torch.no_grad():
        G.decoder_W.mul_(30.0)
    torch.manual_seed(62)
    D = impl.CharCNN(_cnn_params(V, E)).cuda().train()
    (inps, targs, auths, lens), noise = _cfg5_inputs(impl, R)
    g2 = torch.Generator().manual_seed(991)
    xi, xt, xl = impl.synthetic_batch(g2, T, BB - R, V, [int(v) for v in torch.randint(T // 4, T + 1, (BB - R,), generator=g2)])
    big = (torch.cat([inps, xi], 1), torch.cat([targs, xt], 1), torch.cat([auths, torch.zeros(BB - R, dtype=torch.long)]),
           list(lens) + list(xl))
    assert impl.ops.lstm_geometry(BB, H) == (4, 4) and impl.ops.lstm_geometry(R, H)[0] == 1
    res = {}
    for tag, (bi, bt, ba, bl) in (('rows8', (inps, targs, auths, lens)), ('rows256', big)):
        B = bi.shape[1]
        G.zero_grad(); D.zero_grad()
        U = []
        for t in range(T):
            u = torch.rand(B, V + 1, device='cuda', generator=None) if B > R else None
            if u is None:
                u = noise[t].cuda()
            else:
                u[:R] = noise[t].cuda()
            U.append(u)
        G.set_noise(U)
        out = impl.adv_forward_pass(G, D, bi, bl, end_c=1, backprop_for='all', maxlen=T, auths=ba, cycle_compute=False)
        logits, toks, gl = out[0], torch.stack(list(out[4])), out[3]
        loss = 5 * F.binary_cross_entropy_with_logits(logits[:R, 1], torch.ones(R, device='cuda'), reduction='sum')
        scores, _ = G.forward_mltrain(bi, bl, bi, bl, auths=ba)
        ce = torch.stack([F.cross_entropy(scores[:bl[r], r], bt[:bl[r], r].cuda(), reduction='sum') for r in range(R)])
        (loss + ce.sum() / 100.0).backward()
        impl.check_timeouts()
        res[tag] = dict(tok=toks[:, :R].cpu().clone(), gl=gl[:R].cpu().clone(), logits=logits[:R].detach().cpu(),
                        ce=ce.detach().cpu(), gG={k: v.detach().cpu().clone() for k, v in _grads(G).items()},
                        gD={k: v.detach().cpu().clone() for k, v in _grads(D).items()})
        del scores, out, U
        G.set_noise(None)
        torch.cuda.empty_cache()
    a, b = res['rows8'], res['rows256']
    assert torch.equal(a['gl'], b['gl'])
    assert torch.equal(a['tok'], b['tok']), '%d of %d sampled tokens differ' % (int((a['tok'] != b['tok']).sum()), a['tok'].numel())
    assert_close(b['logits'], a['logits'], TOL_OUT, 'cfg5 rows 0-7 of 256: D logits')
    assert_close(b['ce'], a['ce'], TOL_OUT, 'cfg5 rows 0-7 of 256: per-row CE')
    wG = _assert_grad_dicts(b['gG'], a['gG'], 1e-5, 'cfg5 rows 0-7 of 256: G grads')
    wD = _assert_grad_dicts(b['gD'], a['gD'], 1e-5, 'cfg5 rows 0-7 of 256: D grads')
    print('cfg5 B=256 geometry vs the 8-row run: %d tokens exact, logits err %.2e, CE err %.2e, worst G grad err %.2e, '
          'worst D grad err %.2e' % (a['tok'].numel(), rel_err(b['logits'], a['logits']), rel_err(b['ce'], a['ce']), wG, wD))
