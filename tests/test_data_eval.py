"""DataProvider / prepare_data (bit-exact against the oracle restatement) and the evaluation procedures.

CPU part: integer plumbing and the eval procedures over the oracle models.  GPU part: the same procedures
over the HIP-backed modules give the oracle models' metrics on identical weights."""
import copy
import os
import tempfile

import numpy as np
import pytest
import torch

import oracle
from oracle import data_prep


def _dataset(seed=0, n_docs=12):
    rng = np.random.RandomState(seed)
    alphabet = list('abcdefghijklmnop qrstuv')
    docs = []
    for d in range(n_docs):
        sents = []
        for _ in range(rng.randint(2, 6)):
            n = rng.randint(3, 25)
            sents.append(''.join(rng.choice(alphabet, n)).strip() or 'ab')
        text = '2' + '.'.join(sents) + '.'
        docs.append({'text': text, 'tokens': [s for s in sents], 'author': 'A' if d % 2 == 0 else 'B',
                     'split': 'train' if d < n_docs - 4 else 'val', 'actage': 20 + d})
    return {'docs': docs, 'configs': {'start': '2', 'end': '.'}}


def _provider():
    # loaded by path: importing the a4nt_amd package itself needs the built HIP library and a GPU
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('a4nt_dp', os.path.join(here, 'a4nt-author-masking_amd', 'data_provider.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.DataProvider


def _eval_utils():
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('a4nt_eval', os.path.join(here, 'a4nt-author-masking_amd', 'eval_utils.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_prepare_data_bit_exact_against_oracle():
    DP = _provider()
    dp = DP({'dataset': 'speechdata', 'batch_size': 4, 'max_seq_len': 20}, data=_dataset())
    char_to_ix, ix_to_char = dp.createCharVocab(threshold=3)
    assert 0 not in ix_to_char and min(ix_to_char) == 1
    char_to_ix.pop('q', None)                      # force out-of-vocabulary symbols on both sides
    auth_to_ix, ix_to_auth = dp.createAuthorIdx()
    assert auth_to_ix == {'A': 0, 'B': 1}
    for maxlen in (None, 7, 20):
        for leakage in (0., 0.5):
            np.random.seed(5)
            batch = dp.get_sentence_batch(9, split='train', atoms='char')
            b1, b2 = copy.deepcopy(batch), copy.deepcopy(batch)
            np.random.seed(11)
            got = dp.prepare_data(b1, char_to_ix, auth_to_ix, leakage=leakage, maxlen=maxlen)
            np.random.seed(11)
            ref = data_prep.prepare_data(b2, char_to_ix, auth_to_ix, leakage=leakage, maxlen=maxlen)
            for a, b in zip(got[:3], ref[:3]):
                assert a.dtype == torch.int64 and torch.equal(a, b)
            assert got[3] == ref[3] and got[3] == sorted(got[3], reverse=True)
            assert [x['in'] for x in b1] == [x['in'] for x in b2]          # same in-place reorder


def test_sentence_and_document_windows():
    DP = _provider()
    data = _dataset(1)
    dp = DP({'dataset': 'blogdata', 'batch_size': 2, 'max_seq_len': 16}, data=data)
    for cid in range(3):
        n = dp.get_num_sent_doc(cid)
        for j in range(n):
            assert dp.get_rand_sentence(cid, sidx=j) == data_prep.rand_sentence(data['docs'][cid]['text'], j)
    # document windows tile the text: concatenated targets == text[1:]
    for cid in dp.splits['val'][:2]:
        cur = np.ones(len(data['docs']), dtype=np.int64)
        pieces, done = [], False
        while not done:
            c0 = int(cur[cid])
            inp, targ, done, cur = dp.get_nextstring_doc(cid, cur)
            assert (inp, targ, done) == data_prep.nextstring_doc(data['docs'][cid]['text'], c0, 16)[:3]
            pieces.append(targ)
        assert ''.join(pieces) == data['docs'][cid]['text'][1:]
    total = sum(1 for _ in dp.iter_single_doc(split='val'))
    assert total == dp.get_num_seqs(split='val')
    nb = sum(len(b) for b, _ in dp.iter_sentences(split='val', batch_size=3))
    assert nb == sum(dp.get_num_sent_doc(c) for c in dp.splits['val'])
    b = dp.get_sentence_batch(5, split='train', aid='B', sample_by_len=True)
    assert len(b) == 5 and all(x['author'] == 'B' for x in b)
    # age-group filter relabels the authors
    dp2 = DP({'dataset': 'blogdata', 'filterauths': ['24', '40'], 'filtertype': 'agegroup'}, data=_dataset(1))
    assert set(d['author'] for d in dp2.data['docs']) <= {'<24', '<40'}


def _setup(impl, device, DP):
    data = _dataset(2, n_docs=10)
    dp = DP({'dataset': 'speechdata', 'batch_size': 4, 'max_seq_len': 12}, data=data)
    char_to_ix, _ = dp.createCharVocab(threshold=1)
    auth_to_ix, _ = dp.createAuthorIdx()
    V = len(char_to_ix)
    torch.manual_seed(3)
    ep = dict(vocabulary_size=V, embedding_size=16, hidden_size=64, hidden_depth=1, num_output_layers=2,
              mode='classifier', drop_prob_encoder=0.0, drop_prob_decoder=0.0)
    lp = dict(ep, mode='generative')
    gp = dict(vocabulary_size=V, embedding_size=16, enc_hidden_size=64, dec_hidden_size=64, enc_hidden_depth=1,
              dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=4, num_output_layers=2, batch_size=4,
              drop_prob_emb=0.0, drop_prob_encoder=0.0, drop_prob_decoder=0.0, max_seq_len=12)
    ref = (oracle.CharLstm(ep), oracle.CharLstm(lp), oracle.CharTranslator(gp))
    if impl is oracle:
        return dp, char_to_ix, auth_to_ix, ref, gp
    mine = (impl.CharLstm(ep), impl.CharLstm(lp), impl.CharTranslator(gp))
    for a, b in zip(mine, ref):
        a.load_state_dict({k: v.to(device) for k, v in b.state_dict().items()})
    return dp, char_to_ix, auth_to_ix, mine, gp


def _run_evals(ev, dp, c2i, a2i, models, gp):
    clf, lm, tr = models
    np.random.seed(1)
    r1 = ev.eval_classify(dp, clf.eval(), {'atoms': 'char'}, c2i, a2i, split='val', batch_size=4)
    np.random.seed(1)
    r2 = ev.eval_model(dp, lm.eval(), {}, c2i, a2i, split='val')
    r3 = ev.eval_translator(dp, tr, gp, c2i, a2i, split='val', batch_size=4)
    return r1, r2, r3


def test_eval_procedures_cpu_and_checkpoint_roundtrip():
    ev, DP = _eval_utils(), _provider()
    dp, c2i, a2i, models, gp = _setup(oracle, 'cpu', DP)
    r1, r2, r3 = _run_evals(ev, dp, c2i, a2i, models, gp)
    assert 0. <= r1[0] <= 100. and 0. <= r1[1] <= 1. and r1[2]['n_docs'] == len(dp.splits['val'])
    assert 0. <= r2[0] <= 1. and 0. <= r2[1] <= 1.
    assert r3[1] == pytest.approx(np.exp(r3[0])) and r3[0] > 0
    # checkpoint dictionary: written with torch.save, read back with the tensors-only loader
    optG, optE = oracle.make_optimizers(models[2], models[0])
    cp = ev.make_checkpoint(7, gp, {'hidden_size': 64}, {'char_to_ix': c2i, 'auth_to_ix': a2i}, models[2], models[0], optG, optE)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'checkpoint_gan_test.pth.tar')
        torch.save(cp, path)
        back = ev.load_checkpoint(path)
    assert back['iter'] == 7 and back['misc']['char_to_ix'] == c2i
    for k, v in models[2].state_dict().items():
        assert torch.equal(back['state_dict'][k], v)


@pytest.mark.gpu
def test_eval_procedures_hip_match_oracle():
    import a4nt_amd
    ev, DP = a4nt_amd.eval_utils, a4nt_amd.DataProvider
    dp, c2i, a2i, models, gp = _setup(a4nt_amd, 'cuda', DP)
    got = _run_evals(ev, dp, c2i, a2i, models, gp)
    dp_o, c2i_o, a2i_o, models_o, _ = _setup(oracle, 'cpu', DP)
    want = _run_evals(ev, dp_o, c2i_o, a2i_o, models_o, gp)
    assert got[0][0] == want[0][0] and got[0][1] == want[0][1]                 # accuracies: exact
    assert got[1][0] == want[1][0] and abs(got[1][1] - want[1][1]) < 1e-12     # rank, AUC
    assert abs(got[2][0] - want[2][0]) < 1e-5 * max(1.0, abs(want[2][0]))      # CE loss within the parity tolerance
    inps, targs, auths, lens = dp.prepare_data(dp.get_sentence_batch(4), c2i, a2i)
    assert inps.is_pinned() and targs.is_pinned()
    a4nt_amd.check_timeouts()
