#!/usr/bin/env python3
"""Oracle (test infrastructure): golden vectors for the callers' data side, recorded FROM THE REFERENCE'S OWN
utils/data_provider.py and utils/utils.py.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_utils.py

Those two files are Python 2 (print statements, xrange, cPickle ...): they are read as text, passed through
lib2to3 IN MEMORY (nothing of them is written anywhere) and executed as modules, next to the reference's model
files loaded through the shim of oracle/gen_golden.py.  Harness-side shims, none of which touches a computed value:
``np.int = int`` (removed from numpy), an empty ``preproc_dataset`` module (imported at the top of
data_provider.py, only called when the dataset JSON is missing; its own dependency chardet is not installed), and
``nn.CrossEntropyLoss`` returning a 1-element tensor as torch 0.3 did (utils.py reads ``loss.data.cpu().numpy()[0]``),
and the author-id tensor of prepare_data handing out Python numbers when indexed, as 1-D tensors did in torch 0.3
(utils.py compares ``auths[0]`` with numpy arrays).

What is recorded (tests/golden/utils_ref.npz: a synthetic dataset as JSON, call scripts' outputs as JSON / arrays,
state_dicts and metrics): DataProvider's vocabularies, author index, sentence / document batches under fixed numpy
seeds, prepare_data tensors, the document and sentence iterators, length-bucket sampling; eval_model,
eval_classify and eval_translator over reference models on that dataset.  The reference's character-sentence
accessor indexes its sentence argument (``sents[sidx[0]]``) and therefore fails for the plain integers that
iter_sentences / sample_by_len pass: those two paths are recorded with atoms='word', the mode that runs.
tests/test_utils_golden.py replays the same scripts on the product's DataProvider / eval_utils.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
REF = '/root/reference'


def load_py2(path, name):
    """Execute a Python-2 source file as module `name` after an in-memory lib2to3 pass."""
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    with open(path) as f:
        tree = tool.refactor_string(f.read() + '\n', name)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(str(tree), path, 'exec'), mod.__dict__)
    return mod


def dataset(seed=4, n_docs=14):
    rng = np.random.RandomState(seed)
    words = ['ab', 'ba', 'cab', 'dad', 'egg', 'fig', 'go', 'hat', 'in', 'jam', 'kit', 'log', 'map', 'nor', 'oak']
    docs = []
    for d in range(n_docs):
        sents = []
        for _ in range(rng.randint(2, 6)):
            sents.append(' '.join(rng.choice(words, rng.randint(3, 9))))
        text = '2' + '.'.join(sents) + '.'
        docs.append({'text': text, 'tokens': list(sents), 'author': 'A' if d % 2 == 0 else 'B',
                     'split': 'train' if d < n_docs - 5 else 'val', 'actage': 20 + d})
    return {'docs': docs, 'configs': {'start': '2', 'end': '.'}}


def strip(batch):
    """JSON-able view of a batch (ids may be numpy integers)."""
    out = []
    for b in batch:
        out.append({k: (int(v) if isinstance(v, (np.integer,)) else v) for k, v in b.items()})
    return out


def provider_script(dp, rec):
    """The call script both sides run; `rec(name, value)` stores one result."""
    a2i, i2a = dp.createAuthorIdx()
    c2i, i2c = dp.createCharVocab(threshold=3)
    w2i, i2w = dp.createWordVocab(threshold=1)
    rec('author_idx', a2i)
    rec('char_vocab', c2i)
    rec('word_vocab', w2i)
    rec('num_seqs', [int(dp.get_num_seqs(split=s)) for s in ('train', 'val')] + [int(dp.get_num_seqs(maxlen=7, split='val'))])
    rec('num_sents', [int(dp.get_num_sents(split=s)) for s in ('train', 'val')])
    rec('num_sent_doc', [[int(dp.get_num_sent_doc(c, atoms=a)) for c in range(4)] for a in ('char', 'word')])
    rec('min_len_cdist', [int(dp.min_len)] + [int(v) for v in dp.lenCdist])
    k = 0
    for seed, bs, split, aid, leak, maxlen in ((5, 6, 'train', None, 0., None), (6, 9, 'train', 'B', 0.5, 7),
                                               (7, 4, 'val', None, 0., 20), (8, 5, 'val', 'A', 0.3, 3)):
        np.random.seed(seed)
        batch = dp.get_sentence_batch(bs, split=split, atoms='char', aid=aid)
        rec('sb%d_batch' % k, strip(batch))
        inps, targs, auths, lens = dp.prepare_data(batch, c2i, a2i, leakage=leak, maxlen=maxlen)
        rec('sb%d_inps' % k, inps.numpy()); rec('sb%d_targs' % k, targs.numpy()); rec('sb%d_auths' % k, auths.numpy())
        rec('sb%d_lens' % k, [int(v) for v in lens])
        rec('sb%d_order' % k, strip(batch))                      # prepare_data reorders the batch in place
        k += 1
    # (word sentences drawn at random index a list with a 1-element array, which today's numpy refuses; the
    # length-bucket path hands the accessor a plain integer and runs)
    np.random.seed(9)
    wb = dp.get_sentence_batch(5, split='train', atoms='word', sample_by_len=True)
    rec('wb_batch', strip(wb))
    wi, wt, wa, wl = dp.prepare_data(wb, w2i, a2i)
    rec('wb_inps', wi.numpy()); rec('wb_targs', wt.numpy()); rec('wb_lens', [int(v) for v in wl])
    np.random.seed(10)
    rec('rand_len', [int(dp.getRandLen()) for _ in range(25)])
    np.random.seed(11)
    rec('bylen_batch', strip(dp.get_sentence_batch(4, split='train', atoms='word', sample_by_len=True)))
    rec('single_doc', [[b[0]['in'], b[0]['targ'], b[0]['author'], bool(d)] for b, d in dp.iter_single_doc(split='val')])
    np.random.seed(12)
    rec('single_doc_sampled', [[b[0]['in'], b[0]['targ'], bool(d)] for b, d in dp.iter_single_doc(split='train', max_docs=3)])
    rec('iter_sentences_word', [[strip(b), bool(d)] for b, d in dp.iter_sentences(split='val', batch_size=3, atoms='word')])
    rec('iter_bylen_word', [[strip(b), bool(d)] for b, d in dp.iter_sentences_bylen(split='val', batch_size=3, atoms='word',
                                                                                    auths=['A', 'B'])])
    np.random.seed(13)
    rdb = []
    for _ in range(3):
        b, dead = dp.get_rand_doc_batch(3, split='train')
        rdb.append([strip(b), [int(v) for v in dead]])
    rec('rand_doc_batch', rdb)
    np.random.seed(14)
    db = []
    for _ in range(4):
        b, dead = dp.get_doc_batch(split='train')
        db.append([strip(b), [int(v) for v in dead]])
    rec('doc_batch', db)
    return a2i, c2i, w2i


PROVIDER_PARAMS = {'dataset': 'speechdata', 'batch_size': 3, 'max_seq_len': 12}


def model_params(V_char, V_word):
    ep = dict(vocabulary_size=V_word, embedding_size=16, hidden_size=64, hidden_depth=1, num_output_layers=2,
              mode='classifier', drop_prob_encoder=0.0, drop_prob_decoder=0.0, bidir=False, en_residual_conn=0,
              maxpoolrnn=0)
    lp = dict(ep, mode='generative', vocabulary_size=V_char)
    gp = dict(vocabulary_size=V_word, embedding_size=16, enc_hidden_size=64, dec_hidden_size=64, enc_hidden_depth=1,
              dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=4, num_output_layers=2, batch_size=4,
              drop_prob_emb=0.0, drop_prob_encoder=0.0, drop_prob_decoder=0.0, max_seq_len=12, atoms='word',
              maxpoolrnn=0)
    return ep, lp, gp


def main():
    import gen_golden
    ct, cl, cc, mu = gen_golden.install_shim()
    np.int = int                                                   # removed from numpy; the reference uses it as a dtype
    sys.modules['preproc_dataset'] = types.ModuleType('preproc_dataset')
    sys.modules['preproc_dataset'].preproc_dataset = None          # only called when the dataset JSON is missing
    ref_dp = load_py2(os.path.join(REF, 'utils', 'data_provider.py'), 'ref_data_provider')
    ref_ut = load_py2(os.path.join(REF, 'utils', 'utils.py'), 'ref_utils')

    data = dataset()
    out = {'dataset_json': np.array(json.dumps(data))}

    def rec(name, value):
        out[name] = value if isinstance(value, np.ndarray) else np.array(json.dumps(value))

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'data', 'speechdata'))
        with open(os.path.join(d, 'data', 'speechdata', 'dataset.json'), 'w') as f:
            json.dump(data, f)
        os.chdir(d)
        try:
            dp = ref_dp.DataProvider(dict(PROVIDER_PARAMS))
            a2i, c2i, w2i = provider_script(dp, rec)

            # ---- evaluation procedures over reference models ------------------------------------------------
            ep, lp, gp = model_params(len(c2i), len(w2i))
            torch.manual_seed(3)
            clf, lm, tr = cl.CharLstm(ep), cl.CharLstm(lp), ct.CharTranslator(gp)
            for tag, m in (('clf', clf), ('lm', lm), ('tr', tr)):
                for k, v in m.state_dict().items():
                    out['%s.%s' % (tag, k)] = v.detach().numpy().copy()
            out['model_params_json'] = np.array(json.dumps({'ep': ep, 'lp': lp, 'gp': gp}))
            dpe = ref_dp.DataProvider(dict(PROVIDER_PARAMS))       # a fresh provider: no iterator state carried over

            class Ids03(torch.Tensor):                             # torch 0.3: indexing a 1-D tensor gave a Python number
                def __getitem__(self, i):
                    v = torch.Tensor.__getitem__(self, i)
                    return v.item() if isinstance(i, int) and v.dim() == 0 else v
            prep = dpe.prepare_data

            def prepare_data_03(*a, **k):
                inps, targs, auths, lens = prep(*a, **k)
                return inps, targs, auths.as_subclass(Ids03), lens
            dpe.prepare_data = prepare_data_03
            ce = nn.CrossEntropyLoss

            class CE03(ce):                                        # torch 0.3 losses were 1-element tensors
                def forward(self, *a, **k):
                    return super().forward(*a, **k).view(1)
            nn.CrossEntropyLoss = CE03
            try:
                np.random.seed(1)
                r_clf = ref_ut.eval_classify(dpe, clf.eval(), {'atoms': 'word'}, w2i, a2i, split='val')
                np.random.seed(1)
                r_lm = ref_ut.eval_model(dpe, lm.eval(), {}, c2i, a2i, split='val')
                r_tr = ref_ut.eval_translator(dpe, tr, gp, w2i, a2i, split='val')
            finally:
                nn.CrossEntropyLoss = ce
            out['eval_classify'] = np.array([float(r_clf[0]), float(r_clf[1])])
            out['eval_model'] = np.array([float(r_lm[0]), float(r_lm[1])])
            out['eval_translator'] = np.array([float(r_tr[0]), float(r_tr[1])])
        finally:
            os.chdir(cwd)
    path = os.path.join(ROOT, 'tests', 'golden', 'utils_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote %s  %.1f KB' % (path, os.path.getsize(path) / 1024.))
    print('eval_classify', out['eval_classify'], 'eval_model', out['eval_model'], 'eval_translator', out['eval_translator'])


if __name__ == '__main__':
    main()
