"""CPU: the product's DataProvider and evaluation procedures against vectors recorded from the REFERENCE's own
utils/data_provider.py and utils/utils.py (tests/golden/utils_ref.npz, written by oracle/gen_golden_utils.py, which
runs those Python-2 files through an in-memory lib2to3 pass).  The same call script is replayed on the product
(integer / string results exact, metrics to 1e-6)."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import oracle                                   # noqa: E402
from oracle import gen_golden_utils as gg       # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location('a4nt_' + name, os.path.join(ROOT, 'a4nt-author-masking_amd', name + '.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fixture():
    return np.load(os.path.join(HERE, 'golden', 'utils_ref.npz'))


def test_data_provider_replays_the_reference():
    fx = _fixture()
    data = json.loads(str(fx['dataset_json']))
    dp = _load('data_provider').DataProvider(dict(gg.PROVIDER_PARAMS), data=data)
    got = {}

    def rec(name, value):
        got[name] = value
    gg.provider_script(dp, rec)
    assert len(got) >= 35
    for name, value in got.items():
        want = fx[name]
        if isinstance(value, np.ndarray):
            assert value.dtype == np.int64 and want.shape == value.shape and np.array_equal(want, value), name
        else:
            assert json.loads(json.dumps(value)) == json.loads(str(want)), name


def test_eval_procedures_reproduce_the_reference_metrics():
    fx = _fixture()
    data = json.loads(str(fx['dataset_json']))
    DP, ev = _load('data_provider').DataProvider, _load('eval_utils')
    dp = DP(dict(gg.PROVIDER_PARAMS), data=data)
    a2i, c2i, w2i = json.loads(str(fx['author_idx'])), json.loads(str(fx['char_vocab'])), json.loads(str(fx['word_vocab']))
    mp = json.loads(str(fx['model_params_json']))
    models = {}
    for tag, cls, params in (('clf', oracle.CharLstm, mp['ep']), ('lm', oracle.CharLstm, mp['lp']), ('tr', oracle.CharTranslator, mp['gp'])):
        m = cls(params)
        sd = {k[len(tag) + 1:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(tag + '.')}
        res = m.load_state_dict(sd)
        assert not res.missing_keys and not res.unexpected_keys
        models[tag] = m
    np.random.seed(1)
    r_clf = ev.eval_classify(dp, models['clf'].eval(), {'atoms': 'word'}, w2i, a2i, split='val')
    np.random.seed(1)
    r_lm = ev.eval_model(dp, models['lm'].eval(), {}, c2i, a2i, split='val')
    r_tr = ev.eval_translator(dp, models['tr'], mp['gp'], w2i, a2i, split='val')
    np.testing.assert_allclose([r_clf[0], r_clf[1]], fx['eval_classify'], rtol=0, atol=1e-9)       # accuracies
    np.testing.assert_allclose([r_lm[0], r_lm[1]], fx['eval_model'], rtol=0, atol=1e-9)            # mean rank, AUC
    np.testing.assert_allclose([r_tr[0], r_tr[1]], fx['eval_translator'], rtol=1e-6)               # CE, perplexity
