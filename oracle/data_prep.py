"""Oracle (test infrastructure): plain-loop restatement of DataProvider.prepare_data and of the two string
windowing helpers of the reference's utils/data_provider.py (that file has Python-2 print statements and
cannot be imported as it is; oracle/gen_golden_utils.py runs it through lib2to3 in memory and records
tests/golden/utils_ref.npz from it, which pins the product's DataProvider -- and with it this restatement, which
tests/test_data_eval.py compares the product with -- to the reference file).

  * prepare_data           utils/data_provider.py:310-348
  * rand_sentence          utils/data_provider.py:257-272
  * nextstring_doc         utils/data_provider.py:185-200
"""
import numpy as np
import torch


def prepare_data(batch, char_to_ix, auth_to_ix, leakage=0., maxlen=None):
    inp_seqs, targ_seqs, lens, auths = [], [], [], []
    for b in batch:
        inp_seqs.append([char_to_ix[c] for c in b['in'][:maxlen] if c in char_to_ix])
        targ_seqs.append([char_to_ix[c] for c in b['targ'][:maxlen] if c in char_to_ix])
        lens.append(len(inp_seqs[-1]))
        if len(targ_seqs[-1]) != lens[-1]:
            if len(targ_seqs[-1]) < lens[-1]:
                targ_seqs[-1].append(0)
            else:
                inp_seqs[-1].insert(0, 0)
                lens[-1] = lens[-1] + 1
        if np.random.rand() >= leakage:
            auths.append(auth_to_ix[b['author']])
        else:
            auths.append(np.random.choice(list(auth_to_ix.values())))
    max_len = max(lens)
    inp_arr = np.zeros((max_len, len(batch)), dtype=np.int64)
    targ_arr = np.zeros((max_len, len(batch)), dtype=np.int64)
    lens_arr = np.array(lens, dtype=np.int64)
    sort_idx = np.argsort(lens_arr)[::-1]
    lens_arr = lens_arr[sort_idx]
    auths_arr = np.array(auths, dtype=np.int64)[sort_idx]
    batch[:] = [batch[i] for i in sort_idx]
    for i, j in enumerate(sort_idx):
        inp_arr[:lens_arr[i], i] = inp_seqs[j]
        targ_arr[:lens_arr[i], i] = targ_seqs[j]
    return torch.from_numpy(inp_arr), torch.from_numpy(targ_arr), torch.from_numpy(auths_arr.copy()), lens_arr.tolist()


def rand_sentence(text, sidx):
    sents = [st for st in text.split('.') if len(st) > 0]
    s = sents[sidx]
    if s[0] == '2':
        return s, s[1:] + '.'
    return '2' + s, s + '.'


def nextstring_doc(text, cidx, maxlen):
    eidx = cidx + maxlen if cidx + maxlen < len(text) else len(text)
    targ, inp = text[cidx:eidx], text[cidx - 1:eidx - 1]
    done = eidx == len(text)
    return inp, targ, done, (1 if done else eidx)
