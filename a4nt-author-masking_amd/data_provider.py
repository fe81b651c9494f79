"""Py3 DataProvider with the interface of the reference's utils/data_provider.py:13-396: sentence / document
sampling, iteration, vocabulary building and `prepare_data`, the function that turns a batch of strings into
the (inps [T,B] int64, targs [T,B] int64, auths [B] int64, lens list) the model layer takes.

Differences that matter on MI355X: the id matrices are assembled in one pass into PINNED host tensors (when a
GPU is present), so the modules' `x.to(device)` is an asynchronous DMA instead of a pageable copy; everything
else (sorting by length, OOV handling, padding with 0, the `batch[:]` in-place reorder) follows the reference.
The raw-text preprocessing (utils/preproc_dataset.py) is not restated: a dataset is the reference's
`dataset.json` (or the same dict passed in memory).
"""
import json
import os.path as osp
from bisect import bisect
from collections import Counter, defaultdict

import numpy as np
import torch


class DataProvider(object):
    def __init__(self, params, data=None):
        dataset = params['dataset']
        if data is not None:
            self.data = data
        else:
            path = osp.join('data', dataset, params.get('dataset_file', 'dataset.json'))
            with open(path, 'r') as f:
                self.data = json.load(f)
        self.athstr = params.get('authstring') or 'author'
        fa = params.get('filterauths', [])
        if len(fa) > 0:                                               # data_provider.py:25-41
            ftype = params.get('filtertype', 'keep')
            if ftype == 'keep':
                keep = set(fa)
                self.data['docs'] = [d for d in self.data['docs'] if str(d[self.athstr]) in keep]
            elif ftype in ('agegroup', 'agegroup-grt'):
                edges = [int(a) for a in fa]
                for d in self.data['docs']:
                    grp = bisect(edges, d['actage'])
                    if ftype == 'agegroup':
                        d[self.athstr] = '<' + str(edges[grp]) if grp < len(edges) else 'None'
                    else:
                        d[self.athstr] = '>' + str(edges[grp - 1]) if grp > 0 else 'None'
                self.data['docs'] = [d for d in self.data['docs'] if d[self.athstr] != 'None']
        self.splits = defaultdict(list)
        for i, dc in enumerate(self.data['docs']):
            self.splits[dc['split']].append(i)
        self.cur_batch = np.array([-1] * params.get('batch_size', 1), dtype=np.int64)
        self.max_seq_len = params.get('max_seq_len', 100)
        self.cur_char_idx = np.ones(len(self.data['docs']), dtype=np.int64)
        self.hid_cache = {}
        self.use_unk = params.get('use_unk', 0)
        if dataset == 'blogdata':
            self.min_len, self.max_len = 2, 32
        elif dataset == 'speechdata':
            self.min_len, self.max_len = 2, 30
        else:
            raise ValueError('ERROR: Dont know how to do len splitting for this dataset')
        # sentences bucketed by (clipped) word count, for length-matched batches (:69-85)
        self.lenMap, lenHist = {}, {}
        for sp in self.splits:
            self.lenMap[sp] = defaultdict(list)
            lenHist[sp] = defaultdict(int)
            for iid in self.splits[sp]:
                for sid, tkn in enumerate(self.data['docs'][iid]['tokens']):
                    ix = max(min(len(tkn.split()), self.max_len), self.min_len)
                    lenHist[sp][ix] += 1
                    self.lenMap[sp][ix].append((iid, sid))
        self.min_len = min(lenHist['train'].keys())
        if not params.get('uniform_len_sample', 0):
            self.lenCdist = np.cumsum(list(lenHist['train'].values()))
        else:
            self.lenCdist = np.arange(len(lenHist['train'])) + 1
        self._pin = torch.cuda.is_available()

    # -- hidden-state cache of the document-level LM training (:88-99) -----------------------------
    def set_hid_cache(self, idces, hid_state):
        n_hid = hid_state[0].shape[1]
        for i, cid in enumerate(idces):
            self.hid_cache[cid] = [hd.detach()[:, min(i, n_hid - 1):min(i, n_hid - 1) + 1].clone() for hd in hid_state]

    def get_hid_cache(self, idces, hid_state):
        return tuple(torch.cat([self.hid_cache[cid][i] for cid in idces], dim=1) for i in range(len(hid_state)))

    # -- documents ---------------------------------------------------------------------------------
    def get_nextstring_doc(self, i, cur_char_idx, maxlen=-1):
        """Next window of document i: targ = text[c:e], inp = text[c-1:e-1]; wraps at the end (:185-200)."""
        maxlen = self.max_seq_len if maxlen == -1 else maxlen
        cidx = int(cur_char_idx[i])
        txt = self.data['docs'][i]['text']
        eidx = cidx + maxlen if cidx + maxlen < len(txt) else len(txt)
        targ, inp = txt[cidx:eidx], txt[cidx - 1:eidx - 1]
        cur_char_idx[i] = eidx
        done = False
        if cur_char_idx[i] == len(txt):
            cur_char_idx[i] = 1
            done = True
        return inp, targ, done, cur_char_idx

    def iter_single_doc(self, split='train', max_docs=-1):
        cur = np.ones(len(self.data['docs']), dtype=np.int64)
        idxes = np.random.choice(self.splits[split], max_docs) if max_docs > 0 else self.splits[split]
        for cid in idxes:
            dead = False
            while not dead:
                inp, targ, dead, cur = self.get_nextstring_doc(cid, cur)
                yield [{'in': inp, 'targ': targ, 'author': self.data['docs'][cid][self.athstr]}], dead

    def get_num_seqs(self, maxlen=-1, split='train'):
        maxlen = self.max_seq_len if maxlen == -1 else maxlen
        return sum(max(len(self.data['docs'][i]['text']) - 1, 0) // maxlen + 1 for i in self.splits[split])

    def get_num_sents(self, split='train'):
        return sum(len(self.data['docs'][i]['tokens']) for i in self.splits[split])

    def get_num_sent_doc(self, cid, atoms='char'):
        if atoms == 'char':
            return len([st for st in self.data['docs'][cid]['text'].split('.') if len(st) > 0])
        return len(self.data['docs'][cid]['tokens'])

    def get_doc_batch(self, split='train'):
        act_ids = np.where(self.cur_batch >= 0)[0]
        dead_ids = np.where(self.cur_batch < 0)[0]
        self.cur_batch[dead_ids] = np.random.choice(list(set(self.splits[split]) - set(act_ids.tolist())),
                                                    len(dead_ids), replace=False)
        batch = []
        for i, cid in enumerate(self.cur_batch):
            inp, targ, dead, self.cur_char_idx = self.get_nextstring_doc(cid, self.cur_char_idx)
            self.cur_batch[i] = -1 if dead else cid
            batch.append({'in': inp, 'targ': targ, 'author': self.data['docs'][cid][self.athstr], 'id': cid})
        return batch, dead_ids

    def get_rand_doc_batch(self, batch_size, split='train'):
        batch, dead_next = [], []
        for i, cid in enumerate(np.random.choice(self.splits[split], batch_size, replace=False)):
            inp, targ, dead, self.cur_char_idx = self.get_nextstring_doc(cid, self.cur_char_idx)
            batch.append({'in': inp, 'targ': targ, 'author': self.data['docs'][cid][self.athstr], 'id': cid})
            if dead:
                dead_next.append(i)
        return batch, dead_next

    # -- sentences -----------------------------------------------------------------------------------
    def get_rand_sentence_tokenized(self, cid, sidx=None):
        sents = self.data['docs'][cid]['tokens']
        sidx = int(np.random.randint(0, len(sents))) if sidx is None else int(np.asarray(sidx).reshape(-1)[0])
        s = sents[sidx].split()
        return s[:-1], s[1:]

    def get_rand_sentence(self, cid, sidx=None):
        """Character sentence with the start symbol '2' in front of the input and '.' closing the target (:257-272)."""
        sents = [st for st in self.data['docs'][cid]['text'].split('.') if len(st) > 0]
        sidx = int(np.random.randint(0, len(sents))) if sidx is None else int(np.asarray(sidx).reshape(-1)[0])
        s = sents[sidx]
        if s[0] == '2':
            return s, s[1:] + '.'
        return '2' + s, s + '.'

    def _sent(self, atoms):
        return {'char': self.get_rand_sentence, 'word': self.get_rand_sentence_tokenized}[atoms]

    def iter_sentences(self, split='train', batch_size=100, atoms='char'):
        sent, batch = self._sent(atoms), []
        for cid in self.splits[split]:
            n = self.get_num_sent_doc(cid, atoms=atoms)
            for j in range(n):
                inp, targ = sent(cid, sidx=j)
                batch.append({'in': inp, 'targ': targ, 'author': self.data['docs'][cid][self.athstr], 'id': cid, 'sid': j})
                if 'attrib' in self.data['docs'][cid]:
                    batch[-1]['attrib'] = self.data['docs'][cid]['attrib']
                if len(batch) == batch_size:
                    yield batch, j == n - 1
                    batch = []
            if batch:
                yield batch, True
                batch = []

    def iter_sentences_bylen(self, split='train', batch_size=100, atoms='char', auths=None):
        sent, batch = self._sent(atoms), []
        for l in self.lenMap[split]:
            for aid in auths:
                for idx in self.lenMap[split][l]:
                    doc = self.data['docs'][idx[0]]
                    if doc[self.athstr] == aid:
                        inp, targ = sent(idx[0], sidx=idx[1])
                        batch.append({'in': inp, 'targ': targ, 'author': doc[self.athstr], 'id': idx[0], 'sid': idx[1]})
                        if 'attrib' in doc:
                            batch[-1]['attrib'] = doc['attrib']
                    if len(batch) == batch_size:
                        yield batch, False
                        batch = []
                if batch:
                    yield batch, True
                    batch = []

    def getRandLen(self, split='train'):
        rn = np.random.randint(0, self.lenCdist[-1])
        l = 0
        for l in range(len(self.lenCdist)):
            if rn < self.lenCdist[l] and len(self.lenMap[split][l + self.min_len]) > 0:
                break
        return l + self.min_len

    def get_sentence_batch(self, batch_size, split='train', atoms='char', aid=None, sample_by_len=False):
        """batch_size random sentences, optionally of ONE author and of one length bucket (:295-308)."""
        allids = self.lenMap[split][self.getRandLen()] if sample_by_len else self.splits[split]
        if aid:
            allids = [idx for idx in allids if self.data['docs'][idx[0] if sample_by_len else idx][self.athstr] == aid]
        sent, batch = self._sent(atoms), []
        for r in np.random.randint(0, len(allids), batch_size):
            cid, sid = allids[r] if sample_by_len else (allids[r], None)
            inp, targ = sent(cid, sid)
            batch.append({'in': inp, 'targ': targ, 'author': self.data['docs'][cid][self.athstr], 'id': cid})
        return batch

    # -- strings -> padded id matrices (:310-348) ----------------------------------------------------
    def prepare_data(self, batch, char_to_ix, auth_to_ix, leakage=0., maxlen=None):
        b_sz = len(batch)
        inp_seqs, targ_seqs, auths = [], [], []
        lens = np.empty(b_sz, dtype=np.int64)
        auth_ids = list(auth_to_ix.values())
        for k, b in enumerate(batch):
            inp = [char_to_ix[c] for c in b['in'][:maxlen] if c in char_to_ix]
            targ = [char_to_ix[c] for c in b['targ'][:maxlen] if c in char_to_ix]
            # an out-of-vocabulary symbol can shorten one side: pad the target / shift the input (:322-327)
            if len(targ) < len(inp):
                targ.append(0)
            elif len(targ) > len(inp):
                inp.insert(0, 0)
            lens[k] = len(inp)
            inp_seqs.append(inp)
            targ_seqs.append(targ)
            auths.append(auth_to_ix[b['author']] if np.random.rand() >= leakage else np.random.choice(auth_ids))
        max_len = int(lens.max())
        sort_idx = np.argsort(lens)[::-1]                    # longest first, the reference's tie order
        lens_sorted = lens[sort_idx]
        batch[:] = [batch[i] for i in sort_idx]              # callers rely on the in-place reorder
        inps = torch.zeros((max_len, b_sz), dtype=torch.int64, pin_memory=self._pin)
        targs = torch.zeros((max_len, b_sz), dtype=torch.int64, pin_memory=self._pin)
        ia, ta = inps.numpy(), targs.numpy()
        for i, j in enumerate(sort_idx):
            L = lens_sorted[i]
            ia[:L, i] = inp_seqs[j]
            ta[:L, i] = targ_seqs[j]                         # raises like the reference if the two sides still differ
        auths_t = torch.from_numpy(np.asarray(auths, dtype=np.int64)[sort_idx].copy())
        return inps, targs, auths_t, lens_sorted.tolist()

    # -- vocabularies (:350-396) ------------------------------------------------------------------------
    def createAuthorIdx(self):
        author_idx = {}
        for i in self.splits['train']:
            a = self.data['docs'][i][self.athstr]
            if a not in author_idx:
                author_idx[a] = len(author_idx)
        return author_idx, {v: k for k, v in author_idx.items()}

    def _vocab(self, counter, threshold):
        mini, ixto = {}, {}
        for c in counter:                                    # insertion order, as the reference's dict iteration
            if counter[c] >= threshold:
                mini[c] = len(mini) + 1                      # 0 is the padding id
                ixto[mini[c]] = c
        return mini, ixto

    def createCharVocab(self, threshold=5):
        vocab = Counter()
        for i in self.splits['train']:
            vocab.update(self.data['docs'][i]['text'])
        return self._vocab(vocab, threshold)

    def createWordVocab(self, threshold=5):
        vocab = Counter()
        for i in self.splits['train']:
            vocab.update(w for tk in self.data['docs'][i]['tokens'] for w in tk.split())
        mini, ixto = self._vocab(vocab, threshold)
        if self.use_unk:
            mini['UNK'] = len(mini) + 1
            ixto[mini['UNK']] = 'UNK'
            for doc in self.data['docs']:
                doc['tokens'] = [' '.join(w if w in mini else 'UNK' for w in tk.split()) for tk in doc['tokens']]
        return mini, ixto
