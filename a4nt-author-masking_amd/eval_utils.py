"""Py3 restatement of the reference's evaluation procedures (utils/utils.py:14-244) over the model-layer API,
plus the checkpoint dictionary of the training drivers (train_adversarial_translator.py:790-805).

They only call `forward_eval` / `forward_classify` / `forward_mltrain`, `init_hidden` and a DataProvider
(`iter_single_doc`, `iter_sentences`, `prepare_data`), so they run over the HIP-backed modules unchanged.
Printing is replaced by a returned metrics dict next to the reference's return values.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence


def repackage_hidden(h):
    """Detach hidden states from their history (utils.py:14-21)."""
    if h is None:
        return None
    if torch.is_tensor(h):
        return h.detach()
    return tuple(repackage_hidden(v) for v in h)


def eval_model(dp, model, params, char_to_ix, auth_to_ix, split='val', max_docs=-1):
    """Generative multi-head LM as an attribution model: every document is scored window by window under each
    author's output layer; rank of the true author and AUC against a random other author (utils.py:23-85).
    Returns (mean rank, auc, metrics)."""
    from sklearn.metrics import roc_auc_score
    hidden = model.init_hidden(1)
    n_auth = model.num_output_layers
    correct = n_docs = mean_rank = 0.
    cur = np.zeros(n_auth)
    all_scores, all_auths = [], []
    for b_data in dp.iter_single_doc(split=split, max_docs=max_docs):
        done = b_data[1]
        inps, targs, auths, lens = dp.prepare_data(b_data[0], char_to_ix, auth_to_ix)
        output, hidden = model.forward_eval(inps, hidden, compute_softmax=True)
        z = output.detach().cpu().numpy()
        cur = cur + z[np.arange(lens[0]), :, np.squeeze(targs.numpy()[:lens[0]], axis=1)].sum(axis=0)
        if done:
            hidden = tuple(torch.zeros_like(h) for h in hidden)
            a = int(auths[0])
            correct += float(cur.argmax() == a)
            mean_rank += float(np.where(cur.argsort()[::-1] == a)[0][0])
            n_docs += 1.
            all_scores.append(cur[None, :])
            all_auths.append(auths.numpy())
            cur = np.zeros(n_auth)
    all_auths = np.concatenate(all_auths, axis=0)
    all_docs = np.concatenate(all_scores, axis=0)
    all_docs = all_docs - all_docs.mean(axis=1)[:, None]
    n = int(n_docs)
    neg_auths = np.random.randint(0, n_auth, n)
    cols = np.concatenate([all_auths, neg_auths])
    adjusted = ((np.argsort(all_docs[:, cols], axis=0) == np.concatenate([np.arange(n), np.arange(n)])).argmax(axis=0) + 1) / n_docs
    labels = np.concatenate([np.ones(n, dtype=int), np.zeros(n, dtype=int)])
    auc = roc_auc_score(labels, adjusted) if n > 1 else 0.
    metrics = {'n_docs': n_docs, 'accuracy': 100. * correct / n_docs, 'mean_rank': mean_rank / n_docs, 'auc': auc}
    return mean_rank / n_docs, auc, metrics


def eval_classify(dp, model, params, char_to_ix, auth_to_ix, split='val', max_docs=-1, batch_size=1000):
    """Attribution classifier over all sentences of a split: sentence-block and document accuracies
    (utils.py:90-207).  Returns (document accuracy %, block accuracy, metrics)."""
    n_auths = len(auth_to_ix)
    topk = params.get('topk', 5)
    correct = n_docs = correct_blk = n_blks = correct_blk_topk = 0.
    blk_correct_pc, blk_pc = np.zeros(n_auths), np.zeros(n_auths)
    doc_count, doc_correct = np.zeros(n_auths), np.zeros(n_auths)
    cur = np.zeros(n_auths)
    for b_data in dp.iter_sentences(split=split, atoms=params.get('atoms', 'char'), batch_size=batch_size):
        c_sz = len(b_data[0])
        done = b_data[1]
        inps, targs, auths, lens = dp.prepare_data(b_data[0], char_to_ix, auth_to_ix)
        output = model.forward_classify(inps, model.init_hidden(c_sz), compute_softmax=False, predict_mode=True,
                                        lens=lens)[0]
        scores = F.log_softmax(output, dim=-1).detach().cpu().numpy()
        a = auths.numpy()
        hit = scores.argmax(axis=1) == a
        correct_blk += hit.sum()
        correct_blk_topk += (np.where(scores.argsort(axis=1)[:, ::-1] == a[:, None])[1] <= topk).sum()
        n_blks += c_sz
        np.add.at(blk_correct_pc, a, hit)
        np.add.at(blk_pc, a, 1.)
        cur = cur + scores.sum(axis=0)
        if done:
            a0 = int(a[0])
            correct += float(cur.argmax() == a0)
            doc_correct[a0] += float(cur.argmax() == a0)
            doc_count[a0] += 1.
            n_docs += 1.
            cur = np.zeros(n_auths)
    metrics = {'n_docs': n_docs, 'doc_accuracy': 100. * correct / max(n_docs, 1.), 'block_accuracy': correct_blk / n_blks,
               'block_topk_accuracy': correct_blk_topk / n_blks,
               'block_accuracy_per_class': (blk_correct_pc / np.maximum(blk_pc, 1.)).tolist(),
               'doc_accuracy_per_class': (doc_correct / np.maximum(doc_count, 1.)).tolist()}
    return metrics['doc_accuracy'], correct_blk / n_blks, metrics


def eval_translator(dp, model, params, char_to_ix, auth_to_ix, split='val', max_docs=-1, batch_size=100):
    """Teacher-forced reconstruction loss / perplexity of the translator on a split (utils.py:209-244).
    Like the reference it divides by the index of the last batch, not by the number of batches (:235)."""
    model.eval()
    total_loss, i = 0., 0
    with torch.no_grad():
        for i, b_data in enumerate(dp.iter_sentences(split=split, atoms=params.get('atoms', 'char'), batch_size=batch_size)):
            inps, targs, auths, lens = dp.prepare_data(b_data[0], char_to_ix, auth_to_ix, maxlen=params['max_seq_len'])
            output, _ = model.forward_mltrain(inps, lens, inps, lens, None, compute_softmax=False, auths=auths)
            tp = pack_padded_sequence(targs.to(output.device), lens)
            total_loss += float(F.cross_entropy(pack_padded_sequence(output, lens)[0], tp[0]))
    model.train()
    cur_loss = total_loss / max(i, 1)
    return cur_loss, float(np.exp(cur_loss))


def eval_forward_pass(modelGen, modelEval, inps, lens, end_c=0, maxlen=100, auths=None, cycle_compute=True,
                      append_symb=None):
    """The translation pass of the evaluation driver (eval_gantranslate.py:19-53): greedy translation to `auths`,
    classifier probabilities of the translated ids, and the greedy reverse translation of those ids.
    append_symb: [1,1] LongTensor holding the start id.  Returns (probs [B,C] in input order, char_outs, gen_lens
    [, rev_char_outs, rev_gen_lens])."""
    modelGen.eval()
    modelEval.eval()
    b_sz = len(lens)
    with torch.no_grad():
        _, gen_lens, char_outs = modelGen.forward_advers_gen(inps, lens, end_c=end_c, n_max=maxlen, auths=auths,
                                                             soft_samples=False)
        len_sorted, sort_idx = gen_lens.cpu().sort(dim=0, descending=True, stable=True)
        _, rev_sort_idx = sort_idx.sort(dim=0)
        eval_inp = torch.stack(list(char_outs), 0).cpu().index_select(1, sort_idx)
        probs = modelEval.forward_classify(eval_inp, lens=len_sorted.tolist(), compute_softmax=True)[0]
        probs = probs.detach().index_select(0, rev_sort_idx.to(probs.device))
        if not cycle_compute:
            return probs, char_outs, gen_lens
        reverse_inp = torch.cat([append_symb.cpu().repeat(1, b_sz), eval_inp], dim=0)
        _, rev_lens, rev_chars = modelGen.forward_advers_gen(reverse_inp, len_sorted.tolist(), end_c=end_c,
                                                             n_max=maxlen, auths=1 - auths, soft_samples=False)
        rev_dev = rev_sort_idx.to(rev_lens.device)
        rev_chars = [rc.index_select(0, rev_dev) for rc in rev_chars]
    return probs, char_outs, gen_lens, rev_chars, rev_lens


def make_checkpoint(it, params, eval_params, misc, modelGen, modelEval, optimGen, optimEval, val_loss=0., val_pplx=0.,
                    loss=0.):
    """The dictionary the GAN driver saves (train_adversarial_translator.py:791-802); torch.save it."""
    return {'iter': it, 'arch': params, 'eval_arch': eval_params, 'val_loss': val_loss, 'val_pplx': val_pplx,
            'misc': misc, 'state_dict': modelGen.state_dict(), 'state_dict_eval': modelEval.state_dict(),
            'loss': loss, 'optimizerGen': optimGen.state_dict(), 'optimizerEval': optimEval.state_dict()}


def load_checkpoint(path, map_location='cpu'):
    """Load a checkpoint dictionary with the tensors-and-plain-containers loader only (no arbitrary unpickling):
    the reference's .pth.tar files hold state_dicts, the params dicts and vocab maps."""
    return torch.load(path, map_location=map_location, weights_only=True)


def restore_models(cp, CharTranslator, classifier_cls=None):
    """Rebuild (modelGen, modelEval or None) from a checkpoint the way the drivers do (:331-345): only the keys
    present in both state_dicts are taken."""
    G = CharTranslator(cp['arch'])
    st = G.state_dict()
    st.update({k: v for k, v in cp['state_dict'].items() if k in st})
    G.load_state_dict(st)
    E = None
    if classifier_cls is not None and 'state_dict_eval' in cp:
        E = classifier_cls(cp['eval_arch'])
        st = E.state_dict()
        st.update({k: v for k, v in cp['state_dict_eval'].items() if k in st})
        E.load_state_dict(st)
    return G, E
