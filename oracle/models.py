"""Oracle (test infrastructure): pure-PyTorch CPU fp32 restatement of the reference model layer.

Follows, function by function (citations are into /root/reference):
  * sample_gumbel / gumbel_softmax_sample      models/char_translator.py:10-28
  * CharTranslator                             models/char_translator.py:30-401
  * CharLstm (classifier mode)                 models/char_lstm.py:32-164, 263-340
  * CharCNN                                    models/char_cnn.py:8-102
  * get_classifier / packed_mean / packed_add  models/model_utils.py:5-28

Differences from the reference are limited to what a Python-3 / device-agnostic restatement
needs: no ``.cuda()`` in constructors, tensors follow the parameters' device, and every random
draw of the Gumbel sampler goes through an injectable ``NoiseSource`` so fixtures can pin it.
Every constructor option and method of the four model files is restated.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence, PackedSequence


# ----------------------------------------------------------------------------------------------
# noise
# ----------------------------------------------------------------------------------------------
class NoiseSource(object):
    """Uniform [0,1) draws for the Gumbel sampler.

    ``queue``: list of pre-generated tensors, popped front-first (fixtures / parity tests).
    Otherwise draws come from ``generator`` (a torch.Generator) or the global RNG.
    """

    def __init__(self, queue=None, generator=None):
        self.queue = list(queue) if queue is not None else None
        self.generator = generator
        self.drawn = []          # every tensor handed out, in order (lets tests replay it)
        self.record = False

    def uniform(self, shape, device):
        if self.queue is not None:
            u = self.queue.pop(0)
            assert tuple(u.shape) == tuple(shape), (tuple(u.shape), tuple(shape))
            u = u.to(device)
        else:
            u = torch.rand(*shape, generator=self.generator).to(device)
        if self.record:
            self.drawn.append(u.clone())
        return u


def sample_gumbel(shape, noise, device):
    """g = -log(-log(u + eps) + eps), eps = 1e-20 (char_translator.py:10-15)."""
    eps = 1e-20
    u = noise.uniform(shape, device)
    return -torch.log(-torch.log(u + eps) + eps)


# Tests that assert exact tokens set this to a list: every arg-max decision (sampled or greedy) then appends its
# per-row top-1 / top-2 gap, in the units the arg max is taken in, so that "exact" is claimed where the gap is far
# above the fp32 tolerance.
MARGIN_LOG = None


def _log_margin(v):
    if MARGIN_LOG is not None and v.shape[1] > 1:
        t2 = v.detach().topk(2, dim=1).values
        MARGIN_LOG.append(t2[:, 0] - t2[:, 1])


def gumbel_softmax_sample(x, tau, hard, noise):
    """char_translator.py:17-28.  x: [B,V] (already multiplied by softmax_scale); tau: float, or a
    [B,1] tensor when the temperature is learnt (char_translator.py:379)."""
    g = sample_gumbel(x.shape, noise, x.device)
    y = (F.log_softmax(x, dim=1) + g) / tau
    _log_margin(y)
    ysft = F.softmax(y, dim=1)
    if hard:
        idx = ysft.detach().argmax(dim=1, keepdim=True)
        one_hot = torch.zeros_like(ysft).scatter_(1, idx, 1.0) - ysft.detach()
        return one_hot + ysft
    return ysft


# ----------------------------------------------------------------------------------------------
# packed helpers (model_utils.py:14-28)
# ----------------------------------------------------------------------------------------------
def packed_mean(packed):
    data, lens = pad_packed_sequence(packed)
    return data.sum(dim=0) / lens.to(data.dtype).to(data.device).view(-1, 1)


def packed_add(p1, p2):
    d1, lens = pad_packed_sequence(p1)
    d2, _ = pad_packed_sequence(p2)
    return pack_padded_sequence(d1 + d2, lens)


def _fill_forget_bias(lstm, layer_ids, hsz, value):
    # reference quirk: index range is arange(H+1, 2H), index H itself is left alone
    # (char_translator.py:141-161, char_lstm.py:121-134)
    with torch.no_grad():
        for lid in layer_ids:
            for nm in ('bias_ih_l%d' % lid, 'bias_hh_l%d' % lid):
                getattr(lstm, nm)[hsz + 1:2 * hsz] = value


def _build_rec(in_size, hsz, depth, residual):
    if residual:
        sizes = [in_size] + [hsz] * (depth - 1)
        return nn.ModuleList([nn.LSTM(sizes[i], hsz, 1) for i in range(depth)])
    return nn.LSTM(in_size, hsz, depth)


def _run_rec(rec, inp, h_prev, residual, depth):
    """Stacked LSTM with optional residual adds (char_translator.py:175-198, char_lstm.py:142-164)."""
    if not residual:
        return rec(inp) if h_prev is None else rec(inp, h_prev)
    cur = inp
    hs, cs = [], []
    for i in range(depth):
        if h_prev is None:
            out, (h, c) = rec[i](cur)
        else:
            out, (h, c) = rec[i](cur, (h_prev[0][i:i + 1], h_prev[1][i:i + 1]))
        if i > 0:
            cur = packed_add(cur, out) if isinstance(cur, PackedSequence) else cur + out
        else:
            cur = out
        hs.append(h)
        cs.append(c)
    return cur, (torch.cat(hs, 0), torch.cat(cs, 0))


def _unsupported(params, keys):
    for k in keys:
        if params.get(k, 0):
            raise NotImplementedError('oracle: option %r is outside the hot-path scope (SURVEY.md section 8)' % k)


# ----------------------------------------------------------------------------------------------
# translator
# ----------------------------------------------------------------------------------------------
class CharTranslator(nn.Module):
    def __init__(self, params, encoder_only=False, noise=None):
        super().__init__()
        self.encoder_only = encoder_only
        self.no_encoder = params.get('no_encoder', 0)
        self.enc_noise = params.get('apply_noise', 0)
        self.learn_gumbel = params.get('learn_gumbel', False)
        self.split_gen = params.get('split_generators', 0)
        self.encoder_mean_vec = params.get('encoder_mean_vec', 0)
        self.vocab_size = params.get('vocabulary_size', -1) + 1
        self.enc_num_rec_layers = params.get('enc_hidden_depth', -1)
        self.emb_size = params.get('embedding_size', -1)
        self.enc_hidden_size = params.get('enc_hidden_size', -1)
        self.dec_num_rec_layers = params.get('dec_hidden_depth', -1)
        self.dec_hidden_size = params.get('dec_hidden_size', -1)
        self.en_residual = params.get('en_residual_conn', 1)
        self.pad_auth_vec = params.get('pad_auth_vec', 0)
        self.gumb_type = params.get('gumbel_hard', False)
        self.softmax_scale = params.get('softmax_scale', 2.0)
        self.max_pool_rnn = params.get('maxpoolrnn', 0)
        self.noise = noise if noise is not None else NoiseSource()
        self.temp = None

        if self.learn_gumbel:
            self.gumbel_W = nn.Parameter(torch.zeros(self.dec_hidden_size, 1))
            self.gumbel_b = nn.Parameter(torch.zeros(1))
        if self.pad_auth_vec:
            self.auth_emb = nn.Embedding(params.get('num_output_layers', 2), self.pad_auth_vec)
        self.char_emb = nn.Embedding(self.vocab_size, self.emb_size, padding_idx=0)
        self.emb_drop = nn.Dropout(p=params.get('drop_prob_emb', 0.25))
        if not self.no_encoder:
            self.enc_rec_layers = _build_rec(self.emb_size, self.enc_hidden_size, self.enc_num_rec_layers,
                                             self.en_residual)
            self.enc_drop = nn.Dropout(p=params.get('drop_prob_encoder', 0.25))
        if not encoder_only:
            enc_out = self.enc_hidden_size * (1 + self.max_pool_rnn) if not self.no_encoder else 0
            in_sz = enc_out + self.emb_size + self.pad_auth_vec
            self.dec_rec_layers = _build_rec(in_sz, self.dec_hidden_size, self.dec_num_rec_layers,
                                             self.en_residual)
            self.decoder_W = nn.Parameter(torch.zeros(self.dec_hidden_size, self.vocab_size))
            self.decoder_b = nn.Parameter(torch.zeros(self.vocab_size))
            self.dec_drop = nn.Dropout(p=params.get('drop_prob_decoder', 0.25))
            if self.split_gen:
                # second decoder + output layer, used for batches of author 1 (char_translator.py:99-108)
                self.dec_rec_layers_2 = _build_rec(in_sz, self.dec_hidden_size, self.dec_num_rec_layers,
                                                   self.en_residual)
                self.decoder_W_2 = nn.Parameter(torch.zeros(self.dec_hidden_size, self.vocab_size))
                self.decoder_b_2 = nn.Parameter(torch.zeros(self.vocab_size))
        self.init_weights()
        self.zero_hidden_bsz = params['batch_size']

    def init_weights(self):
        if not self.encoder_only:
            with torch.no_grad():
                self.decoder_W.uniform_(-0.001, 0.001)
                self.decoder_b.zero_()
                if self.split_gen:
                    self.decoder_W_2.uniform_(-0.001, 0.001)
                    self.decoder_b_2.zero_()
        if self.learn_gumbel:
            with torch.no_grad():
                self.gumbel_W.uniform_(-0.01, 0.01)
        for rec, depth, hsz in ((getattr(self, 'enc_rec_layers', None), self.enc_num_rec_layers, self.enc_hidden_size),
                                (getattr(self, 'dec_rec_layers', None), self.dec_num_rec_layers,
                                 self.dec_hidden_size),
                                (getattr(self, 'dec_rec_layers_2', None), self.dec_num_rec_layers,
                                 self.dec_hidden_size)):
            if rec is None:
                continue
            if self.en_residual:
                for i in range(depth):
                    _fill_forget_bias(rec[i], [0], hsz, 1.0)
            else:
                _fill_forget_bias(rec, range(depth), hsz, 1.0)

    # -- helpers ---------------------------------------------------------------------------
    def set_noise(self, uniforms):
        """Pin the Gumbel uniforms: one [B,V] tensor per sampling step, consumed in order."""
        self.noise = NoiseSource(queue=uniforms)

    def _dev(self):
        return self.char_emb.weight.device

    def init_hidden(self, bsz):
        w = self.char_emb.weight
        return (w.new_zeros(self.enc_num_rec_layers, bsz, self.enc_hidden_size),
                w.new_zeros(self.enc_num_rec_layers, bsz, self.enc_hidden_size))

    def init_hidden_dec(self, bsz):
        w = self.char_emb.weight
        return (w.new_zeros(self.dec_num_rec_layers, bsz, self.dec_hidden_size),
                w.new_zeros(self.dec_num_rec_layers, bsz, self.dec_hidden_size))

    def _soft_embed(self, x):
        n_steps, b_sz = x.shape[0], x.shape[1]
        return x.reshape(n_steps * b_sz, -1).mm(self.char_emb.weight).view(n_steps, b_sz, -1)

    def _encode(self, emb, lengths, h_prev):
        packed = pack_padded_sequence(emb, lengths)
        return _run_rec(self.enc_rec_layers, packed, h_prev, self.en_residual, self.enc_num_rec_layers)

    def _decoder(self, auths):
        """(rec layers, W, b) of the decoder in use: the second one when split_generators is set and the
        batch's first author id is not 0 (char_translator.py:265,273-275,359-366)."""
        if not self.split_gen or int(auths[0]) == 0:
            return self.dec_rec_layers, self.decoder_W, self.decoder_b
        return self.dec_rec_layers_2, self.decoder_W_2, self.decoder_b_2

    # -- teacher forced (char_translator.py:200-286) -------------------------------------------
    def forward_mltrain(self, inp, lengths_inp, targ, lengths_targ, h_prev=None, compute_softmax=False,
                        auths=None, adv_inp=False, sort_enc=None, adv_targ=False):
        dev = self._dev()
        b_sz = inp.shape[1]
        enc_hidden = None
        if not self.no_encoder:
            if not adv_inp:
                emb = self.emb_drop(self.char_emb(inp.to(dev)))
            else:
                emb = self._soft_embed(inp)
            enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
            if self.max_pool_rnn:
                enc_unp, _ = pad_packed_sequence(enc_out)
                # reference: plain mean over the padded time axis (char_translator.py:227)
                ctxt = torch.cat([enc_unp.mean(dim=0), enc_hidden[0][-1]], dim=-1)
            else:
                ctxt = enc_hidden[0][-1]
            if sort_enc is not None:
                ctxt = ctxt.index_select(0, sort_enc.to(dev))
            if self.pad_auth_vec:
                ctxt = torch.cat([ctxt, self.auth_emb(auths.to(dev))], dim=-1)
            if self.enc_noise:
                # char_translator.py:240-241: N(0,1)/20 noise instead of dropout (global RNG: not pinned)
                ctxt = ctxt + torch.randn(ctxt.shape, device=dev) / 20.
            else:
                ctxt = self.enc_drop(ctxt)

        if not adv_targ:
            targ = targ.to(dev)
            targ_emb = self.emb_drop(self.char_emb(targ))
        else:
            # soft target rows (the language-model loss feeds generated samples, char_translator.py:252-254)
            targ_emb = self.emb_drop(self._soft_embed(targ))
        n_t = targ.shape[0]
        if not self.no_encoder:
            dec_inp = torch.cat([ctxt.unsqueeze(0).expand(n_t, b_sz, ctxt.shape[1]), targ_emb], dim=-1)
        else:
            dec_inp = targ_emb
        packed = pack_padded_sequence(dec_inp, lengths_targ)
        dec_rec, dec_W, dec_b = self._decoder(auths)
        dec_out, dec_hidden = _run_rec(dec_rec, packed, None, self.en_residual, self.dec_num_rec_layers)
        dec_unp, _ = pad_packed_sequence(dec_out)
        dec_unp = self.dec_drop(dec_unp)
        n_t = dec_unp.shape[0]
        score = dec_unp.reshape(n_t * b_sz, -1).mm(dec_W) + dec_b.view(1, -1)
        if compute_softmax:
            score = F.log_softmax(score, dim=1)
        return score.view(n_t, b_sz, self.vocab_size), (enc_hidden, dec_hidden)

    # -- encoder only (char_translator.py:288-312) ----------------------------------------------
    def forward_encode(self, x, lengths_inp, h_prev=None, adv_inp=False):
        emb = self._soft_embed(x) if adv_inp else self.char_emb(x.to(self._dev()))
        enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
        if self.encoder_mean_vec:
            return torch.cat([packed_mean(enc_out), enc_hidden[0][-1]], dim=-1)
        return enc_hidden[0][-1]

    # -- free running generation (char_translator.py:314-401) -----------------------------------
    def forward_advers_gen(self, x, lengths_inp, h_prev=None, n_max=100, end_c=-1, soft_samples=False,
                           temp=0.1, auths=None, adv_inp=False, n_samples=1):
        dev = self._dev()
        b_sz = x.shape[1]
        if not adv_inp:
            x = x.to(dev)
            emb = self.char_emb(x)
        else:
            emb = self._soft_embed(x)
        enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
        if self.max_pool_rnn:
            ctxt = torch.cat([packed_mean(enc_out), enc_hidden[0][-1]], dim=-1)
        else:
            ctxt = enc_hidden[0][-1]
        if self.pad_auth_vec:
            ctxt = torch.cat([ctxt, self.auth_emb(auths.to(dev))], dim=-1)
        targ_emb = self.char_emb(x[0]) if not adv_inp else x[0].mm(self.char_emb.weight)
        dec_bsz = b_sz * n_samples
        ctxt_rep = ctxt.repeat(n_samples, 1)
        dec_inp = torch.cat([ctxt, targ_emb], dim=-1).repeat(n_samples, 1)
        dec_rec, W, b = self._decoder(auths)
        b = b.view(1, -1)
        out, dec_hidden = _run_rec(dec_rec, dec_inp.view(1, dec_bsz, -1), None, self.en_residual,
                                   self.dec_num_rec_layers)
        p_rnn = out.view(dec_bsz, -1)
        char_out, samp_out = [], []
        gen_lens = torch.zeros(dec_bsz, dtype=torch.int32, device=dev)
        prev_done = torch.zeros(dec_bsz, dtype=torch.bool, device=dev)
        for _ in range(n_max):
            dec_out = p_rnn.mm(W) + b
            if soft_samples:
                # learnt temperature: one tau >= 1 per row from the decoder state (char_translator.py:379)
                self.temp = (F.softplus(p_rnn.mm(self.gumbel_W) + self.gumbel_b) + 1.) if self.learn_gumbel else temp
                samp = gumbel_softmax_sample(dec_out * self.softmax_scale, self.temp, self.gumb_type, self.noise)
                emb = samp.mm(self.char_emb.weight)
                pred_c = samp.detach().argmax(dim=-1)
                char_out.append(pred_c)
                # reference zeroes finished rows through .data AFTER emb was recorded
                # (char_translator.py:381-385): the forward emb sees the un-zeroed sample, the
                # weight gradient of that mm sees the zeroed one.
                samp.data.masked_fill_(prev_done.view(-1, 1), 0.)
                samp_out.append(samp)
            else:
                pred_c = dec_out.detach().argmax(dim=-1)
                _log_margin(dec_out)
                char_out.append(pred_c)
                emb = self.char_emb(pred_c)
            gen_lens += (~prev_done).int()
            prev_done = prev_done | (pred_c == end_c)
            if bool(prev_done.all()):
                break
            dec_inp = torch.cat([ctxt_rep, emb], dim=-1).view(1, dec_bsz, -1)
            out, dec_hidden = _run_rec(dec_rec, dec_inp, dec_hidden, self.en_residual, self.dec_num_rec_layers)
            p_rnn = out.view(dec_bsz, -1)
        return samp_out, gen_lens, char_out


# ----------------------------------------------------------------------------------------------
# LSTM attribution classifier (discriminator)
# ----------------------------------------------------------------------------------------------
class CharLstm(nn.Module):
    """models/char_lstm.py:32-340.  Classifier mode (default linear head; optional MLP / CNN-over-hidden heads,
    generic real/fake head, bidirectional LSTM, compression layer) and the generative multi-head LM mode."""

    def __init__(self, params):
        super().__init__()
        self.output_size = params.get('vocabulary_size', -1) + 1
        self.num_output_layers = params.get('num_output_layers', 1)
        self.num_rec_layers = params.get('hidden_depth', -1)
        self.emb_size = params.get('embedding_size', -1)
        self.hidden_size = params.get('hidden_size', -1)
        self.en_residual = params.get('en_residual_conn', 0)
        self.bidir = bool(params.get('bidir', 0))
        self.compression_layer = params.get('compression_layer', 0)
        self.max_pool_rnn = params.get('maxpoolrnn', 0)
        self.classifier = params.get('mode', 'generative') == 'classifier'
        H, D = self.hidden_size, 1 + int(self.bidir)
        if self.compression_layer:
            # vocabulary -> a few learnt "classes" (rows L1-normalised) -> embedding (:47-49, :114-118)
            self.compression_W = nn.Embedding(self.output_size, self.compression_layer)
            self.encoder = nn.Embedding(self.compression_layer, self.emb_size)
        else:
            self.encoder = nn.Embedding(self.output_size, self.emb_size, padding_idx=0)
        self.enc_drop = nn.Dropout(p=params.get('drop_prob_encoder', 0.25))
        if self.en_residual:
            sizes = [self.emb_size] + [H] * (self.num_rec_layers - 1)
            self.rec_layers = nn.ModuleList([nn.LSTM(sizes[i], H, 1, bidirectional=self.bidir)
                                             for i in range(self.num_rec_layers)])
        else:
            self.rec_layers = nn.LSTM(self.emb_size, H, self.num_rec_layers, bidirectional=self.bidir)
        self.head_mlp = self.head_cnn = False
        if self.classifier:
            # reference quirk (:65 vs :309): with maxpoolrnn == 3 decoder_W is sized 2H although the feature is
            # H + L*H wide, so that head only type-checks when L == 1
            decoder_size = H * (1 + int(self.max_pool_rnn in (1, 3))) * D
            if params.get('decoder_mlp', 0):
                self.decoder_W_mlp = nn.Parameter(torch.zeros(decoder_size, params['decoder_mlp']))
                self.decoder_b_mlp = nn.Parameter(torch.zeros(params['decoder_mlp']))
                decoder_size = params['decoder_mlp']
                self.head_mlp = True
            elif params.get('decoder_cnn', 0):
                self.cnn_ks = params['decoder_cnn_ks']
                self.cnn_nfilt = params['decoder_cnn_nfilt']
                self.decoder_cnn = nn.ModuleList([nn.Conv1d(H, self.cnn_nfilt, K, padding=K) for K in self.cnn_ks])
                self.head_cnn = True
                decoder_size = self.cnn_nfilt * len(self.cnn_ks)
            self.decoder_W = nn.Parameter(torch.zeros(decoder_size, self.num_output_layers))
            self.decoder_b = nn.Parameter(torch.zeros(self.num_output_layers))
            if params.get('generic_classifier', False):
                self.generic_W = nn.Parameter(torch.zeros(H * (1 + int(self.max_pool_rnn == 1)), 1))
                self.generic_b = nn.Parameter(torch.zeros(1))
        else:
            # one [H, V] output layer per author (:86-88)
            self.decoder_W = nn.Parameter(torch.zeros(self.num_output_layers, H, self.output_size))
            self.decoder_b = nn.Parameter(torch.zeros(self.num_output_layers, self.output_size))
        self.dec_drop = nn.Dropout(p=params.get('drop_prob_decoder', 0.25))
        self.init_weights()

    def init_weights(self):
        a = math.sqrt(float(self.decoder_W.shape[0]))
        with torch.no_grad():
            if hasattr(self, 'generic_W'):
                self.generic_W.uniform_(-a, a)
                self.generic_b.zero_()
            if self.head_mlp:
                n_in = math.sqrt(float(self.decoder_W_mlp.shape[0]))
                self.decoder_W_mlp.uniform_(-1.73 / n_in, 1.73 / n_in)
                self.decoder_b_mlp.zero_()
            self.decoder_W.uniform_(-1.73 / a, 1.73 / a)
            self.decoder_b.zero_()
            if self.compression_layer:
                w = self.compression_W.weight
                w.div_(w.abs().sum(dim=1, keepdim=True))
        sfx = ['', '_reverse'] if self.bidir else ['']
        with torch.no_grad():
            for i in range(self.num_rec_layers):
                lstm, lid = (self.rec_layers[i], 0) if self.en_residual else (self.rec_layers, i)
                for sx in sfx:
                    for nm in ('bias_ih_l%d%s' % (lid, sx), 'bias_hh_l%d%s' % (lid, sx)):
                        getattr(lstm, nm)[self.hidden_size + 1:2 * self.hidden_size] = 2.0

    def init_hidden(self, bsz):
        w = self.decoder_W
        n = self.num_rec_layers * (1 + int(self.bidir))
        return (w.new_zeros(n, bsz, self.hidden_size), w.new_zeros(n, bsz, self.hidden_size))

    def _rec(self, inp, h_prev):
        return _run_rec(self.rec_layers, inp, h_prev, self.en_residual, self.num_rec_layers)

    # -- generative multi-head language model (:166-261) ---------------------------------------
    def forward(self, x, lengths, h_prev, target_head, compute_softmax=False):
        dev = self.decoder_W.device
        n_steps, b_sz = x.shape[0], x.shape[1]
        emb = self.enc_drop(self.encoder(x.to(dev)))
        rnn_out, hidden = self._rec(pack_padded_sequence(emb, lengths), h_prev)
        rnn_out = self.dec_drop(pad_packed_sequence(rnn_out)[0])
        n_t = rnn_out.shape[0]
        head = target_head.to(dev)
        W = self.decoder_W[head]                                    # [B, H, V]: each row's own output layer
        b = self.decoder_b[head].view(b_sz, 1, self.output_size).expand(b_sz, n_t, self.output_size)
        dec_out = torch.baddbmm(b, rnn_out.transpose(0, 1), W).transpose(0, 1)
        if compute_softmax:
            dec_out = F.softmax(dec_out.reshape(-1, self.output_size), dim=1).view(n_t, b_sz, self.output_size)
        return dec_out, hidden

    def forward_eval(self, x, h_prev, compute_softmax=True):
        """One sequence ([T,1]) scored by every author's output layer -> [T, n_auth, V]."""
        dev = self.decoder_W.device
        n_auth, n_steps = self.num_output_layers, x.shape[0]
        with torch.no_grad():
            emb = self.encoder(x.to(dev))
            rnn_out, hidden = self._rec(emb, h_prev)
            rnn_out = rnn_out.expand(n_steps, n_auth, self.hidden_size)
            b = self.decoder_b.view(n_auth, 1, self.output_size).expand(n_auth, n_steps, self.output_size)
            dec_out = torch.baddbmm(b, rnn_out.transpose(0, 1), self.decoder_W).transpose(0, 1)
            if compute_softmax:
                dec_out = F.softmax(dec_out.reshape(-1, self.output_size), dim=1).view(n_steps, n_auth,
                                                                                       self.output_size)
        return dec_out, hidden

    def forward_gen(self, x, h_prev, target_auth, n_max=100, end_c=-1):
        """Greedy continuation of the seed x ([T,1]) with target_auth's output layer; never stops early (:252)."""
        dev = self.decoder_W.device
        with torch.no_grad():
            rnn_out, hidden = self._rec(self.encoder(x.to(dev)), h_prev)
            W = self.decoder_W[target_auth.to(dev)][0]
            b = self.decoder_b[target_auth.to(dev)].view(1, self.output_size)
            p_rnn = rnn_out[-1]
            char_out = []
            for _ in range(n_max):
                pred_c = (p_rnn.mm(W) + b).argmax(dim=-1)
                char_out.append(pred_c)
                p_rnn, hidden = self._rec(self.encoder(pred_c).unsqueeze(0), hidden)
                p_rnn = p_rnn[-1]
        return char_out

    # -- classifier (:263-340) -----------------------------------------------------------------
    def _embed(self, x, adv_inp, dev):
        n_steps, b_sz = x.shape[0], x.shape[1]
        if self.compression_layer:
            if not adv_inp:
                comp = self.compression_W(x.to(dev)).view(n_steps * b_sz, -1)
                qn = comp.abs().sum(dim=1, keepdim=True) + 1e-8
            else:
                comp = x.reshape(n_steps * b_sz, -1).mm(self.compression_W.weight)
                qn = comp.abs().sum(dim=1, keepdim=True).detach() + 1e-8
            return (comp / qn).mm(self.encoder.weight).view(n_steps, b_sz, -1)
        if not adv_inp:
            return self.encoder(x.to(dev))
        return x.reshape(n_steps * b_sz, -1).mm(self.encoder.weight).view(n_steps, b_sz, -1)

    def forward_classify(self, x, h_prev=None, compute_softmax=False, predict_mode=False, adv_inp=False,
                         lens=None, drop=True):
        dev = self.decoder_W.device
        b_sz = x.shape[1]
        with torch.set_grad_enabled(torch.is_grad_enabled() and not predict_mode):
            emb = self._embed(x, adv_inp, dev)
            if drop:
                emb = self.enc_drop(emb)
            rnn_out, hidden = self._rec(pack_padded_sequence(emb, lens), h_prev)
            if self.head_cnn:
                # Conv1d(padding=K) over the (zero padded) hidden states, leaky_relu, max over time (:321-327);
                # this dropout is applied whatever `drop` says
                unp = pad_packed_sequence(rnn_out)[0].permute(1, 2, 0)
                pooled = [F.leaky_relu(conv(unp)).max(dim=2)[0] for conv in self.decoder_cnn]
                dec_in = self.dec_drop(torch.cat(pooled, dim=1))
                enc_out = dec_in
            else:
                if self.max_pool_rnn == 1:
                    ctxt = torch.cat([packed_mean(rnn_out), hidden[0][-1]], dim=-1)
                elif self.max_pool_rnn == 2:
                    ctxt = pad_packed_sequence(rnn_out)[0].max(dim=0)[0]
                elif self.max_pool_rnn == 3:
                    ctxt = torch.cat([pad_packed_sequence(rnn_out)[0].max(dim=0)[0],
                                      hidden[0].transpose(0, 1).contiguous().view(b_sz, -1)], dim=-1)
                else:
                    ctxt = hidden[0][-1]
                enc_out = self.dec_drop(ctxt) if drop else ctxt
                dec_in = enc_out
                if self.head_mlp:
                    dec_in = torch.tanh(enc_out.mm(self.decoder_W_mlp) + self.decoder_b_mlp.view(1, -1))
            dec_out = dec_in.mm(self.decoder_W) + self.decoder_b.view(1, -1)
            if compute_softmax:
                dec_out = F.softmax(dec_out, dim=-1)
            if hasattr(self, 'generic_W'):
                # real/fake head; note the reference returns the LSTM state, not the features, here (:335-338)
                generic = torch.sigmoid(enc_out.mm(self.generic_W) + self.generic_b.view(1, 1))
                return dec_out, hidden, generic
            return dec_out, enc_out


# ----------------------------------------------------------------------------------------------
# CNN attribution classifier
# ----------------------------------------------------------------------------------------------
class CharCNN(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.n_words = params.get('vocabulary_size', -1) + 1
        self.num_outputs = params.get('num_output_layers', 1)
        self.num_output_layers = self.num_outputs
        self.emb_size = params.get('embedding_size', -1)
        self.encoder = nn.Embedding(self.n_words, self.emb_size, padding_idx=0)
        self.enc_drop = nn.Dropout(p=params.get('drop_prob_encoder', 0.25))
        self.n_layers = params.get('decoder_cnn_nlayers', 1)
        self.decoder_cnn_residual = params.get('decoder_cnn_residual', 0)
        self.cnn_ks = params['decoder_cnn_ks']
        self.cnn_nfilt = params['decoder_cnn_nfilt']
        in_sizes = [self.emb_size] + [self.cnn_nfilt * len(self.cnn_ks)] * (self.n_layers - 1)
        self.decoder_cnn = nn.ModuleList([
            nn.ModuleList([nn.Conv1d(in_sizes[i], self.cnn_nfilt, K, padding=K // 2) for K in self.cnn_ks])
            for i in range(self.n_layers)])
        dsz = self.cnn_nfilt * len(self.cnn_ks)
        self.decoder_W = nn.Parameter(torch.zeros(dsz, self.num_outputs))
        self.decoder_b = nn.Parameter(torch.zeros(self.num_outputs))
        self.dec_drop = nn.Dropout(p=params.get('drop_prob_decoder', 0.25))
        self.init_weights()

    def init_weights(self):
        a = math.sqrt(float(self.decoder_W.shape[0]))
        with torch.no_grad():
            self.decoder_W.uniform_(-1.73 / a, 1.73 / a)
            self.decoder_b.zero_()
            for layer in self.decoder_cnn:
                for conv in layer:
                    conv.weight.uniform_(-0.01, 0.01)
                    conv.bias.zero_()

    def init_hidden(self, bsz):
        return None

    def forward_classify(self, x, h_prev=None, compute_softmax=False, predict_mode=False, adv_inp=False,
                         lens=None, drop=True):
        # ``drop`` is accepted (and ignored, as the reference always applies its dropouts,
        # char_cnn.py:71-73,87) so the adversarial glue can drive either classifier.
        dev = self.decoder_W.device
        n_steps, b_sz = x.shape[0], x.shape[1]
        if not adv_inp:
            emb = self.enc_drop(self.encoder(x.to(dev)))
        else:
            emb = self.enc_drop(x.reshape(n_steps * b_sz, -1).mm(self.encoder.weight).view(n_steps, b_sz, -1))
        cur = emb.permute(1, 2, 0)                     # [B, E, T]
        inputs = [cur]
        for i in range(self.n_layers):
            outs = [F.leaky_relu(conv(inputs[-1])) for conv in self.decoder_cnn[i]]
            max_sz = outs[-1].shape[2]
            padded = []
            for o in outs:
                d = max_sz - o.shape[2]
                left, right = d // 2, int((d / 2. + 0.5) // 1)
                padded.append(F.pad(o, (left, right)))
            nxt = self.dec_drop(torch.cat(padded, dim=1))
            if self.decoder_cnn_residual and i > 0 and (i % 2) == 0:
                prev = inputs[-2]
                nxt = nxt + F.pad(prev, (max_sz - prev.shape[2], 0))
            inputs.append(nxt)
        feat = inputs[-1].max(dim=2)[0]
        dec_out = feat.mm(self.decoder_W) + self.decoder_b.view(1, -1)
        if compute_softmax:
            dec_out = F.softmax(dec_out, dim=-1)
        return dec_out, feat


def get_classifier(params):
    """model_utils.py:5-12."""
    if params.get('modeltype', 'lstm') == 'lstm':
        return CharLstm(params)
    return CharCNN(params)
