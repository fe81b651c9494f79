"""HIP-backed CharCNN attribution classifier (reference models/char_cnn.py:8-102): embedding ->
n_layers x {parallel Conv1d(K, padding=K//2) -> leaky_relu -> pad to the last branch's length ->
concat -> dropout [-> residual every 2nd layer]} -> max over time -> linear.

Each Conv1d branch is ONE launch of a4nt_conv1d_tm: the K taps are shifted views of the time-major [T*B, C]
activations and accumulate in the MFMA accumulators; bias, leaky_relu and the placement into the zero-padded,
concatenated layer output happen in its epilogue, and when the last layer feeds the global max-pool directly
(no active dropout, no residual on that layer) so does the max over time -- the pre-activations are never
written.  Padding positions are not masked (`lens` unused), as in the reference.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .model_utils import ids_to_device
from .gen_decode import SoftSamples, sample_embed


def _conv_launch(x2, T, B, wt, bias, pad, act, slope, out=None, ldo=0, t_lo=0, t_hi=None, t_off=0, beta=0, keys=None):
    """One a4nt_conv1d_tm launch.  x2 [T*B, C] time major; wt [K][C][F] taps; the K taps accumulate in the MFMA
    accumulators of one kernel (no per-tap read-modify-write of the output)."""
    lib = ops._lib.load()
    K, C, Fo = wt.shape
    Tout = T + 2 * pad - K + 1
    t_hi = Tout if t_hi is None else t_hi
    ops.check(lib.a4nt_conv1d_tm(T, B, C, Fo, K, pad, ops._p(x2), x2.stride(0), ops._p(wt), ops._p(bias), int(act),
                                 float(slope), ops._p(out), int(ldo), int(t_lo), int(t_hi), int(t_off), int(beta),
                                 ops._p(keys), ops._s()), 'a4nt_conv1d_tm')
    return Tout


def _conv_grads(x2, T, B, weight, pad, dpre2, Tout, need_dx, need_dw, need_db, dx_acc=None):
    """Gradients of one Conv1d given d(pre) [Tout*B, F]: dx through the same kernel with the flipped, transposed
    taps (added into dx_acc when given), dW as one time-batched TN product per tap over shifted views, db."""
    Fo, C, K = weight.shape
    dx = dx_acc
    if need_dx:
        wflip = weight.permute(2, 0, 1).flip(0).contiguous()                      # [K][F][C]
        if dx is None:
            dx = torch.empty((T * B, C), device=x2.device)
            beta = 0
        else:
            beta = 1
        _conv_launch(dpre2, Tout, B, wflip, None, K - 1 - pad, 0, 0.0, out=dx, ldo=C, beta=beta)
    dw = None
    if need_dw:
        dwk = torch.zeros((K, Fo, C), device=x2.device)
        for k in range(K):
            t0, t1 = max(0, pad - k), min(Tout, T + pad - k)            # output steps whose tap k is inside the input
            if t1 > t0:
                s0 = t0 + k - pad
                dwk[k].copy_(ops.gemm_auto(dpre2[t0 * B:t1 * B], x2[s0 * B:(s0 + t1 - t0) * B], transA=True))
        dw = dwk.permute(1, 2, 0).contiguous()
    db = ops.colsum(dpre2) if need_db else None
    return dx, dw, db


class _Conv1dTM(Function):
    """x [T,B,C] (time major), weight [F,C,K], bias [F], zero padding `pad` (default K//2) on both sides
    -> pre-activations [Tout,B,F] (nn.Conv1d, char_cnn.py:84; char_lstm.py:75,324)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad=None):
        T, B, C = x.shape
        Fo, _, K = weight.shape
        pad = K // 2 if pad is None else int(pad)
        x2 = x.contiguous().view(T * B, C)
        wt = weight.permute(2, 1, 0).contiguous()              # [K][C][F]
        Tout = T + 2 * pad - K + 1
        out = torch.empty((Tout * B, Fo), device=x.device)
        _conv_launch(x2, T, B, wt, bias, pad, 0, 0.0, out=out, ldo=Fo)
        ctx.save_for_backward(x2, weight)
        ctx.dims = (T, B, C, Fo, K, pad, Tout)
        return out.view(Tout, B, Fo)

    @staticmethod
    def backward(ctx, dout):
        x2, weight = ctx.saved_tensors
        T, B, C, Fo, K, pad, Tout = ctx.dims
        d2 = dout.contiguous().view(Tout * B, Fo)
        dx, dw, db = _conv_grads(x2, T, B, weight, pad, d2, Tout, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 ctx.needs_input_grad[2])
        return (dx.view(T, B, C) if dx is not None else None), dw, db, None


def _branch_geometry(T, ks):
    """Per branch (K, pad, Tout, t_lo, t_hi, t_off, n_pad) and the common length: every branch is padded with zeros
    (or cropped, when it is longer) to the LAST kernel size's output length, left d//2, right int(d/2 + 0.5)
    (char_cnn.py:85-86)."""
    touts = [T + 2 * (K // 2) - K + 1 for K in ks]
    max_sz = touts[-1]
    geo = []
    for K, Tout in zip(ks, touts):
        d = max_sz - Tout
        left, right = d // 2, int((d / 2. + 0.5) // 1)
        geo.append((K, K // 2, Tout, max(0, -left), Tout - max(0, -right), max(0, left), max(0, left) + max(0, right)))
    return geo, max_sz


class _ConvLayer(Function):
    """One CharCNN layer (char_cnn.py:83-87): for every kernel size a Conv1d + leaky_relu whose output lands,
    zero-padded / cropped to the common length, in its column slice of the concatenated activations -- one
    launch per branch, nothing else.  apply(x, ks, slope, w_0, b_0, w_1, b_1, ...) -> [Tcommon, B, nb*F]."""

    @staticmethod
    def forward(ctx, x, ks, slope, *wb):
        T, B, C = x.shape
        x2 = x.contiguous().view(T * B, C)
        geo, max_sz = _branch_geometry(T, ks)
        Fo = wb[0].shape[0]
        nb = len(ks)
        out = torch.zeros((max_sz, B, nb * Fo), device=x.device)         # zero: the pad positions
        for i, (K, pad, Tout, t_lo, t_hi, t_off, _) in enumerate(geo):
            wt = wb[2 * i].permute(2, 1, 0).contiguous()
            _conv_launch(x2, T, B, wt, wb[2 * i + 1], pad, 1, slope, out=out.view(max_sz * B, nb * Fo)[:, i * Fo:],
                         ldo=nb * Fo, t_lo=t_lo, t_hi=t_hi, t_off=t_off)
        ctx.save_for_backward(x2, out, *wb[0::2])
        ctx.meta = (T, B, C, Fo, geo, max_sz, slope)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, out = ctx.saved_tensors[:2]
        weights = ctx.saved_tensors[2:]
        T, B, C, Fo, geo, max_sz, slope = ctx.meta
        nb = len(geo)
        # leaky_relu keeps the sign: its derivative is read off the saved output
        dact = torch.where(out > 0, dout, dout * slope)
        dx, grads = None, []
        for i, (K, pad, Tout, t_lo, t_hi, t_off, _) in enumerate(geo):
            dpre = torch.zeros((Tout, B, Fo), device=dout.device)
            dpre[t_lo:t_hi] = dact[t_off:t_off + (t_hi - t_lo), :, i * Fo:(i + 1) * Fo]
            dx, dw, db = _conv_grads(x2, T, B, weights[i], pad, dpre.view(Tout * B, Fo), Tout, ctx.needs_input_grad[0],
                                     ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i], dx_acc=dx)
            grads += [dw, db]
        return (dx.view(T, B, C) if dx is not None else None, None, None) + tuple(grads)


class _ConvLayerMax(Function):
    """The last CharCNN layer when it feeds the global max-pool directly (char_cnn.py:83-92): Conv1d + leaky_relu +
    zero-pad + max over time in the conv kernel's epilogue, one launch per branch; the pre-activations are never
    written.  apply(x, ks, slope, w_0, b_0, ...) -> feat [B, nb*F]."""

    @staticmethod
    def forward(ctx, x, ks, slope, *wb):
        lib = ops._lib.load()
        T, B, C = x.shape
        x2 = x.contiguous().view(T * B, C)
        geo, max_sz = _branch_geometry(T, ks)
        Fo = wb[0].shape[0]
        nb = len(ks)
        feat = torch.empty((B, nb * Fo), device=x.device)
        arg = torch.empty((nb, B, Fo), device=x.device, dtype=torch.int32)
        keys = torch.empty((nb, B, Fo), device=x.device, dtype=torch.int64)
        s = ops._s()
        for i, (K, pad, Tout, t_lo, t_hi, t_off, n_pad) in enumerate(geo):
            wt = wb[2 * i].permute(2, 1, 0).contiguous()
            ops.check(lib.a4nt_maxkey_init(B, Fo, ops._p(keys[i]), int(n_pad > 0), s), 'a4nt_maxkey_init')
            _conv_launch(x2, T, B, wt, wb[2 * i + 1], pad, 1, slope, t_lo=t_lo, t_hi=t_hi, keys=keys[i])
            ops.check(lib.a4nt_maxkey_decode(B, Fo, ops._p(keys[i]), ops._p(feat[:, i * Fo:]), nb * Fo, ops._p(arg[i]), s),
                      'a4nt_maxkey_decode')
        ctx.save_for_backward(x2, feat, arg, *wb[0::2])
        ctx.meta = (T, B, C, Fo, geo, slope)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = ops._lib.load()
        x2, feat, arg = ctx.saved_tensors[:3]
        weights = ctx.saved_tensors[3:]
        T, B, C, Fo, geo, slope = ctx.meta
        nb = len(geo)
        dfeat = dfeat.contiguous()
        dx, grads = None, []
        for i, (K, pad, Tout, t_lo, t_hi, t_off, _) in enumerate(geo):
            dpre = torch.empty((Tout * B, Fo), device=dfeat.device)
            ops.check(lib.a4nt_maxtime_scatter_bwd(Tout, B, Fo, ops._p(dfeat[:, i * Fo:]), nb * Fo, ops._p(feat[:, i * Fo:]),
                                                   nb * Fo, ops._p(arg[i]), float(slope), ops._p(dpre), Fo, ops._s()),
                      'a4nt_maxtime_scatter_bwd')
            dx, dw, db = _conv_grads(x2, T, B, weights[i], pad, dpre, Tout, ctx.needs_input_grad[0],
                                     ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i], dx_acc=dx)
            grads += [dw, db]
        return (dx.view(T, B, C) if dx is not None else None, None, None) + tuple(grads)


class _LreluMaxTime(Function):
    """feat[b,f] = max over time of leaky_relu(pre[t,b,f]) with `pad` zero positions joining the max (the
    CNN-over-hidden-states head of CharLstm, char_lstm.py:324-325)."""

    @staticmethod
    def forward(ctx, pre, pad, slope):
        lib = ops._lib.load()
        Tout, B, Fo = pre.shape
        pre = pre.contiguous()
        feat = torch.empty((B, Fo), device=pre.device)
        arg = torch.empty((B, Fo), device=pre.device, dtype=torch.int32)
        ops.check(lib.a4nt_lrelu_maxtime(Tout, B, Fo, ops._p(pre), Fo, float(slope), int(pad), ops._p(feat), Fo,
                                         ops._p(arg), ops._s()), 'a4nt_lrelu_maxtime')
        ctx.save_for_backward(pre, arg)
        ctx.slope = slope
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = ops._lib.load()
        pre, arg = ctx.saved_tensors
        Tout, B, Fo = pre.shape
        dpre = pre.clone()
        dfeat = dfeat.contiguous()
        ops.check(lib.a4nt_lrelu_maxtime_bwd(Tout, B, Fo, ops._p(dpre), Fo, float(ctx.slope), ops._p(dfeat), Fo,
                                             ops._p(arg), ops._s()), 'a4nt_lrelu_maxtime_bwd')
        return dpre, None, None


class _ConvParams(nn.Module):
    """Holds weight [F,C,K] / bias [F] under nn.Conv1d's parameter names."""

    def __init__(self, cin, nfilt, K, padding=None):
        super().__init__()
        self.K = K
        self.padding = K // 2 if padding is None else padding
        ref = nn.Conv1d(cin, nfilt, K, padding=self.padding)   # consumes the RNG exactly like the reference
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())


class CharCNN(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.n_words = params.get('vocabulary_size', -1) + 1
        self.num_outputs = params.get('num_output_layers', 1)
        self.num_output_layers = self.num_outputs
        self.emb_size = params.get('embedding_size', -1)
        self.encoder = nn.Embedding(self.n_words, self.emb_size, padding_idx=0)
        self.p_enc = params.get('drop_prob_encoder', 0.25)
        self.n_layers = params.get('decoder_cnn_nlayers', 1)
        self.decoder_cnn_residual = params.get('decoder_cnn_residual', 0)
        self.cnn_ks = params['decoder_cnn_ks']
        self.cnn_nfilt = params['decoder_cnn_nfilt']
        in_sizes = [self.emb_size] + [self.cnn_nfilt * len(self.cnn_ks)] * (self.n_layers - 1)
        self.decoder_cnn = nn.ModuleList([
            nn.ModuleList([_ConvParams(in_sizes[i], self.cnn_nfilt, K) for K in self.cnn_ks])
            for i in range(self.n_layers)])
        dsz = self.cnn_nfilt * len(self.cnn_ks)
        self.decoder_W = nn.Parameter(torch.zeros(dsz, self.num_outputs))
        self.decoder_b = nn.Parameter(torch.zeros(self.num_outputs))
        self.p_dec = params.get('drop_prob_decoder', 0.25)
        self.init_weights()
        self.to(params.get('device', 'cuda'))
        if self.decoder_W.is_cuda:            # device='cpu' builds a parameter container only (init / checkpoint tools);
            ops._lib.init_device(self.decoder_W.device.index or 0)      # every forward still needs the GPU

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

    def _drop(self, x, p):
        return F.dropout(x, p, True) if (self.training and p > 0.0) else x

    def forward_classify(self, x, h_prev=None, compute_softmax=False, predict_mode=False, adv_inp=False, lens=None,
                         drop=True):
        # `drop` accepted for the adversarial glue; like the reference the dropouts follow train()/eval() only
        dev = self.decoder_W.device
        T, B = x.shape[0], x.shape[1]
        ctxmgr = torch.no_grad() if predict_mode else torch.enable_grad()
        with ctxmgr:
            if isinstance(x, SoftSamples):
                emb = sample_embed(x, self.encoder.weight)
            elif adv_inp:
                emb = ops.linear_kn(x.reshape(T * B, -1), self.encoder.weight).view(T, B, -1)
            else:
                emb = ops.embedding(ids_to_device(x, dev), self.encoder.weight, 0)
            cur = self._drop(emb, self.p_enc)                      # [T,B,C] time major throughout
            inputs = [cur]
            drop_active = self.training and self.p_dec > 0.0
            feat = None
            for i in range(self.n_layers):
                wb = [t for conv in self.decoder_cnn[i] for t in (conv.weight, conv.bias)]
                res_here = self.decoder_cnn_residual and i > 0 and (i % 2) == 0
                if i == self.n_layers - 1 and not drop_active and not res_here:
                    # conv + leaky_relu + zero-pad + max over time: one launch per kernel size
                    feat = _ConvLayerMax.apply(inputs[-1], tuple(self.cnn_ks), 0.01, *wb)
                    break
                act = _ConvLayer.apply(inputs[-1], tuple(self.cnn_ks), 0.01, *wb)
                max_sz = act.shape[0]
                nxt = self._drop(act, self.p_dec)
                if res_here:
                    prev = inputs[-2]
                    nxt = nxt + F.pad(prev, (0, 0, 0, 0, max_sz - prev.shape[0], 0))
                inputs.append(nxt)
            if feat is None:
                feat = inputs[-1].max(dim=0)[0]
            dec_out = ops.linear_kn(feat, self.decoder_W, self.decoder_b)
            if compute_softmax:
                dec_out = F.softmax(dec_out, dim=-1)
        return dec_out, feat
