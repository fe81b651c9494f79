"""HIP-backed CharCNN attribution classifier (reference models/char_cnn.py:8-102): embedding ->
n_layers x {parallel Conv1d(K, padding=K//2) -> leaky_relu -> pad to the last branch's length ->
concat -> dropout [-> residual every 2nd layer]} -> max over time -> linear.

Each Conv1d runs as K shifted MFMA GEMMs over the time-major [T,B,C] activations (a4nt_sgemm with
accumulate); when the last layer feeds the global max-pool directly (no active dropout, no
residual on that layer) leaky_relu + zero-pad + max-over-time are one fused kernel
(a4nt_lrelu_maxtime).  Padding positions are not masked (`lens` unused), as in the reference.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .gen_decode import SoftSamples, sample_embed


class _Conv1dTM(Function):
    """x [T,B,C] (time major), weight [F,C,K], bias [F], zero padding `pad` (default K//2) on both sides
    -> pre-activations [Tout,B,F]."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad=None):
        T, B, C = x.shape
        Fo, _, K = weight.shape
        pad = K // 2 if pad is None else int(pad)
        Tout = T + 2 * pad - K + 1
        x2 = x.contiguous().view(T * B, C)
        wk = weight.permute(2, 0, 1).contiguous()              # [K][F][C]
        out = bias.view(1, Fo).expand(Tout * B, Fo).contiguous()
        for k in range(K):
            t0, t1 = max(0, pad - k), min(Tout, T + pad - k)    # output steps whose tap k is inside the input
            if t1 > t0:
                s0 = t0 + k - pad
                ops.gemm(x2[s0 * B:(s0 + t1 - t0) * B], wk[k], transB=True, out=out[t0 * B:t1 * B], beta=1)
        ctx.save_for_backward(x2, wk)
        ctx.dims = (T, B, C, Fo, K, pad, Tout)
        return out.view(Tout, B, Fo)

    @staticmethod
    def backward(ctx, dout):
        x2, wk = ctx.saved_tensors
        T, B, C, Fo, K, pad, Tout = ctx.dims
        d2 = dout.contiguous().view(Tout * B, Fo)
        dx = torch.zeros((T * B, C), device=d2.device) if ctx.needs_input_grad[0] else None
        dwk = torch.zeros((K, Fo, C), device=d2.device) if ctx.needs_input_grad[1] else None
        for k in range(K):
            t0, t1 = max(0, pad - k), min(Tout, T + pad - k)
            if t1 <= t0:
                continue
            s0 = t0 + k - pad
            xs, ds = x2[s0 * B:(s0 + t1 - t0) * B], d2[t0 * B:t1 * B]
            if dx is not None:
                ops.gemm(ds, wk[k], out=dx[s0 * B:(s0 + t1 - t0) * B], beta=1)
            if dwk is not None:
                dwk[k].copy_(ops.gemm_auto(ds, xs, transA=True))
        dw = dwk.permute(1, 2, 0).contiguous() if dwk is not None else None
        db = ops.colsum(d2) if ctx.needs_input_grad[2] else None
        return (dx.view(T, B, C) if dx is not None else None), dw, db, None


class _LreluMaxTime(Function):
    """feat[b,f] = max over time of leaky_relu(pre[t,b,f]) with `pad` zero positions joining the max."""

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
                emb = ops.embedding(x.to(dev), self.encoder.weight, 0)
            cur = self._drop(emb, self.p_enc)                      # [T,B,C] time major throughout
            inputs = [cur]
            drop_active = self.training and self.p_dec > 0.0
            feat = None
            for i in range(self.n_layers):
                pres = [_Conv1dTM.apply(inputs[-1], conv.weight, conv.bias) for conv in self.decoder_cnn[i]]
                max_sz = pres[-1].shape[0]
                res_here = self.decoder_cnn_residual and i > 0 and (i % 2) == 0
                if i == self.n_layers - 1 and not drop_active and not res_here:
                    feats = []
                    for p in pres:
                        d = max_sz - p.shape[0]
                        left, right = d // 2, int((d / 2. + 0.5) // 1)
                        # a negative pad crops (FN.pad semantics, char_cnn.py:86: branch longer than the last one)
                        p = p[max(0, -left):p.shape[0] - max(0, -right)]
                        feats.append(_LreluMaxTime.apply(p, max(0, left) + max(0, right), 0.01))
                    feat = torch.cat(feats, dim=1)
                    break
                padded = []
                for p in pres:
                    d = max_sz - p.shape[0]
                    left, right = d // 2, int((d / 2. + 0.5) // 1)
                    padded.append(F.pad(F.leaky_relu(p), (0, 0, 0, 0, left, right)))
                nxt = self._drop(torch.cat(padded, dim=2), self.p_dec)
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
