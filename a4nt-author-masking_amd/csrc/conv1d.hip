// Conv1d over time-major activations + bias + leaky_relu (+ zero-pad placement, + max over time) in ONE launch
// per branch (reference models/char_cnn.py:83-92: Conv1d(K, padding=K//2) -> leaky_relu -> pad to a common
// length -> concat -> max_pool1d over time).
//
//   y[t][b][f] = bias[f] + sum_k sum_c x[t + k - pad][b][c] * w[k][c][f]        (x rows outside [0, T) are zero)
//
// The K taps are K shifted views of the same [T*B, C] matrix, so the whole convolution is one contraction over
// (tap, channel) accumulated in the MFMA accumulators (v_mfma_f32_32x32x2_f32): the output is written once,
// never read-modify-written per tap.  A workgroup tile is 128 output rows x 128 filters; its rows are
// 32 consecutive time steps x 4 consecutive batch rows, laid out so that ONE lane holds all 32 time steps of
// one (batch row, filter) pair in its accumulator registers (C layout of the 32x32 tile: column = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)): the max over time needs no cross-lane traffic, and a
// (batch row, filter) pair receives one 64-bit atomic max per 32 time steps.  The packed key (ordered value
// bits, step) makes that max independent of the order of arrival: results are deterministic.
// The same kernel computes the data gradient (a convolution of d(pre) with the flipped, transposed taps).
#include "common.h"
#include "gemm_stage.h"

#define CBK 16

struct ConvArgs {
    const float* x; const float* w; const float* bias; float* out; unsigned long long* maxkey;
    int T, B, C, F, K, pad, Tout;
    int ldx, ldo;
    int act; float slope;
    int t_lo, t_hi, t_off;      // output steps [t_lo, t_hi) are emitted, at time offset t_off - t_lo
    int beta;
    int tiles_b, tiles_n;
};

// tile-local row -> (time step, batch row) of the tile (see the header comment)
__device__ __forceinline__ int row_tt(int r) { return (r & 3) | (((r >> 3) & 3) << 2) | (((r >> 5) & 1) << 4); }
__device__ __forceinline__ int row_bb(int r) { return ((r >> 2) & 1) | (((r >> 6) & 1) << 1); }

// order-preserving map float -> unsigned (larger float <=> larger unsigned)
__device__ __forceinline__ unsigned f2ord(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
// key of (value, step): larger value wins, then the earlier step; a zero-pad position (step < 0) wins ties
__device__ __host__ __forceinline__ unsigned long long pad_key() { return ((unsigned long long)0x80000000u << 32) | 0xffffffffu; }

__global__ __launch_bounds__(256) void conv1d_kernel(ConvArgs g) {
    constexpr int BM = 128, BN = 128;
    constexpr int ASZ = CBK * (BM + LPAD), BSZ = CBK * (BN + LPAD);
    __shared__ __attribute__((aligned(16))) float As[2][ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[2][BSZ];

    int tile = blockIdx.x;
    const int tn = tile % g.tiles_n; tile /= g.tiles_n;
    const int tb = tile % g.tiles_b; const int tt_ = tile / g.tiles_b;
    const int t0 = 32 * tt_, b0 = 4 * tb, n0 = BN * tn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A staging: thread -> rows r = tid/4 + 64*p, k quad kq; the row's (step, batch row) is fixed for the kernel
    const int kq = (tid & 3) * 4;
    int a_t[2], a_b[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (tid >> 2) + 64 * p;
        a_t[p] = t0 + row_tt(r);
        a_b[p] = b0 + row_bb(r);
    }
    const long long xbytes = ((long long)(g.T * g.B - 1) * g.ldx + g.C) * 4;
    __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, clamp_bytes(xbytes));
    const int nct = (g.C + CBK - 1) / CBK;
    const int nkt = g.K * nct;
    const bool eb = (g.F - n0) < BN;

    f32x4 va[2];
    StageMC<BN, CBK> sb;
    auto load = [&](int kt) {
        const int tap = kt / nct, c0 = (kt - tap * nct) * CBK;
        const int kleft = g.C - c0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ts = a_t[p] + tap - g.pad;                       // input step of this tap
            const bool rv = ts >= 0 && ts < g.T && a_b[p] < g.B;
            const unsigned off = (unsigned)(((long long)ts * g.B + a_b[p]) * g.ldx + c0 + kq) * 4u;
            if (kleft >= CBK) {
                va[p] = bl4(rx, rv ? off : OOB);
            } else {
                f32x4 t;
                t.x = bl1(rx, (rv && kq + 0 < kleft) ? off : OOB);
                t.y = bl1(rx, (rv && kq + 1 < kleft) ? off + 4 : OOB);
                t.z = bl1(rx, (rv && kq + 2 < kleft) ? off + 8 : OOB);
                t.w = bl1(rx, (rv && kq + 3 < kleft) ? off + 12 : OOB);
                va[p] = t;
            }
        }
        const float* wt = g.w + (long long)tap * g.C * g.F;
        if (eb) sb.template load<true>(wt, g.F, n0, g.F, c0, g.C);
        else sb.template load<false>(wt, g.F, n0, g.F, c0, g.C);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (tid >> 2) + 64 * p;
            As[buf][(kq + 0) * (BM + LPAD) + r] = va[p].x;
            As[buf][(kq + 1) * (BM + LPAD) + r] = va[p].y;
            As[buf][(kq + 2) * (BM + LPAD) + r] = va[p].z;
            As[buf][(kq + 3) * (BM + LPAD) + r] = va[p].w;
        }
        sb.store(Bs[buf]);
    };

    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load(kt + 1);
        mma_tile_k<2, 2, CBK>(As[cur], Bs[cur], lane, wm, wn, acc);
        if (kt + 1 < nkt) store(cur ^ 1);
        __syncthreads();
    }

    // epilogue
    const int h = lane >> 5;
    const int b = b0 + (h | ((wave >> 1) << 1));                 // the one batch row of this lane (row_bb of its rows)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + 32 * j + (lane & 31);
        const bool cv = col < g.F && b < g.B;
        const float bv = (g.bias && cv) ? g.bias[col] : 0.f;
        unsigned long long best = 0ull;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int t = t0 + row_tt(R);
                float v = acc[i][j][r] + bv;
                if (g.act) v = v > 0.f ? v : v * g.slope;
                if (cv && t >= g.t_lo && t < g.t_hi && t < g.Tout) {
                    if (g.out) {
                        float* dst = g.out + ((long long)(t - g.t_lo + g.t_off) * g.B + b) * g.ldo + col;
                        *dst = g.beta ? (*dst + v) : v;
                    }
                    const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (unsigned)(0xfffffffeu - (unsigned)t);
                    best = key > best ? key : best;
                }
            }
        if (g.maxkey && best != 0ull) atomicMax(g.maxkey + (long long)b * g.F + col, best);
    }
}

extern "C" int a4nt_conv1d_tm(int T, int B, int C, int F, int K, int pad, const float* x, int ldx, const float* w,
                              const float* bias, int act, float slope, float* out, int ldo, int t_lo, int t_hi,
                              int t_off, int beta, unsigned long long* maxkey, void* stream) {
    const int Tout = T + 2 * pad - K + 1;
    // (pad may be negative: a crop -- the data gradient of a convolution padded by more than K-1)
    if (T <= 0 || B <= 0 || C <= 0 || F <= 0 || K <= 0 || Tout <= 0 || !x || !w || (!out && !maxkey) ||
        ldx < C || (out && ldo < F) || t_lo < 0 || t_hi > Tout || t_lo >= t_hi || t_off < 0) {
        a4nt_set_error("a4nt_conv1d_tm: bad arguments (T=%d B=%d C=%d F=%d K=%d pad=%d, steps [%d,%d) of %d)", T, B, C, F, K,
                       pad, t_lo, t_hi, Tout);
        return A4NT_ERR_ARG;
    }
    if (((long long)T * B * ldx) * 4 >= 0x7fffffffll) {
        a4nt_set_error("a4nt_conv1d_tm: input of %lld bytes exceeds the 2 GiB one buffer descriptor covers", (long long)T * B * ldx * 4);
        return A4NT_ERR_ARG;
    }
    ConvArgs g;
    g.x = x; g.w = w; g.bias = bias; g.out = out; g.maxkey = maxkey;
    g.T = T; g.B = B; g.C = C; g.F = F; g.K = K; g.pad = pad; g.Tout = Tout;
    g.ldx = ldx; g.ldo = ldo; g.act = act; g.slope = slope; g.t_lo = t_lo; g.t_hi = t_hi; g.t_off = t_off; g.beta = beta;
    g.tiles_b = (B + 3) / 4; g.tiles_n = (F + 127) / 128;
    const long long tiles = (long long)((Tout + 31) / 32) * g.tiles_b * g.tiles_n;
    if (tiles > 0x7fffffffll) { a4nt_set_error("a4nt_conv1d_tm: too many tiles"); return A4NT_ERR_ARG; }
    hipLaunchKernelGGL(conv1d_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, g);
    return a4nt_check_launch("conv1d_tm");
}

// ---------------------------------------------------------------------------------------------
// max-over-time keys: initialise, decode, scatter the gradient back
// ---------------------------------------------------------------------------------------------
__global__ void maxkey_init_kernel(long long n, unsigned long long* keys, int with_pad) {
    const unsigned long long v = with_pad ? pad_key() : 0ull;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) keys[i] = v;
}
// feat[b][f] = winning value, argmax[b][f] = winning step (-1: a zero-pad position won)
__global__ void maxkey_decode_kernel(int B, int F, const unsigned long long* keys, float* feat, int ldf, int* argmax) {
    const long long total = (long long)B * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i - (long long)b * F);
        const unsigned long long k = keys[i];
        const unsigned lo = (unsigned)(k & 0xffffffffu);
        feat[(long long)b * ldf + f] = ord2f((unsigned)(k >> 32));
        argmax[i] = lo == 0xffffffffu ? -1 : (int)(0xfffffffeu - lo);
    }
}
// dpre[argmax[b][f]][b][f] = dfeat[b][f] * (feat > 0 ? 1 : slope); everything else of dpre must already be zero
__global__ void maxtime_scatter_kernel(int B, int F, const float* dfeat, int ldd, const float* feat, int ldf,
                                       const int* argmax, float slope, float* dpre, int ldp) {
    const long long total = (long long)B * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i - (long long)b * F);
        const int t = argmax[i];
        if (t < 0) continue;
        const float fv = feat[(long long)b * ldf + f];
        dpre[((long long)t * B + b) * ldp + f] = dfeat[(long long)b * ldd + f] * (fv > 0.f ? 1.f : slope);
    }
}

extern "C" int a4nt_maxkey_init(int B, int F, unsigned long long* keys, int with_pad, void* stream) {
    if (B <= 0 || F <= 0 || !keys) { a4nt_set_error("a4nt_maxkey_init: bad arguments"); return A4NT_ERR_ARG; }
    const long long n = (long long)B * F;
    hipLaunchKernelGGL(maxkey_init_kernel, dim3((unsigned)min((long long)1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, keys, with_pad);
    return a4nt_check_launch("maxkey_init");
}
extern "C" int a4nt_maxkey_decode(int B, int F, const unsigned long long* keys, float* feat, int ldf, int* argmax, void* stream) {
    if (B <= 0 || F <= 0 || !keys || !feat || !argmax || ldf < F) { a4nt_set_error("a4nt_maxkey_decode: bad arguments"); return A4NT_ERR_ARG; }
    const long long n = (long long)B * F;
    hipLaunchKernelGGL(maxkey_decode_kernel, dim3((unsigned)min((long long)1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, F, keys, feat, ldf, argmax);
    return a4nt_check_launch("maxkey_decode");
}
extern "C" int a4nt_maxtime_scatter_bwd(int Tout, int B, int F, const float* dfeat, int ldd, const float* feat, int ldf,
                                        const int* argmax, float slope, float* dpre, int ldp, void* stream) {
    if (B <= 0 || F <= 0 || Tout <= 0 || !dfeat || !feat || !argmax || !dpre || ldp < F) { a4nt_set_error("a4nt_maxtime_scatter_bwd: bad arguments"); return A4NT_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (ldp == F) {
        if (hipMemsetAsync(dpre, 0, (size_t)Tout * B * F * 4, s) != hipSuccess) return a4nt_check_launch("maxtime_scatter memset");
    } else {
        if (hipMemset2DAsync(dpre, (size_t)ldp * 4, 0, (size_t)F * 4, (size_t)Tout * B, s) != hipSuccess) return a4nt_check_launch("maxtime_scatter memset");
    }
    const long long n = (long long)B * F;
    hipLaunchKernelGGL(maxtime_scatter_kernel, dim3((unsigned)min((long long)1024, (n + 255) / 256)), dim3(256), 0, s, B, F, dfeat, ldd, feat, ldf, argmax, slope, dpre, ldp);
    return a4nt_check_launch("maxtime_scatter_bwd");
}
