// Prototype: C[M,N] = A[M,K] * B[N,K]^T with fp32 operands split into three bf16 terms (a = a0 + a1 + a2 exactly)
// and six v_mfma_f32_32x32x16_bf16 products per k-chunk (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0), fp32 accumulate.
// 256x128 workgroup tile, 8 waves of 64x64 (two per SIMD), double-buffered LDS images; waves 0-3 split the next tile
// before their MFMAs and waves 4-7 after them, so that on every SIMD one wave's split runs under the other's MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm3 gemm3.hip ; run: ./gemm3 [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define BKT 32
#define BM 256
#define BN 128
#define IMA (BM * BKT * 2)           // bytes of one split image of the A tile
#define IMB (BN * BKT * 2)
#define BUF (3 * IMA + 3 * IMB)

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// (row, 8-byte chunk c of the row's 64 bytes) -> byte offset inside one split image
__device__ __forceinline__ int slot_off(int row, int c) {
    const int L = row >> 2;
    return L * 256 + ((((row & 3) * 8 + c) ^ (2 * (L & 15))) * 8);
}

// split 4 consecutive k of one row and store the three 8-byte chunks
template <int IM>
__device__ __forceinline__ void split_store(unsigned char* img, int off, float4 v) {
    const unsigned h01 = cvt_pk(v.x, v.y), h23 = cvt_pk(v.z, v.w);
    const float rx = v.x - __uint_as_float(h01 << 16), ry = v.y - __uint_as_float(h01 & 0xffff0000u);
    const float rz = v.z - __uint_as_float(h23 << 16), rw = v.w - __uint_as_float(h23 & 0xffff0000u);
    const unsigned m01 = cvt_pk(rx, ry), m23 = cvt_pk(rz, rw);
    const float sx = rx - __uint_as_float(m01 << 16), sy = ry - __uint_as_float(m01 & 0xffff0000u);
    const float sz = rz - __uint_as_float(m23 << 16), sw = rw - __uint_as_float(m23 & 0xffff0000u);
    const unsigned l01 = cvt_pk(sx, sy), l23 = cvt_pk(sz, sw);
    u2 t;
    t.x = h01; t.y = h23; *reinterpret_cast<u2*>(img + off) = t;
    t.x = m01; t.y = m23; *reinterpret_cast<u2*>(img + IM + off) = t;
    t.x = l01; t.y = l23; *reinterpret_cast<u2*>(img + 2 * IM + off) = t;
}

struct Regs { float4 a[4]; float4 b[2]; };

__global__ __launch_bounds__(512, 1) void gemm3_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int role = wave >> 2;

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int srow = tid >> 3, sc = tid & 7;          // staging: float4 f = tid + 512 p -> row srow + 64 p, chunk sc
    auto gload = [&](Regs& R, int k0) {
        const bool kv = k0 < K;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int gm = m0 + srow + 64 * p;
            R.a[p] = (kv && gm < M) ? *reinterpret_cast<const float4*>(A + (long long)gm * K + k0 + 4 * sc) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int gn = n0 + srow + 64 * p;
            R.b[p] = (kv && gn < N) ? *reinterpret_cast<const float4*>(B + (long long)gn * K + k0 + 4 * sc) : make_float4(0, 0, 0, 0);
        }
    };
    int soff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) soff[p] = slot_off(srow + 64 * p, sc);
    auto unit = [&](const Regs& R, unsigned char* buf, int u) {     // u = 0..11: one float4 of the staged tile
        if (u < 4) split_store<IMA>(buf, soff[u], R.a[u]);
        else split_store<IMB>(buf + 3 * IMA, soff[u - 4], R.b[u - 4]);
    };

    const int r32 = lane & 31, h = lane >> 5;
    int offa[2][2], offb[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) offa[kk][i] = slot_off(wm + 32 * i + r32, 4 * kk + 2 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j) offb[kk][j] = slot_off(wn + 32 * j + r32, 4 * kk + 2 * h);
    }

    auto compute = [&](const unsigned char* cur) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf8 a[2][3], b[2][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][s] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(cur + s * IMA + offa[kk][i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][s] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(cur + 3 * IMA + s * IMB + offb[kk][j]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f16v c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    };
    auto split = [&](const Regs& R, unsigned char* nxt) {
#pragma unroll
        for (int u = 0; u < 6; ++u) unit(R, nxt, u);
    };
    // one K tile: MFMAs from `cur`, the staged registers R split into `nxt`, then R refilled from k = knext
    auto step = [&](const unsigned char* cur, unsigned char* nxt, Regs& R, int knext) {
        if (role == 0) {
            split(R, nxt);
            gload(R, knext);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
        } else {
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            split(R, nxt);
            gload(R, knext);
        }
        __syncthreads();
    };

    Regs R0, R1;
    unsigned char* buf0 = smem;
    unsigned char* buf1 = smem + BUF;
    gload(R0, 0);
    gload(R1, BKT);
    split(R0, buf0);
    gload(R0, 2 * BKT);
    __syncthreads();
    const int nk = (K + BKT - 1) / BKT;
    for (int kt = 0; kt < nk; kt += 2) {
        // tile kt in buf0; R1 = tile kt+1; R0 = tile kt+2
        step(buf0, buf1, R1, (kt + 3) * BKT);
        if (kt + 1 >= nk) break;
        step(buf1, buf0, R0, (kt + 4) * BKT);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) C[(long long)row * N + col] = acc[i][j][r];
            }
        }
}

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 12800, N = argc > 3 ? atoi(argv[2]) : 10001, K = argc > 3 ? atoi(argv[3]) : 512;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& x : hB) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipFuncSetAttribute((const void*)gemm3_nt, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm3_nt, dim3(tiles), dim3(512), 2 * BUF, 0, dA, dB, dC, M, N, K);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm3_nt, dim3(tiles), dim3(512), 2 * BUF, 0, dA, dB, dC, M, N, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("M=%d N=%d K=%d: %.3f ms, %.1f TFLOP/s (fp32-equivalent)\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, e32max = 0;
    for (int t = 0; t < 2000; ++t) {
        const int i = rand() % M, j = rand() % N;
        double ref = 0, sa = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) {
            ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
            sa += fabs((double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k]);
            f = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], f);
        }
        emax = fmax(emax, fabs(hC[(size_t)i * N + j] - ref) / sa);
        e32max = fmax(e32max, fabs((double)f - ref) / sa);
    }
    printf("max |err| / sum|a b|: split-bf16 %.3e   f32 fma chain %.3e\n", emax, e32max);
    return 0;
}
