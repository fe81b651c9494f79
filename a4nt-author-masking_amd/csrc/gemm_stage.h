// Branch-free operand staging shared by the GEMM kernels (sgemm.hip, projgumbel.hip).
#pragma once
#include "common.h"

#define LPAD 4

// Operand staging uses raw buffer loads: out-of-range lanes get an out-of-bounds offset and the
// hardware returns 0, so the loads are branch-free and stay in flight across the MFMA block (a
// per-lane `if (in range) load` makes hipcc wait vmcnt(0) at every merge point, which serialised the
// k-loop on L2 latency).  The descriptor is rebuilt per tile at the tile origin, so 32-bit offsets
// are enough for any matrix size.  Rows need only 4-byte alignment (gfx950 runs in unaligned
// access mode; dwordx4 loads of V = 10001-strided rows are legal).
#define OOB 0x80000000u

__device__ __forceinline__ f32x4 bl4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    f32x4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}
__device__ __forceinline__ float bl1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ unsigned clamp_bytes(long long b) {
    return (unsigned)(b < 0 ? 0 : (b > 0x7fffffffll ? 0x7fffffffll : b));
}

// Stage one [ROWS x BK] operand tile, global element (r, k) at base[r*ld + k]  ("k contiguous").
template <int ROWS, int BKT>
struct StageKC {
    static constexpr int KQ = BKT / 4;            // float4 per row
    static constexpr int RPP = 256 / KQ;          // rows per pass
    static constexpr int NV = ROWS / RPP;
    f32x4 v[NV];
    // EDGE=false: the tile's BK columns are all valid (or the whole tile is past the end: kleft <= 0 -> zeros)
    // EDGE=true : only kleft < BK columns are valid (last, partial K tile): per-element dword loads
    template <bool EDGE>
    __device__ __forceinline__ void load(const float* base, int ld, int r0, int rmax, int k0, int kmax) {
        const int tid = threadIdx.x & 255;      // 256 staging threads per k-group
        const int kq = (tid % KQ) * 4;
        const int nrows = min(ROWS, rmax - r0);
        const int kleft = kmax - k0;
        const float* origin = base + (long long)r0 * ld + k0;
        const long long bytes = kleft > 0 ? ((long long)(nrows - 1) * ld + min(kleft, BKT)) * 4 : 0;
        __amdgpu_buffer_rsrc_t rs = make_rsrc(origin, clamp_bytes(bytes));
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int r = tid / KQ + RPP * p;
            const unsigned off = (unsigned)(r * ld + kq) * 4u;
            const bool rv = r < nrows;
            if (!EDGE) {
                v[p] = bl4(rs, rv ? off : OOB);
            } else {
                f32x4 t;
                t.x = bl1(rs, (rv && kq + 0 < kleft) ? off : OOB);
                t.y = bl1(rs, (rv && kq + 1 < kleft) ? off + 4 : OOB);
                t.z = bl1(rs, (rv && kq + 2 < kleft) ? off + 8 : OOB);
                t.w = bl1(rs, (rv && kq + 3 < kleft) ? off + 12 : OOB);
                v[p] = t;
            }
        }
    }
    __device__ __forceinline__ void store(float* S /* [BKT][ROWS+LPAD] */) const {
        const int tid = threadIdx.x & 255;      // 256 staging threads per k-group
        const int kq = (tid % KQ) * 4;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int r = tid / KQ + RPP * p;
            S[(kq + 0) * (ROWS + LPAD) + r] = v[p].x;
            S[(kq + 1) * (ROWS + LPAD) + r] = v[p].y;
            S[(kq + 2) * (ROWS + LPAD) + r] = v[p].z;
            S[(kq + 3) * (ROWS + LPAD) + r] = v[p].w;
        }
    }
};

// Stage one [BK x COLS] operand tile, global element (k, c) at base[k*ld + c]  ("mn contiguous").
template <int COLS, int BKT>
struct StageMC {
    static constexpr int PER_ROW = COLS / 4;
    static constexpr int NV = (BKT * PER_ROW) / 256;
    f32x4 v[NV];
    // EDGE=false: all COLS columns valid; EDGE=true: only cleft < COLS columns (last tile column / row block)
    template <bool EDGE>
    __device__ __forceinline__ void load(const float* base, int ld, int c0, int cmax, int k0, int kmax) {
        const int tid = threadIdx.x & 255;      // 256 staging threads per k-group
        const int nk = min(BKT, kmax - k0);
        const int cleft = cmax - c0;
        const float* origin = base + (long long)k0 * ld + c0;
        const long long bytes = nk > 0 ? ((long long)(nk - 1) * ld + min(cleft, COLS)) * 4 : 0;
        __amdgpu_buffer_rsrc_t rs = make_rsrc(origin, clamp_bytes(bytes));
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 256 * p;
            const int k = idx / PER_ROW;
            const int c = (idx % PER_ROW) * 4;
            const unsigned off = (unsigned)(k * ld + c) * 4u;
            const bool kv = k < nk;
            if (!EDGE) {
                v[p] = bl4(rs, kv ? off : OOB);
            } else {
                f32x4 t;
                t.x = bl1(rs, (kv && c + 0 < cleft) ? off : OOB);
                t.y = bl1(rs, (kv && c + 1 < cleft) ? off + 4 : OOB);
                t.z = bl1(rs, (kv && c + 2 < cleft) ? off + 8 : OOB);
                t.w = bl1(rs, (kv && c + 3 < cleft) ? off + 12 : OOB);
                v[p] = t;
            }
        }
    }
    __device__ __forceinline__ void store(float* S /* [BK][COLS+LPAD] */) const {
        const int tid = threadIdx.x & 255;      // 256 staging threads per k-group
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + 256 * p;
            const int k = idx / PER_ROW;
            const int c = (idx % PER_ROW) * 4;
            *reinterpret_cast<f32x4*>(&S[k * (COLS + LPAD) + c]) = v[p];
        }
    }
};


// One BKT-deep step of a (64*WM) x (64*WN) workgroup tile from k-major LDS tiles ([BKT][rows + LPAD]): wave (wm, wn)
// owns WM x WN tiles of 32x32 (v_mfma_f32_32x32x2_f32, 16 accumulator registers each).
template <int WM, int WN, int BKT>
__device__ __forceinline__ void mma_tile_k(const float* as, const float* bs, int lane, int wm, int wn, f32x16 (&acc)[WM][WN]) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 2) {
        const int kr = kk + (lane >> 5);
        float a[WM], b[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = as[kr * (BM + LPAD) + wm + 32 * i + (lane & 31)];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = bs[kr * (BN + LPAD) + wn + 32 * j + (lane & 31)];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
