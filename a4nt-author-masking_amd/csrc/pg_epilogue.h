// Epilogue of the fused projection + Gumbel-softmax tile (projgumbel.hip, genpersist.hip): one wave holds RT row tiles
// x 5 column tiles of 16x16 accumulators (C layout: col = lane & 15, row = 4*(lane >> 4) + reg; column tile c of lane
// (fr, fq) is column c0 + 5*fr + c).  It adds the bias, scales, adds the Gumbel noise, stores z (for the rows whose
// backward will run) and reduces per row, over the wave's 80 columns, (max z, arg max z, sum exp((z - max)/tau)) into
// the strip's entry of `part`.
//
// Written for the instruction count (the epilogue was a quarter of the per-step launch): no per-element branches -- the
// stores are raw buffer stores whose out-of-range lanes get an out-of-bounds offset, which the hardware drops -- and the
// reductions over the 16 lanes that share a row are DPP row rotations (one VALU instruction per step) instead of
// ds_bpermute round trips.  Lane fr = 0's results are bit-identical to an xor butterfly: for that lane a rotation by k
// IS the xor partner, and the additions commute.
#pragma once
#include "gemm_stage.h"

__device__ __forceinline__ void pg_st1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, 0);
}

struct PgEpi {
    int M, N;                 // rows, columns of the projection
    float scale, inv_tau; int greedy;
    int m0, c0, fr, fq;       // first row of the wave's tile, first column of the strip, lane coordinates
    __amdgpu_buffer_rsrc_t rz; int z_on, zrow0, zrows, ldz;    // z of this step (rows zrow0.. only), based at row zrow0
    __amdgpu_buffer_rsrc_t rpart; int nstrips, strip;
};

// bv: bias of the lane's 5 columns (0 past the end); gn: the Gumbel noise of element (tile r*5 + c, reg e).
// SC1: write `part` through (in-launch hand-off) instead of plain.
template <int RT, bool SC1>
__device__ __forceinline__ void pg_epilogue(const PgEpi& p, const f32x4 (&acc)[RT][5], const float (&bv)[5],
                                            const float (&gn)[RT * 5][4]) {
    const int colb = p.c0 + 5 * p.fr;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rbase = p.m0 + 16 * r + 4 * p.fq;
        float zv[5][4];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const bool cvalid = colb + c < p.N;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xx = acc[r][c][e] + bv[c];
                const float zz = p.greedy ? xx : (xx * p.scale + gn[r * 5 + c][e]);
                zv[c][e] = (cvalid && rbase + e < p.M) ? zz : -INFINITY;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = rbase + e;
            if (p.z_on) {
                const int zr = row - p.zrow0;
                const bool rok = row < p.M && zr >= 0 && zr < p.zrows;
                const unsigned base = (unsigned)(zr * p.ldz + colb) * 4u;
#pragma unroll
                for (int c = 0; c < 5; ++c) pg_st1(p.rz, (rok && colb + c < p.N) ? base + 4u * c : OOB, zv[c][e]);
            }
            float m = zv[0][e]; int mi = colb;
#pragma unroll
            for (int c = 1; c < 5; ++c)
                if (zv[c][e] > m) { m = zv[c][e]; mi = colb + c; }
            if (m == -INFINITY) mi = 0x7fffffff;
#define PG_ARGMAX_STEP(n)                                                                      \
            {                                                                                  \
                const float om = pg_dpp_f<PG_ROR(n)>(m); const int oi = pg_dpp_i<PG_ROR(n)>(mi); \
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }                         \
            }
            PG_ARGMAX_STEP(8) PG_ARGMAX_STEP(4) PG_ARGMAX_STEP(2) PG_ARGMAX_STEP(1)
#undef PG_ARGMAX_STEP
            float s = 0.f;
            if (!p.greedy) {
#pragma unroll
                for (int c = 0; c < 5; ++c) s += (zv[c][e] == -INFINITY) ? 0.f : expf((zv[c][e] - m) * p.inv_tau);
                s += pg_dpp_f<PG_ROR(8)>(s);
                s += pg_dpp_f<PG_ROR(4)>(s);
                s += pg_dpp_f<PG_ROR(2)>(s);
                s += pg_dpp_f<PG_ROR(1)>(s);
            }
            const f32x4 pv = (f32x4){m, s, __int_as_float(mi), 0.f};
            const unsigned po = (p.fr == 0 && row < p.M) ? (unsigned)(row * p.nstrips + p.strip) * 16u : OOB;
            u32x4 v;
            v.x = __float_as_uint(pv.x); v.y = __float_as_uint(pv.y); v.z = __float_as_uint(pv.z); v.w = 0u;
            __builtin_amdgcn_raw_buffer_store_b128(v, p.rpart, (int)po, 0, SC1 ? 16 : 0);
        }
    }
}
