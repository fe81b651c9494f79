// Shared device helpers for the A4NT gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a4nt_hip.h"     // every definition is compiled against its public prototype

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define A4NT_WAVE 64

// ---------------------------------------------------------------------------------------------
// status codes returned by every C-ABI entry point (include/a4nt_hip.h)
// ---------------------------------------------------------------------------------------------
enum {
    A4NT_OK = 0,
    A4NT_ERR_ARG = 1,        // bad shape / null pointer / unsupported size
    A4NT_ERR_LAUNCH = 2,     // hip launch error
    A4NT_ERR_TIMEOUT = 3,    // a bounded inter-workgroup wait expired (persistent kernels)
};

void a4nt_set_error(const char* fmt, ...);
int a4nt_check_launch(const char* what);
int a4nt_lstm_probe_occupancy(int device, int num_cus);     // lstm.hip, called by a4nt_init

// The sampler's finalisation (projgumbel.hip): merge the per-strip (max, sum exp, argmax) partials of a row, then the
// bookkeeping of one decoding step.  Run by proj_gumbel_final_kernel, or by the next decoder step's prologue (lstm.hip).
#define PG_BC 80          // strip width of the fused projection: five 16-column MFMA tiles per wave
struct PgFinalArgs {
    const float* part; int nstrips; float inv_tau; int greedy; int hard;
    float* stats; long long* tok; float* val_in; float* val_out; int* gen_lens; unsigned char* done; long long end_c;
};

// ---------------------------------------------------------------------------------------------
// inter-workgroup hand-off primitives (persistent LSTM kernels)
//
// Payload: 16-byte write-through (sc1) buffer stores, read back with 16-byte sc1 buffer loads,
// whole 128-B lines written by one store instruction of one wave.  Signal: one lane per
// workgroup, after every storing wave drained (s_waitcnt vmcnt(0)) and the workgroup barrier,
// adds 1 to an agent-scope counter; consumers poll that counter with a relaxed agent-scope
// (sc1) load from ONE lane, then pass a workgroup barrier before any wave loads the payload.
// Placement-independent: XCD co-location is used for speed only.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    f32x4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}
// the same with a wave-uniform part of the offset in an SGPR (no address VGPRs for per-block offsets)
__device__ __forceinline__ f32x4 ld16_sc1_s(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uniform_off, 16);
    f32x4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}
__device__ __forceinline__ void st16_sc1_s(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off, f32x4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)lane_off, (int)uniform_off, 16);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void add_flag(unsigned* p, unsigned v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bounded wait: returns false when the counter did not reach `target` within the spin budget or
// another workgroup already raised the timeout word.  Called by ONE lane.
#define A4NT_SPIN_LIMIT (1u << 24)
__device__ __forceinline__ bool wait_ge(const unsigned* ctr, unsigned target, unsigned* timeout_word) {
    for (unsigned spins = 0;; ++spins) {
        if (ld_flag(ctr) >= target) return true;
        if ((spins & 63u) == 63u && ld_flag(timeout_word) != 0u) return false;
        if (spins > A4NT_SPIN_LIMIT) { st_flag(timeout_word, 1u); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
// Accurate OCML expf/tanhf (not the v_exp fast path): the gate non-linearities sit inside a
// T-step recurrence and parity against the fp32 CPU reference is asserted at 1e-5.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// DPP row rotation inside the 16-lane rows of a wave: lane l reads lane (l + n) % 16 of its row -- one VALU instruction
// where __shfl_xor is an LDS-crossbar round trip.  For lane 0 of a row a rotation by 8 / 4 / 2 / 1 IS the xor partner, so
// a sum reduced by rotations 8, 4, 2, 1 is bit-identical, in lane 0, to the xor butterfly's.
template <int CTRL>
__device__ __forceinline__ float pg_dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int pg_dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
#define PG_ROR(n) (0x120 + (n))       // dpp_ctrl row_ror:n -- lane l of a 16-lane row reads lane (l + n) % 16

