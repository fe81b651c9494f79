// Census: which blocks of a 512-block, 2-per-CU persistent launch share a CU?  (placement is used for speed only)
// build: hipcc -O2 --offload-arch=gfx950 census.hip -o census ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void census(unsigned* out, unsigned* ctr, int nblocks) {
    __shared__ float pad[4096];                      // 16 KB like the sweeps
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID bits 0..3
        out[2 * blockIdx.x] = hwid;
        out[2 * blockIdx.x + 1] = xcc;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {                         // all resident at once?
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nblocks) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (pad[threadIdx.x] < 0) out[0] = 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    unsigned *out, *ctr;
    hipMalloc(&out, n * 8); hipMalloc(&ctr, 4); hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 0, 0, out, ctr, n);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * n); unsigned c = 0;
    hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    printf("arrived %u of %d\n", c, n);
    std::map<unsigned, std::vector<int>> cu;          // key: xcc, se, sh, cu
    for (int b = 0; b < n; ++b) {
        const unsigned id = h[2 * b], xcc = h[2 * b + 1];
        const unsigned cu_id = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 0x7;
        cu[(xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("%zu distinct (xcc,se,sh,cu); first 24:\n", cu.size());
    int k = 0;
    for (auto& kv : cu) {
        if (k++ >= 24) break;
        printf("  xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    std::map<int, int> diffs;
    for (auto& kv : cu) if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
    printf("block-index difference of the two blocks of a CU (count):");
    for (auto& d : diffs) printf(" %d(%d)", d.first, d.second);
    printf("\n");
    return 0;
}
