// probe of v_permlane16_swap_b32 on gfx950 (round 5: the cross-row exchange of the 32-lane block layout at order 4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 2; r++) { printf("r[%d]:", r); for (int i = 0; i < 64; i++) printf(" %u", h[64*r + i]); printf("\n"); }
    return 0;
}
