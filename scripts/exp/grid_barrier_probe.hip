// grid_barrier_probe.hip -- what does a grid-wide barrier cost on MI355X (8 XCDs, non-coherent L2s) inside ONE launch?  The question behind a
// persistent kernel for the fixed-length Chebyshev solves of the shallow-water step (3 launches of ~5-9 us per step today): a step would be
// three phases separated by grid barriers.  Every workgroup: writes a value the NEXT phase's reader (another workgroup, likely another XCD)
// loads, release fence, arrive on a device-scope counter, spin (bounded) until all arrived, acquire fence.  Variants: fence kinds.
//   build: hipcc -O3 --offload-arch=gfx950 grid_barrier_probe.hip -o build_ab/grid_barrier_probe      run: grid_barrier_probe [nwg] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>     // 0: counter only (no data, no fences); 1: __threadfence() both sides + plain data; 2: data through sc1 (system-coherent) accesses
__global__ __launch_bounds__(64) void k_barrier(unsigned* ctr, int iters, double* data, int* bad, double* sink) {
    const unsigned nwg = gridDim.x;
    const int me = blockIdx.x, peer = (blockIdx.x*97 + 31)%nwg;        // a reader far away in block order (another XCD, most of the time)
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        if (MODE >= 1 && threadIdx.x == 0) {
            if (MODE == 2) __builtin_nontemporal_store((double)(it + 1), &data[me]);
            else data[me] = (double)(it + 1);
        }
        if (MODE >= 1) __threadfence();                                   // release: my store before my arrival
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1)*nwg;
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > 20000000L) { *bad = 1; break; }             // (bounded: a lost wakeup must not hang the GPU)
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (MODE >= 1) __threadfence();                                   // acquire
        if (MODE >= 1 && threadIdx.x == 0) {
            const double v = MODE == 2 ? __builtin_nontemporal_load(&data[peer]) : __hip_atomic_load(&data[peer], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < (double)(it + 1)) *bad = 2;                           // the peer's store of THIS iteration must be visible
            acc += v;
        }
        if (*bad) break;
    }
    if (threadIdx.x == 0) sink[me] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    std::vector<int> sizes = {64, 256, 512, 1024};
    if (argc > 1) sizes = {std::atoi(argv[1])};
    unsigned* ctr; double *data, *sink; int* bad;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&data, 4096*8)); CHECK(hipMalloc(&sink, 4096*8)); CHECK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int nwg : sizes) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e30f; int hb = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipMemset(ctr, 0, 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(data, 0, 4096*8));
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_barrier<0>, dim3(nwg), dim3(64), 0, 0, ctr, iters, data, bad, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_barrier<1>, dim3(nwg), dim3(64), 0, 0, ctr, iters, data, bad, sink);
                else hipLaunchKernelGGL(k_barrier<2>, dim3(nwg), dim3(64), 0, 0, ctr, iters, data, bad, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                if (ms < best) best = ms;
                if (hb) break;
            }
            std::printf("workgroups %4d  mode %d (%s): %.2f us per barrier%s\n", nwg, mode,
                        mode == 0 ? "counter only" : (mode == 1 ? "plain data + __threadfence" : "nontemporal data + __threadfence"), 1e3*best/iters,
                        hb == 1 ? "  [SPIN LIMIT HIT]" : (hb == 2 ? "  [STALE DATA SEEN]" : ""));
            std::fflush(stdout);
        }
    }
    return 0;
}
