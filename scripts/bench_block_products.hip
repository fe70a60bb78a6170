// scripts/bench_block_products.hip -- A/B of three ways to multiply batches of small dense FP64 blocks (the 9x9 blocks of the p = 3
// column Schur assembly, the 16x16 blocks of p = 4; eul/VertSolve.cpp:694-767 are ten MatMatMult of such block-diagonal matrices):
//   lds   one block row per lane (16 lanes per product), the right-hand operand published in LDS and read back as broadcasts
//         -- RowBlocks<N>::mul of round 1;
//   dpp   the same row-per-lane layout with BOTH operands in registers: v_fmac_f64_dpp ... row_newbcast:m takes row m of the
//         right-hand block straight from lane m of the 16-lane DPP row -- no LDS, no publishing;
//   mfma  v_mfma_f64_16x16x4_f64 with the blocks zero-padded to 16x16 (one product per wavefront; operands through LDS because the
//         MFMA fragment layout is not the row-per-lane layout the Gauss-Jordan sweeps need).
// Each variant chains R dependent products C <- A . C per block so that the arithmetic, not the loads, is timed.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 scripts/bench_block_products.hip -o /tmp/bbp && /tmp/bbp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

template <int M> struct Bc {
    // acc += a * (b held by lane M of this lane's 16-lane row)
    static __device__ __forceinline__ void fmac(double& acc, double a, double b) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(M));
    }
};

template <int N2, int M> struct DppMul {
    static __device__ __forceinline__ void run(double (&C)[N2], const double (&A)[N2], const double (&B)[N2]) {
#pragma unroll
        for (int j = 0; j < N2; j++) Bc<M>::fmac(C[j], A[M], B[j]);
        if constexpr (M + 1 < N2) DppMul<N2, M + 1>::run(C, A, B);
    }
};

// ---- variant dpp -------------------------------------------------------------------------------------------------------------
template <int N2>
__global__ __launch_bounds__(64) void k_dpp(long long nb, int reps, const double* __restrict__ Ag, const double* __restrict__ Bg, double* __restrict__ Cg) {
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    const long long blk = (long long)blockIdx.x*4 + t;
    const bool act = blk < nb && r < N2;
    const long long b = blk < nb ? blk : nb - 1;
    const int rr = r < N2 ? r : 0;
    double A[N2], B[N2], C[N2];
#pragma unroll
    for (int j = 0; j < N2; j++) { A[j] = Ag[(b*N2 + rr)*N2 + j]; B[j] = Bg[(b*N2 + rr)*N2 + j]; }
    for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int j = 0; j < N2; j++) C[j] = 0.0;
        asm volatile("s_nop 1");                     // VALU write -> DPP read of the same VGPR needs two wait states
        DppMul<N2, 0>::run(C, A, B);
#pragma unroll
        for (int j = 0; j < N2; j++) B[j] = C[j];
    }
    if (act) {
#pragma unroll
        for (int j = 0; j < N2; j++) Cg[(b*N2 + r)*N2 + j] = B[j];
    }
}

// ---- variant lds (round-1 RowBlocks::mul) -----------------------------------------------------------------------------------
template <int N2>
__global__ __launch_bounds__(64) void k_lds(long long nb, int reps, const double* __restrict__ Ag, const double* __restrict__ Bg, double* __restrict__ Cg) {
    __shared__ double sB[4][N2*N2];
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    const long long blk = (long long)blockIdx.x*4 + t;
    const bool act = blk < nb && r < N2;
    const long long b = blk < nb ? blk : nb - 1;
    const int rr = r < N2 ? r : 0;
    double A[N2], B[N2], C[N2];
#pragma unroll
    for (int j = 0; j < N2; j++) { A[j] = Ag[(b*N2 + rr)*N2 + j]; B[j] = Bg[(b*N2 + rr)*N2 + j]; }
    for (int it = 0; it < reps; it++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (r < N2) {
#pragma unroll
            for (int j = 0; j < N2; j++) sB[t][r*N2 + j] = B[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the stores above have landed
#pragma unroll
        for (int j = 0; j < N2; j++) C[j] = 0.0;
#pragma unroll
        for (int m = 0; m < N2; m++) {
            const double am = A[m];
#pragma unroll
            for (int j = 0; j < N2; j++) C[j] += am*sB[t][m*N2 + j];
        }
#pragma unroll
        for (int j = 0; j < N2; j++) B[j] = C[j];
    }
    if (act) {
#pragma unroll
        for (int j = 0; j < N2; j++) Cg[(b*N2 + r)*N2 + j] = B[j];
    }
}

// ---- variant mfma: one 16x16 (zero-padded) product per wavefront and step ------------------------------------------------------
// v_mfma_f64_16x16x4_f64: A operand lane l holds A[i = l&15][k = l>>4], B operand B[k = l>>4][j = l&15];
// C/D 4 doubles per lane: col = l&15, row = (l>>4) + 4*reg   (cdna_hip_programming.md section 3)
template <int N2>
__global__ __launch_bounds__(64) void k_mfma(long long nb, int reps, const double* __restrict__ Ag, const double* __restrict__ Bg, double* __restrict__ Cg) {
    __shared__ double sB[16*16];
    const int l = threadIdx.x, i = l&15, kk = l>>4;
    const long long b = blockIdx.x;
    constexpr int KS = (N2 + 3)/4;
    double a[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) { const int k = 4*s + kk; a[s] = (i < N2 && k < N2) ? Ag[(b*N2 + i)*N2 + k] : 0.0; }
    for (int x = l; x < 256; x += 64) { const int rr = x/16, cc = x%16; sB[x] = (rr < N2 && cc < N2) ? Bg[(b*N2 + rr)*N2 + cc] : 0.0; }
    __syncthreads();
    v4d acc;
    for (int it = 0; it < reps; it++) {
        acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const double bv = sB[(4*s + kk)*16 + i];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bv, acc, 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; g++) sB[(kk + 4*g)*16 + i] = acc[g];          // the result becomes the next right-hand operand
        __syncthreads();
    }
    for (int x = l; x < N2*N2; x += 64) Cg[b*N2*N2 + x] = sB[(x/N2)*16 + x%N2];
}

template <int N2>
void run(long long nb, int reps) {
    const size_t cnt = (size_t)nb*N2*N2;
    std::vector<double> A(cnt), B(cnt), C(cnt), ref(cnt);
    srand(7);
    for (size_t x = 0; x < cnt; x++) { A[x] = (rand()/(double)RAND_MAX - 0.5)*0.5; B[x] = rand()/(double)RAND_MAX - 0.5; }
    const long long ncheck = 64;
    for (long long b = 0; b < ncheck; b++) {                 // CPU reference of the chained product on a few blocks
        std::vector<double> cur(B.begin() + b*N2*N2, B.begin() + (b + 1)*N2*N2), nxt(N2*N2);
        for (int it = 0; it < reps; it++) {
            for (int r = 0; r < N2; r++) for (int j = 0; j < N2; j++) { double s = 0.0; for (int m = 0; m < N2; m++) s = std::fma(A[(b*N2 + r)*N2 + m], cur[m*N2 + j], s); nxt[r*N2 + j] = s; }
            cur = nxt;
        }
        std::copy(cur.begin(), cur.end(), ref.begin() + b*N2*N2);
    }
    double *dA, *dB, *dC;
    CK(hipMalloc(&dA, cnt*8)); CK(hipMalloc(&dB, cnt*8)); CK(hipMalloc(&dC, cnt*8));
    CK(hipMemcpy(dA, A.data(), cnt*8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), cnt*8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"lds ", "dpp ", "mfma"};
    for (int v = 0; v < 3; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(dC, 0, cnt*8));
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL((k_lds<N2>), dim3((unsigned)((nb + 3)/4)), dim3(64), 0, 0, nb, reps, dA, dB, dC);
            if (v == 1) hipLaunchKernelGGL((k_dpp<N2>), dim3((unsigned)((nb + 3)/4)), dim3(64), 0, 0, nb, reps, dA, dB, dC);
            if (v == 2) hipLaunchKernelGGL((k_mfma<N2>), dim3((unsigned)nb), dim3(64), 0, 0, nb, reps, dA, dB, dC);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        CK(hipMemcpy(C.data(), dC, cnt*8, hipMemcpyDeviceToHost));
        double err = 0.0, nrm = 0.0;
        for (size_t x = 0; x < (size_t)ncheck*N2*N2; x++) { err += (C[x] - ref[x])*(C[x] - ref[x]); nrm += ref[x]*ref[x]; }
        const double useful = 2.0*N2*N2*N2*(double)nb*reps;
        printf("N2=%2d %s  %9.3f ms  %8.2f useful TFLOP/s  (%.0f products, %d chained)  rel err %.1e\n", N2, names[v], best, useful/best/1e9,
               (double)nb*reps, reps, std::sqrt(err/nrm));
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main() {
    printf("batched FP64 block products C <- A.C, 64 chained products per block (FP64 vector / matrix peak of MI355X: 78.6 TFLOP/s)\n");
    run<9>(103680, 64);
    run<16>(65536, 64);
    return 0;
}
