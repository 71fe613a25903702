// How fast can 8 waves of a CU stream 1-KiB weight fragments from L2/MALL when ALL workgroups walk the same
// addresses in the same order (the k_stack access pattern)?  Ring of D fragments-groups in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int NF, int D, int MF, int SAME>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ W, float* out, long long* cyc, int ksteps, size_t wg_stride) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[NF][MF > 0 ? MF : 1];
    for (int a = 0; a < NF; ++a) for (int b = 0; b < (MF > 0 ? MF : 1); ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    // fragment (s, wave, nf) at ((s*8 + wave)*NF + nf)*64 + lane  : a k-step of the whole workgroup is contiguous
    const uint4* Wq = W + (SAME ? 0 : (size_t)blockIdx.x * wg_stride) + (size_t)wave * NF * 64 + lane;
    bf16x8 xf = {1, 1, 1, 1, 1, 1, 1, 1};
    uint4 ring[D][NF];
    uint4 sink = make_uint4(0, 0, 0, 0);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int p = 0; p < D - 1; ++p)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) ring[p][nf] = Wq[((size_t)p * 8 * NF + nf) * 64];
    for (int s0 = 0; s0 < ksteps; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = s0 + u;
            if (s + D - 1 < ksteps)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) ring[(u + D - 1) % D][nf] = Wq[((size_t)(s + D - 1) * 8 * NF + nf) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if (MF > 0) {
                    const bf16x8 wf = __builtin_bit_cast(bf16x8, ring[u][nf]);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = MFMA16(wf, xf, acc[nf][mf]);
                } else {
                    sink.x ^= ring[u][nf].x; sink.y ^= ring[u][nf].w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = (float)(sink.x ^ sink.y);
    for (int a = 0; a < NF; ++a) for (int b = 0; b < (MF > 0 ? MF : 1); ++b) r += acc[a][b][0] + acc[a][b][3];
    out[(size_t)blockIdx.x * 512 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NF, int D, int MF, int SAME>
void run(const char* name, const uint4* W, float* out, long long* cyc, int grid, size_t bytes) {
    const int ksteps = 960;                                 // multiple of every D used; 960 * 8 * NF KiB per workgroup
    const size_t per_wg = (size_t)ksteps * 8 * NF * 64;     // uint4 elements
    if (!SAME && per_wg * 16 * grid > bytes) { printf("%-40s skipped (buffer)\n", name); return; }
    hipMemset(cyc, 0, 8 * 4096);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NF, D, MF, SAME>), dim3(grid), dim3(512), 0, 0, W, out, cyc, ksteps, per_wg);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid; ++i) sum += (double)h[i];
    const double cycles = sum / grid;
    printf("%-40s grid %4d: %6.1f B/clk/CU  (%7.1f cycles per k-step, %d KiB in flight per CU)\n", name, grid,
           (double)ksteps * 8 * NF * 1024 / cycles, cycles / ksteps, (D - 1) * 8 * NF);
}

int main() {
    const size_t bytes = 8ull << 30;
    uint4* W; float* out; long long* cyc;
    if (hipMalloc(&W, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(W, 0x3c, bytes);
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&cyc, 4096 * 8);
    for (int grid : {256, 512}) {
        printf("--- all workgroups walk the SAME %s weights (k_stack pattern), pure streaming\n", "cold");
        run<3, 4, 0, 1>("NF=3 D=4  no MFMA", W, out, cyc, grid, bytes);
        run<3, 8, 0, 1>("NF=3 D=8  no MFMA", W, out, cyc, grid, bytes);
        run<3, 16, 0, 1>("NF=3 D=16 no MFMA", W, out, cyc, grid, bytes);
        run<2, 6, 0, 1>("NF=2 D=6  no MFMA", W, out, cyc, grid, bytes);
        run<4, 3, 0, 1>("NF=4 D=3  no MFMA", W, out, cyc, grid, bytes);
        run<4, 8, 0, 1>("NF=4 D=8  no MFMA", W, out, cyc, grid, bytes);
        printf("--- same, with 4 MFMAs per fragment (MT=64 pacing)\n");
        run<3, 4, 4, 1>("NF=3 D=4  MF=4", W, out, cyc, grid, bytes);
        run<3, 8, 4, 1>("NF=3 D=8  MF=4", W, out, cyc, grid, bytes);
        run<2, 6, 4, 1>("NF=2 D=6  MF=4", W, out, cyc, grid, bytes);
        run<4, 3, 4, 1>("NF=4 D=3  MF=4", W, out, cyc, grid, bytes);
        run<4, 6, 4, 1>("NF=4 D=6  MF=4", W, out, cyc, grid, bytes);
        printf("--- every workgroup its OWN addresses (HBM streaming)\n");
        run<3, 4, 0, 0>("NF=3 D=4  no MFMA own", W, out, cyc, grid, bytes);
        run<3, 8, 0, 0>("NF=3 D=8  no MFMA own", W, out, cyc, grid, bytes);
    }
    return 0;
}
