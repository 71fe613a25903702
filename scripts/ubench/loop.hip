// Micro-benchmark of the barrier-free "resident activation in LDS x streamed weight" loop (fc1-style).
// variants: bit0 = no global weight loads (registers reused), bit1 = no LDS reads, NFW = weight frags per k-step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int NFW, int MF, int D, int VAR, int NW, int LAY>
__global__ __launch_bounds__(NW * 64) void k(const uint4* __restrict__ W, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lr = lane & 15;
    for (int i = tid; i < 64 * 1024 / 4; i += NW * 64) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    constexpr int KS = 16;
    f32x4 acc[NFW][MF];
    for (int a = 0; a < NFW; ++a) for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    const size_t NFR = (size_t)NW * NFW;                 // fragments per k-step of this workgroup
    const uint4* Wq = W + (size_t)(blockIdx.x % 8) * NFR * KS * 64 + lane;
    auto widx = [&](int nf, int ks) -> size_t {
        return LAY ? ((size_t)ks * NFR + wave * NFW + nf) * 64 : (((size_t)wave * NFW + nf) * KS + ks) * 64;
    };
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint4 wq[D][NFW];
#pragma unroll
        for (int p = 0; p < D - 1; ++p)
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) wq[p][nf] = Wq[widx(nf, p)];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (!(VAR & 1) && s + D - 1 < KS)
#pragma unroll
                for (int nf = 0; nf < NFW; ++nf) wq[(s + D - 1) % D][nf] = Wq[widx(nf, s + D - 1)];
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 xf[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                if (VAR & 2) xf[mf] = __builtin_bit_cast(bf16x8, wq[0][0]);
                else xf[mf] = *reinterpret_cast<const bf16x8*>(smem + (mf * 16 + lr) * 1024 + (((4 * s + g) ^ lr) << 4));
            }
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, wq[(VAR & 1) ? (s % (D - 1)) : (s % D)][nf]);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = MFMA16(wf, xf[mf], acc[nf][mf]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    f32x4 s = {0, 0, 0, 0};
    for (int a = 0; a < NFW; ++a) for (int b = 0; b < MF; ++b) s += acc[a][b];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NFW, int MF, int D, int VAR, int NW, int LAY>
void run(const char* name, const uint4* W, float* out, long long* cyc, int grid) {
    const int iters = 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NFW, MF, D, VAR, NW, LAY>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NFW, MF, D, VAR, NW, LAY>), dim3(grid), dim3(NW * 64), 65536, 0, W, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4096];
    hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid; ++i) sum += (double)h[i];
    const double per_kstep = sum / grid / iters / 16.0;
    printf("%-44s grid %4d: %7.1f cycles/k-step  (MFMA floor for 2 waves/SIMD: %d)\n", name, grid, per_kstep, NFW * MF * 16 * (NW / 4));
}

int main() {
    uint4* W; float* out; long long* cyc;
    hipMalloc(&W, 64 << 20); hipMemset(W, 0x3c, 64 << 20);
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&cyc, 4096 * 8);
    for (int grid : {1, 256, 512}) {
        run<2, 4, 4, 0, 8, 0>("fc1-like NFW=2 MF=4 D=4", W, out, cyc, grid);
        run<2, 4, 8, 0, 8, 0>("fc1-like NFW=2 MF=4 D=8", W, out, cyc, grid);
        run<2, 4, 4, 1, 8, 0>("  no weight loads", W, out, cyc, grid);
        run<2, 4, 4, 2, 8, 0>("  no LDS reads", W, out, cyc, grid);
        run<2, 4, 4, 3, 8, 0>("  neither (MFMA only)", W, out, cyc, grid);
        run<4, 4, 3, 0, 8, 0>("fc2-like NFW=4 MF=4 D=3", W, out, cyc, grid);
        run<4, 4, 3, 0, 8, 1>("fc2-like NFW=4 MF=4 D=3 [ks][n] layout", W, out, cyc, grid);
        run<4, 4, 4, 0, 8, 1>("fc2-like NFW=4 MF=4 D=4 [ks][n] layout", W, out, cyc, grid);
        run<2, 4, 4, 0, 8, 1>("fc1-like NFW=2 MF=4 D=4 [ks][n] layout", W, out, cyc, grid);
        run<3, 4, 4, 0, 8, 1>("qkv-like NFW=3 MF=4 D=4 [ks][n] layout", W, out, cyc, grid);
        run<4, 8, 3, 0, 8, 1>("gemm128-like NFW=4 MF=8 D=3 [ks][n] layout", W, out, cyc, grid);
        run<4, 8, 3, 0, 8, 0>("gemm128-like NFW=4 MF=8 D=3", W, out, cyc, grid);
        run<3, 4, 4, 0, 8, 0>("qkv-like NFW=3 MF=4 D=4", W, out, cyc, grid);
        run<4, 4, 4, 0, 4, 0>("4 waves NFW=4 MF=4 D=4", W, out, cyc, grid);
        run<8, 4, 3, 0, 4, 0>("4 waves NFW=8 MF=4 D=3", W, out, cyc, grid);
    }
    return 0;
}
