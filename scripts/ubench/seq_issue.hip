// Micro-benchmark of k_seq's issue budget: ONE wave per SIMD, 4 waves per CU, a chunk = 16 v_mfma_f32_32x32x16_bf16 with the fillers
// the step kernel carries between them.  What does each filler cost when it is added to / removed from the stream?
//   NACC  independent accumulators (2 = pair512, 12 = wide)
//   WAIT  s_waitcnt lgkmcnt per N MFMAs (0: no LDS reads at all; 1: one wait per MFMA, window 4; 2 / 4: one wait per 2 / 4 MFMAs,
//         reads issued in groups, window 2 x group)
//   DMA   0 none; 1 one global_load_lds per 4 MFMAs with s_mov m0 + s_nop each; 2 m0 written once per chunk; 3 all four at the chunk start
//   BAR   s_waitcnt vmcnt + s_barrier per chunk
//   SALU  0: the ring's slot / tape-position arithmetic as the step kernel does it (~17 scalar instructions per chunk); -1: fixed
//         addresses (racy, timing only); n > 0: n extra scalar instructions
//   VALU  v_fma_f32 fillers per MFMA (independent chains; PK = 1: v_pk_fma_f32 on register pairs instead);  TRN  v_exp_f32 fillers per MFMA
// Prints shader cycles per MFMA (median over workgroups) and the wall-clock rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kLds = 160 * 1024, kSlots = 5, kChunk = 16 * 1024;

template <int OFF>
__device__ __forceinline__ void dma(const char* src, unsigned dst, unsigned lane16, bool set_m0) {
    if (set_m0) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%c3" :: "v"(lane16), "s"(dst), "s"(src), "i"(OFF) : "memory");
    else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" :: "v"(lane16), "s"(src), "i"(OFF) : "memory");
}

template <int NACC, int WAIT, int DMA, int BAR, int SALU, int VALU, int TRN, int SGAP = 0, int PK = 0>
__global__ __launch_bounds__(256) void k(const char* __restrict__ tape, unsigned n_chunks, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kLds / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (float)((i * 2654435761u) >> 24);
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    bf16x8 B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) B[i][e] = (__bf16)(0.01f * (float)((lane + 3 * e + i) & 15));
    float fv[8], fk0 = 1.0001f, fk1 = 0.25f;
    unsigned sjunk = 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 fp[8], fpk = {1.0001f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) fp[i] = f32x2{0.5f + 0.001f * (float)(lane + i), 0.25f};
    asm volatile("" : "+v"(fpk));
    asm volatile("" : "+v"(fk0), "+v"(fk1), "+s"(sjunk));
#pragma unroll
    for (int i = 0; i < 8; ++i) fv[i] = 0.5f + 0.001f * (float)(lane + i);
    constexpr int G = WAIT == 0 ? 1 : WAIT, WIN = WAIT <= 1 ? 4 : 2 * WAIT;
    bf16x8 win[WIN];
    const unsigned lane16 = (unsigned)lane * 16;
#pragma unroll
    for (int i = 0; i < WIN; ++i) win[i] = *reinterpret_cast<const bf16x8*>(lds + lane16 + i * 1024);
    unsigned slot = 0, pos = 0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        // chunk boundary
        unsigned s1, cur, nxt, dst, pos1;
        const char* src;
        if (SALU >= 0) {
            s1 = slot + 1 == kSlots ? 0u : slot + 1;
            const unsigned sd = slot == 0 ? kSlots - 1 : slot - 1;
            cur = lane16 + slot * kChunk; nxt = lane16 + s1 * kChunk;
            dst = sd * kChunk + wave * 4096;
            src = tape + ((size_t)pos * 16 + wave * 4) * 1024;
            pos1 = pos + 1 == n_chunks ? 0u : pos + 1;
        } else {
            s1 = 0; cur = lane16; nxt = lane16 + kChunk; dst = 4 * kChunk + wave * 4096; src = tape + wave * 4096; pos1 = 0;
        }
        asm volatile("" : "+s"(dst), "+s"(src), "+s"(pos1));
        if (SALU > 0) {
            unsigned junk = pos1;
#pragma unroll
            for (int i = 0; i < SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(junk));
            asm volatile("" :: "s"(junk));
        }
        if (BAR) {
            if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        FENCE();
        if (DMA == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst));
        if (DMA == 3) { dma<0>(src, dst, lane16, true); dma<1024>(src, dst, lane16, false); dma<2048>(src, dst, lane16, false); dma<3072>(src, dst, lane16, false); }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if ((DMA == 1 || DMA == 2) && (f & 3) == 2) {
                switch (f >> 2) {
                    case 0: dma<0>(src, dst, lane16, DMA == 1); break;
                    case 1: dma<1024>(src, dst, lane16, DMA == 1); break;
                    case 2: dma<2048>(src, dst, lane16, DMA == 1); break;
                    default: dma<3072>(src, dst, lane16, DMA == 1); break;
                }
            }
            bf16x8 a;
            if (WAIT == 0) a = B[(f + 1) & 3];
            else {
                a = win[f % WIN];
                if (G == 1) {
                    const int g = f + WIN;
                    win[f % WIN] = *reinterpret_cast<const bf16x8*>(lds + (g < 16 ? cur + g * 1024 : nxt + (g - 16) * 1024));
                }
            }
            acc[f % NACC] = MFMA32(a, B[f & 3], acc[f % NACC]);
            if (WAIT > 1 && (f % G) == G - 1) {
                // group refill: the G slots just consumed take fragments f + 1 + G .. (issued together -> one s_waitcnt per group)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int g = f + 1 - G + j + WIN;
                    win[(f + 1 - G + j) % WIN] = *reinterpret_cast<const bf16x8*>(lds + (g < 16 ? cur + g * 1024 : nxt + (g - 16) * 1024));
                }
            }
            // (asm volatile: plain expressions are re-associated and sunk across the scheduling fences)
#pragma unroll
            for (int i = 0; i < VALU; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(fp[(f + i) & 7]) : "v"(fpk));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[(f + i) & 7]) : "v"(fk0), "v"(fk1));
            }
#pragma unroll
            for (int i = 0; i < TRN; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(fv[(f + i + 4) & 7]));
#pragma unroll
            for (int i = 0; i < SGAP; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sjunk));
            FENCE();
        }
        slot = s1; pos = pos1;
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[a][v];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fv[i];
    s += (float)sjunk;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fp[i][0] + fp[i][1];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int WAIT, int DMA, int BAR, int SALU, int VALU, int TRN, int SGAP = 0, int PK = 0>
void run(const char* tape, unsigned n_chunks, float* out, long long* cyc) {
    auto kern = k<NACC, WAIT, DMA, BAR, SALU, VALU, TRN, SGAP, PK>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    const int grid = 256, iters = 2000;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, 0, tape, n_chunks, out, cyc, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, 0, tape, n_chunks, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per = (double)c[grid / 2] / (iters * 16.0);
    const double tf = (double)grid * 4 * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("NACC %2d WAIT %d DMA %d BAR %d SALU %2d VALU %d TRN %d SGAP %d PK %d : %6.2f cycles/MFMA  (%.0f TFLOP/s wall, %.2f GHz eff)\n", NACC, WAIT, DMA, BAR, SALU, VALU,
           TRN, SGAP, PK, per, tf, (double)c[grid / 2] / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const unsigned n_chunks = 2256;
    char* tape; float* out; long long* cyc;
    hipMalloc(&tape, (size_t)n_chunks * kChunk);
    hipMemset(tape, 0x3c, (size_t)n_chunks * kChunk);
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    //   NACC WAIT DMA BAR SALU VALU TRN SGAP PK
    run<12, 1, 1, 1, -1, 0, 0>(tape, n_chunks, out, cyc);         // reads + DMA + barrier, fixed addresses: the floor of the stream
    run<12, 1, 1, 1, -1, 2, 0>(tape, n_chunks, out, cyc);         // + v_fma fillers
    run<12, 1, 1, 1, -1, 4, 0>(tape, n_chunks, out, cyc);
    run<12, 1, 1, 1, -1, 1, 0, 0, 1>(tape, n_chunks, out, cyc);   // + v_pk_fma fillers (two values each)
    run<12, 1, 1, 1, -1, 2, 0, 0, 1>(tape, n_chunks, out, cyc);
    run<12, 1, 1, 1, -1, 3, 0, 0, 1>(tape, n_chunks, out, cyc);
    run<12, 1, 1, 1, -1, 4, 0, 0, 1>(tape, n_chunks, out, cyc);
    run<12, 1, 1, 1, -1, 3, 1>(tape, n_chunks, out, cyc);         // half a GELU per gap, scalar
    run<12, 1, 1, 1, -1, 2, 1, 0, 1>(tape, n_chunks, out, cyc);   // the same work packed (2 pk + 1 transcendental per element pair... per gap)
    run<12, 1, 1, 1, -1, 3, 2, 0, 1>(tape, n_chunks, out, cyc);   // a whole GELU per gap, packed: 6 pk + 4 trans per 2 elements = 3 + 2 per gap
    run<12, 1, 1, 1, -1, 6, 2>(tape, n_chunks, out, cyc);         // a whole GELU per gap, scalar
    return 0;
}
