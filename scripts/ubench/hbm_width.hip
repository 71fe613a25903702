// How fast does the chip stream a 234 MB fp32 tensor from HBM with 4-byte loads per lane (a wave instruction = 256 contiguous bytes: the access of a
// "thread = channel" kernel over [position][64] rows) against 16-byte loads per lane (1 KB per wave instruction)?  And the same for stores.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_width.hip -o /tmp/hbm_width && /tmp/hbm_width
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int VEC, int D>       // VEC floats per lane and load, D loads in flight per thread
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ x, size_t n_vec, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i + (D - 1) * stride < n_vec; i += D * stride) {
        float v[D][VEC];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (VEC == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(x + (i + d * stride) * 4); v[d][0] = t[0]; v[d][1 % VEC] = t[1]; v[d][2 % VEC] = t[2]; v[d][3 % VEC] = t[3]; }
            else v[d][0] = x[i + d * stride];
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += v[d][e];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_write(float* __restrict__ y, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
        if (VEC == 4) *reinterpret_cast<f32x4*>(y + i * 4) = f32x4{1.f, 2.f, 3.f, 4.f};
        else y[i] = 1.f;
    }
}

template <typename F>
float timed(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main() {
    const size_t n = (size_t)32 * 14331 * 64 * 2;          // 234 MB
    float *x, *out; hipMalloc(&x, n * 4); hipMalloc(&out, 64); hipMemset(x, 0, n * 4);
    for (int grid : {1024, 2048, 4096, 8192}) {
        const float r1 = timed([&] { hipLaunchKernelGGL((k_read<1, 4>), dim3(grid), dim3(256), 0, 0, x, n, out); });
        const float r1d = timed([&] { hipLaunchKernelGGL((k_read<1, 16>), dim3(grid), dim3(256), 0, 0, x, n, out); });
        const float r4 = timed([&] { hipLaunchKernelGGL((k_read<4, 4>), dim3(grid), dim3(256), 0, 0, x, n / 4, out); });
        const float w1 = timed([&] { hipLaunchKernelGGL((k_write<1>), dim3(grid), dim3(256), 0, 0, x, n); });
        const float w4 = timed([&] { hipLaunchKernelGGL((k_write<4>), dim3(grid), dim3(256), 0, 0, x, n / 4); });
        printf("grid %5d: read 4 B/lane x4 in flight %6.1f us (%.2f TB/s)  x16 in flight %6.1f us (%.2f TB/s)  16 B/lane x4 %6.1f us (%.2f TB/s) | write 4 B/lane %6.1f us (%.2f TB/s)  16 B/lane %6.1f us (%.2f TB/s)\n",
               grid, r1 * 1e3, n * 4 / r1 / 1e9, r1d * 1e3, n * 4 / r1d / 1e9, r4 * 1e3, n * 4 / r4 / 1e9, w1 * 1e3, n * 4 / w1 / 1e9, w4 * 1e3, n * 4 / w4 / 1e9);
    }
    return 0;
}
