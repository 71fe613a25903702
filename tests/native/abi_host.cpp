// A native caller of the C ABI (include/syn_hip.h) with no Python and no torch in the process: what a C / C++ host of the reference's path would link.
// Draws 4096 N(0,1) values with syn_randn, converts 1000 axis-angle rotations to the 6D form and back, prints checksums the pytest side compares with
// the same calls made through ctypes.   hipcc -I include tests/native/abi_host.cpp -L syntalker_amd/csrc -lsyn_hip -o abi_host
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "syn_hip.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, syn_last_error()); return 1; } } while (0)

int main() {
    if (syn_version() != SYN_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", syn_version(), SYN_ABI_VERSION); return 1; }
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return 2;
    const int n = 4096, joints = 1000;
    float *noise, *aa, *d6, *back;
    hipMalloc(&noise, n * 4); hipMalloc(&aa, joints * 12); hipMalloc(&d6, joints * 24); hipMalloc(&back, joints * 12);
    CHECK(syn_randn(noise, n, 1234ull, 999ull, 8, s));
    std::vector<float> h(joints * 3);
    for (int i = 0; i < joints * 3; ++i) h[i] = 0.001f * (float)((i * 7919) % 2001 - 1000);          // angles up to ~1.7 rad
    hipMemcpyAsync(aa, h.data(), joints * 12, hipMemcpyHostToDevice, s);
    CHECK(syn_axis_angle_to_rot6d(aa, joints, d6, s));
    CHECK(syn_rot6d_to_axis_angle(d6, joints, back, s));
    std::vector<float> hn(n), hb(joints * 3), h6(joints * 6);
    hipMemcpyAsync(hn.data(), noise, n * 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hb.data(), back, joints * 12, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h6.data(), d6, joints * 24, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return 3;
    double sum = 0, sq = 0, s6 = 0, err = 0;
    for (float v : hn) { sum += v; sq += (double)v * v; }
    for (float v : h6) s6 += v;
    for (int i = 0; i < joints * 3; ++i) { double e = hb[i] - h[i]; if (e < 0) e = -e; if (e > err) err = e; }
    printf("randn_sum %.9e randn_sq %.9e rot6d_sum %.9e roundtrip_max_err %.3e\n", sum, sq, s6, err);
    // an error path: the ABI reports through its status + syn_last_error, never by aborting
    if (syn_randn(noise, 3, 0, 0, 0, s) == 0) { fprintf(stderr, "syn_randn accepted n = 3\n"); return 4; }
    printf("error_text %s\n", syn_last_error());
    return 0;
}
