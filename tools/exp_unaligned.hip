#include <hip/hip_runtime.h>
#include <cstdio>
typedef double nt_pair __attribute__((ext_vector_type(2)));
__global__ void k(double* out, int off) {
    nt_pair v; v.x = 1.0 + threadIdx.x; v.y = -1.0 - threadIdx.x;
    __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(out + off + 2 * threadIdx.x));
}
__global__ void k2(const double* in, double* out, int off) {
    nt_pair v = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(in + off + 2 * threadIdx.x));
    out[2 * threadIdx.x] = v.x; out[2 * threadIdx.x + 1] = v.y;
}
int main() {
    double *d, *e; hipMalloc(&d, 4096); hipMalloc(&e, 4096); hipMemset(d, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1);       // 8-byte aligned, not 16
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, e, 1);
    hipError_t err = hipDeviceSynchronize();
    double h[140]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (h[1 + 2 * i] != 1.0 + i || h[2 + 2 * i] != -1.0 - i) ++bad;
    printf("err=%d bad=%d first=%g %g %g\n", (int)err, bad, h[0], h[1], h[2]);
    return 0;
}
