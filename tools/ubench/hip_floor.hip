// hip_floor -- what ANY HIP process pays before and after its own work on this box: runtime + driver initialisation, one stream,
// one allocation, one (empty) kernel launch from this code object, teardown.  bench.py's e2e_cli leg runs it next to gps_test so
// that the front end's wall clock can be read against the floor it cannot go below.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_nop(int* p) { if (p && threadIdx.x == 1234567) *p = 0; }
int main() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { printf("no device\n"); return 2; }
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* d = nullptr;
    (void)hipMalloc((void**)&d, 1 << 20);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d);
    (void)hipStreamSynchronize(s);
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
    printf("ok\n");
    return 0;
}
