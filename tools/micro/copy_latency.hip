// Latency of a small device-to-host / host-to-device copy behind a short kernel: pageable against pinned host memory.
// hipcc --offload-arch=gfx950 -O2 tools/micro/copy_latency.hip -o gpurun_out/copy_latency && gpurun_out/copy_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k_touch(unsigned* p) { p[threadIdx.x] += 1; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    void* pinned;
    hipHostMalloc(&pinned, 1 << 20, hipHostMallocDefault);
    void* pageable = malloc(1 << 20);
    const int reps = 2000;
    for (size_t bytes : {256, 4096, 65536, 524288}) {
        for (int mode = 0; mode < 4; ++mode) {            // 0: D2H pageable  1: D2H pinned (+memcpy)  2: H2D pageable  3: H2D pinned (+memcpy)
            for (int warm = 0; warm < 2; ++warm) {
                double t0 = now_us();
                for (int i = 0; i < reps; ++i) {
                    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d);
                    if (mode == 0) hipMemcpyAsync(pageable, d, bytes, hipMemcpyDeviceToHost, s);
                    if (mode == 1) hipMemcpyAsync(pinned, d, bytes, hipMemcpyDeviceToHost, s);
                    if (mode == 2) hipMemcpyAsync(d, pageable, bytes, hipMemcpyHostToDevice, s);
                    if (mode == 3) { memcpy(pinned, pageable, bytes); hipMemcpyAsync(d, pinned, bytes, hipMemcpyHostToDevice, s); }
                    hipStreamSynchronize(s);
                    if (mode == 1) memcpy(pageable, pinned, bytes);
                }
                double t1 = now_us();
                if (warm) printf("bytes=%zu mode=%d  %.1f us per (kernel + copy + sync)\n", bytes, mode, (t1 - t0) / reps);
            }
        }
    }
    // kernel + sync alone
    double t0 = now_us();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); }
    printf("kernel + sync alone: %.1f us\n", (now_us() - t0) / reps);
    return 0;
}
