// What a large device allocation costs on this machine, and which way of getting the memory is cheapest: the set-up of a
// fixed-base table (2.5 GB at w = 16, 17 GB at w = 19) is dominated by it.
// hipcc --offload-arch=gfx950 -O2 tools/micro/alloc_latency.hip -o gpurun_out/alloc_latency && gpurun_out/alloc_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k_fill(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{1, 2, 3, 4};
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e__)); (void)hipGetLastError(); } } while (0)
static void touch(void* d, size_t bytes, const char* what) {
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now_ms();
        hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, 0, (uint4*)d, bytes / 16);
        CK(hipDeviceSynchronize());
        double t1 = now_ms();
        printf("    %s fill pass %d: %.1f ms (%.0f GB/s)\n", what, rep, t1 - t0, bytes / (t1 - t0) / 1e6);
    }
}
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    size_t fre = 0, tot = 0;
    CK(hipMemGetInfo(&fre, &tot));
    printf("free %.1f GB of %.1f GB\n", fre / 1e9, tot / 1e9);
    const size_t sizes[] = {(size_t)256 << 20, (size_t)2560 << 20, (size_t)17 << 30, (size_t)34 << 30};
    for (size_t bytes : sizes) {
        printf("== %.2f GB\n", bytes / 1e9);
        for (int rep = 0; rep < 2; ++rep) {
            void* d = nullptr;
            double t0 = now_ms();
            CK(hipMalloc(&d, bytes));
            double t1 = now_ms();
            printf("  hipMalloc: %.1f ms\n", t1 - t0);
            if (d && rep == 0) touch(d, bytes, "hipMalloc");
            t0 = now_ms();
            CK(hipFree(d));
            t1 = now_ms();
            printf("  hipFree: %.1f ms\n", t1 - t0);
        }
        {   // stream-ordered allocator with a pool that keeps what it is given back
            hipMemPool_t pool;
            CK(hipDeviceGetDefaultMemPool(&pool, 0));
            uint64_t keep = UINT64_MAX;
            CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
            for (int rep = 0; rep < 3; ++rep) {
                void* d = nullptr;
                double t0 = now_ms();
                CK(hipMallocAsync(&d, bytes, 0));
                CK(hipStreamSynchronize(0));
                double t1 = now_ms();
                printf("  hipMallocAsync (rep %d): %.1f ms\n", rep, t1 - t0);
                if (d && rep == 0) touch(d, bytes, "async");
                t0 = now_ms();
                CK(hipFreeAsync(d, 0));
                CK(hipStreamSynchronize(0));
                t1 = now_ms();
                printf("  hipFreeAsync: %.1f ms\n", t1 - t0);
            }
            CK(hipMemPoolTrimTo(pool, 0));
        }
        {   // virtual memory management: reserve, create, map, set access
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            size_t sz = (bytes + gran - 1) / gran * gran;
            void* va = nullptr;
            double t0 = now_ms();
            CK(hipMemAddressReserve(&va, sz, 0, nullptr, 0));
            double t1 = now_ms();
            hipMemGenericAllocationHandle_t h{};
            CK(hipMemCreate(&h, sz, &prop, 0));
            double t2 = now_ms();
            CK(hipMemMap(va, sz, 0, h, 0));
            double t3 = now_ms();
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, sz, &acc, 1));
            double t4 = now_ms();
            printf("  VMM gran %zu: reserve %.1f create %.1f map %.1f access %.1f ms\n", gran, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
            if (va) touch(va, bytes, "vmm");
            t0 = now_ms();
            CK(hipMemUnmap(va, sz));
            CK(hipMemRelease(h));
            CK(hipMemAddressFree(va, sz));
            t1 = now_ms();
            printf("  VMM teardown %.1f ms\n", t1 - t0);
        }
    }
    // many threads allocating in parallel: does the driver serialise?
    return 0;
}
