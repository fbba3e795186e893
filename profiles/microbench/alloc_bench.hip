#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipSetDevice(0);
    hipFree(0);
    const size_t GB = 1ull << 30;
    for (int round = 0; round < 3; round++) {
        std::vector<void *> p;
        double t0 = now();
        for (int i = 0; i < 6; i++) { void *q; hipMalloc(&q, 2 * GB); p.push_back(q); }
        double t1 = now();
        for (auto q : p) hipMemsetAsync(q, 0, 2 * GB, 0);
        hipDeviceSynchronize();
        double t2 = now();
        for (auto q : p) hipFree(q);
        double t3 = now();
        printf("round %d: 6 x hipMalloc(2 GiB) %.2f ms, first touch (memset) %.2f ms, 6 x hipFree %.2f ms\n", round, t1 - t0, t2 - t1, t3 - t2);
    }
    hipMemPool_t pool;
    hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t thr = UINT64_MAX;
    hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int round = 0; round < 3; round++) {
        std::vector<void *> p;
        double t0 = now();
        for (int i = 0; i < 6; i++) { void *q; hipMallocAsync(&q, 2 * GB, 0); p.push_back(q); }
        hipDeviceSynchronize();
        double t1 = now();
        for (auto q : p) hipMemsetAsync(q, 0, 2 * GB, 0);
        hipDeviceSynchronize();
        double t2 = now();
        for (auto q : p) hipFreeAsync(q, 0);
        hipDeviceSynchronize();
        double t3 = now();
        printf("pool round %d: 6 x hipMallocAsync(2 GiB) %.2f ms, memset %.2f ms, 6 x hipFreeAsync %.2f ms\n", round, t1 - t0, t2 - t1, t3 - t2);
    }
    { void *q; double t0 = now(); hipMalloc(&q, 12 * GB); double t1 = now(); hipFree(q); double t2 = now(); printf("one hipMalloc(12 GiB) %.2f ms, free %.2f ms\n", t1 - t0, t2 - t1); }
    return 0;
}
