// Census (GPU box): where does the hardware place the workgroups of a (H, B) grid?  Each workgroup records its XCC id, SE /
// CU id and start time; the host prints workgroups per CU and the spread.  Usage: census <grid_x> <grid_y> <threads> <lds_bytes> <spin_us>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void census(unsigned* out, long long* t0, int spin_cycles) {
    extern __shared__ float smem[];
    const long long start = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int w = blockIdx.y * gridDim.x + blockIdx.x;
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
        t0[w] = start;
    }
    smem[threadIdx.x] = threadIdx.x;
    const long long c0 = clock64();
    while (clock64() - c0 < spin_cycles) __builtin_amdgcn_s_sleep(8);
    if (smem[threadIdx.x] < 0) out[0] = 0;
}

int main(int argc, char** argv) {
    const int gx = atoi(argv[1]), gy = atoi(argv[2]), threads = atoi(argv[3]), lds = atoi(argv[4]), spin_us = atoi(argv[5]);
    const int n = gx * gy;
    unsigned* d; long long* dt;
    hipMalloc(&d, sizeof(unsigned) * 2 * n); hipMalloc(&dt, sizeof(long long) * n);
    hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) {
        census<<<dim3(gx, gy), threads, lds>>>(d, dt, spin_us * 2000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * n); std::vector<long long> ht(n);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), dt, sizeof(long long) * n, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    long long tmin = ht[0], tmax = ht[0];
    for (int w = 0; w < n; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(w);
        if (ht[w] < tmin) tmin = ht[w];
        if (ht[w] > tmax) tmax = ht[w];
    }
    std::map<size_t, int> hist;
    for (auto& kv : per_cu) hist[kv.second.size()]++;
    printf("grid %dx%d threads %d lds %d: %zu distinct CUs; start spread %.2f us (100 MHz wall clock)\n", gx, gy, threads, lds,
           per_cu.size(), (tmax - tmin) / 100.0);
    for (auto& kv : hist) printf("  %d CUs hold %zu workgroups\n", kv.second, kv.first);
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 6) break;
        printf("  cu key %05x:", kv.first);
        for (int w : kv.second) printf(" %d", w);
        printf("\n");
    }
    return 0;
}
