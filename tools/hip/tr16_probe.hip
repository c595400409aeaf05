// GPU-box probe: (1) what ds_read_b64_tr_b16 returns for a given set of lane addresses; (2) the operand / accumulator lane maps of
// v_mfma_f32_32x32x16_bf16 -- checked against a scalar reference with an asymmetric operand.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__global__ void tr_probe(u16* out, int row_stride) {
    // LDS image: element e holds the value e (as u16); lane a of a 16-lane group addresses row (a >> 2), cols 4 (a & 3) .. + 3
    // of a 4 x 16 block with row stride `row_stride` elements; group gq = lane >> 4 uses rows 4 gq .. 4 gq + 3
    __shared__ __attribute__((aligned(16))) u16 img[4096];
    for (int e = threadIdx.x; e < 4096; e += 64) img[e] = static_cast<u16>(e);
    __syncthreads();
    const int lane = threadIdx.x, a = lane & 15, gq = lane >> 4;
    const u16* p = img + (4 * gq + (a >> 2)) * row_stride + 4 * (a & 3);
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
    unsigned long long raw = __builtin_bit_cast(unsigned long long, v);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = static_cast<u16>(raw >> (16 * j));
}

__global__ void mfma_probe(const float* A, const float* B, float* C) {
    // A [32][16] row-major, B [16][32] row-major (fp32 holding bf16-exact values), C [32][32]
    const int l = threadIdx.x, i = l & 31, hi = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = static_cast<__bf16>(A[i * 16 + 8 * hi + j]);        // A[i][k = 8 hi + j]
        b[j] = static_cast<__bf16>(B[(8 * hi + j) * 32 + i]);      // B[k = 8 hi + j][n = i]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = c[r];
}

int main() {
    u16* d; hipMalloc(&d, 256 * 2);
    for (int stride : {16, 80}) {
        tr_probe<<<1, 64>>>(d, stride);
        std::vector<u16> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("row_stride %d: lane -> 4 values (element index = row * stride + col)\n", stride);
        for (int lane = 0; lane < 64; ++lane) {
            printf("  lane %2d:", lane);
            for (int j = 0; j < 4; ++j) printf("  %4d=(r%d,c%d)", h[lane * 4 + j], h[lane * 4 + j] / stride, h[lane * 4 + j] % stride);
            printf("\n");
        }
        // hypothesis: lane (gq, i) elem j = block row (4 gq + j), col i
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j)
            bad += h[lane * 4 + j] != (4 * (lane >> 4) + j) * stride + (lane & 15);
        printf("  hypothesis elem j = (row 4 gq + j, col lane & 15): %s\n", bad ? "NO" : "yes");
    }
    std::vector<float> A(512), B(512), C(1024), R(1024, 0.f);
    for (int x = 0; x < 512; ++x) { A[x] = static_cast<float>((x * 7) % 13 - 6); B[x] = static_cast<float>((x * 5) % 11 - 5); }
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
    float *dA, *dB, *dC; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int x = 0; x < 1024; ++x) bad += C[x] != R[x];
    printf("mfma_f32_32x32x16_bf16 lane maps (A[i][8 hi + j], B[8 hi + j][n], C row (r&3)+8(r>>2)+4hi col l&31): %s (%d mismatches)\n",
           bad ? "WRONG" : "ok", bad);
    return 0;
}
