// Probe (not product): pins, with exact small-integer data, the three hardware layouts csrc/fset.hip relies on:
//   1. v_mfma_f32_32x32x16_bf16 operand / result lane maps,
//   2. an accumulator tile used as the next MFMA's A operand (Z = X^T . B) and its permuted k order,
//   3. ds_read_b64_tr_b16 (transposed LDS read) as the B operand of such a product, with the 320-byte row stride.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_layout.hip -o tools/probes/mfma_layout ; prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A[32][16], B[16][32] row-major floats (exact in bf16); D[32][32]; then X = D (values kept < 256),
// B2[32][32] given; Z[32][32] = X^T . B2 with B2 read (a) from registers in the permuted order, (b) via tr reads.
__global__ void probe(const float* A, const float* B, const float* B2, float* D, float* Z1, float* Z2) {
    __shared__ __attribute__((aligned(16))) bf16 img[32 * 160];  // [row k][n], row stride 160 elements = 320 B
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (bf16)A[r * 16 + 8 * h + j], b[j] = (bf16)B[(8 * h + j) * 32 + r];
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    // X = acc: column r on the lane, rows in the registers. Z = X^T . B2, reduction over X's 32 rows = 2 k-steps.
    for (int i = l; i < 32 * 32; i += 64) img[(i >> 5) * 160 + (i & 31)] = (bf16)B2[i];
    __syncthreads();
    f32x16 z1 = {}, z2 = {};
    for (int s = 0; s < 2; ++s) {
        bf16x8 xa, b1;
        for (int j = 0; j < 8; ++j) {
            xa[j] = (bf16)acc[8 * s + j];
            const int row = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            b1[j] = (bf16)B2[row * 32 + r];
        }
        z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, b1, z1, 0, 0, 0);
        // tr reads: 16-lane group g = l >> 4 -> (h = g >> 1, column block 16 (g & 1)); lane 4q + p of the group supplies
        // the address of row R0 + q, columns c0 + 4p .. + 3; lane i of the group receives column c0 + i of the 4 rows
        const int g = l >> 4, i16 = l & 15, q = i16 >> 2, p = i16 & 3, c0 = 16 * (g & 1);
        bf16x8 b2;
        for (int half = 0; half < 2; ++half) {
            const int R0 = 16 * s + 8 * half + 4 * (g >> 1);
            bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(img + (R0 + q) * 160 + c0 + 4 * p));
            for (int j = 0; j < 4; ++j) b2[4 * half + j] = t[j];
        }
        z2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, b2, z2, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        Z1[row * 32 + r] = z1[i], Z2[row * 32 + r] = z2[i];
    }
}

int main() {
    std::vector<float> A(32 * 16), B(16 * 32), B2(32 * 32), D(32 * 32), Z1(32 * 32), Z2(32 * 32);
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 5 - 2);
    for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 11 + (j > k)) % 4 - 1);  // asymmetric
    for (int k = 0; k < 32; ++k)
        for (int j = 0; j < 32; ++j) B2[k * 32 + j] = (float)((k * 3 + j * 13 + (j > 2 * k)) % 7 - 3);
    float *dA, *dB, *dB2, *dD, *dZ1, *dZ2;
    hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dB2, B2.size() * 4);
    hipMalloc(&dD, D.size() * 4), hipMalloc(&dZ1, Z1.size() * 4), hipMalloc(&dZ2, Z2.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB2, B2.data(), B2.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dB2, dD, dZ1, dZ2);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(Z1.data(), dZ1, Z1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(Z2.data(), dZ2, Z2.size() * 4, hipMemcpyDeviceToHost);
    int bad_d = 0, bad_z1 = 0, bad_z2 = 0;
    float maxd = 0;
    std::vector<float> Dr(32 * 32);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
            Dr[i * 32 + j] = s;
            maxd = fmaxf(maxd, fabsf(s));
            bad_d += D[i * 32 + j] != s;
        }
    for (int f = 0; f < 32; ++f)
        for (int n = 0; n < 32; ++n) {
            float s = 0;
            for (int k = 0; k < 32; ++k) s += Dr[k * 32 + f] * B2[k * 32 + n];
            bad_z1 += Z1[f * 32 + n] != s;
            bad_z2 += Z2[f * 32 + n] != s;
        }
    printf("max |D| = %g (must be < 256 for exact bf16)\n", maxd);
    printf("%s mfma_f32_32x32x16_bf16 lane maps (%d wrong)\n", bad_d ? "FAIL" : "PASS", bad_d);
    printf("%s accumulator tile as A operand, permuted k order (%d wrong)\n", bad_z1 ? "FAIL" : "PASS", bad_z1);
    printf("%s ds_read_b64_tr_b16 as the B operand of that product (%d wrong)\n", bad_z2 ? "FAIL" : "PASS", bad_z2);
    return (bad_d || bad_z1 || bad_z2) ? 1 : 0;
}
