// Micro-test of the MFMA chain recorded in DESIGN.md section 5: a v_mfma_f32_16x16x16_bf16 that takes the result registers of a
// v_mfma_f32_16x16x32_bf16 as its accumulator input (head dim 48 = 32 + 16 as one 32-deep and one 16-deep step).
//   B  builtins, back to back (what the attention kernels were first written as)
//   N  builtins with `s_nop 7; s_nop 7` forced between the two (sched_barrier + asm volatile)
//   A  one inline-asm block with fixed registers, back to back (nothing can be inserted by the compiler)
//   AN the same with s_nop 7 x 2 between the two instructions
// Reference: fp32 dot products on the host-visible inputs (bf16 products are exact in fp32; sums of 48 terms).
// Build / run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_chain.hip -o gpurun_out/mfma_chain && gpurun_out/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A [16 rows][48 k], B [16 cols][48 k] per wave (row-major, k contiguous); C[i][j] = sum_k A[i][k] B[j][k]
template <int VARIANT>
__global__ void __launch_bounds__(256) chain_kernel(const __bf16* A, const __bf16* B, float* C, int iters) {
  const int lane = threadIdx.x & 63, w = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const __bf16* a = A + (size_t)w * 16 * 48;
  const __bf16* b = B + (size_t)w * 16 * 48;
  const int r = lane & 15, g = lane >> 4;
  // 16x16x32 operands: lane holds k = 8g .. 8g+7 of row r; 16x16x16: k = 32 + 4g .. 32 + 4g + 3
  bf16x8 a32 = *reinterpret_cast<const bf16x8*>(a + r * 48 + 8 * g), b32 = *reinterpret_cast<const bf16x8*>(b + r * 48 + 8 * g);
  bf16x4 a16 = *reinterpret_cast<const bf16x4*>(a + r * 48 + 32 + 4 * g), b16 = *reinterpret_cast<const bf16x4*>(b + r * 48 + 32 + 4 * g);
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {  // (the same product every time: a wrong pass shows in the final value)
    f32x4 c = {0, 0, 0, 0};
    if (VARIANT == 0) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b32, a32, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, b16),
                                                    __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, a16), c, 0, 0, 0);
    } else if (VARIANT == 1) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b32, a32, c, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c));
      __builtin_amdgcn_sched_barrier(0);
      c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, b16),
                                                    __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, a16), c, 0, 0, 0);
    } else if (VARIANT == 2) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 7\n\ts_nop 7"
                   : "=&v"(c) : "v"(b32), "v"(a32), "v"(b16), "v"(a16));
    } else {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 7\n\ts_nop 7"
                   : "=&v"(c) : "v"(b32), "v"(a32), "v"(b16), "v"(a16));
    }
    acc = c;
  }
  // first operand = rows of B, second = rows of A: acc[e] = sum_k B[4g+e][k] A[r][k] = C[r][4g+e]
  for (int e = 0; e < 4; ++e) C[(size_t)w * 256 + r * 16 + (4 * g + e)] = acc[e];
}

int main() {
  const int waves = 256 * 4 * 4, iters = 64;
  size_t n = (size_t)waves * 16 * 48;
  __bf16 *hA = (__bf16*)malloc(n * 2), *hB = (__bf16*)malloc(n * 2);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (size_t i = 0; i < n; ++i) { hA[i] = (__bf16)rnd(); hB[i] = (__bf16)rnd(); }
  __bf16 *dA, *dB; float* dC;
  hipMalloc(&dA, n * 2); hipMalloc(&dB, n * 2); hipMalloc(&dC, (size_t)waves * 256 * 4);
  hipMemcpy(dA, hA, n * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 2, hipMemcpyHostToDevice);
  float* hC = (float*)malloc((size_t)waves * 256 * 4);
  const char* names[4] = {"B  builtins, back to back", "N  builtins, s_nop 7 x 2 between", "A  inline asm, back to back", "AN inline asm, s_nop 7 x 2 between"};
  for (int v = 0; v < 4; ++v) {
    double worst = 0; long bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(dC, 0, (size_t)waves * 256 * 4);
      if (v == 0) chain_kernel<0><<<waves / 4, 256>>>(dA, dB, dC, iters);
      if (v == 1) chain_kernel<1><<<waves / 4, 256>>>(dA, dB, dC, iters);
      if (v == 2) chain_kernel<2><<<waves / 4, 256>>>(dA, dB, dC, iters);
      if (v == 3) chain_kernel<3><<<waves / 4, 256>>>(dA, dB, dC, iters);
      hipMemcpy(hC, dC, (size_t)waves * 256 * 4, hipMemcpyDeviceToHost);
      for (int w = 0; w < waves; ++w)
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            float ref = 0;
            for (int k = 0; k < 48; ++k) ref += (float)hA[((size_t)w * 16 + i) * 48 + k] * (float)hB[((size_t)w * 16 + j) * 48 + k];
            const double err = fabs((double)hC[(size_t)w * 256 + i * 16 + j] - ref);
            if (err > 1e-3) ++bad;
            if (err > worst) worst = err;
          }
    }
    printf("%-40s wrong elements in 3 launches: %ld of %ld   worst abs error %.3g\n", names[v], bad, 3L * waves * 256, worst);
  }
  return 0;
}
