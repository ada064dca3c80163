// Hardware layout probe for gfx950 (test infrastructure, not product code).
// Verifies the MFMA operand/accumulator lane maps and the ds_read_b64_tr_b16 gather
// that the kernels in camradepth_amd/csrc rely on. Build: hipcc --offload-arch=gfx950 -O2 -o probe probe_layouts.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_mfma32(const float* A, const float* B, float* D) {  // A[32][16], B[16][32] row-major, D[32][32]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = acc[r];
  }
}
__global__ void k_mfma16(const float* A, const float* B, float* D) {  // A[16][32], B[32][16], D[16][16]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = (__bf16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r, col = l & 15;
    D[row * 16 + col] = acc[r];
  }
}
__global__ void k_tr(int mode, short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  int off;
  if (mode == 0) off = l * 4;                                                     // contiguous 8 B per lane
  else if (mode == 1) off = (l >> 4) * 256 + ((l & 15) >> 2) * 64 + (l & 3) * 4;  // rows of 16 elems at stride 64
  else off = (l >> 4) * 1024 + ((l & 15) >> 2) * 128 + (l & 3) * 4 + 16;          // stride 128, col offset 16
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  srand(1);
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (auto& v : A) v = (float)(rand() % 9 - 4);
    for (auto& v : B) v = (float)(rand() % 7 - 3);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (size_t i = 0; i < D.size(); ++i) bad += (D[i] != R[i]);
    printf("mfma_f32_32x32x16_bf16 layout hypothesis: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  }
  {
    std::vector<float> A(16 * 32), B(32 * 16), D(16 * 16), R(16 * 16, 0.f);
    for (auto& v : A) v = (float)(rand() % 9 - 4);
    for (auto& v : B) v = (float)(rand() % 7 - 3);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (size_t i = 0; i < D.size(); ++i) bad += (D[i] != R[i]);
    printf("mfma_f32_16x16x32_bf16 layout hypothesis: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  }
  for (int mode = 0; mode < 3; ++mode) {
    short* d; CK(hipMalloc(&d, 256 * 2)); std::vector<short> h(256);
    k_tr<<<1, 64>>>(mode, d); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int exp;
      if (mode == 0) exp = (l & 15) + j * 16 + (l >> 4) * 64;
      else if (mode == 1) exp = (l >> 4) * 256 + j * 64 + (l & 15);
      else exp = (l >> 4) * 1024 + j * 128 + 16 + (l & 15);
      bad += (h[l * 4 + j] != exp);
    }
    printf("ds_read_b64_tr_b16 mode %d hypothesis: %s (%d mismatches)\n", mode, bad ? "FAIL" : "PASS", bad);
    if (bad) { for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch %s CUs %d clock %d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  return 0;
}
