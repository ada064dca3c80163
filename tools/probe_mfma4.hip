// Hardware probe (round 6): the operand / result layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave) -- the
// instruction the MFMA form of the depthwise 3x3 convolution rests on (block = channel).  Checks the layout assumed in
// csrc/encoder_ops.hip against a host model on random small integers and prints which of the candidate layouts holds.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma4.hip -o tools/probe_mfma4 && tools/probe_mfma4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const s16x4* a, const s16x4* b, f32x4* d) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  d[threadIdx.x] = acc;
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
  float A[64][4], B[64][4], D[64][4];
  unsigned short ha[64][4], hb[64][4];
  srand(1);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) { A[l][k] = (float)(rand() % 7 - 3); B[l][k] = (float)(rand() % 7 - 3); ha[l][k] = bf(A[l][k]); hb[l][k] = bf(B[l][k]); }
  void *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(D));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const s16x4*)da, (const s16x4*)db, (f32x4*)dd);
  hipMemcpy(D, dd, sizeof(D), hipMemcpyDeviceToHost);
  // candidate 0: block = lane / 4; A row i = lane % 4 (its 4 values = k); B column j = lane % 4 (its 4 values = k); D: lane = 4 block + j, register = i
  // candidate 1: same operands; D: lane = 4 block + i, register = j
  for (int cand = 0; cand < 2; ++cand) {
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      float ref = 0;
      for (int kk = 0; kk < 4; ++kk) ref += A[4 * blk + i][kk] * B[4 * blk + j][kk];
      const float got = cand == 0 ? D[4 * blk + j][i] : D[4 * blk + i][j];
      bad += got != ref;
    }
    printf("candidate %d: %s (%d mismatches)\n", cand, bad ? "no" : "YES", bad);
  }
  return 0;
}
