// How fast can a channels-last [D][H][W][8] fp32 volume (2 GB at M1) be read and written when the access order is the conv kernels'
// (a workgroup owns a TX x TY column and marches along z) instead of linear?  Pure copy, no halo, no LDS: the ceiling the tiled
// CostRegNet layers can reach on the HBM side.  hipcc --offload-arch=gfx950 -O3 tile_copy.hip -o tile_copy && ./tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int TX, int TY, int DEPTH>
__global__ __launch_bounds__(256) void tile_copy(const float4* __restrict__ in, float4* __restrict__ out, int D, int H, int W, int tiles_x,
                                                 int nseg) {
  const int nwg = gridDim.x;
  int b = blockIdx.x;
  {  // XCD-aware remap like cds_xcd_remap
    const int q = nwg / 8, r = nwg % 8, xcd = b % 8, idx = b / 8;
    b = (nwg >= 16) ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
  }
  const int seg = b % nseg;
  const int tile = b / nseg;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int zseg = (D + nseg - 1) / nseg;
  const int z0 = seg * zseg, z1 = min(D, z0 + zseg);
  constexpr int VOX = TX * TY, PER = (VOX + 255) / 256;
  for (int z = z0; z < z1; z += DEPTH) {
    float4 a[DEPTH][PER], c[DEPTH][PER];
#pragma unroll
    for (int dz = 0; dz < DEPTH; ++dz)
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int v = k * 256 + threadIdx.x;
        const int x = tx * TX + v % TX, y = ty * TY + v / TX;
        const size_t o = (((size_t)min(z + dz, z1 - 1) * H + y) * W + x) * 2;
        a[dz][k] = in[o];
        c[dz][k] = in[o + 1];
      }
#pragma unroll
    for (int dz = 0; dz < DEPTH; ++dz)
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int v = k * 256 + threadIdx.x;
        const int x = tx * TX + v % TX, y = ty * TY + v / TX;
        if (z + dz < z1) {
          const size_t o = (((size_t)(z + dz) * H + y) * W + x) * 2;
          out[o] = a[dz][k];
          out[o + 1] = c[dz][k];
        }
      }
  }
}
__global__ void linear_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
template <int TX, int TY, int DEPTH>
void run(const float4* in, float4* out, int D, int H, int W, int nseg) {
  const int tx = W / TX, ty = H / TY;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tile_copy<TX, TY, DEPTH>), dim3(tx * ty * nseg), dim3(256), 0, 0, in, out, D, H, W, tx, nseg);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((tile_copy<TX, TY, DEPTH>), dim3(tx * ty * nseg), dim3(256), 0, 0, in, out, D, H, W, tx, nseg);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 2.0 * D * H * W * 32.0;
  printf("tile %3d x %2d, %d planes in flight, %d z-segments (%5d workgroups): %.3f ms = %.2f TB/s (read + write)\n", TX, TY, DEPTH, nseg,
         tx * ty * nseg, ms, bytes / ms / 1e9);
}
int main() {
  const int D = 192, H = 512, W = 640;
  const size_t n4 = (size_t)D * H * W * 2;
  float4 *in, *out;
  hipMalloc(&in, n4 * 16); hipMalloc(&out, n4 * 16);
  hipMemset(in, 1, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(linear_copy, dim3(256 * 16), dim3(256), 0, 0, in, out, n4);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(linear_copy, dim3(256 * 16), dim3(256), 0, 0, in, out, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("linear float4 copy: %.3f ms = %.2f TB/s (read + write)\n", ms, 2.0 * n4 * 16 / ms / 1e9);
  run<32, 8, 1>(in, out, D, H, W, 1);
  run<32, 8, 3>(in, out, D, H, W, 1);
  run<32, 8, 3>(in, out, D, H, W, 4);
  run<32, 8, 6>(in, out, D, H, W, 4);
  run<64, 4, 3>(in, out, D, H, W, 4);
  run<64, 8, 3>(in, out, D, H, W, 4);
  run<128, 2, 3>(in, out, D, H, W, 4);
  run<128, 8, 2>(in, out, D, H, W, 4);
  run<32, 8, 3>(in, out, D, H, W, 16);
  return 0;
}
