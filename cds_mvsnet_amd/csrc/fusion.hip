// Depth-map filtering + fusion (reference: fusion.py:7-114 and test.py:334-351).
// One thread per reference pixel.  For every source view: project the pixel (with the reference depth) into the source
// view, bilinearly sample — zero padding, align_corners=True — the map "source pixel -> (x, y, depth) re-projected into
// the reference view" (computed on the fly at the four taps from the probability-filtered source depth instead of being
// materialised as a [V,3,h,w] tensor), then apply the pixel-distance / relative-depth / in-range tests, count
// the consistent views and average the consistent depths (ave_fusion).  Output: fused depth, final mask
// (photometric AND geometric) and the world-space point of every pixel.
// Camera block per view (100 floats, row-major), chain "image -> camera -> world -> other camera -> image":
//   [0]  Kinv_ref 3x3  [9]  Einv_ref 4x4  [25] E_src 4x4  [41] K_src 3x3       (reference -> source)
//   [50] Kinv_src 3x3  [59] Einv_src 4x4  [75] E_ref 4x4  [91] K_ref 3x3       (source -> reference)
#include "cds_common.hpp"

namespace {

struct Xyd {
  float x, y, d;
};

// img (x,y,1) with depth -> image coordinates in the other view and depth in the other camera (fusion.py:24-46)
__device__ __forceinline__ Xyd chain(const float* __restrict__ m, float px, float py, float depth) {
  // idx_img2cam: Kinv @ pix, normalised by its last component, times depth
  float cx = m[0] * px + m[1] * py + m[2];
  float cy = m[3] * px + m[4] * py + m[5];
  float cz = m[6] * px + m[7] * py + m[8];
  const float n = cz + 1e-9f;
  cx = cx / n * depth;
  cy = cy / n * depth;
  cz = cz / n * depth;
  // idx_cam2world: Einv @ (cx,cy,cz,1), normalised
  const float* e = m + 9;
  float wx = e[0] * cx + e[1] * cy + e[2] * cz + e[3];
  float wy = e[4] * cx + e[5] * cy + e[6] * cz + e[7];
  float wz = e[8] * cx + e[9] * cy + e[10] * cz + e[11];
  float ww = e[12] * cx + e[13] * cy + e[14] * cz + e[15];
  const float nw = ww + 1e-9f;
  wx /= nw; wy /= nw; wz /= nw; ww /= nw;
  // idx_world2cam: E_other @ world, normalised
  const float* f = m + 25;
  float ox = f[0] * wx + f[1] * wy + f[2] * wz + f[3] * ww;
  float oy = f[4] * wx + f[5] * wy + f[6] * wz + f[7] * ww;
  float oz = f[8] * wx + f[9] * wy + f[10] * wz + f[11] * ww;
  float ow = f[12] * wx + f[13] * wy + f[14] * wz + f[15] * ww;
  const float no = ow + 1e-9f;
  ox /= no; oy /= no; oz /= no; ow /= no;
  // idx_cam2img: K @ (xyz / w), normalised by z
  const float nq = ow + 1e-9f;
  const float qx = ox / nq, qy = oy / nq, qz = oz / nq;
  const float* k = m + 41;
  float ix = k[0] * qx + k[1] * qy + k[2] * qz;
  float iy = k[3] * qx + k[4] * qy + k[5] * qz;
  float iz = k[6] * qx + k[7] * qy + k[8] * qz;
  const float ni = iz + 1e-9f;
  Xyd r;
  r.x = ix / ni;
  r.y = iy / ni;
  r.d = oz;  // depth in the other camera (srcs2ref_idx_cam[..., 2])
  return r;
}

__device__ __forceinline__ bool prob_ok(const float* __restrict__ conf, size_t hw, size_t p, float t0, float t1, float t2) {
  return conf[p] > t0 && conf[hw + p] > t1 && conf[2 * hw + p] > t2;
}

__global__ __launch_bounds__(256) void depth_fusion_kernel(const float* __restrict__ ref_depth,
                                                           const float* __restrict__ ref_conf,
                                                           const float* __restrict__ src_depths,
                                                           const float* __restrict__ src_confs,
                                                           const float* __restrict__ cams, float* __restrict__ fused,
                                                           float* __restrict__ mask_out, float* __restrict__ points,
                                                           float* __restrict__ view_masks, int V, int h, int w, float t0,
                                                           float t1, float t2, float dist_thresh, float depth_thresh,
                                                           float vthresh) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t hw = (size_t)h * w;
  if (p >= h * w) return;
  const int y = p / w, x = p % w;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float rd = ref_depth[p];
  float sum_m = 0.f, sum_d = 0.f;
  for (int v = 0; v < V; ++v) {
    const float* __restrict__ m = cams + (size_t)v * 100;
    const float* __restrict__ sd = src_depths + (size_t)v * hw;
    const float* __restrict__ sc = src_confs + (size_t)v * 3 * hw;
    const Xyd q = chain(m, px, py, rd);
    float gx = q.x / (float)w * 2.0f - 1.0f;
    float gy = q.y / (float)h * 2.0f - 1.0f;
    gx = fminf(fmaxf(gx, -1.1f), 1.1f);
    gy = fminf(fmaxf(gy, -1.1f), 1.1f);
    const bool in_range = (gx >= -1.0f) && (gx <= 1.0f) && (gy >= -1.0f) && (gy <= 1.0f);
    const float ix = (gx + 1.0f) * 0.5f * (float)(w - 1), iy = (gy + 1.0f) * 0.5f * (float)(h - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wx0 = 1.0f - wx1, wy1 = iy - y0f, wy0 = 1.0f - wy1;
    const int x0 = (int)x0f, y0 = (int)y0f;
    float rx = 0.f, ry = 0.f, rz = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int xs = x0 + (t & 1), ys = y0 + (t >> 1);
      if (xs < 0 || xs >= w || ys < 0 || ys >= h) continue;   // zero padding
      const size_t sp = (size_t)ys * w + xs;
      const float d_s = prob_ok(sc, hw, sp, t0, t1, t2) ? sd[sp] : 0.f;   // src_depths *= prob_mask (test.py:336-338)
      const Xyd r = chain(m + 50, (float)xs + 0.5f, (float)ys + 0.5f, d_s);
      const float wt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
      rx = fmaf(r.x, wt, rx);
      ry = fmaf(r.y, wt, ry);
      rz = fmaf(r.d, wt, rz);
    }
    const float dx = rx - px, dy = ry - py;
    const bool dist_ok = sqrtf(dx * dx + dy * dy) < dist_thresh;
    const bool depth_ok = fabsf(rd - rz) < fmaxf(rd, rz) * depth_thresh;
    const float mv = (in_range && dist_ok && depth_ok) ? 1.0f : 0.0f;
    if (view_masks) view_masks[(size_t)v * hw + p] = mv;
    sum_m += mv;
    sum_d = fmaf(rz, mv, sum_d);
  }
  const bool vis_ok = sum_m >= vthresh - 1.1f;
  const float ave = (sum_d + rd) / (sum_m + 1.0f);
  const bool keep = vis_ok && prob_ok(ref_conf, hw, p, t0, t1, t2);
  fused[p] = ave;
  mask_out[p] = keep ? 1.0f : 0.0f;
  // world point of the fused depth (test.py:348-350): Einv_ref @ (Kinv_ref @ pix / z * depth)
  const float* m = cams;
  float cx = m[0] * px + m[1] * py + m[2], cy = m[3] * px + m[4] * py + m[5], cz = m[6] * px + m[7] * py + m[8];
  const float n = cz + 1e-9f;
  cx = cx / n * ave; cy = cy / n * ave; cz = cz / n * ave;
  const float* e = m + 9;
  const float ww = e[12] * cx + e[13] * cy + e[14] * cz + e[15] + 1e-9f;
  points[p] = (e[0] * cx + e[1] * cy + e[2] * cz + e[3]) / ww;
  points[hw + p] = (e[4] * cx + e[5] * cy + e[6] * cz + e[7]) / ww;
  points[2 * hw + p] = (e[8] * cx + e[9] * cy + e[10] * cz + e[11]) / ww;
}

}  // namespace

extern "C" int cds_depth_fusion_f32(const float* ref_depth, const float* ref_conf, const float* src_depths,
                                    const float* src_confs, const float* cams, float* fused, float* mask, float* points,
                                    float* view_masks, int V, int h, int w, const float* prob_thresh_host,
                                    float dist_thresh, float depth_thresh, float view_thresh, void* stream) {
  if (!ref_depth || !ref_conf || !src_depths || !src_confs || !cams || !fused || !mask || !points || !prob_thresh_host ||
      V < 1 || h < 1 || w < 1)
    return CDS_EINVAL;
  hipLaunchKernelGGL(depth_fusion_kernel, dim3(cds_ceil_div(h * w, 256)), dim3(256), 0, (hipStream_t)stream, ref_depth,
                     ref_conf, src_depths, src_confs, cams, fused, mask, points, view_masks, V, h, w, prob_thresh_host[0],
                     prob_thresh_host[1], prob_thresh_host[2], dist_thresh, depth_thresh, view_thresh);
  return cds_launch_status();
}
