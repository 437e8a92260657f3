// Training loss of CDS-MVSNet on the device (models/losses.py:6-48) and the feature-distance targets (models/model.py:202-207):
// a handful of launches per step instead of ~90 ATen launches forward and ~150 backward (the config-5 step is launch-bound on
// its small tensors).  All sums are fp64 and in a fixed order (per-workgroup records reduced by a tree whose shape depends on
// the tensor sizes only): bit-reproducible, no atomics.
#include "cds_common.hpp"
#include "feat_common.hpp"

namespace {

constexpr int LOSS_MAX_BLOCKS = 1024;   // records per pass: a workgroup of the next pass reduces them with 4 loads per thread

// sum of v over the workgroup (256 threads), valid in every thread
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  v = wave_sum_f64(v);
  __syncthreads();                       // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// records [n][NF] -> NF sums, valid in every thread; fixed order for a given n
template <int NF>
__device__ __forceinline__ void reduce_records(const double* __restrict__ rec, int n, double* out, double* red) {
  double acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = 0.0;
  for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] += rec[(size_t)i * NF + f];
#pragma unroll
  for (int f = 0; f < NF; ++f) out[f] = block_sum_f64(acc[f], red);
}

__device__ __forceinline__ float smooth_l1(float x) {
  const float a = fabsf(x);
  return a < 1.f ? 0.5f * x * x : a - 0.5f;
}

// Pass A over the pixels of a stage: records [blocks][4] = (count, sum smooth-L1, sum norm_curv, sum_d target) over mask > 0.5.
__global__ __launch_bounds__(256) void loss_pixel_kernel(const float* __restrict__ depth, const float* __restrict__ gt,
                                                         const float* __restrict__ mask, const float* __restrict__ nc,
                                                         const float* __restrict__ target, const float* __restrict__ interval,
                                                         int B, int hw, int Dp, double* __restrict__ rec) {
  __shared__ double red[4];
  double cnt = 0.0, sl = 0.0, cv = 0.0, ps = 0.0;
  const int total = B * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (!(mask[i] > 0.5f)) continue;
    const int b = i / hw, p = i - b * hw;
    const float iv = interval[b];
    cnt += 1.0;
    sl += (double)smooth_l1(depth[i] / iv - gt[i] / iv);
    if (nc) cv += (double)nc[i];
    if (target) {
      const float* __restrict__ t = target + (size_t)b * Dp * hw + p;
      float s = 0.f;                                   // 0 / 1 values: exact in fp32 up to 2^24 planes
      for (int d = 0; d < Dp; ++d) s += t[(size_t)d * hw];
      ps += (double)s;
    }
  }
  cnt = block_sum_f64(cnt, red);
  sl = block_sum_f64(sl, red);
  cv = block_sum_f64(cv, red);
  ps = block_sum_f64(ps, red);
  if (threadIdx.x == 0) {
    double* r = rec + (size_t)blockIdx.x * 4;
    r[0] = cnt; r[1] = sl; r[2] = cv; r[3] = ps;
  }
}

// F.binary_cross_entropy_with_logits(x, t, pos_weight = pw), elementwise (ATen's formula)
__device__ __forceinline__ float bce_logits(float x, float t, float pw) {
  const float lw = 1.f + (pw - 1.f) * t;
  return (1.f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}

// Pass B over the feature-distance volume: records [blocks][1] = sum of the balanced BCE over mask > 0.5 (all Dp planes).
__global__ __launch_bounds__(256) void loss_bce_kernel(const float* __restrict__ dist, const float* __restrict__ target,
                                                       const float* __restrict__ mask, const double* __restrict__ recA, int nA, int B,
                                                       int hw, int Dp, double* __restrict__ rec) {
  __shared__ double red[4];
  double a[4];
  reduce_records<4>(recA, nA, a, red);
  const double n = a[0] * Dp;
  const float pw = (float)(n - a[3]) / (float)a[3];                 // neg / pos (losses.py:30-33), fp32 like the reference
  double s = 0.0;
  const size_t total = (size_t)B * Dp * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / ((size_t)Dp * hw));
    const int p = (int)(i % hw);
    if (!(mask[(size_t)b * hw + p] > 0.5f)) continue;
    s += (double)bce_logits(dist[i], target[i], pw);
  }
  s = block_sum_f64(s, red);
  if (threadIdx.x == 0) rec[blockIdx.x] = s;
}

struct LossStages {
  const double* recA[4];
  const double* recB[4];
  int nA[4], nB[4], Dp[4];
  float weight[4];       // dlossw (1 when absent); the refined-depth stage carries 2 (losses.py:42)
  int has_feat[4], has_curv[4];
  int n;                 // stages incl. the refined-depth one
};

// total loss, the last depth loss, and per stage (count, pos_weight, n) for the backward pass.  One workgroup.
__global__ __launch_bounds__(256) void loss_final_kernel(LossStages st, float* __restrict__ total, float* __restrict__ depth_loss,
                                                         double* __restrict__ scalars) {
  __shared__ double red[4];
  float tot = 0.f, dl = 0.f;
  for (int s = 0; s < st.n; ++s) {
    double a[4], bce[1] = {0.0};
    reduce_records<4>(st.recA[s], st.nA[s], a, red);
    if (st.has_feat[s]) reduce_records<1>(st.recB[s], st.nB[s], bce, red);
    const float cnt = (float)a[0];
    dl = (float)a[1] / cnt;
    float term = dl;
    const double n = a[0] * st.Dp[s];
    if (st.has_feat[s]) term += 5.f * ((float)bce[0] / (float)n);
    if (st.has_curv[s]) term += 0.1f * ((float)a[2] / cnt);
    tot += st.weight[s] * term;
    if (threadIdx.x == 0) {
      scalars[4 * s] = a[0];
      scalars[4 * s + 1] = st.has_feat[s] ? (double)((float)(n - a[3]) / (float)a[3]) : 0.0;
      scalars[4 * s + 2] = n;
    }
  }
  if (threadIdx.x == 0) {
    *total = tot;
    *depth_loss = dl;
  }
}

// Backward of one stage: g = d total (device scalar) -> gdepth [B][hw], gnc [B][hw] (or NULL), gdist [B][Dp][hw] (or NULL).
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ gt,
                                                       const float* __restrict__ mask, const float* __restrict__ dist,
                                                       const float* __restrict__ target, const float* __restrict__ interval,
                                                       const float* __restrict__ gtotal, const double* __restrict__ scalars, float weight,
                                                       int B, int hw, int Dp, float* __restrict__ gdepth, float* __restrict__ gnc,
                                                       float* __restrict__ gdist) {
  const float g = gtotal[0] * weight;
  const float cnt = (float)scalars[0], pw = (float)scalars[1], n = (float)scalars[2];
  const int planes = gdist ? Dp : 1;
  const size_t total = (size_t)B * planes * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / ((size_t)planes * hw));
    const int d = (int)((i / hw) % planes);
    const int p = (int)(i % hw);
    const size_t px = (size_t)b * hw + p;
    const bool m = mask[px] > 0.5f;
    if (gdist) {
      float v = 0.f;
      if (m) {
        const float x = dist[i], t = target[i];
        const float sig_neg = 1.f / (1.f + expf(x));                 // sigmoid(-x)
        v = g * 5.f * ((1.f - t) - (1.f + (pw - 1.f) * t) * sig_neg) / n;
      }
      gdist[i] = v;
    }
    if (d == 0) {
      float v = 0.f;
      if (m) {
        const float iv = interval[b];
        const float x = depth[px] / iv - gt[px] / iv;
        const float dx = fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f);
        v = g * dx / iv / cnt;
      }
      gdepth[px] = v;
      if (gnc) gnc[px] = m ? g * 0.1f / cnt : 0.f;
    }
  }
}

// target[b][d] = |hyp[b][d] - gt[b]| / (di[b] scale) < thresh for d < D, plane D = 1  (models/model.py:202-207)
__global__ __launch_bounds__(256) void feat_target_kernel(const float* __restrict__ hyp, const float* __restrict__ gt,
                                                          const float* __restrict__ di, float scale, float thresh, int B, int D,
                                                          int hw, float* __restrict__ target) {
  const size_t total = (size_t)B * (D + 1) * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / ((size_t)(D + 1) * hw));
    const int d = (int)((i / hw) % (D + 1));
    const int p = (int)(i % hw);
    float v = 1.f;
    if (d < D) v = fabsf(hyp[((size_t)b * D + d) * hw + p] - gt[(size_t)b * hw + p]) / (di[b] * scale) < thresh ? 1.f : 0.f;
    target[i] = v;
  }
}

inline int loss_blocks(size_t n) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)LOSS_MAX_BLOCKS ? (size_t)LOSS_MAX_BLOCKS : b));
}

}  // namespace

extern "C" int cds_loss_records(long long elements) { return elements < 1 ? 1 : loss_blocks((size_t)elements); }

// One stage of final_loss, forward part 1.  depth, gt, mask [B][h*w]; norm_curv [B][h*w] or NULL; dist, target [B][Dp][h*w] or both
// NULL (the refined-depth stage); interval [B] on the device.  recA: cds_loss_records(B h w) x 4 doubles, recB:
// cds_loss_records(B Dp h w) doubles (NULL without dist).
extern "C" int cds_loss_stage_f32(const float* depth, const float* gt, const float* mask, const float* norm_curv, const float* dist,
                                  const float* target, const float* interval, int B, int hw, int Dp, double* recA, double* recB,
                                  void* stream) {
  if (!depth || !gt || !mask || !interval || !recA || B < 1 || hw < 1 || (dist != nullptr) != (target != nullptr) ||
      (dist && (!recB || Dp < 1)))
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nA = loss_blocks((size_t)B * hw);
  hipLaunchKernelGGL(loss_pixel_kernel, dim3(nA), dim3(256), 0, st, depth, gt, mask, norm_curv, target, interval, B, hw, Dp, recA);
  if (dist) {
    const int nB = loss_blocks((size_t)B * Dp * hw);
    hipLaunchKernelGGL(loss_bce_kernel, dim3(nB), dim3(256), 0, st, dist, target, mask, recA, nA, B, hw, Dp, recB);
  }
  return cds_launch_status();
}

// Forward part 2 over all stages: n_stages <= 4 entries of recA / recB (NULL without the feature term) / pixels = B h w / Dp / weight /
// has_curv (the stage was given norm_curv).  term = depth + 5 feature + 0.1 curvature (losses.py:36), total = sum weight x term.
// total, depth_loss: device scalars; scalars: [n_stages][4] doubles kept for cds_loss_stage_bwd_f32.
extern "C" int cds_loss_final_f32(const double* const* recA, const double* const* recB, const long long* pixels, const int* Dp,
                                  const float* weight, const int* has_curv, int n_stages, float* total, float* depth_loss, double* scalars,
                                  void* stream) {
  if (!recA || !recB || !pixels || !Dp || !weight || !has_curv || n_stages < 1 || n_stages > 4 || !total || !depth_loss || !scalars) return CDS_EINVAL;
  LossStages st{};
  st.n = n_stages;
  for (int s = 0; s < n_stages; ++s) {
    if (!recA[s] || pixels[s] < 1) return CDS_EINVAL;
    st.recA[s] = recA[s];
    st.recB[s] = recB[s];
    st.has_feat[s] = recB[s] != nullptr;
    st.Dp[s] = st.has_feat[s] ? Dp[s] : 1;
    st.nA[s] = loss_blocks((size_t)pixels[s]);
    st.nB[s] = st.has_feat[s] ? loss_blocks((size_t)pixels[s] * Dp[s]) : 0;
    st.weight[s] = weight[s];
    st.has_curv[s] = has_curv[s];
  }
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, st, total, depth_loss, scalars);
  return cds_launch_status();
}

// Backward of one stage.  gtotal: device scalar d(loss); scalars: this stage's row of cds_loss_final_f32's output.
extern "C" int cds_loss_stage_bwd_f32(const float* depth, const float* gt, const float* mask, const float* dist, const float* target,
                                      const float* interval, const float* gtotal, const double* scalars, float weight, int B, int hw,
                                      int Dp, float* gdepth, float* gnc, float* gdist, void* stream) {
  if (!depth || !gt || !mask || !interval || !gtotal || !scalars || !gdepth || B < 1 || hw < 1 || (gdist && (!dist || !target || Dp < 1)))
    return CDS_EINVAL;
  const size_t total = (size_t)B * (gdist ? Dp : 1) * hw;
  const size_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0, (hipStream_t)stream, depth, gt,
                     mask, dist, target, interval, gtotal, scalars, weight, B, hw, Dp, gdepth, gnc, gdist);
  return cds_launch_status();
}

// The feature-distance targets of a stage: hyp [B][D][h*w], gt [B][h*w], di [B] (device: the depth interval; scale: the stage's) ->
// target [B][D + 1][h*w].
extern "C" int cds_feat_target_f32(const float* hyp, const float* gt, const float* di, float scale, float thresh, int B, int D, int hw,
                                   float* target, void* stream) {
  if (!hyp || !gt || !di || !target || B < 1 || D < 1 || hw < 1) return CDS_EINVAL;
  const size_t total = (size_t)B * (D + 1) * hw;
  const size_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(feat_target_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0, (hipStream_t)stream, hyp,
                     gt, di, scale, thresh, B, D, hw, target);
  return cds_launch_status();
}
