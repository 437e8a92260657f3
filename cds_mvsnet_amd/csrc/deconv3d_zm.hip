// K4: ConvTranspose3d k3 s2 p1 op1 (+ folded BN shift, ReLU, U-Net residual; models/module.py:125-160, :310-312) to Cout = 16 with
// Cin = 32 - conv9 of CostRegNet - as a z-marching, class-per-wave kernel in split-bf16 arithmetic.
//
// Output voxel o = 2 a + p per axis: parity 0 takes kernel tap 1 of input cell a, parity 1 takes tap 2 of cell a and tap 0 of cell
// a + 1.  The 8 parity classes (pz, py, px) are 8 small convolutions over the SAME 2 x 2 x 2 cell neighbourhood with different
// weights, so (1) a class is given to one consumer wave, which keeps ALL its weights in registers for the whole march (4 rounds x
// 1 or 2 matrix operands: nothing but input cells is read from LDS in the loop), and (2) the K-steps are organised by cell offset,
// not by class: K-slot g = (dy, dx) of a 32-deep K-step at dz = 0, a second K-step at dz = 1 for the pz = 1 classes; slots a class
// does not use carry zero weights (12 MFMA K-steps per round over the 8 classes instead of 9, 2 operand reads per round and
// N-tile instead of 9: the tiled kernel this replaces ran its matrix pipe at 20 %).
// A workgroup owns a column of 16 x 8 input cells and marches along z over a ring of three input cell planes in LDS (all four
// 8-channel rounds of a plane resident, exact three-way bf16 split done by the 4 producer waves on the way in): one barrier per cell
// plane; every input cell is read from HBM once per column (+ the one-cell halo on the high sides).  Each SIMD holds one pz = 0 and
// one pz = 1 consumer wave (1 : 2 matrix work) and a producer.
#include "sbf_common.hpp"

namespace {

template <int ROUNDS_>
struct DZC {
  static constexpr int ROUNDS = ROUNDS_;
  static constexpr int TX = 16, TY = 8;                 // cells per plane: one N-tile per cell row
  static constexpr int IX = TX + 1, IY = TY + 1, IXP = 18;
  static constexpr int ROUNDB = IY * IXP * POSB;
  static constexpr int SLOTB = ROUNDS * ROUNDB;         // one input cell plane: [round][row][col][term][8] bf16
  static constexpr int NSLOT = 3;
  static constexpr int CW = 8, PW = 4, THREADS = (CW + PW) * 64;
  static constexpr int LDS = NSLOT * SLOTB;
  static constexpr int NITEM = IY * IX * ROUNDS;
  static constexpr int IPT = (NITEM + PW * 64 - 1) / (PW * 64);
};

// F16: split-f16 arithmetic (sbf_common.hpp): two fp16 terms, three products per K-step, tensor scales from device bounds.
template <int ROUNDS, bool F16>
__global__ __launch_bounds__((DZC<ROUNDS>::THREADS), 3) void deconv3d_zm_kernel(
    const float* __restrict__ x, const uint4* __restrict__ wcls, const float* __restrict__ bias, const float* __restrict__ skip,
    float* __restrict__ out, int D, int H, int W, int tiles_x, int ncols, int seg_len, int act, const float* __restrict__ in_bound,
    float w_inv, float* __restrict__ out_bound) {
  using C = DZC<ROUNDS>;
  constexpr int NT = F16 ? 2 : 3;
  const float xs = F16 ? sf16_scale(in_bound[0]) : 1.0f;
  const float out_mul = F16 ? w_inv / xs : 1.0f;
  constexpr int Cin = 8 * ROUNDS, Cout = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = cds_xcd_remap(blockIdx.x, gridDim.x);
  const int col = wg % ncols, seg = wg / ncols;
  const int tx_i = col % tiles_x, ty_i = col / tiles_x;
  const int a0 = seg * seg_len, a1 = min(D, a0 + seg_len);
  if (a0 >= a1) return;
  const int X0 = tx_i * C::TX, Y0 = ty_i * C::TY;
  const int Ho = 2 * H, Wo = 2 * W;

  if (wave >= C::CW) {
    // ============================== producers ==============================
    __builtin_amdgcn_s_setprio(3);               // A/B at M1: producers at 0 / 1 / 3: 355 / 351 / 345 us
    const int ptid = tid - C::CW * 64;
    int s_src[C::IPT], s_dst[C::IPT];
#pragma unroll
    for (int h = 0; h < C::IPT; ++h) {
      const int it = h * C::PW * 64 + ptid;
      const int rd = it % ROUNDS, p = it / ROUNDS;
      const int row = p / C::IX, c = p - row * C::IX;
      const int gy = Y0 + row, gx = X0 + c;
      const bool ok = it < C::NITEM && gy < H && gx < W;
      s_src[h] = ok ? ((gy * W + gx) * Cin + rd * 8) : -1;
      s_dst[h] = it < C::NITEM ? rd * C::ROUNDB + (row * C::IXP + c) * POSB : -1;
    }
    float4 va[C::IPT], vb[C::IPT];
    auto issue = [&](int plane) {
      const bool pok = plane < D;
      const float* __restrict__ xp = x + (size_t)min(plane, D - 1) * H * W * Cin;
#pragma unroll
      for (int h = 0; h < C::IPT; ++h) {
        const bool ok = pok && s_src[h] >= 0;
        const float* src = xp + (ok ? s_src[h] : 0);
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        va[h] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        vb[h] = ok ? b : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto deposit = [&](int plane) {
      unsigned char* base = lds + (plane % C::NSLOT) * C::SLOTB;
#pragma unroll
      for (int h = 0; h < C::IPT; ++h)
        if (s_dst[h] >= 0) {
          if (F16) split_store8_f16(base + s_dst[h], va[h], vb[h], xs);
          else split_store8(base + s_dst[h], va[h], vb[h]);
        }
    };
    issue(a0);
    deposit(a0);
    issue(a0 + 1);
    deposit(a0 + 1);
    issue(a0 + 2);
    __syncthreads();                                  // #0: planes a0, a0 + 1 staged
    for (int a = a0; a < a1; ++a) {
      deposit(a + 2);                                 // slot of plane a - 1, which nobody reads any more
      issue(a + 3);
      __syncthreads();
    }
    return;
  }

  // ============================== consumers: wave = parity class ==============================
  const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
  if (pz) __builtin_amdgcn_s_setprio(1);         // twice the matrix work of its pz = 0 SIMD neighbour: 345 -> 329 us at M1
  const int j = lane & 15, g = lane >> 4;
  BV wlo[ROUNDS][3], whi[ROUNDS][3];                  // this class's weights: K-step at dz = 0 and (pz = 1) at dz = 1
  {
    const uint4* __restrict__ wp = wcls + (size_t)wave * ROUNDS * 2 * 3 * 64 + lane;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        wlo[rd][k].u = wp[((rd * 2 + 0) * 3 + k) * 64];
        whi[rd][k].u = wp[((rd * 2 + 1) * 3 + k) * 64];
      }
  }
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int b_off = (((g >> 1) * C::IXP) + j + (g & 1)) * POSB;       // slot g = (dy, dx)
  const int xv = 2 * (X0 + j) + px;
  const bool x_ok = X0 + j < W;
  const size_t zstride = (size_t)Ho * Wo * Cout;
  float amax = 0.f;                                   // split-f16: running maximum of the magnitudes this lane stores
  __syncthreads();                                    // #0
  for (int a = a0; a < a1; ++a) {
    const unsigned char* lo = lds + (a % C::NSLOT) * C::SLOTB + b_off;
    const unsigned char* hi = lds + ((a + 1) % C::NSLOT) * C::SLOTB + b_off;
    const size_t zbase = (size_t)(2 * a + pz) * zstride;
    const int nrows = min(C::TY, H - Y0);
    float4 skn = make_float4(0.f, 0.f, 0.f, 0.f);
    auto row_off = [&](int n) { return ((size_t)(2 * (Y0 + n) + py) * Wo + xv) * Cout + 4 * g; };
    if (skip && x_ok) skn = *reinterpret_cast<const float4*>(skip + zbase + row_off(0));
    asm volatile("" ::"v"(skn.x), "v"(skn.y), "v"(skn.z), "v"(skn.w));   // (the row loop's head then joins two states without pending loads)
    for (int n = 0; n < nrows; ++n) {
      const float4 sk = skn;
      if (skip && x_ok && n + 1 < nrows) skn = *reinterpret_cast<const float4*>(skip + zbase + row_off(n + 1));
      f32x4 acc[1] = {F16 ? (f32x4){0.f, 0.f, 0.f, 0.f} : (f32x4){bv.x, bv.y, bv.z, bv.w}};
      const unsigned char* lo_n = lo + n * C::IXP * POSB;
      const unsigned char* hi_n = hi + n * C::IXP * POSB;
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        BV b[1][3];
        b[0][0].u = *reinterpret_cast<const uint4*>(lo_n + rd * C::ROUNDB);
        b[0][1].u = *reinterpret_cast<const uint4*>(lo_n + rd * C::ROUNDB + 16);
        if constexpr (F16) {
          SF16_TERMS(acc, 0, 1, wlo[rd], b);
        } else {
          b[0][2].u = *reinterpret_cast<const uint4*>(lo_n + rd * C::ROUNDB + 32);
          SBF_TERMS(acc, 0, 1, wlo[rd], b);
        }
        if (pz) {
          BV c[1][3];
          c[0][0].u = *reinterpret_cast<const uint4*>(hi_n + rd * C::ROUNDB);
          c[0][1].u = *reinterpret_cast<const uint4*>(hi_n + rd * C::ROUNDB + 16);
          if constexpr (F16) {
            SF16_TERMS(acc, 0, 1, whi[rd], c);
          } else {
            c[0][2].u = *reinterpret_cast<const uint4*>(hi_n + rd * C::ROUNDB + 32);
            SBF_TERMS(acc, 0, 1, whi[rd], c);
          }
        }
      }
      // the next row's residual is taken delivery of BEFORE this row's store is issued (gfx9: one vmcnt for loads and stores, out of
      // order with respect to each other -> a wait for a load with a younger store in flight is a full drain)
      asm volatile("" ::"v"(skn.x), "v"(skn.y), "v"(skn.z), "v"(skn.w));
      if (x_ok) {
        const f32x4 r = acc[0];
        float4 o = F16 ? make_float4(r.x * out_mul + bv.x, r.y * out_mul + bv.y, r.z * out_mul + bv.z, r.w * out_mul + bv.w)
                       : make_float4(r.x, r.y, r.z, r.w);
        if (act == CDS_ACT_RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        if (skip) o = make_float4(sk.x + o.x, sk.y + o.y, sk.z + o.z, sk.w + o.w);
        if (F16) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        sbf_store4(out + zbase + row_off(n), o);
      }
    }
    __syncthreads();
  }
  if (F16) sf16_publish_bound(amax, out_bound);
}

}  // namespace

// ConvTranspose3d k3 s2 p1 op1 (+bias +ReLU +residual) 32 -> 16 in split-bf16 arithmetic, channels-last: x [D][H][W][32] ->
// out [2D][2H][2W][16]; weight_cls from ops.split_pack_deconv_cls (int16 [8 classes][4 rounds][2][3][64][8]).
static int dzm_entry(const float* x, const void* weight_cls, const float* bias, const float* skip, float* out, int Cin, int Cout, int D,
                     int H, int W, int act, const float* in_bound, float w_inv, float* out_bound, void* stream) {
  if (!x || !weight_cls || !out || Cin != 32 || Cout != 16 || D < 1 || H < 1 || W < 1) return CDS_EINVAL;
  if ((long)2 * H * 2 * W * Cout >= (1l << 31) || (long)H * W * Cin >= (1l << 31)) return CDS_EINVAL;   // in-plane offsets are 32-bit
  using C = DZC<4>;
  hipStream_t st = (hipStream_t)stream;
  const int tiles_x = cds_ceil_div(W, C::TX), tiles_y = cds_ceil_div(H, C::TY);
  const int ncols = tiles_x * tiles_y;
  const char* nseg_e = getenv("CDS_DZM_NSEG");   // A/B and test knob, read per launch
  int best = 1;
  double best_cost = 1e30;
  for (int n = 1; n <= min(D, 24); ++n) {        // whole rounds of 256 single-resident workgroups; 2 planes of priming per segment
    const int len = cds_ceil_div(D, n);
    const int n_eff = cds_ceil_div(D, len);
    const double cost = (double)cds_ceil_div(ncols * n_eff, 256) * (len + 1.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
  }
  if (nseg_e && atoi(nseg_e) > 0) best = min(atoi(nseg_e), D);
  const int seg_len = cds_ceil_div(D, best);
  const int nseg = cds_ceil_div(D, seg_len);
  if (in_bound) {
    static std::atomic<unsigned long long> lds_ok_h{0};
    if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(deconv3d_zm_kernel<4, true>), 160 * 1024, lds_ok_h)) return e_lds;
    hipLaunchKernelGGL((deconv3d_zm_kernel<4, true>), dim3(ncols * nseg), dim3(C::THREADS), C::LDS, st, x,
                       reinterpret_cast<const uint4*>(weight_cls), bias, skip, out, D, H, W, tiles_x, ncols, seg_len, act, in_bound, w_inv,
                       out_bound);
    return cds_launch_status();
  }
  static std::atomic<unsigned long long> lds_ok{0};
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(deconv3d_zm_kernel<4, false>), 160 * 1024, lds_ok)) return e_lds;
  hipLaunchKernelGGL((deconv3d_zm_kernel<4, false>), dim3(ncols * nseg), dim3(C::THREADS), C::LDS, st, x,
                     reinterpret_cast<const uint4*>(weight_cls), bias, skip, out, D, H, W, tiles_x, ncols, seg_len, act, nullptr, 1.0f,
                     nullptr);
  return cds_launch_status();
}

extern "C" int cds_deconv3d_zm_f32(const float* x, const void* weight_cls, const float* bias, const float* skip, float* out,
                                   int Cin, int Cout, int D, int H, int W, int act, void* stream) {
  return dzm_entry(x, weight_cls, bias, skip, out, Cin, Cout, D, H, W, act, nullptr, 1.0f, nullptr, stream);
}

// The same layer in SPLIT-F16 arithmetic (sbf_common.hpp): weight_cls from ops.split_pack_deconv_cls(..., f16=True), w_inv_scale = 1 / its
// weight scale, in_bound a DEVICE scalar >= max |x|, out_bound a zeroed DEVICE scalar that receives max |out| (or NULL).
extern "C" int cds_deconv3d_zm_sf16_f32(const float* x, const void* weight_cls, const float* bias, const float* skip, float* out,
                                        int Cin, int Cout, int D, int H, int W, int act, const float* in_bound, float w_inv_scale,
                                        float* out_bound, void* stream) {
  if (!in_bound || !(w_inv_scale > 0.f)) return CDS_EINVAL;
  return dzm_entry(x, weight_cls, bias, skip, out, Cin, Cout, D, H, W, act, in_bound, w_inv_scale, out_bound, stream);
}
