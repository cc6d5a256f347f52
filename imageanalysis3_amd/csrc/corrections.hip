// Elementwise pre-corrections of correct_fov_image (reference: io_tools/load.py:337-384):
//   z-shift       corrections.py:479-487  out = u16( f32(im) / median_z * median_all )        (float32 arithmetic)
//   bleedthrough  io_tools/load.py:348-370 out_i = u16(clip( sum_j im_j * P[i,j] ))            (P (C,C,X,Y) f32|f64)
//   illumination  io_tools/load.py:373-384 out = u16( f32(im) / P[ch] )                        (P (X,Y) f32|f64)
// NumPy semantics kept: products/quotients are float32 when the profile is float32 and float64 when it is
// float64; sums run in channel order; float -> uint16 is C truncation (out-of-range values wrap as on x86).
// All three are single-pass HBM-bound streams; the medians use a 3-pass radix select (11+11+10 bits of the
// order-preserving uint32 key of the float32 value), batched over the Z planes plus the whole stack, for the
// two middle ranks NumPy averages.
#include "ia3_rt.h"
#include <math.h>
#include <vector>

using namespace ia3rt;

namespace {

__device__ __forceinline__ uint16_t to_u16(double t) {   // numpy .astype(np.uint16) of a float on x86-64
  if (!(fabs(t) < 2147483648.0)) return 0;   // cvttss2si/cvttsd2si r32: out-of-range -> INT_MIN -> low 16 bits 0
  return (uint16_t)(int)t;
}
template <class T> __device__ __forceinline__ float ldf(const T* p, size_t i) { return (float)p[i]; }

__device__ __forceinline__ uint32_t fkey(float v) {      // order-preserving key
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

constexpr int RB = 2048;  // buckets per pass (11 bits; last pass uses 10)
struct SelState {         // per problem (plane z or whole stack) and rank r
  uint32_t prefix;        // key bits decided so far (high bits)
  unsigned long long k;   // remaining rank inside the current prefix group
};

// pass p: histogram of the next digit among voxels whose decided high bits match the problem's prefix.
// problems: index 2*z + r for planes, 2*Z + r for the whole stack.
template <class T>
__global__ __launch_bounds__(256) void select_hist_k(const T* __restrict__ im, int Z, size_t plane, int pass,
                                                     const SelState* __restrict__ st, unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[4][RB];
  const int z = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * RB; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  const uint32_t dmask = pass == 2 ? 0x3ffu : 0x7ffu;
  const int hi_shift = pass == 0 ? 32 : (pass == 1 ? 21 : 10);
  uint32_t pre[4];
  pre[0] = st[2 * z].prefix; pre[1] = st[2 * z + 1].prefix; pre[2] = st[2 * Z].prefix; pre[3] = st[2 * Z + 1].prefix;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    const uint32_t key = fkey(ldf(im, (size_t)z * plane + i));
    const uint32_t d = (key >> shift) & dmask;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bool match = hi_shift >= 32 ? true : ((key >> hi_shift) == (pre[q] >> hi_shift));
      if (match) atomicAdd(&h[q][d], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RB; i += 256) {
    if (h[0][i]) atomicAdd(&hist[((size_t)(2 * z) * RB) + i], h[0][i]);
    if (h[1][i]) atomicAdd(&hist[((size_t)(2 * z + 1) * RB) + i], h[1][i]);
    if (h[2][i]) atomicAdd(&hist[((size_t)(2 * Z) * RB) + i], h[2][i]);
    if (h[3][i]) atomicAdd(&hist[((size_t)(2 * Z + 1) * RB) + i], h[3][i]);
  }
}
// one thread per problem: pick the bucket holding rank k, extend the prefix
__global__ void select_pick_k(SelState* st, const unsigned int* __restrict__ hist, int n_prob, int pass) {
  int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_prob) return;
  const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  const int nb = pass == 2 ? 1024 : RB;
  unsigned long long k = st[q].k, cum = 0;
  int b = 0;
  for (; b < nb; ++b) {
    unsigned long long c = hist[(size_t)q * RB + b];
    if (cum + c > k) break;
    cum += c;
  }
  if (b >= nb) b = nb - 1;
  st[q].prefix |= (uint32_t)b << shift;
  st[q].k = k - cum;
}
// medians[z] (Z planes) and medians[Z] (whole stack): float32 mean of the two middle order statistics
__global__ void select_finish_k(const SelState* __restrict__ st, int Z, float* __restrict__ med) {
  int z = blockIdx.x * 64 + threadIdx.x;
  if (z > Z) return;
  float a = fkey_inv(st[2 * z].prefix), b = fkey_inv(st[2 * z + 1].prefix);
  med[z] = (a + b) / 2.0f;
}

template <class T>
__global__ __launch_bounds__(256) void zshift_apply_k(const T* __restrict__ im, int Z, size_t plane,
                                                      const float* __restrict__ med, uint16_t* __restrict__ out) {
  const int z = blockIdx.y;
  const float mz = med[z], mall = med[Z];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    float v = ldf(im, (size_t)z * plane + i);
    out[(size_t)z * plane + i] = to_u16((double)((v / mz) * mall));
  }
}

template <class P>
__global__ __launch_bounds__(256) void illum_k(const uint16_t* __restrict__ im, int Z, size_t plane,
                                               const P* __restrict__ prof, uint16_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    const P pv = prof[i];
    for (int z = 0; z < Z; ++z) {
      // float32(im) / profile : float32 for a float32 profile, float64 for a float64 profile (NumPy promotion)
      P q = (P)(float)im[(size_t)z * plane + i] / pv;
      out[(size_t)z * plane + i] = to_u16((double)q);
    }
  }
}

constexpr int MAXC = 8;
struct ChanPtrs { const uint16_t* in[MAXC]; uint16_t* out[MAXC]; };

template <class P>
__global__ __launch_bounds__(256) void bleed_k(ChanPtrs ch, int C, int Z, size_t plane, const P* __restrict__ prof) {
  // prof[(i*C + j) * plane + xy]
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    for (int z = 0; z < Z; ++z) {
      P v[MAXC];
      for (int j = 0; j < C; ++j) v[j] = (P)ch.in[j][(size_t)z * plane + i];
      for (int a = 0; a < C; ++a) {
        P s = v[0] * prof[((size_t)a * C) * plane + i];
        for (int j = 1; j < C; ++j) s = s + v[j] * prof[((size_t)a * C + j) * plane + i];
        s = s > (P)65535 ? (P)65535 : s;
        s = s < (P)0 ? (P)0 : s;
        ch.out[a][(size_t)z * plane + i] = to_u16((double)s);
      }
    }
  }
}


// Four neighbouring voxels per thread (8-byte uint16 loads / stores, 16-byte profile loads) with the profile values
// held in registers across the z loop: the one-voxel-per-lane kernels above moved 128 bytes per wave instruction and
// re-read the nine profile planes for every z (2.6 ms for three 2048 x 2048 x 50 channels = 1 TB/s).
struct alignas(8) U16x4 { uint16_t v[4]; };
template <class P> struct alignas(sizeof(P) * 4) Px4 { P v[4]; };

template <class P>
__global__ __launch_bounds__(256) void illum4_k(const uint16_t* __restrict__ im, int Z, size_t plane,
                                                const P* __restrict__ prof, uint16_t* __restrict__ out) {
  const size_t nq = plane / 4;
  for (size_t qd = (size_t)blockIdx.x * 256 + threadIdx.x; qd < nq; qd += (size_t)gridDim.x * 256) {
    const size_t i = qd * 4;
    const Px4<P> pv = *reinterpret_cast<const Px4<P>*>(prof + i);
#pragma unroll 2
    for (int z = 0; z < Z; ++z) {
      const U16x4 a = *reinterpret_cast<const U16x4*>(im + (size_t)z * plane + i);
      U16x4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[k] = to_u16((double)((P)(float)a.v[k] / pv.v[k]));
      *reinterpret_cast<U16x4*>(out + (size_t)z * plane + i) = r;
    }
  }
}

template <class P>
__global__ __launch_bounds__(256) void bleed3x4_k(ChanPtrs ch, int Z, size_t plane, const P* __restrict__ prof) {
  const size_t nq = plane / 4;
  for (size_t qd = (size_t)blockIdx.x * 256 + threadIdx.x; qd < nq; qd += (size_t)gridDim.x * 256) {
    const size_t i = qd * 4;
    Px4<P> pf[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 3; ++j) pf[a][j] = *reinterpret_cast<const Px4<P>*>(prof + ((size_t)a * 3 + j) * plane + i);
#pragma unroll 2
    for (int z = 0; z < Z; ++z) {
      U16x4 in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) in[j] = *reinterpret_cast<const U16x4*>(ch.in[j] + (size_t)z * plane + i);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        U16x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          P sum = (P)in[0].v[k] * pf[a][0].v[k];
          sum = sum + (P)in[1].v[k] * pf[a][1].v[k];
          sum = sum + (P)in[2].v[k] * pf[a][2].v[k];
          sum = sum > (P)65535 ? (P)65535 : sum;
          sum = sum < (P)0 ? (P)0 : sum;
          r.v[k] = to_u16((double)sum);
        }
        *reinterpret_cast<U16x4*>(ch.out[a] + (size_t)z * plane + i) = r;
      }
    }
  }
}

// prof_dtype: 1 float32, else float64.  The four-wide kernels need plane % 4 == 0 (then every row start is aligned).
static void launch_illum(int prof_dtype, unsigned gx, hipStream_t st, const uint16_t* im, int Z, size_t plane, const void* prof,
                         uint16_t* out) {
  const bool wide = plane % 4 == 0 && ((uintptr_t)prof % 32) == 0 && ((uintptr_t)im % 8) == 0 && ((uintptr_t)out % 8) == 0;
  if (prof_dtype == 1) {
    if (wide) hipLaunchKernelGGL((illum4_k<float>), dim3(gx), dim3(256), 0, st, im, Z, plane, (const float*)prof, out);
    else hipLaunchKernelGGL((illum_k<float>), dim3(gx), dim3(256), 0, st, im, Z, plane, (const float*)prof, out);
  } else {
    if (wide) hipLaunchKernelGGL((illum4_k<double>), dim3(gx), dim3(256), 0, st, im, Z, plane, (const double*)prof, out);
    else hipLaunchKernelGGL((illum_k<double>), dim3(gx), dim3(256), 0, st, im, Z, plane, (const double*)prof, out);
  }
}
static void launch_bleed(int prof_dtype, unsigned gx, hipStream_t st, const ChanPtrs& ch, int C, int Z, size_t plane, const void* prof) {
  bool wide = C == 3 && plane % 4 == 0 && ((uintptr_t)prof % 32) == 0;
  for (int j = 0; j < C && wide; ++j) wide = ((uintptr_t)ch.in[j] % 8) == 0 && ((uintptr_t)ch.out[j] % 8) == 0;
  if (prof_dtype == 1) {
    if (wide) hipLaunchKernelGGL((bleed3x4_k<float>), dim3(gx), dim3(256), 0, st, ch, Z, plane, (const float*)prof);
    else hipLaunchKernelGGL((bleed_k<float>), dim3(gx), dim3(256), 0, st, ch, C, Z, plane, (const float*)prof);
  } else {
    if (wide) hipLaunchKernelGGL((bleed3x4_k<double>), dim3(gx), dim3(256), 0, st, ch, Z, plane, (const double*)prof);
    else hipLaunchKernelGGL((bleed_k<double>), dim3(gx), dim3(256), 0, st, ch, C, Z, plane, (const double*)prof);
  }
}

// ---- DaxProcesser variants (classes/preprocess.py:464-680): same stages with a min-max rescale to the full uint16
// range.  Two passes over the inputs: (1) the corrected value of every voxel is formed and only its min / max are
// kept (fixed grid of partials + one block; min/max do not depend on the order), (2) it is formed again, rescaled
// exactly as NumPy evaluates `(im - min) / (max - min) * 65535 + 0`, clipped and truncated to uint16.
constexpr int MM_BLOCKS = 1024;
template <class P> __device__ __forceinline__ P illum_val(const uint16_t* im, const P* prof, size_t z, size_t plane, size_t i) {
  return (P)(float)im[z * plane + i] / prof[i];          // im.astype(np.float32) / pf[None]: float32, or float64 for a float64 profile
}
template <class P> __device__ __forceinline__ double bleed_val(const ChanPtrs& ch, int C, int a, const P* prof, size_t z, size_t plane, size_t i) {
  double acc = 0.0;                                     // np.zeros(image_size): float64 accumulator (:511)
  for (int j = 0; j < C; ++j) acc = acc + (double)((P)ch.in[j][z * plane + i] * prof[((size_t)a * C + j) * plane + i]);
  return acc;
}
struct MinMax { double mn, mx; };
template <class V> __device__ __forceinline__ void block_minmax(V mn, V mx, MinMax* part) {
  __shared__ double smn[256], smx[256];
  smn[threadIdx.x] = (double)mn; smx[threadIdx.x] = (double)mx;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      smn[threadIdx.x] = smn[threadIdx.x + k] < smn[threadIdx.x] ? smn[threadIdx.x + k] : smn[threadIdx.x];
      smx[threadIdx.x] = smx[threadIdx.x + k] > smx[threadIdx.x] ? smx[threadIdx.x + k] : smx[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x].mn = smn[0]; part[blockIdx.x].mx = smx[0]; }
}
template <class P>
__global__ __launch_bounds__(256) void illum_minmax_k(const uint16_t* __restrict__ im, int Z, size_t plane, const P* __restrict__ prof, MinMax* part) {
  P mn = INFINITY, mx = -INFINITY;
  const size_t n = plane * Z;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)MM_BLOCKS * 256) {
    const P q = illum_val<P>(im, prof, v / plane, plane, v % plane);
    mn = q < mn ? q : mn; mx = q > mx ? q : mx;
  }
  block_minmax(mn, mx, part);
}
template <class P>
__global__ __launch_bounds__(256) void bleed_minmax_k(ChanPtrs ch, int C, int a, int Z, size_t plane, const P* __restrict__ prof, MinMax* part) {
  double mn = INFINITY, mx = -INFINITY;
  const size_t n = plane * Z;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)MM_BLOCKS * 256) {
    const double q = bleed_val<P>(ch, C, a, prof, v / plane, plane, v % plane);
    mn = q < mn ? q : mn; mx = q > mx ? q : mx;
  }
  block_minmax(mn, mx, part);
}
__global__ __launch_bounds__(1024) void minmax_final_k(const MinMax* __restrict__ part, MinMax* out) {
  __shared__ double smn[1024], smx[1024];
  smn[threadIdx.x] = threadIdx.x < MM_BLOCKS ? part[threadIdx.x].mn : INFINITY;
  smx[threadIdx.x] = threadIdx.x < MM_BLOCKS ? part[threadIdx.x].mx : -INFINITY;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      smn[threadIdx.x] = smn[threadIdx.x + k] < smn[threadIdx.x] ? smn[threadIdx.x + k] : smn[threadIdx.x];
      smx[threadIdx.x] = smx[threadIdx.x + k] > smx[threadIdx.x] ? smx[threadIdx.x + k] : smx[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out->mn = smn[0]; out->mx = smx[0]; }
}
template <class V> __device__ __forceinline__ uint16_t rescale_store(V q, V mn, V mx, int rescale) {
  if (rescale) { q = q - mn; q = q / (mx - mn); q = q * (V)65535; q = q + (V)0; }
  q = q < (V)0 ? (V)0 : q;            // np.clip(a_min=0, a_max=65535); NaN passes through as in NumPy
  q = q > (V)65535 ? (V)65535 : q;
  return to_u16((double)q);
}
template <class P>
__global__ __launch_bounds__(256) void illum_rescale_k(const uint16_t* __restrict__ im, int Z, size_t plane, const P* __restrict__ prof,
                                                       const MinMax* __restrict__ mm, int rescale, uint16_t* __restrict__ out) {
  const P mn = (P)mm->mn, mx = (P)mm->mx;
  const size_t n = plane * Z;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
    out[v] = rescale_store<P>(illum_val<P>(im, prof, v / plane, plane, v % plane), mn, mx, rescale);
}
template <class P>
__global__ __launch_bounds__(256) void bleed_rescale_k(ChanPtrs ch, int C, int a, int Z, size_t plane, const P* __restrict__ prof,
                                                       const MinMax* __restrict__ mm, int rescale) {
  const double mn = mm->mn, mx = mm->mx;
  const size_t n = plane * Z;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
    ch.out[a][v] = rescale_store<double>(bleed_val<P>(ch, C, a, prof, v / plane, plane, v % plane), mn, mx, rescale);
}

}  // namespace

namespace ia3k {

// med[z] for the Z planes and med[Z] for the whole stack (device array of Z+1 floats): np.median in float32
int stack_medians(const ia3_stack* s, float* dmed) {
  hipStream_t st = stream();
  const int Z = s->Z;
  const size_t plane = (size_t)s->X * s->Y, n = plane * Z;
  const int n_prob = 2 * (Z + 1);
  Scratch dst((size_t)n_prob * sizeof(SelState)), dh((size_t)n_prob * RB * sizeof(unsigned int));
  if (!dst.p || !dh.p) return IA3_ENOMEM;
  std::vector<SelState> hs(n_prob);
  for (int z = 0; z <= Z; ++z) {
    const unsigned long long cnt = z < Z ? plane : n;
    hs[2 * z] = SelState{0u, (cnt - 1) / 2};
    hs[2 * z + 1] = SelState{0u, cnt / 2};
  }
  hipError_t e = hipMemcpyAsync(dst.p, hs.data(), hs.size() * sizeof(SelState), hipMemcpyHostToDevice, st);
  unsigned gx = (unsigned)((plane + 256 * 16 - 1) / (256 * 16));
  if (gx < 1) gx = 1;
  {
    ProfScope ps("zshift_median");
    for (int pass = 0; pass < 3 && e == hipSuccess; ++pass) {
      e = hipMemsetAsync(dh.p, 0, (size_t)n_prob * RB * sizeof(unsigned int), st);
      if (s->dtype == IA3_F32) hipLaunchKernelGGL((select_hist_k<float>), dim3(gx, Z), dim3(256), 0, st, (const float*)s->d, Z, plane, pass, (const SelState*)dst.p, dh.as<unsigned int>());
      else hipLaunchKernelGGL((select_hist_k<uint16_t>), dim3(gx, Z), dim3(256), 0, st, (const uint16_t*)s->d, Z, plane, pass, (const SelState*)dst.p, dh.as<unsigned int>());
      hipLaunchKernelGGL(select_pick_k, dim3((n_prob + 63) / 64), dim3(64), 0, st, dst.as<SelState>(), (const unsigned int*)dh.p, n_prob, pass);
    }
    hipLaunchKernelGGL(select_finish_k, dim3((Z + 64) / 64), dim3(64), 0, st, (const SelState*)dst.p, Z, dmed);
  }
  if (e == hipSuccess) e = hipGetLastError();
  // hs is read by the (pageable, hence already staged) H2D copy above; nothing else host-side is pending
  if (e != hipSuccess) return set_error(IA3_EHIP, "median selection failed: %s", hipGetErrorString(e));
  return IA3_OK;
}

int stack_median_all(const ia3_stack* s, float* med_all) {
  Scratch dmed((size_t)(s->Z + 1) * sizeof(float));
  if (!dmed.p) return IA3_ENOMEM;
  int rc = stack_medians(s, dmed.as<float>()); if (rc) return rc;
  IA3_HIP(hipMemcpyAsync(med_all, dmed.as<float>() + s->Z, sizeof(float), hipMemcpyDeviceToHost, stream()));
  IA3_HIP(hipStreamSynchronize(stream()));
  return IA3_OK;
}

}  // namespace ia3k

extern "C" {

// corrections.py:479-487 Z_Shift_Correction(im.astype(float32), dtype=uint16): out (uint16) = im / median_z * median
int ia3_z_shift_correction(const void* im, int dtype, int Z, int X, int Y, void* out_u16, float* medians_out) {
  ia3_stack* s = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  hipStream_t st = stream();
  const size_t plane = (size_t)X * Y, n = plane * Z;
  Scratch dmed((size_t)(Z + 1) * sizeof(float)), dout(n * sizeof(uint16_t));
  if (!dmed.p || !dout.p) { ia3_stack_free(s); return IA3_ENOMEM; }
  rc = ia3k::stack_medians(s, dmed.as<float>());
  if (rc) { ia3_stack_free(s); return rc; }
  unsigned gx = (unsigned)((plane + 256 * 16 - 1) / (256 * 16));
  if (gx < 1) gx = 1;
  {
    ProfScope ps("zshift_apply");
    if (dtype == IA3_F32) hipLaunchKernelGGL((zshift_apply_k<float>), dim3(gx, Z), dim3(256), 0, st, (const float*)s->d, Z, plane, (const float*)dmed.p, dout.as<uint16_t>());
    else hipLaunchKernelGGL((zshift_apply_k<uint16_t>), dim3(gx, Z), dim3(256), 0, st, (const uint16_t*)s->d, Z, plane, (const float*)dmed.p, dout.as<uint16_t>());
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out_u16, dout.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && medians_out) e = hipMemcpyAsync(medians_out, dmed.p, (size_t)(Z + 1) * sizeof(float), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  ia3_stack_free(s);
  if (e != hipSuccess) return set_error(IA3_EHIP, "z-shift correction failed: %s", hipGetErrorString(e));
  return IA3_OK;
}

// io_tools/load.py:373-384: out = (im.astype(float32) / profile[None]).astype(uint16); profile (X,Y) f32 (1) or f64 (2)
int ia3_illumination_correct(const void* im_u16, int Z, int X, int Y, const void* profile, int prof_dtype, void* out_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im_u16 || !profile || !out_u16) return set_error(IA3_EINVAL, "null argument");
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  hipStream_t st = stream();
  const size_t plane = (size_t)X * Y, n = plane * Z, pb = plane * (prof_dtype == 1 ? 4 : 8);
  Scratch din(n * 2), dout(n * 2), dp(pb);
  if (!din.p || !dout.p || !dp.p) return IA3_ENOMEM;
  IA3_HIP(hipMemcpyAsync(din.p, im_u16, n * 2, hipMemcpyHostToDevice, st));
  IA3_HIP(hipMemcpyAsync(dp.p, profile, pb, hipMemcpyHostToDevice, st));
  unsigned gx = (unsigned)((plane + 255) / 256);
  {
    ProfScope ps("illumination");
    launch_illum(prof_dtype, gx, st, din.as<uint16_t>(), Z, plane, dp.p, dout.as<uint16_t>());
  }
  IA3_KCHECK();
  IA3_HIP(hipMemcpyAsync(out_u16, dout.p, n * 2, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}

// io_tools/load.py:348-370: out_i = clip(sum_j ims_j * profile[i,j]).astype(uint16); profile (C,C,X,Y)
int ia3_bleedthrough_correct(const void* const* ims_u16, int C, int Z, int X, int Y, const void* profile, int prof_dtype,
                             void* const* outs_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!ims_u16 || !profile || !outs_u16) return set_error(IA3_EINVAL, "null argument");
  if (C < 1 || C > MAXC) return set_error(IA3_EUNSUPPORTED, "1..%d channels supported, got %d", MAXC, C);
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  hipStream_t st = stream();
  const size_t plane = (size_t)X * Y, n = plane * Z, pb = plane * C * C * (prof_dtype == 1 ? 4 : 8);
  Scratch din(n * 2 * C), dout(n * 2 * C), dp(pb);
  if (!din.p || !dout.p || !dp.p) return IA3_ENOMEM;
  ChanPtrs ch;
  for (int j = 0; j < C; ++j) {
    if (!ims_u16[j] || !outs_u16[j]) return set_error(IA3_EINVAL, "null channel pointer");
    ch.in[j] = din.as<uint16_t>() + (size_t)j * n;
    ch.out[j] = dout.as<uint16_t>() + (size_t)j * n;
    IA3_HIP(hipMemcpyAsync((void*)ch.in[j], ims_u16[j], n * 2, hipMemcpyHostToDevice, st));
  }
  IA3_HIP(hipMemcpyAsync(dp.p, profile, pb, hipMemcpyHostToDevice, st));
  unsigned gx = (unsigned)((plane + 255) / 256);
  {
    ProfScope ps("bleedthrough");
    launch_bleed(prof_dtype, gx, st, ch, C, Z, plane, dp.p);
  }
  IA3_KCHECK();
  for (int j = 0; j < C; ++j) IA3_HIP(hipMemcpyAsync(outs_u16[j], ch.out[j], n * 2, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}


// ---- device-resident variants (the chain of io_tools/load.py:323-384 on stacks that stay in HBM) ---------------
// profiles are device buffers made by ia3_buffer_upload (constant over a run: uploaded once)

int ia3_z_shift_correction_dev(const ia3_stack* im, ia3_stack* out_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !out_u16) return set_error(IA3_EINVAL, "null stack");
  if (out_u16->dtype != IA3_U16 || out_u16->Z != im->Z || out_u16->X != im->X || out_u16->Y != im->Y)
    return set_error(IA3_EINVAL, "output must be a uint16 stack of the input's shape");
  hipStream_t st = stream();
  const int Z = im->Z;
  const size_t plane = (size_t)im->X * im->Y;
  Scratch dmed((size_t)(Z + 1) * sizeof(float));
  if (!dmed.p) return IA3_ENOMEM;
  rc = ia3k::stack_medians(im, dmed.as<float>()); if (rc) return rc;
  unsigned gx = (unsigned)((plane + 256 * 16 - 1) / (256 * 16));
  if (gx < 1) gx = 1;
  {
    ProfScope ps("zshift_apply");
    if (im->dtype == IA3_F32) hipLaunchKernelGGL((zshift_apply_k<float>), dim3(gx, Z), dim3(256), 0, st, (const float*)im->d, Z, plane, (const float*)dmed.p, (uint16_t*)out_u16->d);
    else hipLaunchKernelGGL((zshift_apply_k<uint16_t>), dim3(gx, Z), dim3(256), 0, st, (const uint16_t*)im->d, Z, plane, (const float*)dmed.p, (uint16_t*)out_u16->d);
  }
  IA3_KCHECK();
  IA3_HIP(hipStreamSynchronize(st));   // dmed returns to the pool
  return IA3_OK;
}

int ia3_illumination_correct_dev(const ia3_stack* im_u16, const void* profile_dev, int prof_dtype, ia3_stack* out_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im_u16 || !profile_dev || !out_u16) return set_error(IA3_EINVAL, "null argument");
  if (im_u16->dtype != IA3_U16 || out_u16->dtype != IA3_U16) return set_error(IA3_EINVAL, "uint16 stacks expected");
  if (out_u16->Z != im_u16->Z || out_u16->X != im_u16->X || out_u16->Y != im_u16->Y) return set_error(IA3_EINVAL, "shape mismatch");
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  const size_t plane = (size_t)im_u16->X * im_u16->Y;
  unsigned gx = (unsigned)((plane + 255) / 256);
  ProfScope ps("illumination");
  launch_illum(prof_dtype, gx, stream(), (const uint16_t*)im_u16->d, im_u16->Z, plane, profile_dev, (uint16_t*)out_u16->d);
  IA3_KCHECK();
  return IA3_OK;
}

int ia3_bleedthrough_correct_dev(ia3_stack* const* ims_u16, int C, const void* profile_dev, int prof_dtype,
                                 ia3_stack* const* outs_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!ims_u16 || !profile_dev || !outs_u16) return set_error(IA3_EINVAL, "null argument");
  if (C < 1 || C > MAXC) return set_error(IA3_EUNSUPPORTED, "1..%d channels supported, got %d", MAXC, C);
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  ChanPtrs ch;
  for (int j = 0; j < C; ++j) {
    if (!ims_u16[j] || !outs_u16[j]) return set_error(IA3_EINVAL, "null channel");
    if (ims_u16[j]->dtype != IA3_U16 || outs_u16[j]->dtype != IA3_U16) return set_error(IA3_EINVAL, "uint16 stacks expected");
    if (ims_u16[j]->Z != ims_u16[0]->Z || ims_u16[j]->X != ims_u16[0]->X || ims_u16[j]->Y != ims_u16[0]->Y ||
        outs_u16[j]->Z != ims_u16[0]->Z || outs_u16[j]->X != ims_u16[0]->X || outs_u16[j]->Y != ims_u16[0]->Y)
      return set_error(IA3_EINVAL, "shape mismatch");
    for (int k = 0; k < C; ++k)
      if (outs_u16[j]->d == ims_u16[k]->d) return set_error(IA3_EINVAL, "outputs must not alias inputs (every output mixes all inputs)");
    ch.in[j] = (const uint16_t*)ims_u16[j]->d;
    ch.out[j] = (uint16_t*)outs_u16[j]->d;
  }
  const size_t plane = (size_t)ims_u16[0]->X * ims_u16[0]->Y;
  unsigned gx = (unsigned)((plane + 255) / 256);
  ProfScope ps("bleedthrough");
  launch_bleed(prof_dtype, gx, stream(), ch, C, ims_u16[0]->Z, plane, profile_dev);
  IA3_KCHECK();
  return IA3_OK;
}


// classes/preprocess.py:605-680 DaxProcesser._corr_illumination: float32 (float64 for a float64 profile) quotient,
// optional rescale of [min, max] to [0, 65535], clip, truncation.  In place allowed.
int ia3_illumination_rescale_dev(const ia3_stack* im_u16, const void* profile_dev, int prof_dtype, int rescale, ia3_stack* out_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im_u16 || !profile_dev || !out_u16) return set_error(IA3_EINVAL, "null argument");
  if (im_u16->dtype != IA3_U16 || out_u16->dtype != IA3_U16) return set_error(IA3_EINVAL, "uint16 stacks expected");
  if (out_u16->Z != im_u16->Z || out_u16->X != im_u16->X || out_u16->Y != im_u16->Y) return set_error(IA3_EINVAL, "shape mismatch");
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  hipStream_t st = stream();
  const int Z = im_u16->Z;
  const size_t plane = (size_t)im_u16->X * im_u16->Y, n = plane * Z;
  Scratch part((MM_BLOCKS + 1) * sizeof(MinMax));
  if (!part.p) return IA3_ENOMEM;
  MinMax* pm = part.as<MinMax>();
  unsigned blocks = (unsigned)((n + 255) / 256); if (blocks > 256 * 32) blocks = 256 * 32;
  ProfScope ps("illumination_rescale");
  if (prof_dtype == 1) {
    if (rescale) hipLaunchKernelGGL((illum_minmax_k<float>), dim3(MM_BLOCKS), dim3(256), 0, st, (const uint16_t*)im_u16->d, Z, plane, (const float*)profile_dev, pm);
    if (rescale) hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(1024), 0, st, (const MinMax*)pm, pm + MM_BLOCKS);
    hipLaunchKernelGGL((illum_rescale_k<float>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)im_u16->d, Z, plane, (const float*)profile_dev, (const MinMax*)(pm + MM_BLOCKS), rescale, (uint16_t*)out_u16->d);
  } else {
    if (rescale) hipLaunchKernelGGL((illum_minmax_k<double>), dim3(MM_BLOCKS), dim3(256), 0, st, (const uint16_t*)im_u16->d, Z, plane, (const double*)profile_dev, pm);
    if (rescale) hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(1024), 0, st, (const MinMax*)pm, pm + MM_BLOCKS);
    hipLaunchKernelGGL((illum_rescale_k<double>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)im_u16->d, Z, plane, (const double*)profile_dev, (const MinMax*)(pm + MM_BLOCKS), rescale, (uint16_t*)out_u16->d);
  }
  IA3_KCHECK();
  return IA3_OK;
}

// classes/preprocess.py:464-541 DaxProcesser._corr_bleedthrough: float64 accumulation of the float32 (profile dtype)
// products in channel order, optional rescale, clip, truncation.  Outputs must not alias inputs.
int ia3_bleedthrough_rescale_dev(ia3_stack* const* ims_u16, int C, const void* profile_dev, int prof_dtype, int rescale,
                                 ia3_stack* const* outs_u16) {
  int rc = ensure_init(); if (rc) return rc;
  if (!ims_u16 || !profile_dev || !outs_u16) return set_error(IA3_EINVAL, "null argument");
  if (C < 1 || C > MAXC) return set_error(IA3_EUNSUPPORTED, "1..%d channels supported, got %d", MAXC, C);
  if (prof_dtype != 1 && prof_dtype != 2) return set_error(IA3_EINVAL, "profile dtype must be float32 (1) or float64 (2)");
  ChanPtrs ch;
  for (int j = 0; j < C; ++j) {
    if (!ims_u16[j] || !outs_u16[j]) return set_error(IA3_EINVAL, "null channel");
    if (ims_u16[j]->dtype != IA3_U16 || outs_u16[j]->dtype != IA3_U16) return set_error(IA3_EINVAL, "uint16 stacks expected");
    if (ims_u16[j]->Z != ims_u16[0]->Z || ims_u16[j]->X != ims_u16[0]->X || ims_u16[j]->Y != ims_u16[0]->Y ||
        outs_u16[j]->Z != ims_u16[0]->Z || outs_u16[j]->X != ims_u16[0]->X || outs_u16[j]->Y != ims_u16[0]->Y)
      return set_error(IA3_EINVAL, "shape mismatch");
    for (int k = 0; k < C; ++k)
      if (outs_u16[j]->d == ims_u16[k]->d) return set_error(IA3_EINVAL, "outputs must not alias inputs (every output mixes all inputs)");
    ch.in[j] = (const uint16_t*)ims_u16[j]->d;
    ch.out[j] = (uint16_t*)outs_u16[j]->d;
  }
  hipStream_t st = stream();
  const int Z = ims_u16[0]->Z;
  const size_t plane = (size_t)ims_u16[0]->X * ims_u16[0]->Y, n = plane * Z;
  Scratch part((MM_BLOCKS + 1) * sizeof(MinMax));
  if (!part.p) return IA3_ENOMEM;
  MinMax* pm = part.as<MinMax>();
  unsigned blocks = (unsigned)((n + 255) / 256); if (blocks > 256 * 32) blocks = 256 * 32;
  ProfScope ps("bleedthrough_rescale");
  for (int a = 0; a < C; ++a) {
    if (prof_dtype == 1) {
      if (rescale) hipLaunchKernelGGL((bleed_minmax_k<float>), dim3(MM_BLOCKS), dim3(256), 0, st, ch, C, a, Z, plane, (const float*)profile_dev, pm);
      if (rescale) hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(1024), 0, st, (const MinMax*)pm, pm + MM_BLOCKS);
      hipLaunchKernelGGL((bleed_rescale_k<float>), dim3(blocks), dim3(256), 0, st, ch, C, a, Z, plane, (const float*)profile_dev, (const MinMax*)(pm + MM_BLOCKS), rescale);
    } else {
      if (rescale) hipLaunchKernelGGL((bleed_minmax_k<double>), dim3(MM_BLOCKS), dim3(256), 0, st, ch, C, a, Z, plane, (const double*)profile_dev, pm);
      if (rescale) hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(1024), 0, st, (const MinMax*)pm, pm + MM_BLOCKS);
      hipLaunchKernelGGL((bleed_rescale_k<double>), dim3(blocks), dim3(256), 0, st, ch, C, a, Z, plane, (const double*)profile_dev, (const MinMax*)(pm + MM_BLOCKS), rescale);
    }
  }
  IA3_KCHECK();
  IA3_HIP(hipStreamSynchronize(st));   // `part` goes back to the pool
  return IA3_OK;
}

}  // extern "C"
