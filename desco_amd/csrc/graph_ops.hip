// HBM-bound row kernels of the DeSCo hot path (gfx950): CSR gather-aggregate, segmented pooling,
// count head, row scatter, small-K input projection.  One 64-lane wavefront per destination row;
// feature rows are 64 fp32 = 256 B, read and written as whole coalesced rows.
#include "common_device.hpp"

namespace desco {

template <int S>
struct VecOf;
template <>
struct VecOf<1> {
  using type = float;
};
template <>
struct VecOf<2> {
  using type = float2;
};
template <>
struct VecOf<4> {
  using type = float4;
};

__device__ __forceinline__ void vadd(float& a, const float b) { a += b; }
__device__ __forceinline__ void vadd(float2& a, const float2 b) {
  a.x += b.x;
  a.y += b.y;
}
__device__ __forceinline__ void vadd(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }
__device__ __forceinline__ void vzero(float2& a) { a = make_float2(0.f, 0.f); }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

// SAGEConv message + aggregate (gnn_model.py:392-394, 402-404) for S relation slots per row.
// Batched, branch-free form of the gather (the lessons of shmp_layer.hip applied to the stand-alone
// kernel of the training path): a 16-lane group (float4 per lane) owns one destination row, a wave
// four rows; per slot the first W = 8/S sources are fetched unconditionally -- absent ones read a
// row of zeros -- so eight 16-byte loads per lane are in flight before the first add; longer slots
// continue W sources at a time.  vcol must be readable at index 0 even when there is no edge.
__device__ __attribute__((aligned(16))) float gather_zero_row[64];

// ADD (S = 1 only): out[row] = extra[row] + sum (out rows ldo apart; extra may alias out row for row)
template <int S, bool ADD = false>
__global__ __launch_bounds__(256) void csr_gather_sum_v2_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const int32_t* __restrict__ vrowptr,
                                                                const int32_t* __restrict__ vcol,
                                                                int64_t num_rows,
                                                                float* out, const float* extra = nullptr,
                                                                int64_t ld_extra = 0, int64_t ldo = 64) {
  constexpr int W = 8 / S;
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int64_t row_raw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool live = row_raw < num_rows;
  const int64_t row = live ? row_raw : num_rows - 1;
  const float* xc = x + 4 * l16;
  const float* zr = gather_zero_row + 4 * l16;
  int e[S], n[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    e[s] = vrowptr[row * S + s];
    n[s] = vrowptr[row * S + s + 1];
  }
  float4 acc[S];
  float4 v[S][W];
  // first step of every slot: S*W = 8 loads in flight
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const bool ok = e[s] + k < n[s];
      const int j = vcol[ok ? e[s] + k : (n[s] > 0 ? n[s] - 1 : 0)];
      v[s][k] = *reinterpret_cast<const float4*>(ok ? xc + (int64_t)j * ldx : zr);
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    acc[s] = v[s][0];
#pragma unroll
    for (int k = 1; k < W; ++k) {
      acc[s].x += v[s][k].x;
      acc[s].y += v[s][k].y;
      acc[s].z += v[s][k].z;
      acc[s].w += v[s][k].w;
    }
    e[s] += W;
  }
  // longer slots: W more sources per trip, all lane groups of the wave in step
#pragma unroll
  for (int s = 0; s < S; ++s) {
    while (__any(e[s] < n[s])) {
      float4 t[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const bool ok = e[s] + k < n[s];
        const int j = vcol[ok ? e[s] + k : (n[s] > 0 ? n[s] - 1 : 0)];
        t[k] = *reinterpret_cast<const float4*>(ok ? xc + (int64_t)j * ldx : zr);
      }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[s].x += t[k].x;
        acc[s].y += t[k].y;
        acc[s].z += t[k].z;
        acc[s].w += t[k].w;
      }
      e[s] += W;
    }
  }
  if (live) {
    if constexpr (ADD) {
      const float4 b = *reinterpret_cast<const float4*>(extra + row * ld_extra + 4 * l16);
      acc[0].x += b.x;
      acc[0].y += b.y;
      acc[0].z += b.z;
      acc[0].w += b.w;
      *reinterpret_cast<float4*>(out + row * ldo + 4 * l16) = acc[0];
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s)
        *reinterpret_cast<float4*>(out + (row * S + s) * 64 + 4 * l16) = acc[s];
    }
  }
}

// Backward of one SHMP layer w.r.t. its input rows (training trunk): for row i of X_l
//   out[i] = mask_i * ( seed_i + D[i, self block] + sum_{v in T(i)} Dv[v] )
// seed_i = dpool[seg_id[i]] for count rows (broadcast of the pooled gradient), dcanon[i - Nc] for canonical rows;
// D = dZ Wt^T [N, (S+1) 64] (slot blocks + self block; the self block sits at column self_off of the row's type);
// Dv = D viewed as virtual rows of 64; T(i) = the virtual rows that gathered row i (transposed index);
// mask = relu'(X_l[i]) (NULL for the input layer, whose rows come out of a Linear).  16 lanes x float4 per row,
// eight virtual rows in flight per lane group.
__global__ __launch_bounds__(256) void shmp_bwd_dx_kernel(const float* __restrict__ D, int64_t ldd,
                                                          const int32_t* __restrict__ t_rowptr,
                                                          const int32_t* __restrict__ t_col, int64_t num_rows,
                                                          int64_t num_count, int off_count, int off_canon,
                                                          const float* __restrict__ dpool, int64_t ld_pool,
                                                          const int32_t* __restrict__ seg_id,
                                                          const float* __restrict__ dcanon, int64_t ld_canon,
                                                          const float* __restrict__ relu_src, float mask_scale,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int64_t row_raw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool live = row_raw < num_rows;
  const int64_t row = live ? row_raw : num_rows - 1;
  const float* dv = D + 4 * l16;
  const float* zr = gather_zero_row + 4 * l16;
  int e = t_rowptr[row];
  const int n = t_rowptr[row + 1];
  const bool is_count = row < num_count;
  float4 acc;
  if (is_count)
    acc = *reinterpret_cast<const float4*>(dpool + (int64_t)seg_id[row] * ld_pool + 4 * l16);
  else if (dcanon)
    acc = *reinterpret_cast<const float4*>(dcanon + (row - num_count) * ld_canon + 4 * l16);
  else
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float4 sb = *reinterpret_cast<const float4*>(D + row * ldd + (is_count ? off_count : off_canon) + 4 * l16);
    acc.x += sb.x;
    acc.y += sb.y;
    acc.z += sb.z;
    acc.w += sb.w;
  }
  while (__any(e < n)) {
    float4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = e + k < n;
      const int j = t_col[ok ? e + k : (n > 0 ? n - 1 : 0)];
      t[k] = *reinterpret_cast<const float4*>(ok ? dv + (int64_t)j * 64 : zr);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc.x += t[k].x;
      acc.y += t[k].y;
      acc.z += t[k].z;
      acc.w += t[k].w;
    }
    e += 8;
  }
  if (live) {
    if (relu_src) {
      const float4 c = *reinterpret_cast<const float4*>(relu_src + row * 64 + 4 * l16);
      // (mask_scale: relu_src = dropout(relu(z)) is positive exactly where the element was kept and z > 0, and the
      //  gradient passes through the dropout's factor 1 / (1 - p) there -- no mask to regenerate for a relu)
      acc.x = c.x > 0.f ? acc.x * mask_scale : 0.f;
      acc.y = c.y > 0.f ? acc.y * mask_scale : 0.f;
      acc.z = c.z > 0.f ? acc.z * mask_scale : 0.f;
      acc.w = c.w > 0.f ? acc.w * mask_scale : 0.f;
    }
    *reinterpret_cast<float4*>(out + row * 64 + 4 * l16) = acc;
  }
}

// dst[i, 0:64 j] += src[i, 0:64 j]   (row strides ldd / lds; 16 lanes x float4 per 64 columns)
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, int64_t ldd,
                                                       const float* __restrict__ src, int64_t lds,
                                                       int64_t num_rows, int chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / chunks;
  if (row >= num_rows) return;
  const int c = (int)(i % chunks) * 4;
  float4 a = *reinterpret_cast<const float4*>(dst + row * ldd + c);
  const float4 b = *reinterpret_cast<const float4*>(src + row * lds + c);
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
  *reinterpret_cast<float4*>(dst + row * ldd + c) = a;
}

// global_add_pool over contiguous segments (+ one extra row per segment, added last to mirror the
// reference's cat([count, canonical]) order, gnn_model.py:88-89, 107).
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, int64_t ldx,
                                                          int ncols, int nchunks,
                                                          const int32_t* __restrict__ seg_ptr,
                                                          int64_t num_seg,
                                                          const float* __restrict__ extra,
                                                          int64_t ld_extra, float* __restrict__ out,
                                                          int64_t ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
  const int64_t b = idx / nchunks;
  if (b >= num_seg) return;
  const int c = (int)(idx % nchunks) * 64 + lane;
  if (c >= ncols) return;
  const int r0 = seg_ptr[b], r1 = seg_ptr[b + 1];
  float acc0 = 0.f, acc1 = 0.f;
  int r = r0;
  for (; r + 1 < r1; r += 2) {
    acc0 += x[(int64_t)r * ldx + c];
    acc1 += x[(int64_t)(r + 1) * ldx + c];
  }
  if (r < r1) acc0 += x[(int64_t)r * ldx + c];
  acc0 += acc1;
  if (extra) acc0 += extra[b * ld_extra + c];
  out[b * ldo + c] = acc0;
}

// streaming (read-once) 16-byte load: keeps the layer outputs from displacing the next kernel's
// working set in L2 / MALL
typedef float desco_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const desco_f4v v = __builtin_nontemporal_load(reinterpret_cast<const desco_f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// 64-column form (every pooling block of the SHMP path): a 16-lane group (float4 per lane) owns
// one segment and keeps four row loads in flight; a wave serves four consecutive segments, so it
// streams one contiguous row range.  Rows past the end of a segment re-read its last row and are
// discarded (branch-free loads, one wait per four).
__global__ __launch_bounds__(256) void segment_sum64_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int32_t* __restrict__ seg_ptr,
                                                            int64_t num_seg,
                                                            const float* __restrict__ extra,
                                                            int64_t ld_extra, float* __restrict__ out,
                                                            int64_t ldo, int64_t layer_stride = 0) {
  // blockIdx.y = layer (desco_segment_sum_layers_f32): x advances by layer_stride floats, extra and out by one
  // 64-column block per layer
  x += (int64_t)blockIdx.y * layer_stride;
  out += (int64_t)blockIdx.y * 64;
  if (extra) extra += (int64_t)blockIdx.y * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * 4 + (lane >> 4);
  const bool live = b < num_seg;
  const int r0 = live ? seg_ptr[b] : 0, r1 = live ? seg_ptr[b + 1] : 0;
  const float* xc = x + 4 * (lane & 15);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  for (int r = r0; r < r1; r += 4) {
    const int last = r1 - 1;
    const float4 v0 = nt_load4(xc + (int64_t)r * ldx);
    const float4 v1 = nt_load4(xc + (int64_t)(r + 1 < r1 ? r + 1 : last) * ldx);
    const float4 v2 = nt_load4(xc + (int64_t)(r + 2 < r1 ? r + 2 : last) * ldx);
    const float4 v3 = nt_load4(xc + (int64_t)(r + 3 < r1 ? r + 3 : last) * ldx);
    const float m1 = r + 1 < r1 ? 1.f : 0.f, m2 = r + 2 < r1 ? 1.f : 0.f, m3 = r + 3 < r1 ? 1.f : 0.f;
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += r + 1 < r1 ? v1.x : 0.f; a1.y += r + 1 < r1 ? v1.y : 0.f;
    a1.z += r + 1 < r1 ? v1.z : 0.f; a1.w += r + 1 < r1 ? v1.w : 0.f;
    a2.x += r + 2 < r1 ? v2.x : 0.f; a2.y += r + 2 < r1 ? v2.y : 0.f;
    a2.z += r + 2 < r1 ? v2.z : 0.f; a2.w += r + 2 < r1 ? v2.w : 0.f;
    a3.x += r + 3 < r1 ? v3.x : 0.f; a3.y += r + 3 < r1 ? v3.y : 0.f;
    a3.z += r + 3 < r1 ? v3.z : 0.f; a3.w += r + 3 < r1 ? v3.w : 0.f;
    (void)m1; (void)m2; (void)m3;
  }
  if (!live) return;
  float4 o = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                         (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  if (extra) {
    const float4 e = *reinterpret_cast<const float4*>(extra + b * ld_extra + 4 * (lane & 15));
    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
  }
  *reinterpret_cast<float4*>(out + b * ldo + 4 * (lane & 15)) = o;
}

// Second half of the fused pooling (desco_shmp_layer_pool_bf16x6_f32): the layer kernel left one
// partial row per (wave tile, segment) -- tile t's partials at slots slot_base[t] + k, k = the
// number of segment ends in the tile before the segment's first row there.  A 16-lane group
// (float4 per lane) owns one segment and adds its partials in tile order (a COX2-sized segment spans
// 1-3 tiles, a 600-row Syn segment 38-39): fixed order, no atomics.
__global__ __launch_bounds__(256) void pool_reduce_kernel(const float* __restrict__ part,
                                                          const uint32_t* __restrict__ bits,
                                                          const int32_t* __restrict__ slot_base,
                                                          const int32_t* __restrict__ seg_ptr,
                                                          int64_t num_seg,
                                                          const float* __restrict__ extra,
                                                          int64_t ld_extra, float* __restrict__ out,
                                                          int64_t ldo, int tsh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * 4 + (lane >> 4);
  if (b >= num_seg) return;
  const int r0 = seg_ptr[b], r1 = seg_ptr[b + 1];
  const float* pc = part + 4 * (lane & 15);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r1 > r0) {
    const int t0 = r0 >> tsh, t1 = (r1 - 1) >> tsh;                 // tiles of 1 << tsh rows
    for (int t = t0; t <= t1; ++t) {
      const int first = r0 > (t << tsh) ? r0 - (t << tsh) : 0;      // the segment's first row inside tile t
      const int k = __popc(bits[t] & ((1u << first) - 1u));
      const float4 v = *reinterpret_cast<const float4*>(pc + (int64_t)(slot_base[t] + k) * 64);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (extra) {
    const float4 e = *reinterpret_cast<const float4*>(extra + b * ld_extra + 4 * (lane & 15));
    a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
  }
  *reinterpret_cast<float4*>(out + b * ldo + 4 * (lane & 15)) = a;
}

// The same reduce for up to 8 layers' partial arrays in ONE launch (blockIdx.y = layer): the layers share the tile index
// and the segments, so InferencePipeline reduces all of a block's pooled layers at the end of the layer loop instead
// of one 0.2 ms launch per layer (seven launches fewer per pass; a pass over a real-size dataset is launch-bound).
struct PoolMulti {
  const float* part[8];
  const float* extra[8];
  float* out[8];
};
__global__ __launch_bounds__(256) void pool_reduce_multi_kernel(PoolMulti pm, const uint32_t* __restrict__ bits,
                                                                const int32_t* __restrict__ slot_base,
                                                                const int32_t* __restrict__ seg_ptr, int64_t num_seg,
                                                                int64_t ld_extra, int64_t ldo, int tsh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * 4 + (lane >> 4);
  if (b >= num_seg) return;
  const int l = blockIdx.y;
  const float* __restrict__ part = pm.part[l];
  const float* __restrict__ extra = pm.extra[l];
  const int r0 = seg_ptr[b], r1 = seg_ptr[b + 1];
  const float* pc = part + 4 * (lane & 15);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r1 > r0) {
    const int t0 = r0 >> tsh, t1 = (r1 - 1) >> tsh;
    for (int t = t0; t <= t1; ++t) {
      const int first = r0 > (t << tsh) ? r0 - (t << tsh) : 0;
      const int k = __popc(bits[t] & ((1u << first) - 1u));
      const float4 v = *reinterpret_cast<const float4*>(pc + (int64_t)(slot_base[t] + k) * 64);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (extra) {
    const float4 e = *reinterpret_cast<const float4*>(extra + b * ld_extra + 4 * (lane & 15));
    a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
  }
  *reinterpret_cast<float4*>(pm.out[l] + b * ldo + 4 * (lane & 15)) = a;
}

// count head, separable form of lightning_model.py:176-193, 210-221:
//   out[b,q] = b2 + sum_c w2[c] leaky(T[b,c] + Qh[q,c]),   leaky(z) = slope z + (1 - slope) relu(z)
//            = b2 + slope (w2.T[b]) + slope (w2.Qh[q]) + sum_c (1 - slope) w2[c] relu(T[b,c] + Qh[q,c])
// so the inner loop is add, max, fma (two packed + two scalar VALU per element pair).  One thread
// per target row b keeps all NQ query accumulators in registers.  A block's 256 T rows stream
// through LDS in 16-column chunks: loaded coalesced (4 lanes x 16 B per row), read back one row per
// thread (row stride 80 B: conflict-free ds_read_b128); the Qh rows and w2 are wave-uniform
// ds_read_b128 broadcasts.  T crosses HBM once.  (One thread per (b, q) pair re-fetched the T row 29
// times through the vector memory path, which then set the time.)  blockIdx.y selects a group of NQ
// queries: small batches are launched with NQ = 8 and four query groups to fill the chip.
constexpr int HEAD_MAXQ = 32, HEAD_MAXHID = 256, HEAD_TS = 20;
template <int NQ>
__global__ __launch_bounds__(256) void count_head_kernel(const float* __restrict__ t, int64_t ldt,
                                                         const float* __restrict__ qh, int64_t ldq,
                                                         int hid, const float* __restrict__ w2,
                                                         float b2, const float* __restrict__ b2_dev,
                                                         float slope, int exp2m1,
                                                         float* __restrict__ out, int64_t ldo,
                                                         int64_t num_b, int num_q) {
  __shared__ __attribute__((aligned(16))) float qt[NQ * HEAD_MAXHID];     // [q - q0][hid], rows >= num_q zero
  __shared__ __attribute__((aligned(16))) float ws[HEAD_MAXHID];          // w2
  __shared__ __attribute__((aligned(16))) float wr[HEAD_MAXHID];          // (1 - slope) w2
  __shared__ __attribute__((aligned(16))) float tch[256 * HEAD_TS];       // T chunk [256 rows][16 (+4 pad)]
  __shared__ float sq[HEAD_MAXQ];                                         // b2 + slope (w2.Qh[q])
  const int tid = threadIdx.x;
  const int q0 = blockIdx.y * NQ;  // this block's query group
  if (b2_dev) b2 = *b2_dev;        // bias read on the device (training: no host copy of a parameter)
  for (int i = tid; i < NQ * hid; i += 256) {
    const int q = i / hid, c = i - q * hid;
    qt[i] = q0 + q < num_q ? qh[(int64_t)(q0 + q) * ldq + c] : 0.f;
  }
  for (int i = tid; i < hid; i += 256) {
    ws[i] = w2[i];
    wr[i] = (1.f - slope) * w2[i];
  }
  __syncthreads();
  if (tid < NQ) {
    float s = 0.f;
    for (int c = 0; c < hid; ++c) s = fmaf(ws[c], qt[tid * hid + c], s);
    sq[tid] = slope * s + b2;
  }
  const int nch = hid / 16;
  const int lrow = tid >> 2, lpart = 4 * (tid & 3);      // coalesced load map: rows lrow + 64 p
  for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < num_b; b0 += (int64_t)gridDim.x * 256) {
    desco_f2 acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc[j] = desco_f2{0.f, 0.f};
    desco_f2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
    float4 nv0, nv1, nv2, nv3;                            // next chunk in flight
#define HEAD_LOAD(ch_)                                                                         \
  {                                                                                            \
    const float* p_ = t + (ch_) * 16 + lpart;                                                  \
    const int64_t r0_ = b0 + lrow, last_ = num_b - 1;                                          \
    nv0 = *reinterpret_cast<const float4*>(p_ + (r0_ < num_b ? r0_ : last_) * ldt);            \
    nv1 = *reinterpret_cast<const float4*>(p_ + (r0_ + 64 < num_b ? r0_ + 64 : last_) * ldt);  \
    nv2 = *reinterpret_cast<const float4*>(p_ + (r0_ + 128 < num_b ? r0_ + 128 : last_) * ldt); \
    nv3 = *reinterpret_cast<const float4*>(p_ + (r0_ + 192 < num_b ? r0_ + 192 : last_) * ldt); \
  }
    HEAD_LOAD(0)
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();                                    // the previous chunk has been consumed
      *reinterpret_cast<float4*>(tch + lrow * HEAD_TS + lpart) = nv0;
      *reinterpret_cast<float4*>(tch + (lrow + 64) * HEAD_TS + lpart) = nv1;
      *reinterpret_cast<float4*>(tch + (lrow + 128) * HEAD_TS + lpart) = nv2;
      *reinterpret_cast<float4*>(tch + (lrow + 192) * HEAD_TS + lpart) = nv3;
      __syncthreads();
      if (ch + 1 < nch) HEAD_LOAD(ch + 1)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c = ch * 16 + 4 * g4;
        const float4 tv = *reinterpret_cast<const float4*>(tch + tid * HEAD_TS + 4 * g4);
        const float4 wv = *reinterpret_cast<const float4*>(ws + c);
        const float4 rv = *reinterpret_cast<const float4*>(wr + c);
        const desco_f2 t01 = {tv.x, tv.y}, t23 = {tv.z, tv.w};
        const desco_f2 r01 = {rv.x, rv.y}, r23 = {rv.z, rv.w};
        s01 = __builtin_elementwise_fma(t01, desco_f2{wv.x, wv.y}, s01);
        s23 = __builtin_elementwise_fma(t23, desco_f2{wv.z, wv.w}, s23);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const float4 qv = *reinterpret_cast<const float4*>(qt + j * hid + c);
          desco_f2 z01 = t01 + desco_f2{qv.x, qv.y}, z23 = t23 + desco_f2{qv.z, qv.w};
          z01.x = fmaxf(z01.x, 0.f);
          z01.y = fmaxf(z01.y, 0.f);
          z23.x = fmaxf(z23.x, 0.f);
          z23.y = fmaxf(z23.y, 0.f);
          acc[j] = __builtin_elementwise_fma(z01, r01, acc[j]);
          acc[j] = __builtin_elementwise_fma(z23, r23, acc[j]);
        }
      }
    }
#undef HEAD_LOAD
    const int64_t b = b0 + tid;
    if (b < num_b) {
      const float st = slope * ((s01.x + s01.y) + (s23.x + s23.y));
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        if (q0 + j < num_q) {
          const float v = (acc[j].x + acc[j].y) + (st + sq[j]);
          out[b * ldo + q0 + j] = exp2m1 ? exp2f(v) - 1.f : v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src,
                                                           int64_t lds,
                                                           const int32_t* __restrict__ rows,
                                                           int64_t num_src, int ncols,
                                                           float* __restrict__ dst, int64_t ldd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_src * ncols) return;
  const int64_t b = i / ncols;
  const int c = (int)(i % ncols);
  dst[(int64_t)rows[b] * ldd + c] = src[b * lds + c];
}

__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* __restrict__ feat,
                                                            int64_t ldf, int k,
                                                            const float* __restrict__ wt,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int64_t ldo,
                                                            int64_t m, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * n) return;
  const int64_t r = i / n;
  const int c = (int)(i % n);
  float acc = bias ? bias[c] : 0.f;
  for (int j = 0; j < k; ++j) acc = fmaf(feat[r * ldf + j], wt[(int64_t)j * n + c], acc);
  out[r * ldo + c] = acc;
}

__global__ __launch_bounds__(256) void rowdot_add_kernel(const float* __restrict__ y, int64_t ldy,
                                                         int ncols, const float* __restrict__ w,
                                                         float b, const float* __restrict__ add,
                                                         float* __restrict__ out, int64_t num_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= num_rows) return;
  float acc = 0.f;
  for (int c = lane; c < ncols; c += 64) acc += y[r * ldy + c] * w[c];
  acc = wave_sum(acc);
  if (lane == 0) out[r] = (add ? add[r] : 0.f) + acc + b;
}

// First SHMP layer when every node of a type carries the same input row (all-zero node features:
// pre_mp output = its bias, workload.py:431-440): the aggregate of slot s is deg_s(i) * x0_src(s),
// so the layer is  out[i] = act( coef[S] + sum_s deg_s(i) * coef[s] ) (+ extra[i]) -- no gather, no GEMM.
// 16 lanes x float4 per row, four rows per wave and step, persistent grid-stride over row groups:
// the coefficient rows stay in registers and every store is a full 256-B row.
constexpr int DA_MAXS = 4;
__global__ __launch_bounds__(256) void degree_affine_kernel(const int32_t* __restrict__ vrowptr,
                                                            int64_t row0, int64_t num_rows, int S,
                                                            const float* __restrict__ coef, int act,
                                                            float slope,
                                                            const float* __restrict__ extra,
                                                            int64_t ld_extra,
                                                            float* __restrict__ out, int64_t ldo,
                                                            float* __restrict__ row_absmax) {
  const int c4 = 4 * (threadIdx.x & 15);
  float4 cf[DA_MAXS];
#pragma unroll
  for (int s = 0; s < DA_MAXS; ++s)
    cf[s] = s < S ? *reinterpret_cast<const float4*>(coef + s * 64 + c4)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 c0 = *reinterpret_cast<const float4*>(coef + S * 64 + c4);
  const int64_t stride = (int64_t)gridDim.x * 16;
  // four independent rows per trip (row pointers of all four are loaded before the first store)
  for (int64_t i0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i0 < num_rows; i0 += 4 * stride) {
    int dptr[4][DA_MAXS + 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride < num_rows ? i0 + u * stride : num_rows - 1;
      const int32_t* vp = vrowptr + (row0 + i) * S;
#pragma unroll
      for (int s = 0; s <= DA_MAXS; ++s) dptr[u][s] = vp[s <= S ? s : S];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= num_rows) break;
      float4 acc = c0;
#pragma unroll
      for (int s = 0; s < DA_MAXS; ++s) {
        if (s < S) {
          const float d = (float)(dptr[u][s + 1] - dptr[u][s]);
          // (explicit fmas: the same chain as degree_affine_pool_kernel and the layer kernel's SELFDEG rows -- left to
          //  -ffp-contract hipcc fuses some of these sites and not others)
          acc.x = fmaf(d, cf[s].x, acc.x);
          acc.y = fmaf(d, cf[s].y, acc.y);
          acc.z = fmaf(d, cf[s].z, acc.z);
          acc.w = fmaf(d, cf[s].w, acc.w);
        }
      }
      acc.x = apply_act(acc.x, act, slope);
      acc.y = apply_act(acc.y, act, slope);
      acc.z = apply_act(acc.z, act, slope);
      acc.w = apply_act(acc.w, act, slope);
      if (extra) {
        const float4 e = *reinterpret_cast<const float4*>(extra + i * ld_extra + c4);
        acc.x += e.x;
        acc.y += e.y;
        acc.z += e.z;
        acc.w += e.w;
      }
      *reinterpret_cast<float4*>(out + (row0 + i) * ldo + c4) = acc;
      if (row_absmax) {          // (wave-uniform) the row's largest |value|: 16 lanes x 4 columns
        uint32_t v_ = __float_as_uint(fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));
        uint32_t o_ = __builtin_amdgcn_update_dpp(0u, v_, 0xB1, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x4E, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x141, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x140, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        if ((threadIdx.x & 15) == 0) row_absmax[i] = __uint_as_float(v_);
      }
    }
  }
}

// Transposed index of a SYMMETRIC S-slot virtual-row CSR (the backward gather's index), computed
// row-locally: edge (dst j <- src k, slot s) has the mirror edge (dst k <- src j) in slot
// 2*(j is canonical) + (s & 1) (the tride bit belongs to the undirected edge), so the virtual rows
// that READ row j are { k*S + mirror_slot } over j's own sources k -- the same count, hence
// t_rowptr[j] = vrowptr[S*j].  Ascending virtual-row order = ascending k: an S-way merge of the
// row's S sorted slot lists (every neighbour sits in exactly one slot).  One thread per row.
__global__ __launch_bounds__(256) void vcsr_transpose_sym_kernel(const int32_t* __restrict__ vrowptr,
                                                                 const int32_t* __restrict__ vcol,
                                                                 int64_t num_rows, int S,
                                                                 int64_t num_count,
                                                                 int32_t* __restrict__ t_rowptr,
                                                                 int32_t* __restrict__ t_col) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > num_rows) return;
  if (j == num_rows) {
    t_rowptr[j] = vrowptr[S * num_rows];
    return;
  }
  int c[4], n[4];
  for (int s = 0; s < 4; ++s) {
    c[s] = s < S ? vrowptr[S * j + s] : 0;
    n[s] = s < S ? vrowptr[S * j + s + 1] : 0;
  }
  int out = c[0];
  t_rowptr[j] = out;
  const int jbit = (S == 4 && j >= num_count) ? 2 : 0;
  for (;;) {
    int best = -1, bk = 0x7fffffff;
    for (int s = 0; s < S; ++s) {
      if (c[s] < n[s]) {
        const int k = vcol[c[s]];
        if (k < bk) {
          bk = k;
          best = s;
        }
      }
    }
    if (best < 0) break;
    const int ms = S == 4 ? jbit + (best & 1) : best;
    t_col[out++] = bk * S + ms;
    ++c[best];
  }
}

// seg_id[r] = b for rows seg_ptr[b] <= r < seg_ptr[b+1] (one thread per segment)
__global__ __launch_bounds__(256) void segment_ids_kernel(const int32_t* __restrict__ seg_ptr,
                                                          int64_t num_seg, int32_t* __restrict__ seg_id) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= num_seg) return;
  for (int r = seg_ptr[b]; r < seg_ptr[b + 1]; ++r) seg_id[r] = (int32_t)b;
}

inline bool grid_ok(int64_t blocks) { return blocks > 0 && blocks <= INT32_MAX; }

}  // namespace desco

using namespace desco;

extern "C" int desco_csr_gather_sum_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                        const int32_t* vcol, int64_t num_rows, int slots,
                                        float* out, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  if (!x || !vrowptr || !out || num_rows < 0 || ldx % 4 || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || !(slots == 1 || slots == 2 || slots == 4))
    return fail(DESCO_EINVAL, "desco_csr_gather_sum_f32: bad argument");
  if (!vcol) return fail(DESCO_EINVAL, "desco_csr_gather_sum_f32: vcol must be readable at index 0");
  const int64_t blocks = (num_rows + 15) / 16;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_csr_gather_sum_f32: too many rows");
  hipStream_t st = (hipStream_t)stream;
  if (slots == 4)
    hipLaunchKernelGGL(csr_gather_sum_v2_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx,
                       vrowptr, vcol, num_rows, out);
  else if (slots == 2)
    hipLaunchKernelGGL(csr_gather_sum_v2_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx,
                       vrowptr, vcol, num_rows, out);
  else
    hipLaunchKernelGGL(csr_gather_sum_v2_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx,
                       vrowptr, vcol, num_rows, out);
  return launch_status("desco_csr_gather_sum_f32");
}

extern "C" int desco_csr_gather_sum_add_f32(const float* x, int64_t ldx, const int32_t* rowptr,
                                            const int32_t* col, int64_t num_rows, const float* extra,
                                            int64_t ld_extra, float* out, int64_t ldo,
                                            desco_stream_t stream) {
  if (num_rows == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (!x || !rowptr || !col || !extra || !out || num_rows < 0 || ldx % 4 || ld_extra % 4 || ldo % 4 ||
      !al16(x) || !al16(extra) || !al16(out))
    return fail(DESCO_EINVAL, "desco_csr_gather_sum_add_f32: bad argument");
  const int64_t blocks = (num_rows + 15) / 16;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_csr_gather_sum_add_f32: too many rows");
  hipLaunchKernelGGL((csr_gather_sum_v2_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rowptr, col, num_rows, out, extra, ld_extra, ldo);
  return launch_status("desco_csr_gather_sum_add_f32");
}

extern "C" int desco_shmp_bwd_dx_f32(const float* d, int64_t ldd, const int32_t* t_rowptr, const int32_t* t_col,
                                     int64_t num_rows, int64_t num_count, int self_off_count,
                                     int self_off_canon, const float* dpool, int64_t ld_pool,
                                     const int32_t* seg_id, const float* dcanon, int64_t ld_canon,
                                     const float* relu_src, float mask_scale, float* out, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (!d || !t_rowptr || !t_col || !dpool || !seg_id || !out || num_rows < 0 || num_count < 0 ||
      num_count > num_rows || ldd % 4 || ld_pool % 4 ||
      (dcanon && ld_canon % 4) || self_off_count % 4 || self_off_canon % 4 || !al16(d) || !al16(dpool) ||
      !al16(out) || (dcanon && !al16(dcanon)) || (relu_src && !al16(relu_src)))
    return fail(DESCO_EINVAL, "desco_shmp_bwd_dx_f32: bad argument");
  const int64_t blocks = (num_rows + 15) / 16;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_shmp_bwd_dx_f32: too many rows");
  hipLaunchKernelGGL(shmp_bwd_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d, ldd,
                     t_rowptr, t_col, num_rows, num_count, self_off_count, self_off_canon, dpool, ld_pool,
                     seg_id, dcanon, ld_canon, relu_src, mask_scale, out);
  return launch_status("desco_shmp_bwd_dx_f32");
}

extern "C" int desco_add_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t num_rows,
                                  int ncols, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (!dst || !src || num_rows < 0 || ncols <= 0 || ncols % 4 || ldd % 4 || lds % 4 || !al16(dst) || !al16(src))
    return fail(DESCO_EINVAL, "desco_add_rows_f32: bad argument");
  const int chunks = ncols / 4;
  const int64_t blocks = (num_rows * chunks + 255) / 256;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_add_rows_f32: too many rows");
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, ldd,
                     src, lds, num_rows, chunks);
  return launch_status("desco_add_rows_f32");
}

extern "C" int desco_segment_sum_f32(const float* x, int64_t ldx, int ncols,
                                     const int32_t* seg_ptr, int64_t num_seg, const float* extra,
                                     int64_t ld_extra, float* out, int64_t ldo,
                                     desco_stream_t stream) {
  if (num_seg == 0) return 0;
  if (!seg_ptr || !out || num_seg < 0 || ncols <= 0)
    return fail(DESCO_EINVAL, "desco_segment_sum_f32: bad argument");
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (ncols == 64 && ldx % 4 == 0 && ldo % 4 == 0 && al16(x) && al16(out) &&
      (!extra || (ld_extra % 4 == 0 && al16(extra)))) {
    const int64_t blocks64 = (num_seg + 15) / 16;
    if (!grid_ok(blocks64)) return fail(DESCO_EINVAL, "desco_segment_sum_f32: too many segments");
    hipLaunchKernelGGL(segment_sum64_kernel, dim3((unsigned)blocks64), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, seg_ptr, num_seg, extra, ld_extra, out, ldo);
    return launch_status("desco_segment_sum_f32");
  }
  const int nch = (ncols + 63) / 64;
  const int64_t blocks = (num_seg * nch + 3) / 4;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_segment_sum_f32: too many segments");
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, ncols, nch, seg_ptr, num_seg, extra, ld_extra, out, ldo);
  return launch_status("desco_segment_sum_f32");
}

extern "C" int desco_segment_sum_layers_f32(const float* x, int64_t ldx, int64_t layer_stride, int num_layers,
                                           const int32_t* seg_ptr, int64_t num_seg, const float* extra,
                                           int64_t ld_extra, float* out, int64_t ldo, desco_stream_t stream) {
  if (num_seg == 0 || num_layers == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (!x || !seg_ptr || !out || num_seg < 0 || num_layers < 0 || num_layers > 65535 || ldx % 4 || ldo % 4 ||
      layer_stride % 4 || !al16(x) || !al16(out) || (extra && (ld_extra % 4 || !al16(extra))))
    return fail(DESCO_EINVAL, "desco_segment_sum_layers_f32: bad argument");
  const int64_t blocks64 = (num_seg + 15) / 16;
  if (!grid_ok(blocks64)) return fail(DESCO_EINVAL, "desco_segment_sum_layers_f32: too many segments");
  hipLaunchKernelGGL(segment_sum64_kernel, dim3((unsigned)blocks64, (unsigned)num_layers), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, seg_ptr, num_seg, extra, ld_extra, out, ldo, layer_stride);
  return launch_status("desco_segment_sum_layers_f32");
}

extern "C" int desco_pool_reduce_f32(const float* part, const uint32_t* pool_bits,
                                     const int32_t* pool_slot, const int32_t* seg_ptr, int64_t num_seg,
                                     const float* extra, int64_t ld_extra, float* out, int64_t ldo,
                                     int tile_rows, desco_stream_t stream) {
  if (num_seg == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (tile_rows != 16 && tile_rows != 32)
    return fail(DESCO_EINVAL, "desco_pool_reduce_f32: tile_rows must be desco_shmp_pool_tile_rows()");
  if (!part || !pool_bits || !pool_slot || !seg_ptr || !out || num_seg < 0 || ldo % 4 || !al16(part) ||
      !al16(out) || (extra && (ld_extra % 4 || !al16(extra))))
    return fail(DESCO_EINVAL, "desco_pool_reduce_f32: bad argument");
  const int64_t blocks = (num_seg + 15) / 16;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_pool_reduce_f32: too many segments");
  hipLaunchKernelGGL(pool_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, part,
                     pool_bits, pool_slot, seg_ptr, num_seg, extra, ld_extra, out, ldo, tile_rows == 16 ? 4 : 5);
  return launch_status("desco_pool_reduce_f32");
}

extern "C" int desco_pool_reduce_multi_f32(int num, const float* const* pool_parts, const uint32_t* pool_bits,
                                           const int32_t* pool_slot, const int32_t* seg_ptr, int64_t num_seg,
                                           const float* const* extras, int64_t ld_extra, float* const* outs, int64_t ldo,
                                           int tile_rows, desco_stream_t stream) {
  if (num == 0 || num_seg == 0) return 0;
  auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  if (num < 0 || num > 8 || !pool_parts || !outs || (tile_rows != 16 && tile_rows != 32) || !pool_bits || !pool_slot ||
      !seg_ptr || num_seg < 0 || ldo % 4 || (extras && ld_extra % 4))
    return fail(DESCO_EINVAL, "desco_pool_reduce_multi_f32: bad argument (1..8 layers)");
  PoolMulti pm;
  for (int i = 0; i < 8; ++i) {
    const int j = i < num ? i : 0;
    pm.part[i] = pool_parts[j];
    pm.extra[i] = extras ? extras[j] : nullptr;
    pm.out[i] = outs[j];
    if (!pm.part[i] || !pm.out[i] || !al16(pm.part[i]) || !al16(pm.out[i]) || (pm.extra[i] && !al16(pm.extra[i])))
      return fail(DESCO_EINVAL, "desco_pool_reduce_multi_f32: NULL or misaligned layer operand");
  }
  const int64_t blocks = (num_seg + 15) / 16;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_pool_reduce_multi_f32: too many segments");
  hipLaunchKernelGGL(pool_reduce_multi_kernel, dim3((unsigned)blocks, (unsigned)num), dim3(256), 0, (hipStream_t)stream,
                     pm, pool_bits, pool_slot, seg_ptr, num_seg, ld_extra, ldo, tile_rows == 16 ? 4 : 5);
  return launch_status("desco_pool_reduce_multi_f32");
}

extern "C" int desco_count_head_f32(const float* t, int64_t ldt, const float* qh, int64_t ldq,
                                    int hid, const float* w2, float b2, const float* b2_dev,
                                    float slope, int exp2_minus_1, float* out, int64_t ldo,
                                    int64_t num_b, int num_q, desco_stream_t stream) {
  if (num_b == 0 || num_q == 0) return 0;
  if (!t || !qh || !w2 || !out || num_b < 0 || num_q < 0 || num_q > HEAD_MAXQ || hid <= 0 ||
      hid % 64 || hid > HEAD_MAXHID || ldt % 4 || (reinterpret_cast<uintptr_t>(t) & 15))
    return fail(DESCO_EINVAL, "desco_count_head_f32: bad argument (num_q <= 32, hid%64, hid <= 256)");
  int64_t blocks = (num_b + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 512)           // small batch: four times the blocks, a quarter of the queries each
    hipLaunchKernelGGL(count_head_kernel<8>, dim3((unsigned)blocks, (unsigned)((num_q + 7) / 8)), dim3(256), 0,
                       (hipStream_t)stream, t, ldt, qh, ldq, hid, w2, b2, b2_dev, slope, exp2_minus_1, out, ldo,
                       num_b, num_q);
  else if (num_q == 29)       // the standard 29 queries of sizes 3-5 (data.py:37): no padded accumulators
    hipLaunchKernelGGL(count_head_kernel<29>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t,
                       ldt, qh, ldq, hid, w2, b2, b2_dev, slope, exp2_minus_1, out, ldo, num_b, num_q);
  else
    hipLaunchKernelGGL(count_head_kernel<HEAD_MAXQ>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t,
                       ldt, qh, ldq, hid, w2, b2, b2_dev, slope, exp2_minus_1, out, ldo, num_b, num_q);
  return launch_status("desco_count_head_f32");
}

extern "C" int desco_scatter_rows_f32(const float* src, int64_t lds, const int32_t* rows,
                                      int64_t num_src, int ncols, float* dst, int64_t ldd,
                                      desco_stream_t stream) {
  if (num_src == 0 || ncols == 0) return 0;
  if (!src || !rows || !dst || num_src < 0 || ncols < 0)
    return fail(DESCO_EINVAL, "desco_scatter_rows_f32: bad argument");
  const int64_t blocks = (num_src * ncols + 255) / 256;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_scatter_rows_f32: too many elements");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, src, lds, rows, num_src, ncols, dst, ldd);
  return launch_status("desco_scatter_rows_f32");
}

extern "C" int desco_linear_smallk_f32(const float* feat, int64_t ldf, int k, const float* wt,
                                       const float* bias, float* out, int64_t ldo, int64_t m,
                                       int n, desco_stream_t stream) {
  if (m == 0 || n == 0) return 0;
  if (!feat || !wt || !out || m < 0 || n < 0 || k < 0)
    return fail(DESCO_EINVAL, "desco_linear_smallk_f32: bad argument");
  const int64_t blocks = (m * n + 255) / 256;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_linear_smallk_f32: too many elements");
  hipLaunchKernelGGL(linear_smallk_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, feat, ldf, k, wt, bias, out, ldo, m, n);
  return launch_status("desco_linear_smallk_f32");
}

extern "C" int desco_rowdot_add_f32(const float* y, int64_t ldy, int ncols, const float* w, float b,
                                    const float* add, float* out, int64_t num_rows,
                                    desco_stream_t stream) {
  if (num_rows == 0) return 0;
  if (!y || !w || !out || num_rows < 0 || ncols <= 0)
    return fail(DESCO_EINVAL, "desco_rowdot_add_f32: bad argument");
  const int64_t blocks = (num_rows + 3) / 4;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_rowdot_add_f32: too many rows");
  hipLaunchKernelGGL(rowdot_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     y, ldy, ncols, w, b, add, out, num_rows);
  return launch_status("desco_rowdot_add_f32");
}

// The closed-form first layer with its global_add_pool fused in (round 6): the count rows of X_1 used to be written by
// degree_affine_kernel and read back once by segment_sum64 for their pooled block -- 9.9 GB of reads per Syn_1827 x2 pass
// for sums of rows the kernel had in registers.  Here a wave walks whole 16-row tiles (the tiles of the fused
// pooling of the SHMP layer kernel: same pool_bits / pool_slot index, same partial-sum format, so the layer joins the
// others in pool_reduce): a wave computes and stores the tile's rows one column per lane and runs the layer kernel's
// running sum down the rows in row order (wave-uniform control flow).
namespace desco {
template <int CS>      // CS = the slot count as a constant (lane selects of the v_readlane are then immediates), or 0
__global__ __launch_bounds__(256) void degree_affine_pool_kernel(const int32_t* __restrict__ vrowptr, int64_t num_rows,
                                                                 int S_, const float* __restrict__ coef, int act,
                                                                 float slope, float* __restrict__ out, int64_t ldo,
                                                                 const uint32_t* __restrict__ pool_bits,
                                                                 const int32_t* __restrict__ pool_slot,
                                                                 float* __restrict__ pool_part) {
  // One WAVE per 16-row tile, one lane per column, no LDS and no barrier (round 6; the first form staged the tile in
  // LDS for wave 0's running sum behind a barrier per tile and wrote at 3.2-3.5 TB/s): the tile's 16 S slot degrees are
  // the differences of 16 S + 1 consecutive row pointers -- lane l holds difference l, a row's S of them reach the
  // vector unit as wave-uniform operands (v_readlane) -- a row leaves as one 256-byte store, and the lane's running sum
  // over the rows IS the column's pooled partial (same order as before: bit-identical results).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = CS ? CS : S_;
  float cf[DA_MAXS];
#pragma unroll
  for (int s = 0; s < DA_MAXS; ++s) cf[s] = s < S ? coef[s * 64 + lane] : 0.f;
  const float c0 = coef[S * 64 + lane];
  const int64_t ntiles = (num_rows + 15) / 16;
  const int64_t last = num_rows * S;                       // index of the final row pointer
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t e0 = t * 16 * S + lane;
    const int p0 = vrowptr[e0 < last ? e0 : last], p1 = vrowptr[e0 + 1 < last ? e0 + 1 : last];
    const int dvi = __float_as_int((float)(p1 - p0));      // degree of (row lane / S, slot lane % S); 0 past the end
    const uint32_t E = __builtin_amdgcn_readfirstlane(pool_bits[t]);
    int slot = __builtin_amdgcn_readfirstlane(pool_slot[t]);
    const int nr = (int)((num_rows - t * 16) < 16 ? (num_rows - t * 16) : 16);
    float* o = out ? out + t * 16 * ldo + lane : nullptr;
    float* pp = pool_part + lane;
    float run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < nr) {                                        // (wave-uniform)
        float acc = c0;
#pragma unroll
        for (int s = 0; s < DA_MAXS; ++s)
          if (s < S) acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(dvi, r * S + s)), cf[s], acc);
        acc = apply_act(acc, act, slope);
        if (o) o[r * ldo] = acc;
        run += acc;
        if ((E >> r) & 1u) {                               // row r ends its segment (wave-uniform)
          pp[(int64_t)slot * 64] = run;
          ++slot;
          run = 0.f;
        }
      }
    }
    if (nr > 0 && !((E >> (nr - 1)) & 1u)) pp[(int64_t)slot * 64] = run;
  }
}
}  // namespace desco

extern "C" int desco_degree_affine_pool_f32(const int32_t* vrowptr, int64_t num_rows, int slots, const float* coef,
                                            int act, float slope, float* out, int64_t ldo, const uint32_t* pool_bits,
                                            const int32_t* pool_slot, float* pool_part, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  if (!vrowptr || !coef || !pool_bits || !pool_slot || !pool_part || num_rows < 0 || slots < 1 || slots > DA_MAXS ||
      (out && (ldo % 4 || mis16(out))) || mis16(coef) || mis16(pool_part))
    return fail(DESCO_EINVAL, "desco_degree_affine_pool_f32: bad argument (slots <= 4, 16-byte rows)");
  int64_t blocks = ((num_rows + 15) / 16 + 3) / 4;         // four wave tiles per block
  if (blocks > 8 * 256) blocks = 8 * 256;
  if (slots == 4)
    hipLaunchKernelGGL(degree_affine_pool_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vrowptr,
                       num_rows, slots, coef, act, slope, out, ldo, pool_bits, pool_slot, pool_part);
  else
    hipLaunchKernelGGL(degree_affine_pool_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vrowptr,
                       num_rows, slots, coef, act, slope, out, ldo, pool_bits, pool_slot, pool_part);
  return launch_status("desco_degree_affine_pool_f32");
}

extern "C" int desco_degree_affine_f32(const int32_t* vrowptr, int64_t row0, int64_t num_rows,
                                       int slots, const float* coef, int act, float slope,
                                       const float* extra, int64_t ld_extra, float* out,
                                       int64_t ldo, float* row_absmax, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  if (!vrowptr || !coef || !out || row0 < 0 || num_rows < 0 || slots < 1 || slots > DA_MAXS ||
      ldo % 4 || mis16(out) || mis16(coef) || (extra && (ld_extra % 4 || mis16(extra))))
    return fail(DESCO_EINVAL, "desco_degree_affine_f32: bad argument (slots <= 4, 16-byte rows)");
  int64_t blocks = (num_rows + 15) / 16;
  if (blocks > 8 * 256) blocks = 8 * 256;
  hipLaunchKernelGGL(degree_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     vrowptr, row0, num_rows, slots, coef, act, slope, extra, ld_extra, out, ldo, row_absmax);
  return launch_status("desco_degree_affine_f32");
}

extern "C" int desco_vcsr_transpose_sym(const int32_t* vrowptr, const int32_t* vcol, int64_t num_rows,
                                        int slots, int64_t num_count, int32_t* t_rowptr,
                                        int32_t* t_col, desco_stream_t stream) {
  if (!vrowptr || !t_rowptr || num_rows < 0 || !(slots == 1 || slots == 2 || slots == 4) ||
      num_count < 0 || num_count > num_rows)
    return fail(DESCO_EINVAL, "desco_vcsr_transpose_sym: bad argument (slots in {1,2,4})");
  const int64_t blocks = (num_rows + 1 + 255) / 256;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_vcsr_transpose_sym: too many rows");
  hipLaunchKernelGGL(vcsr_transpose_sym_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, vrowptr, vcol, num_rows, slots, num_count, t_rowptr, t_col);
  return launch_status("desco_vcsr_transpose_sym");
}

extern "C" int desco_segment_ids(const int32_t* seg_ptr, int64_t num_seg, int32_t* seg_id,
                                 desco_stream_t stream) {
  if (num_seg == 0) return 0;
  if (!seg_ptr || !seg_id || num_seg < 0) return fail(DESCO_EINVAL, "desco_segment_ids: bad argument");
  const int64_t blocks = (num_seg + 255) / 256;
  if (!grid_ok(blocks)) return fail(DESCO_EINVAL, "desco_segment_ids: too many segments");
  hipLaunchKernelGGL(segment_ids_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     seg_ptr, num_seg, seg_id);
  return launch_status("desco_segment_ids");
}
