// Fused SHMP layer for gfx950: SAGEConv gather-aggregate for every relation slot, the per-relation
// Linear, the to_hetero sum, the update Linear and the ReLU (gnn_model.py:262-264, 273, 389-395) in
// ONE launch; the aggregates never touch HBM.
//
//   out[i] = relu( sum_{s<sm} (sum_{e in vrow(i,s)} x[vcol[e]]) * Wt[s] + x[i] * Wt[sm] + bias
//                  + sum_{sm<=s<sm+st} sum_{e in vrow(i,s)} ytab[vcol[e]-ytab_row0][(s-sm)*64 : +64] )
//
// "MFMA slots" (s < sm) are gathered into LDS and multiplied on the matrix cores; "table slots"
// are the same linear map re-associated, (sum_j x_j) W = sum_j (x_j W): their sources were already
// multiplied by W (ytab), so they are a row add in the epilogue.  The host uses the table form for
// the canonical->count relations (at most one source per row), cutting K from 320 to 192.
//
// Structure (wave-autonomous, persistent, software-pipelined):
//   * block = 8 waves = one CU (2 waves per SIMD); all (sm+1) 64x64 weight blocks are loaded into
//     LDS ONCE per block and stay resident while the block strides over 256-row tiles;
//   * every wave owns 32 destination rows of a tile end to end and never meets a block barrier in
//     the tile loop: while one wave of a SIMD waits on its gather the other feeds the MFMA pipe;
//   * the CSR slice of the wave's NEXT tile (row pointers, then up to WCAP source ids) is
//     prefetched into a second private LDS buffer under the current tile's work, so the only
//     dependent global access on the critical path is the 256-B feature row of a neighbour
//     (16 lanes x float4 per row, 8 rows in flight per lane group);
//   * table slots are folded into the accumulator INIT (their loads fly under the first gather);
//   * a K block is multiplied through a private [32][65] A image (64 MFMAs: 32x64 output, two
//     accumulators share the A fragment);
//   * HBM traffic per row and layer: one 256-B read of x, one 256-B write, ~20 B of indices
//     (neighbour re-reads hit L2: a neighborhood's rows are contiguous).
#include "common_device.hpp"

namespace desco {

constexpr int WR = 32;        // rows per wave
constexpr int NW = 8;         // waves per block (2 per SIMD)
constexpr int AH = 65;        // A image row stride (floats): conflict-free ds_read_b32 over rows
constexpr int CS = 68;        // C staging row stride (floats, 16-B aligned rows)
constexpr int MAXS = 4;       // relation slots stored per row
constexpr int RPN = WR * MAXS + 1;
// source ids staged per wave (longer slices fall back to global); sized so that KB resident weight
// blocks + 8 wave regions fit the 160 KB LDS
constexpr int wcap_for(int kb) { return kb >= 4 ? 128 : 256; }   // multiples of 64
constexpr int wave_lds_for(int kb) { return WR * CS + 2 * RPN + 2 * wcap_for(kb); }   // floats

struct ShmpArgs {
  const float* x;
  int64_t ldx;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t row0, num_rows;
  int S, sm, st;
  const float* wt;
  const float* bias;
  const float* ytab;
  int64_t ldy, ytab_row0;
  float* out;
  int64_t ldo;
};

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

template <int KB, int ST>   // KB = sm + 1 resident weight blocks (1..4), ST table slots (0..2)
__global__ __launch_bounds__(NW * 64) void shmp_layer_f32_kernel(ShmpArgs g) {
  constexpr int WCAP = wcap_for(KB);
  constexpr int WAVE_LDS = wave_lds_for(KB);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bimg = lds;                                       // [KB*64][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Aw = lds + KB * 64 * 64 + wave * WAVE_LDS;        // [32][65]
  int* rpb = reinterpret_cast<int*>(Aw + WR * CS);         // 2 x [32*S+1] row pointers (absolute)
  int* ecb = rpb + 2 * RPN;                                // 2 x [WCAP] source ids

  // ---- resident weights -------------------------------------------------------------------
  for (int i = tid; i < KB * 1024; i += NW * 64)
    *reinterpret_cast<float4*>(Bimg + 4 * i) = *reinterpret_cast<const float4*>(g.wt + 4 * i);
  __syncthreads();

  const int grp = lane >> 4, l16 = lane & 15;
  const int S = g.S;
  const int nslot = WR * S + 1;                            // <= 129: at most 3 per lane
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);

  // prologue: indices of this wave's first tile
  int64_t tile = blockIdx.x;
  {
    const int64_t w0 = tile * (NW * WR) + wave * WR;
    if (tile < ntiles && w0 < g.num_rows) {
      const int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
      const int nptr = nr * S + 1;
      for (int i = lane; i < nslot; i += 64)
        rpb[i] = g.vrowptr[(g.row0 + w0) * S + (i < nptr ? i : nptr - 1)];
      const int eb = rpb[0], ecnt = rpb[WR * S] - eb;
      for (int i = lane; i < ecnt && i < WCAP; i += 64) ecb[i] = g.vcol[eb + i];
    }
  }

  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const int64_t w0 = tile * (NW * WR) + wave * WR;       // first row of this wave (relative)
    if (w0 >= g.num_rows) continue;
    const int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
    const int64_t grow0 = g.row0 + w0;
    int* rp = rpb + cur * RPN;
    int* ec = ecb + cur * WCAP;
    int* rpn = rpb + (cur ^ 1) * RPN;
    int* ecn = ecb + (cur ^ 1) * WCAP;

    // ---- prefetch the row pointers of the next tile (registers now, LDS later) --------------
    const int64_t tn = tile + gridDim.x;
    const int64_t w0n = tn * (NW * WR) + wave * WR;
    const bool has_next = tn < ntiles && w0n < g.num_rows;
    int p0 = 0, p1 = 0, p2 = 0;
    if (has_next) {
      const int nrn = (int)((g.num_rows - w0n) < WR ? (g.num_rows - w0n) : WR);
      const int nptr = nrn * S + 1;
      const int32_t* src = g.vrowptr + (g.row0 + w0n) * S;
      p0 = src[lane < nptr ? lane : nptr - 1];
      if (lane + 64 < nslot) p1 = src[lane + 64 < nptr ? lane + 64 : nptr - 1];
      if (lane + 128 < nslot) p2 = src[lane + 128 < nptr ? lane + 128 : nptr - 1];
    }
    int qn[WCAP / 64];         // source ids of the next tile (registers until the tile ends)
#pragma unroll
    for (int i = 0; i < WCAP / 64; ++i) qn[i] = 0;
    int ebn = 0, ecntn = 0;

    const int ebase = rp[0];
    const int cl = lane & 31;
    // ---- accumulator init: bias + table slots.  Round k adds the k-th source of every
    // (row, table slot); all loads of a round are independent, so they fly together (and under
    // the first gather).  Absent sources read table row 0 and are discarded by a select.
    f32x16 acc0, acc1;
    {
      const float bv0 = g.bias ? g.bias[cl] : 0.f, bv1 = g.bias ? g.bias[32 + cl] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc0[i] = bv0;
        acc1[i] = bv1;
      }
    }
    if (ST > 0) {
      for (int round = 0;; ++round) {
        bool more = false;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
#pragma unroll
          for (int ts = 0; ts < ST; ++ts) {
            const int v = r * S + g.sm + ts;
            const int e = rp[v] - ebase + round, e1 = rp[v + 1] - ebase;
            const bool ok = e < e1;
            more |= e + 1 < e1;
            const int64_t j = ok ? (int64_t)(e < WCAP ? ec[e < WCAP ? e : 0] : g.vcol[ebase + e]) -
                                       g.ytab_row0
                                 : 0;
            const float* y = g.ytab + j * g.ldy + ts * 64 + cl;
            const float y0 = y[0], y1 = y[32];
            acc0[reg] += ok ? y0 : 0.f;
            acc1[reg] += ok ? y1 : 0.f;
          }
        }
        if (!__any(more)) break;
      }
    }

#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      // ---- gather K block kb (slot kb, or the row itself for kb == KB-1) ----------------------
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0,
             a6 = a0, a7 = a0;
      if (kb == KB - 1) {
        const float* xs = g.x + grow0 * g.ldx + 4 * l16;
        // rows beyond nr re-read the wave's last valid row (never stored)
#define DESCO_SELF(av_, it_)                                                   \
  {                                                                            \
    const int r_ = (it_) * 4 + grp;                                            \
    av_ = *reinterpret_cast<const float4*>(xs + (int64_t)(r_ < nr ? r_ : nr - 1) * g.ldx); \
  }
        DESCO_SELF(a0, 0) DESCO_SELF(a1, 1) DESCO_SELF(a2, 2) DESCO_SELF(a3, 3)
        DESCO_SELF(a4, 4) DESCO_SELF(a5, 5) DESCO_SELF(a6, 6) DESCO_SELF(a7, 7)
#undef DESCO_SELF
      } else {
        // edge cursors of the 8 rows this lane group serves (row = it*4 + grp)
        int c0, c1, c2, c3, c4, c5, c6, c7, n0, n1, n2, n3, n4, n5, n6, n7;
#define DESCO_CUR(c_, n_, it_)                        \
  {                                                   \
    const int v_ = ((it_) * 4 + grp) * S + kb;        \
    c_ = rp[v_] - ebase;                              \
    n_ = rp[v_ + 1] - ebase;                          \
  }
        DESCO_CUR(c0, n0, 0) DESCO_CUR(c1, n1, 1) DESCO_CUR(c2, n2, 2) DESCO_CUR(c3, n3, 3)
        DESCO_CUR(c4, n4, 4) DESCO_CUR(c5, n5, 5) DESCO_CUR(c6, n6, 6) DESCO_CUR(c7, n7, 7)
#undef DESCO_CUR
        const float* xb = g.x + 4 * l16;
        while (__any((c0 < n0) | (c1 < n1) | (c2 < n2) | (c3 < n3) | (c4 < n4) | (c5 < n5) |
                     (c6 < n6) | (c7 < n7))) {
#define DESCO_STEP(av_, c_, n_)                                                          \
  if (c_ < n_) {                                                                         \
    const int64_t j_ = c_ < WCAP ? ec[c_] : g.vcol[ebase + c_];                          \
    f4add(av_, *reinterpret_cast<const float4*>(xb + j_ * g.ldx));                       \
    ++c_;                                                                                \
  }
          DESCO_STEP(a0, c0, n0) DESCO_STEP(a1, c1, n1) DESCO_STEP(a2, c2, n2)
          DESCO_STEP(a3, c3, n3) DESCO_STEP(a4, c4, n4) DESCO_STEP(a5, c5, n5)
          DESCO_STEP(a6, c6, n6) DESCO_STEP(a7, c7, n7)
#undef DESCO_STEP
        }
      }
      if (kb == 0 && has_next) {
        // next tile's row pointers have landed: publish them, then fetch its source ids
        rpn[lane] = p0;
        if (lane + 64 < nslot) rpn[lane + 64] = p1;
        if (lane + 128 < nslot) rpn[lane + 128] = p2;
        ebn = rpn[0];
        ecntn = rpn[WR * S] - ebn;
#pragma unroll
        for (int i = 0; i < WCAP / 64; ++i)
          if (lane + 64 * i < ecntn) qn[i] = g.vcol[ebn + lane + 64 * i];
      }
      // ---- write the A image (row = it*4 + grp, 4 floats at 4*l16), 64 MFMAs -------------------
#define DESCO_PUT(av_, it_)                              \
  {                                                      \
    float* d_ = Aw + ((it_) * 4 + grp) * AH + 4 * l16;   \
    d_[0] = av_.x;                                       \
    d_[1] = av_.y;                                       \
    d_[2] = av_.z;                                       \
    d_[3] = av_.w;                                       \
  }
      DESCO_PUT(a0, 0) DESCO_PUT(a1, 1) DESCO_PUT(a2, 2) DESCO_PUT(a3, 3)
      DESCO_PUT(a4, 4) DESCO_PUT(a5, 5) DESCO_PUT(a6, 6) DESCO_PUT(a7, 7)
#undef DESCO_PUT
      {
        // A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
        const float* as = Aw + (lane & 31) * AH + (lane >> 5);
        const float* bs = Bimg + (kb * 64 + (lane >> 5)) * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
          const float a = as[2 * kk];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs[2 * kk * 64], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs[2 * kk * 64 + 32], acc1, 0, 0, 0);
        }
      }
    }

    if (has_next) {   // publish the next tile's source ids
#pragma unroll
      for (int i = 0; i < WCAP / 64; ++i)
        if (lane + 64 * i < ecntn) ecn[lane + 64 * i] = qn[i];
    }

    // ---- epilogue: relu, transpose through the (now free) A image as [32][68], then whole-row
    // float4 stores (4 rows x 256 B per wave instruction instead of 2 x 128 B dword stores).
    // C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const float v0 = acc0[reg], v1 = acc1[reg];
      Aw[r * CS + cl] = v0 > 0.f ? v0 : 0.f;
      Aw[r * CS + 32 + cl] = v1 > 0.f ? v1 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + grp;
      if (r < nr)
        *reinterpret_cast<float4*>(g.out + (grow0 + r) * g.ldo + 4 * l16) =
            *reinterpret_cast<const float4*>(Aw + r * CS + 4 * l16);
    }
  }
}

}  // namespace desco

extern "C" int desco_shmp_layer_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                    const int32_t* vcol, int64_t row0, int64_t num_rows,
                                    int slots_stored, int slots_mfma, int slots_table,
                                    const float* wt, const float* bias, const float* ytab,
                                    int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo,
                                    desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x || !vrowptr || !wt || !out || row0 < 0 || num_rows < 0 || slots_mfma < 0 ||
      slots_mfma > 3 || slots_table < 0 || slots_mfma + slots_table > slots_stored || slots_stored < 1 ||
      slots_stored > MAXS || slots_table > 2 || (slots_table > 0 && !ytab) || ldx % 4 || ldo % 4 || mis16(x) || mis16(wt) || mis16(out) ||
      x == out)
    return fail(DESCO_EINVAL, "desco_shmp_layer_f32: bad argument (slots_mfma <= 3, slots_table <= 2)");
  const int64_t ntiles = (num_rows + NW * WR - 1) / (NW * WR);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const int kb = slots_mfma + 1;
  const size_t shmem = sizeof(float) * ((size_t)kb * 64 * 64 + (size_t)NW * wave_lds_for(kb));
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  ShmpArgs g{x,   ldx,  vrowptr, vcol, row0,      num_rows, slots_stored, slots_mfma, slots_table,
             wt,  bias, ytab,    ldy,  ytab_row0, out,      ldo};
  hipStream_t st = (hipStream_t)stream;
#define DESCO_LAUNCH(KB_, ST_)                                                                   \
  {                                                                                              \
    static bool attr_set = false;                                                                \
    if (!attr_set) {                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_layer_f32_kernel<KB_, ST_>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
      attr_set = true;                                                                           \
    }                                                                                            \
    hipLaunchKernelGGL((shmp_layer_f32_kernel<KB_, ST_>), dim3(grid), dim3(NW * 64), shmem, st,  \
                       g);                                                                       \
  }
#define DESCO_LAUNCH_ST(KB_)                    \
  switch (slots_table) {                        \
    case 0: DESCO_LAUNCH(KB_, 0) break;         \
    case 1: DESCO_LAUNCH(KB_, 1) break;         \
    default: DESCO_LAUNCH(KB_, 2) break;        \
  }
  switch (kb) {
    case 1: DESCO_LAUNCH_ST(1) break;
    case 2: DESCO_LAUNCH_ST(2) break;
    case 3: DESCO_LAUNCH_ST(3) break;
    default: DESCO_LAUNCH_ST(4) break;
  }
#undef DESCO_LAUNCH_ST
#undef DESCO_LAUNCH
  return launch_status("desco_shmp_layer_f32");
}
